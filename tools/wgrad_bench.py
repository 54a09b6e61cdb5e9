#!/usr/bin/env python
"""Weight-gradient kernels on the 3x3 / stride-1 layer shapes of yolo3_darknet53 (B = 64 @416 by default): the generic
k_conv_wgrad against the halo-ring kernel (VD_WGRAD_HALO, vd_wgrad_halo.hip), fp32 tensors in the fp16-split arithmetic
and bf16-stored operands.  TFLOP/s are algorithmic (2 * Ci * Co * 9 * pixels), slab reduction included.
usage: python tools/wgrad_bench.py [--batch 64] [--iters 10] [--size 416] [--only 0,1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops

# cin, cout, spatial divisor (map = size / div)
SHAPES = [(128, 256, 8), (256, 512, 16), (512, 1024, 32), (64, 128, 4), (256, 128, 8), (512, 256, 16)]


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    B = a.batch
    ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    shapes = SHAPES if a.only is None else [SHAPES[int(i)] for i in a.only.split(",")]
    print("%-24s %10s %10s %10s %10s   %s" % ("layer", "f16x2", "f16x2 halo", "bf16", "bf16 halo", "max |halo - generic| / max |generic|"))
    for cin, cout, div in shapes:
        hin = a.size // div
        x = torch.randn(B, hin, hin, cin, device="cuda")
        dy = torch.randn(B, hin, hin, cout, device="cuda")
        xb, dyb = x.to(torch.bfloat16), dy.to(torch.bfloat16)
        ax, ad = ops.amax(x), ops.amax(dy)
        flops = 2.0 * cin * cout * 9 * hin * hin * B
        res, outs = [], {}
        for name, xx, dd, mode in (("g", x, dy, "f16x2"), ("h", x, dy, "f16x2h"), ("bg", xb, dyb, False), ("bh", xb, dyb, "halo")):
            dwp = torch.empty(cout, 9 * cin, device="cuda")
            kw = dict(amax_in=ax, amax_dout=ad) if xx.dtype == torch.float32 else {}
            t = timeit(lambda: ops.conv_wgrad(xx, dd, dwp, ws, k=3, stride=1, pad=1, Co=cout, split=mode, **kw), a.iters)
            res.append(flops / t / 1e9)
            outs[name] = dwp
        d1 = float((outs["h"] - outs["g"]).abs().max() / outs["g"].abs().max())
        d2 = float((outs["bh"] - outs["bg"]).abs().max() / outs["bg"].abs().max())
        print("3x3 %4d->%-4d @%-3d       %10.1f %10.1f %10.1f %10.1f   %.1e  %.1e" % (cin, cout, hin, *res, d1, d2), flush=True)


if __name__ == "__main__":
    main()
