#!/usr/bin/env python
"""Stream-K form of k_conv_igemm (VD_CONV_STREAMK) against the one-tile-per-workgroup form: bit-equality of the outputs and
launch times, forward and data gradient, on yolo3_darknet53 layer shapes (fp16-split arithmetic).
usage: python tools/sk_bench.py [--batch 64] [--iters 10] [--tiles 1,5,...] [--only i,j]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import lib as L
from viddet_amd import ops

SHAPES = [  # cin, cout, k, stride, spatial_in
    (128, 256, 3, 1, 52), (256, 512, 3, 1, 26), (512, 1024, 3, 1, 13), (64, 128, 3, 1, 104),
    (1024, 512, 1, 1, 13), (512, 256, 1, 1, 26), (256, 128, 1, 1, 52), (512, 256, 3, 1, 26), (1024, 512, 3, 1, 13),
    (256, 512, 3, 2, 52),
]


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
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tiles", default="1,5,2,6,11,12")
    ap.add_argument("--only", default=None)
    ap.add_argument("--nohalo", action="store_true")
    a = ap.parse_args()
    B = a.batch
    shapes = SHAPES if a.only is None else [SHAPES[int(i)] for i in a.only.split(",")]
    skws = ops.streamk_workspace()
    bad = 0
    for cin, cout, k, s, hin in shapes:
        pad = k // 2
        ho = (hin + 2 * pad - k) // s + 1
        x = torch.randn(B, hin, hin, cin, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        wp = torch.empty(cout, k * k * cin, device="cuda")
        ops.pack_weight_fwd(w, wp, cout)
        res = torch.randn(B, ho, ho, cout, device="cuda")
        y0, y1 = torch.empty(B, ho, ho, cout, device="cuda"), torch.empty(B, ho, ho, cout, device="cuda")
        ax, aw = ops.amax(x), ops.amax(wp)
        flops = 2.0 * cin * cout * k * k * ho * ho * B
        split = "f16x2nh" if a.nohalo else "f16x2"
        for tile in [int(t) for t in a.tiles.split(",")]:
            kw = dict(k=k, stride=s, pad=pad, Co=cout, tile=tile, split=split, amax_in=ax, amax_w=aw, residual=res, leaky=True,
                      scale=torch.ones(cout, device="cuda"), shift=torch.zeros(cout, device="cuda"))
            ds = []
            ops.conv_fwd(x, wp, y1, streamk_ws=skws, desc_out=ds, **kw)
            used = bool(L.load().vd_conv_igemm_streamk(C.byref(ds[0])))
            ops.conv_fwd(x, wp, y0, **kw)
            torch.cuda.synchronize()
            same = bool(torch.equal(y0, y1))
            t0 = timeit(lambda: ops.conv_fwd(x, wp, y0, **kw), a.iters)
            t1 = timeit(lambda: ops.conv_fwd(x, wp, y1, streamk_ws=skws, **kw), a.iters) if used else float("nan")
            same2 = bool(torch.equal(y0, y1))
            bad += int(used and not (same and same2))
            print("%dx%d s%d %4d->%-4d @%-3d tile %2d  classic %.4f ms %6.1f TF | stream-K %s %.4f ms %6.1f TF  x%.3f  %s" % (
                k, k, s, cin, cout, hin, tile, t0, flops / t0 / 1e9, "on " if used else "n/a", t1, flops / t1 / 1e9, t0 / t1,
                "bit-identical" if (same and same2) else "MISMATCH max|d| %.3e" % float((y0 - y1).abs().max())), flush=True)
    print("mismatches: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
