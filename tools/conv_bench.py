#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on representative yolo3_darknet53 layer shapes (B=64 @416).
usage: python tools/conv_bench.py [--batch 64] [--iters 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops

SHAPES = [  # cin, cout, k, stride, spatial_in
    (128, 256, 3, 1, 52), (256, 512, 3, 1, 26), (512, 1024, 3, 1, 13), (64, 128, 3, 1, 104), (32, 64, 3, 1, 208),
    (1024, 512, 1, 1, 13), (512, 256, 1, 1, 26), (256, 128, 1, 1, 52), (64, 32, 1, 1, 208),
    (256, 512, 3, 2, 52), (32, 64, 3, 2, 416),
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
    ap.add_argument("--tile", type=int, default=0, help="forward/dgrad tile variant (0 = heuristic)")
    ap.add_argument("--split", action="store_true", help="split-operand fp32 products (VD_MATH_SPLIT)")
    ap.add_argument("--math", default=None, choices=["native", "split", "f16x2", "bf16"],
                    help="product arithmetic (overrides --split): f16x2 = two-way fp16 split (VD_MATH_F16X2)")
    ap.add_argument("--only", default=None, help="comma list of shape indices")
    ap.add_argument("--check", action="store_true", help="print the max/rms error of the forward output against an fp64 CPU conv")
    a = ap.parse_args()
    B = a.batch
    if a.math is not None:
        a.split = {"native": False, "split": True, "f16x2": "f16x2", "bf16": "bf16"}[a.math]
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    shapes = SHAPES if a.only is None else [SHAPES[int(i)] for i in a.only.split(",")]
    print("%-28s %9s %9s %9s   (TFLOP/s; fp32 MFMA peak 157.3)" % ("layer", "fwd", "dgrad", "wgrad"))
    for cin, cout, k, s, hin in shapes:
        pad = k // 2
        ho = (hin + 2 * pad - k) // s + 1
        x = torch.randn(B, hin, hin, cin, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        wp = torch.empty(cout, k * k * cin, device="cuda")
        ops.pack_weight_fwd(w, wp, cout)
        y = torch.empty(B, ho, ho, cout, device="cuda")
        dy = torch.randn(B, ho, ho, cout, device="cuda")
        dx = torch.empty_like(x)
        dwp = torch.empty_like(wp)
        flops = 2.0 * cin * cout * k * k * ho * ho * B
        t_f = timeit(lambda: ops.conv_fwd(x, wp, y, k=k, stride=s, pad=pad, Co=cout, tile=a.tile, split=a.split), a.iters)
        plans = ops.dgrad_plans(k, pad, s, hin, hin)
        packs = []
        for pl in plans:
            wpk = torch.empty(cin, len(pl["taps"]) * cout, device="cuda")
            ops.pack_weight_dgrad(wp, wpk, Co=cout, Co_pad=cout, Ci=cin, kd=1, kh=k, kw=k, tap_ids=pl["tap_ids"],
                                  src_packed=True)
            packs.append(wpk)

        def dgrad():
            for pl, wpk in zip(plans, packs):
                ops.conv_igemm(dy, wpk, dx, N=B, Hi=ho, Wi=ho, Ci=cout, Hg=pl["Hg"], Wg=pl["Wg"], in_stride=1,
                               taps=pl["taps"], Ho=hin, Wo=hin, Co=cin, ldo=cin, out_stride=s, out_oy=pl["py"],
                               out_ox=pl["px"], tile=a.tile, split=a.split)
        t_d = timeit(dgrad, a.iters)
        t_w = timeit(lambda: ops.conv_wgrad(x, dy, dwp, ws, k=k, stride=s, pad=pad, Co=cout, split=a.split), a.iters)
        name = "%dx%d s%d %4d->%-4d @%d" % (k, k, s, cin, cout, hin)
        err = ""
        if a.check:
            nb = 2
            ref = torch.nn.functional.conv2d(x[:nb].permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), stride=s, padding=pad)
            got = y[:nb].permute(0, 3, 1, 2).double().cpu()
            e = (got - ref)
            err = "  max|err| %.2e rms %.2e (out rms %.2f)" % (e.abs().max(), e.pow(2).mean().sqrt(), ref.pow(2).mean().sqrt())
            if B <= 8:          # weight gradient against fp64 autograd on the host
                xd = x.permute(0, 3, 1, 2).double().cpu()
                wd = w.double().cpu().requires_grad_(True)
                torch.nn.functional.conv2d(xd, wd, stride=s, padding=pad).backward(dy.permute(0, 3, 1, 2).double().cpu())
                dw = torch.empty_like(w)
                ops.unpack_weight(dwp, dw)
                ew = dw.double().cpu() - wd.grad
                err += "  wgrad rms err %.2e (rms %.2f)" % (ew.pow(2).mean().sqrt(), wd.grad.pow(2).mean().sqrt())
        print("%-28s %9.1f %9.1f %9.1f" % (name, flops / t_f / 1e9, flops / t_d / 1e9, flops / t_w / 1e9) + err)


if __name__ == "__main__":
    main()
