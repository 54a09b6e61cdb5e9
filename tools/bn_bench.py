#!/usr/bin/env python
"""Achieved HBM rate of the BatchNorm apply / backward-apply kernels alone (no side-stream neighbour), on the activation
shapes of the batch-64 416x416 training step.  usage: python tools/bn_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops

SHAPES = [(64 * 208 * 208, 64), (64 * 104 * 104, 128), (64 * 52 * 52, 256), (64 * 26 * 26, 512), (64 * 13 * 13, 1024),
          (64 * 208 * 208, 32), (64 * 52 * 52, 128)]


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def main():
    tot_a = tot_b = 0.0
    for M, C in SHAPES:
        x = torch.randn(M, C, device="cuda")
        dy = torch.randn(M, C, device="cuda")
        y = torch.empty_like(x)
        sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
        mu, iv = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5
        sums2 = torch.randn(2 * C, device="cuda", dtype=torch.float64)
        ta = timeit(lambda: ops.bn_apply_leaky(x, sc, sh, None, y, M, C))
        tb = timeit(lambda: ops.bn_bwd_apply(x, dy, sc, sh, mu, iv, sums2, float(M), M, C, y))
        gb = M * C * 4 / 1e9
        tot_a += ta; tot_b += tb
        print("M %8d C %4d | apply %.1f us %.2f TB/s | bwd_apply %.1f us %.2f TB/s" % (M, C, ta * 1e3, 2 * gb / ta, tb * 1e3, 3 * gb / tb))
    print("sum apply %.3f ms, bwd_apply %.3f ms" % (tot_a, tot_b))


if __name__ == "__main__":
    main()
