#!/usr/bin/env python
"""Stream-K form of k_conv_igemm_bf16 against the one-tile-per-workgroup form: bit-equality and launch times on the 608x608
detect shapes (batch 32).  usage: python tools/sk_bench_bf16.py [--batch 32] [--tiles 6,7,8,9,2]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import lib as L
from viddet_amd import ops
from sk_bench import timeit

SHAPES = [(512, 1024, 3, 19), (256, 512, 3, 38), (128, 256, 3, 76), (1024, 512, 1, 19), (512, 256, 1, 38), (256, 128, 1, 76)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tiles", default="6,7,8,9,2,1")
    a = ap.parse_args()
    B = a.batch
    skws = ops.streamk_workspace()
    bad = 0
    for cin, cout, k, hw in SHAPES:
        x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
        wt = torch.randn(cout, k * k * cin, device="cuda") * 0.05
        wb = torch.empty(cout, k * k * cin, device="cuda", dtype=torch.bfloat16)
        ops.pack_weight_bf16(wt, wb, Co=cout, Co_pad=cout, Ci=cin, Ci_pad=cin, T=k * k)
        res = torch.randn(B, hw, hw, cout, device="cuda").to(torch.bfloat16)
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        flops = 2.0 * cin * cout * k * k * hw * hw * B
        for tile in [int(t) for t in a.tiles.split(",")]:
            for nohalo in ((False, True) if k == 3 else (True,)):
                outs = []
                kw = dict(N=B, Hi=hw, Wi=hw, Ci=cin, Hg=hw, Wg=hw, in_stride=1, taps=ops.fwd_taps(k, k // 2), Ho=hw, Wo=hw, Co=cout, ldo=cout,
                          scale=sc, shift=sh, residual=res, ldr=cout, leaky=True, tile=tile, nohalo=nohalo)
                y0, y1 = torch.empty(B, hw, hw, cout, device="cuda", dtype=torch.bfloat16), torch.empty(B, hw, hw, cout, device="cuda", dtype=torch.bfloat16)
                ops.conv_igemm_bf16(x, wb, y0, **kw)
                d = ops.conv_igemm_bf16(x, wb, y1, streamk_ws=skws, **kw)
                used = bool(L.load().vd_conv_igemm_bf16_streamk(C.byref(d), 0))
                torch.cuda.synchronize()
                same = bool(torch.equal(y0, y1))
                t0 = timeit(lambda: ops.conv_igemm_bf16(x, wb, y0, **kw), 10)
                t1 = timeit(lambda: ops.conv_igemm_bf16(x, wb, y1, streamk_ws=skws, **kw), 10) if used else float("nan")
                bad += int(used and not same)
                print("%dx%d %4d->%-4d @%-2d tile %2d %-7s classic %.4f ms %6.0f TF | stream-K %s %.4f ms %6.0f TF x%.3f %s" % (
                    k, k, cin, cout, hw, tile, "generic" if nohalo else "halo", t0, flops / t0 / 1e9, "on " if used else "n/a", t1,
                    flops / t1 / 1e9, t0 / t1, "bit-identical" if same else "MISMATCH"), flush=True)
    print("mismatches: %d, polls given up: %d" % (bad, int(skws.view(torch.int32)[2047])))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
