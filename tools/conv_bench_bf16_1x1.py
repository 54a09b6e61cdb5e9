#!/usr/bin/env python
"""The 1x1 layers of the bf16-storage training step (batch 64, 416x416) on every tile of k_conv_igemm_bf16: TFLOP/s and the
achieved rate on the algorithmic bytes (input + output + weights, 2 B each) - these layers are HBM-bound."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops, lib as L

SHAPES = [(64, 32, 208), (128, 64, 104), (256, 128, 52), (512, 256, 26), (1024, 512, 13), (768, 256, 26), (384, 128, 52),
          (32, 64, 208), (128, 256, 52)]          # the last two: data-gradient shapes of the first (Ci = 32: two-tap layout) / third
B = int(os.environ.get("B", "64"))
tiles = [int(t) for t in os.environ.get("TILES", "1,2,3,4,5,6,7,10,11,12,13,14,15").split(",")]
lib = L.load()
print("%-22s " % "layer" + " ".join("%6s" % ("t%d" % t) for t in tiles) + "   (GB/s on algorithmic bytes)")
for cin, cout, hin in SHAPES:
    x = torch.randn(B, hin, hin, cin, device="cuda").to(torch.bfloat16)
    wb = (torch.randn(cout, cin, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, hin, hin, cout, dtype=torch.bfloat16, device="cuda")
    d = L.ConvDesc()
    d.in_, d.wp, d.out = x.data_ptr(), wb.data_ptr(), y.data_ptr()
    d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = B, hin, hin, cin, hin, hin, 1
    ops._set_taps(d, ops.fwd_taps(1, 0))
    d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.ldo, d.ldr = 1, hin, hin, cout, 1, cout, cout
    d.flags, d.slope = 0, 0.1
    nbytes = 2.0 * (B * hin * hin * (cin + cout) + cin * cout)
    res = []
    for t in tiles:
        if cin == 32 and t not in (10, 11, 13):
            res.append(float("nan"))
            continue
        if (cout <= 32 and t not in (12, 15)) or (cout > 32 and t in (12, 15)):
            res.append(float("nan"))
            continue
        d.tile = t
        st = L.stream_ptr()
        ok = True
        for _ in range(2):
            ok = ok and lib.vd_conv_igemm_bf16(C.byref(d), 0, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.vd_conv_igemm_bf16(C.byref(d), 0, st)
        e1.record()
        torch.cuda.synchronize()
        res.append(nbytes / (e0.elapsed_time(e1) / 10) / 1e6 if ok else float("nan"))
    print("%-22s " % ("1x1 %4d->%-4d @%d" % (cin, cout, hin)) + " ".join("%6.0f" % r for r in res))
