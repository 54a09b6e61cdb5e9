#!/usr/bin/env python
"""Micro-benchmark of the bf16 forward conv kernel on yolo3_darknet53 layer shapes (B=32 @608)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops, lib as L

SHAPES = [(128, 256, 3, 1, 76), (256, 512, 3, 1, 38), (512, 1024, 3, 1, 19), (64, 128, 3, 1, 152),
          (1024, 512, 1, 1, 19), (512, 256, 1, 1, 38), (256, 128, 1, 1, 76), (256, 512, 3, 2, 76)]
B = int(os.environ.get("B", "32"))
tiles = [int(t) for t in os.environ.get("TILES", "1,2,3,4,5").split(",")]
lib = L.load()
print("%-26s " % "layer" + " ".join("tile%d" % t for t in tiles) + "   (TFLOP/s, bf16 peak 2500)")
for cin, cout, k, s, hin in SHAPES:
    pad = k // 2
    ho = (hin + 2 * pad - k) // s + 1
    x = torch.randn(B, hin, hin, cin, device="cuda").to(torch.bfloat16)
    wb = (torch.randn(cout, k * k * cin, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, ho, ho, cout, dtype=torch.bfloat16, device="cuda")
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    d = L.ConvDesc()
    d.in_, d.wp, d.out = x.data_ptr(), wb.data_ptr(), y.data_ptr()
    d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = B, hin, hin, cin, ho, ho, s
    ops._set_taps(d, ops.fwd_taps(k, pad))
    d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.ldo, d.ldr = 1, ho, ho, cout, 1, cout, cout
    d.scale, d.shift, d.flags, d.slope = sc.data_ptr(), sh.data_ptr(), 3, 0.1
    flops = 2.0 * cin * cout * k * k * ho * ho * B
    res = []
    for t in tiles:
        d.tile = t
        st = L.stream_ptr()
        for _ in range(2):
            lib.vd_conv_igemm_bf16(C.byref(d), 0, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.vd_conv_igemm_bf16(C.byref(d), 0, st)
        e1.record()
        torch.cuda.synchronize()
        res.append(flops / (e0.elapsed_time(e1) / 10) / 1e9)
    print("%-26s " % ("%dx%d s%d %4d->%-4d @%d" % (k, k, s, cin, cout, hin)) + " ".join("%5.0f" % r for r in res))
