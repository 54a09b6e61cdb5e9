#!/usr/bin/env python
"""Fixed cost versus K-loop rate of a 3x3 forward launch: time the same output geometry at several input widths (the
K loop is 9*cin/32 steps, everything else - prologue, epilogue, launch ramp, tail round - is the same) and fit a line.
usage: conv_ksweep.py cout hin [tile] [batch]       env VD_PROBE_ZERO=1 for all-zero operands"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from viddet_amd import ops

cout, hin = int(sys.argv[1]), int(sys.argv[2])
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
zero = os.environ.get("VD_PROBE_ZERO") == "1"
k, s, pad = 3, 1, 1
for split in ("f16x2", "f16x2nh"):
    xs, ys = [], []
    for cin in (32, 64, 128, 256, 512):
        x = torch.zeros(B, hin, hin, cin, device="cuda") if zero else torch.randn(B, hin, hin, cin, device="cuda")
        w = torch.zeros(cout, cin, k, k, device="cuda") if zero else torch.randn(cout, cin, k, k, device="cuda") * 0.05
        wp = torch.empty(cout, k * k * cin, device="cuda")
        ops.pack_weight_fwd(w, wp, cout)
        y = torch.empty(B, hin, hin, cout, device="cuda")
        ax, aw = ops.amax(x), ops.amax(wp)
        ts = []
        for i in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_fwd(x, wp, y, k=k, stride=s, pad=pad, Co=cout, tile=tile, split=split, amax_in=ax, amax_w=aw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        t = sorted(ts[1:])[2]
        steps = 9 * cin // 32
        xs.append(steps)
        ys.append(t * 1e3)
        print("%-8s cout %d @%d cin %4d  K-steps %4d  %8.1f us  %6.1f TF" % (split, cout, hin, cin, steps, t * 1e3,
                                                                          2.0 * cin * cout * 9 * hin * hin * B / t / 1e9))
    a, b = np.polyfit(xs, ys, 1)
    fl_step = 2.0 * 32 * cout * hin * hin * B
    print("%-8s fit: %.1f us fixed + %.3f us per K-step  (loop alone = %.1f TF)  data=%s" % (split, b, a, fl_step / a / 1e6, "zero" if zero else "random"))
