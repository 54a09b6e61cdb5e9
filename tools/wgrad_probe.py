#!/usr/bin/env python
"""Time ONE 3x3 / stride-1 weight-gradient shape in one kernel form (events, median of several launches): the launch the
PMC passes of tools/pmc_wgrad.sh count.  usage: wgrad_probe.py cin cout hin [mode: f16x2 | f16x2h | bf16 | bf16h] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops

cin, cout, hin = [int(v) for v in sys.argv[1:4]]
mode = sys.argv[4] if len(sys.argv) > 4 else "f16x2h"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
x = torch.randn(B, hin, hin, cin, device="cuda")
dy = torch.randn(B, hin, hin, cout, device="cuda")
kw = {}
if mode.startswith("bf16"):
    x, dy, split = x.to(torch.bfloat16), dy.to(torch.bfloat16), ("halo" if mode == "bf16h" else False)
else:
    split, kw = mode, dict(amax_in=ops.amax(x), amax_dout=ops.amax(dy))
dwp = torch.empty(cout, 9 * cin, device="cuda")
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
ts = []
for i in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_wgrad(x, dy, dwp, ws, k=3, stride=1, pad=1, Co=cout, split=split, **kw)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = sorted(ts[2:])
t = ts[len(ts) // 2]
print("wgrad 3x3 %d->%d @%d batch %d mode %s  %.3f ms  %.1f TFLOP/s (slab reduction included)" % (
    cin, cout, hin, B, mode, t, 2.0 * cin * cout * 9 * hin * hin * B / t / 1e9))
