#!/usr/bin/env python
"""Run ONE conv shape a few times (for rocprofv3 --pmc passes).  usage: conv_one.py cin cout k stride hin [fwd|wgrad] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops

cin, cout, k, s, hin = [int(v) for v in sys.argv[1:6]]
which = sys.argv[6] if len(sys.argv) > 6 else "fwd"
B = int(sys.argv[7]) if len(sys.argv) > 7 else 64
pad = k // 2
ho = (hin + 2 * pad - k) // s + 1
x = torch.randn(B, hin, hin, cin, device="cuda")
w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
wp = torch.empty(cout, k * k * cin, device="cuda")
ops.pack_weight_fwd(w, wp, cout)
y = torch.empty(B, ho, ho, cout, device="cuda")
dy = torch.randn(B, ho, ho, cout, device="cuda")
dwp = torch.empty_like(wp)
ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for _ in range(5):
    if which == "fwd":
        ops.conv_fwd(x, wp, y, k=k, stride=s, pad=pad, Co=cout)
    else:
        ops.conv_wgrad(x, dy, dwp, ws, k=k, stride=s, pad=pad, Co=cout)
torch.cuda.synchronize()
