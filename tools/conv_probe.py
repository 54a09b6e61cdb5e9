#!/usr/bin/env python
"""Time ONE forward conv shape (events, median of several launches) — used with VD_IGEMM_PROBE timing probes.
usage: conv_probe.py cin cout k stride hin [tile] [split: 0 | 1 | f16x2 | f16x2nh | bf16] [batch]
VD_PROBE_STREAMK=1: the persistent stream-K form of the launch (fp16 split only)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops

cin, cout, k, s, hin = [int(v) for v in sys.argv[1:6]]
tile = int(sys.argv[6]) if len(sys.argv) > 6 else 0
split = (sys.argv[7] if sys.argv[7] in ('f16x2', 'f16x2nh', 'bf16') else bool(int(sys.argv[7]))) if len(sys.argv) > 7 else False
B = int(sys.argv[8]) if len(sys.argv) > 8 else 64
pad = k // 2
ho = (hin + 2 * pad - k) // s + 1
zero = os.environ.get("VD_PROBE_ZERO") == "1"        # all-zero operands: the DVFS check (same cycles, less power)
x = torch.zeros(B, hin, hin, cin, device="cuda") if zero else torch.randn(B, hin, hin, cin, device="cuda")
w = torch.zeros(cout, cin, k, k, device="cuda") if zero else torch.randn(cout, cin, k, k, device="cuda") * 0.05
wp = torch.empty(cout, k * k * cin, device="cuda")
ops.pack_weight_fwd(w, wp, cout)
y = torch.empty(B, ho, ho, cout, device="cuda")
skws = ops.streamk_workspace() if os.environ.get("VD_PROBE_STREAMK") == "1" else None
ts = []
for i in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_fwd(x, wp, y, k=k, stride=s, pad=pad, Co=cout, tile=tile, split=split, streamk_ws=skws)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = sorted(ts[2:])
t = ts[len(ts) // 2]
print("lib=%s tile=%d split=%s streamk=%s  %.3f ms  %.1f TFLOP/s" % (os.path.basename(os.environ.get("VD_LIB", "default")), tile, split, skws is not None, t,
                                                        2.0 * cin * cout * k * k * ho * ho * B / t / 1e9))
