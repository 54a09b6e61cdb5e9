#!/usr/bin/env python
"""The two stem kernels alone (batch 64, 416x416): forward (fp32 output with fused statistics, as in training) and weight
gradient; algorithmic bytes = the 32-channel fp32 tensor written / read once plus the frames.  VD_STEM_MFMA=0: VALU forward.
usage: python tools/stem_bench.py [batch] [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops
from viddet_amd import lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
x = torch.randn(B, 3, S, S, device="cuda")
w = torch.randn(32, 3, 3, 3, device="cuda") * 0.2
wp = torch.zeros(32, 32, device="cuda")
ops.pack_weight_fwd(w, wp, 32)
out = torch.empty(B, S, S, 32, device="cuda")
nb = L.load().vd_stem_conv_blocks(B, S, S)
part = torch.empty(nb, 64, device="cuda")
dz = torch.randn(B, S, S, 32, device="cuda")
dwp = torch.empty(32, 32, device="cuda")
ws = torch.empty(max(16, L.load().vd_stem_wgrad_ws_bytes(B, S, S)), dtype=torch.uint8, device="cuda")


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


gb = (B * S * S * 32 * 4 + B * 3 * S * S * 4) / 1e9
tf = timeit(lambda: ops.stem_conv(x, wp, out, stats_part=part))
tw = timeit(lambda: ops.stem_wgrad(x, dz, dwp, ws))
print("stem forward %.3f ms (%.2f TB/s)   stem weight gradient %.3f ms (%.2f TB/s)   [%.2f GB each]" % (tf, gb / tf, tw, gb / tw, gb))
