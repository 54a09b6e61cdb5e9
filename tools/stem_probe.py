import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from viddet_amd import ops
from viddet_amd import lib as L
B, S = 64, 416
x = torch.randn(B, 3, S, S, device="cuda")
w = torch.randn(32, 3, 3, 3, device="cuda") * 0.2
wp = torch.zeros(32, 32, device="cuda"); ops.pack_weight_fwd(w, wp, 32)
out = torch.empty(B, S, S, 32, device="cuda")
outb = torch.empty(B, S, S, 32, device="cuda", dtype=torch.bfloat16)
nb = L.load().vd_stem_conv_blocks(B, S, S)
part = torch.empty(nb, 64, device="cuda")
sc = torch.ones(32, device="cuda"); sh = torch.zeros(32, device="cuda")
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("fp32 out + stats  %.3f ms" % timeit(lambda: ops.stem_conv(x, wp, out, stats_part=part)))
print("fp32 out, no stats %.3f ms" % timeit(lambda: ops.stem_conv(x, wp, out)))
print("fp32 out, affine+leaky %.3f ms" % timeit(lambda: ops.stem_conv(x, wp, out, scale=sc, shift=sh, leaky=True)))
print("bf16 out (MFMA), affine+leaky %.3f ms" % timeit(lambda: ops.stem_conv(x, wp, outb, scale=sc, shift=sh, leaky=True)))
print("bf16 out + stats %.3f ms" % timeit(lambda: ops.stem_conv(x, wp, outb, stats_part=part)))
y = torch.empty_like(out)
print("copy 1.4 GB fp32 (torch) %.3f ms" % timeit(lambda: y.copy_(out)))
