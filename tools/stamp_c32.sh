#!/bin/bash
# developer tool: phase timeline of k_conv3x3_c32_bf16 (vd_conv_c32_bf16.hip) - builds the kernel with -DVD_C32_STAMP=1 into
# build_dbg/libviddet_c32stamp.so and prints the cycle counts between the phase boundaries of one steady-state tile:
# tile start -> barrier -> LDS stores + barrier -> requests issued -> 18 MFMA steps -> epilogue.   usage (GPU box): bash tools/stamp_c32.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p build_dbg
F="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -fno-slp-vectorize -Iinclude"
/opt/rocm/bin/hipcc $F -DVD_C32_STAMP=1 -c viddet_amd/csrc/vd_conv_c32_bf16.hip -o build_dbg/vd_conv_c32_stamp.o
(cd viddet_amd/csrc && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_dbg/libviddet_c32stamp.so vd_conv.o vd_conv_sk.o vd_conv_par.o \
    vd_wgrad_halo.o vd_conv_bf16.o vd_conv_bf16_sk.o ../../build_dbg/vd_conv_c32_stamp.o vd_stem.o vd_bn.o vd_pointwise.o vd_yolo.o vd_api.o)
VD_LIB=build_dbg/libviddet_c32stamp.so python - <<'PY'
import ctypes as C
import numpy as np, torch
from viddet_amd import ops, lib as L
lib = L.load()
lib.vd_debug_c32_stamps.argtypes = [C.c_void_p]; lib.vd_debug_c32_stamps.restype = C.c_int
for (n, h, s) in ((32, 304, 1), (32, 608, 2)):
    ho = h // s
    x = torch.randn(n, h, h, 32, device="cuda").to(torch.bfloat16)
    wb = (torch.randn(64, 288, device="cuda") / 17).to(torch.bfloat16)
    res = torch.randn(n, ho, ho, 64, device="cuda").to(torch.bfloat16)
    out = torch.empty(n, ho, ho, 64, device="cuda", dtype=torch.bfloat16)
    sc, sh = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    geo = dict(N=n, Hi=h, Wi=h, Ci=32, Hg=ho, Wg=ho, in_stride=s, taps=ops.fwd_taps(3, 1), Ho=ho, Wo=ho, Co=64, ldo=64, tile=16)
    for _ in range(3):
        ops.conv_igemm_bf16(x, wb, out, scale=sc, shift=sh, residual=res, ldr=64, leaky=True, **geo)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_igemm_bf16(x, wb, out, scale=sc, shift=sh, residual=res, ldr=64, leaky=True, **geo)
    e1.record(); e1.synchronize()
    st = (C.c_ulonglong * 16)()
    assert lib.vd_debug_c32_stamps(st) == 0
    t = [int(v) for v in st[:6]]
    names = ["wait at the barrier", "LDS stores + barrier", "requests issued", "18 MFMA steps", "epilogue"]
    print("stride %d, %dx%d: %.3f ms per launch; one tile (cycle counter ticks):" % (s, h, h, e0.elapsed_time(e1) / 10),
          ", ".join("%s %d" % (nm, t[i + 1] - t[i]) for i, nm in enumerate(names)), " total %d" % (t[5] - t[0]))
PY
