#!/usr/bin/env python
"""The two first-stage launches of the bf16 detect path alone (vd_conv_c32_bf16.hip, tile 16; batch 32, 608 x 608 frames): the
stride-2 conv on the stem's stored map, the stride-1 conv with its residual, and the fused stem + stride-2 launch - timings,
and the process to put under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (tools/pmc_c32.sh).  usage: python tools/c32_probe.py [reps=12]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops, lib as L


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    lib = L.load()
    n, H = 32, 608
    g = torch.Generator(device="cuda").manual_seed(1)
    frames = torch.randn(n, 3, H, H, device="cuda", generator=g)
    wp0 = torch.zeros(32, 32, device="cuda")
    wp0[:, :27] = torch.randn(32, 27, device="cuda", generator=g) / 5
    sc0, sh0 = torch.rand(32, device="cuda", generator=g) + 0.5, torch.randn(32, device="cuda", generator=g)
    a = torch.empty(n, H, H, 32, dtype=torch.bfloat16, device="cuda")
    L.check(lib.vd_stem_conv(frames.data_ptr(), wp0.data_ptr(), a.data_ptr(), 32, n, H, H, sc0.data_ptr(), sh0.data_ptr(), 0.1, 1 | 2, 1,
                             None, L.stream_ptr()), "vd_stem_conv")
    wb = (torch.randn(64, 288, device="cuda", generator=g) / 17).to(torch.bfloat16)
    sc, sh = torch.rand(64, device="cuda", generator=g) + 0.5, torch.randn(64, device="cuda", generator=g)
    h2 = H // 2
    b = torch.empty(n, h2, h2, 64, dtype=torch.bfloat16, device="cuda")
    x1 = torch.randn(n, h2, h2, 32, device="cuda", generator=g).to(torch.bfloat16)
    res = torch.randn(n, h2, h2, 64, device="cuda", generator=g).to(torch.bfloat16)
    c = torch.empty_like(b)
    geo2 = dict(N=n, Hi=H, Wi=H, Ci=32, Hg=h2, Wg=h2, in_stride=2, taps=ops.fwd_taps(3, 1), Ho=h2, Wo=h2, Co=64, ldo=64, tile=16)
    geo1 = dict(N=n, Hi=h2, Wi=h2, Ci=32, Hg=h2, Wg=h2, in_stride=1, taps=ops.fwd_taps(3, 1), Ho=h2, Wo=h2, Co=64, ldo=64, tile=16)
    d2 = ops.conv_igemm_bf16(a, wb, b, scale=sc, shift=sh, leaky=True, **geo2)

    def t(f):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    t2 = t(lambda: ops.conv_igemm_bf16(a, wb, b, scale=sc, shift=sh, leaky=True, **geo2))
    t1 = t(lambda: ops.conv_igemm_bf16(x1, wb, c, scale=sc, shift=sh, residual=res, ldr=64, leaky=True, **geo1))
    ts = t(lambda: lib.vd_stem_conv(frames.data_ptr(), wp0.data_ptr(), a.data_ptr(), 32, n, H, H, sc0.data_ptr(), sh0.data_ptr(), 0.1, 1 | 2, 1,
                                    None, L.stream_ptr()))
    tf = t(lambda: lib.vd_stem_conv_c32_bf16(frames.data_ptr(), wp0.data_ptr(), sc0.data_ptr(), sh0.data_ptr(), 0.1, C.byref(d2), L.stream_ptr()))
    mb2 = (a.numel() + b.numel()) * 2 / 1e6
    mb1 = (x1.numel() + 2 * c.numel()) * 2 / 1e6
    mbf = (frames.numel() * 4 + b.numel() * 2) / 1e6
    print("stride 2 (608 -> 304): %.3f ms, %.0f MB algorithmic = %.2f TB/s" % (t2, mb2, mb2 / t2 / 1e3))
    print("stride 1 (304 x 304, + residual): %.3f ms, %.0f MB = %.2f TB/s" % (t1, mb1, mb1 / t1 / 1e3))
    print("stem alone (k_stem_fwd, bf16 out): %.3f ms;  stem + stride 2 in one launch: %.3f ms, %.0f MB = %.2f TB/s  (two launches: %.3f ms)" % (
        ts, tf, mbf, mbf / tf / 1e3, ts + t2))


if __name__ == "__main__":
    main()
