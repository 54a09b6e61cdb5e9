#!/usr/bin/env python
"""Phase timeline of one k_conv_igemm / k_conv_igemm_bf16 workgroup (developer aid).

Build the stamped library first (s_memtime stamps compiled in with -DVD_STAMP=1):  tools/build_stamp_lib.sh
then:  VD_LIB=build_dbg/libviddet_stamp.so python tools/stamp_conv.py [fp32|bf16]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import lib as L, ops

CASES = [  # n, ci, h, co, k, stride, tiles
    (32, 32, 304, 64, 3, 1, (10, 11)), (32, 64, 304, 32, 1, 1, (10, 12)), (32, 32, 608, 64, 3, 2, (10, 11)),
    (32, 64, 152, 128, 3, 1, (2, 3)), (32, 128, 76, 256, 3, 1, (3, 8)), (32, 256, 38, 512, 3, 1, (7, 8)),
]


F32_CASES = [  # n, ci, h, co, k, stride, (tile, split) list   (batch 64 @ 416)
    (64, 32, 208, 64, 3, 1, ((3, True), (4, True), (6, False))), (64, 64, 208, 32, 1, 1, ((7, False),)),
    (64, 64, 104, 128, 3, 1, ((1, True), (2, True))), (64, 128, 52, 256, 3, 1, ((1, True), (5, True))),
    (64, 256, 52, 128, 1, 1, ((1, True), (2, True))), (64, 256, 26, 512, 3, 1, ((1, True),)),
]


def main_f32():
    lib = L.load()
    lib.vd_debug_stamps_f32.argtypes = [C.c_void_p]
    lib.vd_debug_stamps_f32.restype = C.c_int
    for n, ci, h, co, k, s, variants in F32_CASES:
        p = k // 2
        ho = (h + 2 * p - k) // s + 1
        x = torch.randn(n, h, h, ci, device="cuda")
        w = torch.randn(co, ci, k, k, device="cuda") * 0.05
        wp = torch.empty(co, k * k * ci, device="cuda")
        ops.pack_weight_fwd(w, wp, co)
        y = torch.empty(n, ho, ho, co, device="cuda")
        for tile, split in variants:
            fn = lambda: ops.conv_fwd(x, wp, y, k=k, stride=s, pad=p, Co=co, tile=tile, split=split)
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            st = (C.c_ulonglong * 16)()
            assert lib.vd_debug_stamps_f32(st) == 0
            t = [int(v) for v in st]
            dt = [t[i + 1] - t[i] for i in range(6)]
            fl = 2.0 * n * ho * ho * co * k * k * ci / ms / 1e9
            print("ci%4d co%4d k%d s%d %3d^2 tile %d %s: %.3f ms %4.0f TF | ticks rowinfo %d  gload-issue %d  first-tile %d  "
                  "k-loop %d  epilogue %d  stats %d  total %d" % (ci, co, k, s, h, tile, "split" if split else "native", ms, fl,
                                                                 dt[0], dt[1], dt[2], dt[3], dt[4], dt[5], t[6] - t[0]), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "fp32":
        return main_f32()
    lib = L.load()
    lib.vd_debug_stamps.argtypes = [C.c_void_p]
    lib.vd_debug_stamps.restype = C.c_int
    for n, ci, h, co, k, s, tiles in CASES:
        p = k // 2
        ho = (h + 2 * p - k) // s + 1
        x = torch.randn(n, h, h, ci, device="cuda").to(torch.bfloat16)
        wb = (torch.randn(co, k * k * ci, device="cuda") * 0.05).to(torch.bfloat16)
        out = torch.empty(n, ho, ho, co, dtype=torch.bfloat16, device="cuda")
        res = torch.randn(n, ho, ho, co, device="cuda").to(torch.bfloat16)
        sc, sh = torch.ones(co, device="cuda"), torch.zeros(co, device="cuda")
        for tile in tiles:
            d = L.ConvDesc()
            d.in_, d.wp, d.out = x.data_ptr(), wb.data_ptr(), out.data_ptr()
            d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = n, h, h, ci, ho, ho, s
            ops._set_taps(d, ops.fwd_taps(k, p))
            d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.ldo, d.ldr, d.tile = 1, ho, ho, co, 1, co, co, tile
            d.scale, d.shift, d.residual = sc.data_ptr(), sh.data_ptr(), res.data_ptr()
            d.flags, d.slope = 1 | 2 | 4, 0.1
            for _ in range(2):
                L.check(lib.vd_conv_igemm_bf16(C.byref(d), 0, L.stream_ptr()), "conv")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib.vd_conv_igemm_bf16(C.byref(d), 0, L.stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            st = (C.c_ulonglong * 16)()
            assert lib.vd_debug_stamps(st) == 0
            t = [int(v) for v in st]
            dt = [t[i + 1] - t[i] for i in range(5)]
            fl = 2.0 * n * ho * ho * co * k * k * ci / ms / 1e9
            print("ci%4d co%4d k%d s%d %3d^2 tile %2d: %.3f ms %4.0f TF | ticks rowinfo %d  gload-issue %d  first-tile %d  "
                  "k-loop %d  epilogue %d  total %d" % (ci, co, k, s, h, tile, ms, fl, dt[0], dt[1], dt[2], dt[3], dt[4],
                                                         t[5] - t[0]), flush=True)


if __name__ == "__main__":
    main()
