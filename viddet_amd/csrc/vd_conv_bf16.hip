// vd_conv_bf16.hip — bf16 storage / bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate) variant of the
// tap-list implicit-GEMM convolution, for the inference configs of BASELINE.json that ask for bf16
// (configs[1]; the reference itself is fp32-only, so this path is judged against the fp32 oracle with a
// stated bf16 tolerance).
//
// Same GEMM view and the same 16-byte-chunk loader as vd_conv.hip: a K-step is one 128-byte run of one
// tap per pixel = 64 bf16 channels, LDS rows are 128 B + 16 B pad (conflict-free ds_read_b128), and the
// MFMA operand map is the bf16 one (guide §3): lane l holds A[row l&31][k = 8*(l>>5) .. +7] = exactly the
// 16 bytes it reads at byte offset 32*kc + 16*(l>>5) of its row, so ONE ds_read_b128 feeds ONE MFMA of K=16.
// Activations NHWC bf16 with the channel count padded to a multiple of 64; weights packed [Co_pad][T*Ci] bf16;
// epilogue (BN-eval fold / bias, LeakyReLU, residual) in fp32, output bf16 (layers) or fp32 (prediction heads,
// so the decode / NMS kernels are shared with the fp32 path).
#include "vd_conv_igemm_bf16.h"

int vd_igemm_bf16_sk_dispatch(const vd_conv_desc& d, int tile, hipStream_t s, bool query_only);     // vd_conv_bf16_sk.hip
bool vd_conv_c32_bf16_ok(const vd_conv_desc& d, bool out_f32);                                       // vd_conv_c32_bf16.hip
void vd_conv_c32_bf16_launch(const vd_conv_desc& d, hipStream_t s);
int64_t vd_conv_c32_bf16_tiles(const vd_conv_desc& d);

namespace {

template <bool OUT_F32>
void dispatch_b(const vd_conv_desc& d, hipStream_t s) {
    int tile = d.tile;
    if (d.Ci == 32) {                                          // two taps per K-step; these layers have Co = 64
        // 16: the first-stage patch kernel (vd_conv_c32_bf16.hip: 2-D patches staged once, weights in registers) where it applies
        if (tile == 16 && vd_conv_c32_bf16_ok(d, OUT_F32)) return vd_conv_c32_bf16_launch(d, s);
        if (tile == 11) return launch_b<4, 1, 2, 2, OUT_F32, true>(d, s);
        if (tile == 13) return launch_b<2, 2, 2, 1, OUT_F32, true>(d, s);
        if (tile == 14) return launch_b<2, 2, 1, 1, OUT_F32, true>(d, s);     // 64 x 64, four waves (short-K 1x1: see tile 14 below)
        return launch_b<4, 2, 2, 1, OUT_F32, true>(d, s);
    }
    if (tile <= 0 || tile > 15) tile = d.Co <= 32 ? 12 : (d.Co <= 64 ? 10 : 2);
    // VD_CONV_STREAMK / VD_CONV_SPLITK: the persistent stream-K grid, or the split-K grid, of the same tile
    // (vd_conv_bf16_sk.hip) where one of them applies
    if (!OUT_F32 && (d.flags & (VD_CONV_STREAMK | VD_CONV_SPLITK)) && vd_igemm_bf16_sk_dispatch(d, tile, s, false) == 0) return;
    switch (tile) {
        // 14, 15: small four-wave tiles (37 / 28 KB of LDS: four or five workgroups per CU) for the short-K 1x1 layers, which
        // are HBM-bound and ran at half the HBM rate on the large tiles: one workgroup's load latency and output stores
        // hide behind its neighbours'
        case 14: return launch_b<2, 2, 1, 1, OUT_F32>(d, s);  //  64 x 64, 4 waves of 32x32
        case 15: return launch_b<4, 1, 1, 1, OUT_F32>(d, s);  // 128 x 32, 4 waves of 32x32
        case 10: return launch_b<4, 2, 2, 1, OUT_F32>(d, s);  // 256 x 64, 8 waves of 64x32
        case 11: return launch_b<4, 1, 2, 2, OUT_F32>(d, s);  // 256 x 64, 4 waves of 64x64
        case 12: return launch_b<4, 1, 2, 1, OUT_F32>(d, s);  // 256 x 32, 4 waves of 64x32
        case 13: return launch_b<2, 2, 2, 1, OUT_F32>(d, s);  // 128 x 64, 4 waves of 64x32 (two blocks / CU)
        case 8: return launch_b<2, 4, 4, 2, OUT_F32>(d, s);   // 256 x 256, 8 waves of 128x64 (1 block / CU)
        case 9: return launch_b<4, 2, 2, 4, OUT_F32>(d, s);   // 256 x 256, 8 waves of 64x128 (1 block / CU)
        case 6: return launch_b<4, 2, 2, 2, OUT_F32>(d, s);   // 256 x 128, 8 waves of 64x64 (1 block / CU)
        case 7: return launch_b<2, 4, 2, 2, OUT_F32>(d, s);   // 128 x 256, 8 waves of 64x64 (1 block / CU)
        case 1: return launch_b<2, 2, 2, 2, OUT_F32>(d, s);   // 128 x 128, 4 waves of 64x64
        case 2: return launch_b<4, 2, 1, 2, OUT_F32>(d, s);   // 128 x 128, 8 waves of 32x64
        case 3: return launch_b<2, 4, 2, 1, OUT_F32>(d, s);   // 128 x 128, 8 waves of 64x32
        case 4: return launch_b<2, 2, 1, 2, OUT_F32>(d, s);   //  64 x 128, 4 waves
        default: return launch_b<2, 4, 1, 1, OUT_F32>(d, s);  //  64 x 128, 8 waves
    }
}

__global__ void k_pack_bf16(const float* __restrict__ src, __bf16* __restrict__ dst, int Co, int Co_pad, int Ci,
                            int Ci_pad, int T) {
    // src fp32 fwd-packed [>=Co][T*Ci] -> dst bf16 [Co_pad][T*Ci_pad], zero padded rows / channels
    const int64_t total = (int64_t)Co_pad * T * Ci_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci_pad);
        const int64_t r = i / Ci_pad;
        const int t = (int)(r % T);
        const int co = (int)(r / T);
        const float v = (co < Co && ci < Ci) ? src[((int64_t)co * T + t) * Ci + ci] : 0.f;
        dst[i] = (__bf16)v;
    }
}

// no padding anywhere: a plain fp32 -> bf16 conversion, 8 elements (32 B in, 16 B out) per thread
__global__ void k_cvt_bf16(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n8, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (__bf16)a[e]; o[4 + e] = (__bf16)b[e]; }
        reinterpret_cast<bf16x8*>(dst)[i] = o;
    }
    const int64_t t = n8 * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // tail (n not a multiple of 8)
    if (t < n) dst[t] = (__bf16)src[t];
}

__global__ void k_stem_im2col_bf16(const float* __restrict__ in, __bf16* __restrict__ col, int N, int H, int W, int nchw) {
    // col [N,H,W,64] bf16: entries (ky*3+kx)*3+c for k < 27, zero above
    const int64_t total = (int64_t)N * H * W * 8;          // one 16-byte chunk (8 bf16) per thread
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 7);
        const int64_t pix = i >> 3;
        const int x = (int)(pix % W);
        const int64_t t = pix / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = q * 8 + e;
            float val = 0.f;
            if (k < 27) {
                const int c = k % 3, tap = k / 3;
                const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    val = nchw ? in[(((int64_t)n * 3 + c) * H + iy) * W + ix] : in[(((int64_t)n * H + iy) * W + ix) * 3 + c];
            }
            v[e] = (__bf16)val;
        }
        reinterpret_cast<bf16x8*>(col)[i] = v;
    }
}

}  // namespace

extern "C" {

// rows of the GEMM one workgroup covers, per tile variant (dispatch_b below)
static int bf16_tile_bm(const vd_conv_desc& d) {
    int tile = d.tile;
    if (d.Ci == 32) return tile == 14 ? 64 : (tile == 13 ? 128 : 256);
    if (tile <= 0 || tile > 15) tile = d.Co <= 32 ? 12 : (d.Co <= 64 ? 10 : 2);
    switch (tile) {
        case 10: case 11: case 12: case 8: case 9: case 6: return 256;
        case 13: case 7: case 1: case 2: case 3: case 15: return 128;
        default: return 64;
    }
}

int vd_conv_igemm_bf16_streamk(const vd_conv_desc* d, int out_f32) {
    if (!d || out_f32 || !(d->flags & (VD_CONV_STREAMK | VD_CONV_SPLITK)) || d->Ci == 32) return 0;
    int tile = d->tile;
    if (tile <= 0 || tile > 15) tile = d->Co <= 32 ? 12 : (d->Co <= 64 ? 10 : 2);
    const int r = vd_igemm_bf16_sk_dispatch(*d, tile, nullptr, true);
    return r == 0 ? 1 : (r == 2 ? 2 : 0);
}

int vd_conv_igemm_bf16_mtiles(const vd_conv_desc* d) {
    if (!d) return 0;
    if (d->Ci == 32 && d->tile == 16 && vd_conv_c32_bf16_ok(*d, false)) return (int)vd_conv_c32_bf16_tiles(*d);      // one row per patch
    return (int)vd_cdiv((int64_t)d->N * d->Hg * d->Wg, bf16_tile_bm(*d));
}

int vd_conv_igemm_bf16(const vd_conv_desc* d, int out_f32, void* stream) {
    VD_REQUIRE(d && d->in && d->wp && d->out, "vd_conv_igemm_bf16: null pointer");
    VD_REQUIRE(d->Ci == 32 || (d->Ci > 0 && d->Ci % 64 == 0), "vd_conv_igemm_bf16: Ci=%d must be 32 or a positive multiple of 64", d->Ci);
    VD_REQUIRE(d->T >= 1 && d->T <= VD_MAX_TAPS && d->Kfr >= 1 && d->N % d->Kfr == 0, "vd_conv_igemm_bf16: bad taps");
    VD_REQUIRE(d->out_stride >= 1 && d->out_oy >= 0 && d->out_ox >= 0 && (d->Hg - 1) * d->out_stride + d->out_oy < d->Ho &&
               (d->Wg - 1) * d->out_stride + d->out_ox < d->Wo, "vd_conv_igemm_bf16: output geometry");
    VD_REQUIRE(d->ldo >= d->Co && (int64_t)d->N * d->Hi * d->Wi < (1ll << 31) && (int64_t)d->N * d->Hg * d->Wg < (1ll << 31),
               "vd_conv_igemm_bf16: bad sizes");
    VD_REQUIRE(!(d->flags & VD_EPI_RESIDUAL) || (d->residual && d->ldr >= d->Co), "vd_conv_igemm_bf16: residual missing");
    VD_REQUIRE(!d->in_scale, "vd_conv_igemm_bf16: the in-load transform is an fp32-path feature");
    VD_REQUIRE(!(d->flags & (VD_CONV_STREAMK | VD_CONV_SPLITK)) || !d->sk_ws || ((uintptr_t)d->sk_ws % 16 == 0 && d->sk_ws_bytes >= VD_SK_HEADER_BYTES),
               "vd_conv_igemm_bf16: stream-K / split-K workspace misaligned or smaller than its header");
    if (d->bs_part) {
        const int t_ = (d->Ci == 32) ? 0 : ((d->tile <= 0 || d->tile > 15) ? (d->Co <= 32 ? 12 : (d->Co <= 64 ? 10 : 2)) : d->tile);
        VD_REQUIRE(!out_f32 && t_ != 8 && t_ != 9, "vd_conv_igemm_bf16: fused backward reductions need a bf16 output and a tile other than 8 / 9");
        VD_REQUIRE(d->bs_z && d->bs_scale && d->bs_shift && d->bs_mean && d->bs_invstd, "vd_conv_igemm_bf16: bs_* pointers missing");
        VD_REQUIRE(d->ldo % 4 == 0 && d->Co % 4 == 0 && (uintptr_t)d->out % 16 == 0 && (uintptr_t)d->bs_z % 8 == 0 &&
                   (((uintptr_t)d->bs_scale | (uintptr_t)d->bs_shift | (uintptr_t)d->bs_mean | (uintptr_t)d->bs_invstd) % 16 == 0) &&
                   (!(d->flags & VD_EPI_RESIDUAL) || (d->ldr % 4 == 0 && (uintptr_t)d->residual % 8 == 0)) &&
                   (!(d->flags & VD_EPI_AFFINE) || (((uintptr_t)d->scale | (uintptr_t)d->shift) % 16 == 0)),
                   "vd_conv_igemm_bf16: fused backward reductions need the vector epilogue (aligned rows, Co % 4 == 0)");
    }
    static const int probe = getenv("VD_IGEMM_PROBE") ? atoi(getenv("VD_IGEMM_PROBE")) : 0;
    vd_conv_desc dd = *d;
    dd.flags |= (probe & 7) << 8;
    if (out_f32) dispatch_b<true>(dd, (hipStream_t)stream);
    else dispatch_b<false>(dd, (hipStream_t)stream);
    VD_CHECK_LAUNCH("vd_conv_igemm_bf16");
    return VD_OK;
}

#if VD_STAMP
int vd_debug_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

int vd_pack_weight_bf16(const float* wp_f32, void* wp_bf16, int Co, int Co_pad, int Ci, int Ci_pad, int T, void* stream) {
    VD_REQUIRE(wp_f32 && wp_bf16 && Co > 0 && Co_pad >= Co && Ci > 0 && Ci_pad >= Ci && T > 0, "vd_pack_weight_bf16: bad args");
    const int64_t total = (int64_t)Co_pad * T * Ci_pad;
    if (Co == Co_pad && Ci == Ci_pad && (uintptr_t)wp_f32 % 16 == 0 && (uintptr_t)wp_bf16 % 16 == 0) {
        const int64_t n8 = total / 8;
        const int nbv = (int)(vd_cdiv(n8 + 8, 256) < 4096 ? vd_cdiv(n8 + 8, 256) : 4096);
        hipLaunchKernelGGL(k_cvt_bf16, dim3(nbv), dim3(256), 0, (hipStream_t)stream, wp_f32, (__bf16*)wp_bf16, n8, total);
        VD_CHECK_LAUNCH("vd_pack_weight_bf16/convert");
        return VD_OK;
    }
    const int nb = (int)(vd_cdiv(total, 256) < 4096 ? vd_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(k_pack_bf16, dim3(nb), dim3(256), 0, (hipStream_t)stream, wp_f32, (__bf16*)wp_bf16, Co, Co_pad, Ci, Ci_pad, T);
    VD_CHECK_LAUNCH("vd_pack_weight_bf16");
    return VD_OK;
}

int vd_stem_im2col_bf16(const float* in, void* col, int N, int H, int W, int nchw, void* stream) {
    VD_REQUIRE(in && col && N > 0 && H > 0 && W > 0, "vd_stem_im2col_bf16: bad args");
    const int64_t total = (int64_t)N * H * W * 8;
    const int nb = (int)(vd_cdiv(total, 256) < 8192 ? vd_cdiv(total, 256) : 8192);
    hipLaunchKernelGGL(k_stem_im2col_bf16, dim3(nb), dim3(256), 0, (hipStream_t)stream, in, (__bf16*)col, N, H, W, nchw);
    VD_CHECK_LAUNCH("vd_stem_im2col_bf16");
    return VD_OK;
}

}  // extern "C"
