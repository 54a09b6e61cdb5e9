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
#include "vd_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int ROW_B = 144;              // LDS row: 128 B of K + 16 B pad
constexpr int KCH = 64;                 // bf16 channels per K-step

__device__ __attribute__((aligned(64))) float g_zero_page_b[64];

struct RowInfoB {
    int64_t off;     // element offset (bf16) of (pixel of tap (0,0,0), channel 8*(tid&7))
    unsigned mask;
};

template <int WM, int WN, int TM, int TN, bool OUT_F32>
__global__ __launch_bounds__(WM * WN * 64) void k_conv_igemm_bf16(const vd_conv_desc p, const int64_t zd_in,
                                                                  const int64_t zd_w) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;
    constexpr int AP = BM / RPP, BP = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile vs loader");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;                       // [2][BM][ROW_B]
    unsigned char* Bs = smem_b + 2 * BM * ROW_B;      // [2][BN][ROW_B]
    const __bf16* in = reinterpret_cast<const __bf16*>(p.in);
    const __bf16* wp = reinterpret_cast<const __bf16*>(p.wp);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t M = (int64_t)p.N * p.Hg * p.Wg;
    const int ntile = (p.Co + BN - 1) / BN;
    const int lid = vd_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % ntile, tile_m = lid / ntile;
    const int Ktot = p.T * p.Ci;
    const int lrow = tid >> 3;
    const int lc8 = (tid & 7) * 8;            // bf16 element offset of this lane's 16-byte chunk

    RowInfoB ri[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int64_t m = (int64_t)tile_m * BM + lrow + RPP * i;
        ri[i].off = 0;
        ri[i].mask = 0u;
        if (m < M) {
            const unsigned mu = (unsigned)m;
            const unsigned t = mu / (unsigned)p.Wg;
            const int gx = (int)(mu - t * (unsigned)p.Wg);
            const unsigned n_ = t / (unsigned)p.Hg;
            const int gy = (int)(t - n_ * (unsigned)p.Hg);
            const int n = (int)n_;
            const int iy0 = gy * p.in_stride, ix0 = gx * p.in_stride;
            const int fz0 = n % p.Kfr;
            ri[i].off = (int64_t)((n * p.Hi + iy0) * p.Wi + ix0) * p.Ci + lc8;
            unsigned mk = 0u;
            for (int t2 = 0; t2 < p.T; ++t2) {
                const bool ok = (unsigned)(iy0 + p.dy[t2]) < (unsigned)p.Hi && (unsigned)(ix0 + p.dx[t2]) < (unsigned)p.Wi &&
                                (unsigned)(fz0 + p.dz[t2]) < (unsigned)p.Kfr;
                mk |= ok ? (1u << t2) : 0u;
            }
            ri[i].mask = mk;
        }
    }
    int64_t boff[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = tile_n * BN + lrow + RPP * i;
        boff[i] = (n < p.Co) ? (int64_t)n * Ktot + lc8 : (int64_t)-1;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // PD register sets of gathered tiles + a branch-free steady-state loop: see k_conv_igemm (vd_conv.hip)
    constexpr int PD = (WM * WN == 4 && TM * TN == 4) ? 2 : 3;
    f32x4 ra[PD][AP], rb[PD][BP];
    int t_tap = 0, c0 = 0;

    auto tap_off = [&](int t) -> int64_t {
        return (int64_t)((p.dz[t] * p.Hi + p.dy[t]) * p.Wi + p.dx[t]) * p.Ci;
    };
    int64_t tap_soff = tap_off(0);
    auto gload = [&](f32x4 (&ra)[AP], f32x4 (&rb)[BP]) {
        const int64_t soff = tap_soff + c0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool ok = (ri[i].mask >> t_tap) & 1u;
            const int64_t sel = ok ? ri[i].off + soff : zd_in;
            ra[i] = *reinterpret_cast<const f32x4*>(in + sel);
        }
        const int64_t koff = (int64_t)t_tap * p.Ci + c0;
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int64_t sel = boff[i] >= 0 ? boff[i] + koff : zd_w;
            rb[i] = *reinterpret_cast<const f32x4*>(wp + sel);
        }
        // taps innermost: the T taps of one channel chunk touch (almost) the same pixels, shifted (see vd_conv.hip)
        ++t_tap;
        if (t_tap >= p.T) {
            t_tap = 0;
            c0 += KCH;
        }
        tap_soff = tap_off(t_tap);
    };
    auto lstore = [&](int buf, const f32x4 (&ra)[AP], const f32x4 (&rb)[BP]) {
        unsigned char* a = As + buf * BM * ROW_B;
        unsigned char* b = Bs + buf * BN * ROW_B;
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(a + (lrow + RPP * i) * ROW_B + (tid & 7) * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(b + (lrow + RPP * i) * ROW_B + (tid & 7) * 16) = rb[i];
    };
    auto compute = [&](int buf) {
        const unsigned char* a = As + buf * BM * ROW_B + (wm * TM * 32 + (lane & 31)) * ROW_B + 16 * (lane >> 5);
        const unsigned char* b = Bs + buf * BN * ROW_B + (wn * TN * 32 + (lane & 31)) * ROW_B + 16 * (lane >> 5);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(a + mi * 32 * ROW_B + kc * 32);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8*>(b + ni * 32 * ROW_B + kc * 32);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    const int nks = p.T * (p.Ci / KCH);
    gload(ra[0], rb[0]);
    lstore(0, ra[0], rb[0]);
    __syncthreads();
    constexpr int UN = (PD % 2 == 0) ? PD : 2 * PD;
#pragma unroll
    for (int d = 1; d < PD; ++d)
        if (d < nks) gload(ra[d], rb[d]);
    int ks = 0;
    if (nks >= PD) {
        for (; ks + UN + PD <= nks; ks += UN) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                gload(ra[u % PD], rb[u % PD]);
                compute(u & 1);
                lstore((u + 1) & 1, ra[(u + 1) % PD], rb[(u + 1) % PD]);
                __syncthreads();
            }
        }
    }
    for (; ks < nks; ks += UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (ks + u < nks) {
                if (ks + u + PD < nks) gload(ra[u % PD], rb[u % PD]);
                compute(u & 1);
                if (ks + u + 1 < nks) lstore((u + 1) & 1, ra[(u + 1) % PD], rb[(u + 1) % PD]);
                __syncthreads();
            }
        }
    }

    // ---- epilogue (fp32 math; direct geometry only: this path serves forward convs).  Per wave, one 32x32
    // accumulator tile at a time is transposed through a private LDS patch so that each lane owns 4 consecutive
    // columns of 4 rows: scale/shift/LeakyReLU/residual on 4-vectors and 8-byte (bf16 x4) or 16-byte (fp32 heads)
    // stores instead of 2 bytes per lane.
    const __bf16* res = reinterpret_cast<const __bf16*>(p.residual);
    constexpr int SLD = 36;
    float* stg = reinterpret_cast<float*>(smem_b) + wave * (32 * SLD);
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    const bool vec_ok = (p.ldo % 4 == 0) && ((uintptr_t)p.out % 16 == 0) &&
                        (!(p.flags & VD_EPI_RESIDUAL) || ((p.ldr % 4 == 0) && ((uintptr_t)p.residual % 8 == 0)));
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int col = tile_n * BN + wn * TN * 32 + ni * 32 + ec4;
        const int nvalid = p.Co - col;
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.flags & VD_EPI_AFFINE) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nvalid) {
                    if (p.scale) sc[e] = p.scale[col + e];
                    if (p.shift) sh[e] = p.shift[col + e];
                }
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + (lane & 31)] = acc[mi][ni][r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = erow + 8 * i;
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * SLD + ec4);
                const int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + row;
                if (nvalid <= 0 || m >= M) continue;
                if (p.flags & VD_EPI_AFFINE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
                }
                if (p.flags & VD_EPI_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
                }
                if (nvalid >= 4 && vec_ok) {
                    if (p.flags & VD_EPI_RESIDUAL) {
                        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(res + m * p.ldr + col);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    }
                    if (OUT_F32) *reinterpret_cast<f32x4*>(p.out + m * p.ldo + col) = v;
                    else {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + m * p.ldo + col) = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < nvalid) {
                            float t = v[e];
                            if (p.flags & VD_EPI_RESIDUAL) t += (float)res[m * p.ldr + col + e];
                            if (OUT_F32) p.out[m * p.ldo + col + e] = t;
                            else reinterpret_cast<__bf16*>(p.out)[m * p.ldo + col + e] = (__bf16)t;
                        }
                }
            }
        }
    }
}

const float* zero_page_b() {
    static const float* zp = nullptr;
    if (!zp) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_page_b)) != hipSuccess) q = nullptr;
        zp = (const float*)q;
    }
    return zp;
}

template <int WM, int WN, int TM, int TN, bool OUT_F32>
void launch_b(const vd_conv_desc& d, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int lds = 2 * (BM + BN) * ROW_B;
    static bool attr_done = false;
    auto kfn = k_conv_igemm_bf16<WM, WN, TM, TN, OUT_F32>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const int64_t M = (int64_t)d.N * d.Hg * d.Wg;
    const int64_t nblk = vd_cdiv(M, BM) * vd_cdiv(d.Co, BN);
    const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page_b());
    const int64_t zd_in = zp - reinterpret_cast<const __bf16*>(d.in);
    const int64_t zd_w = zp - reinterpret_cast<const __bf16*>(d.wp);
    hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(WM * WN * 64), lds, s, d, zd_in, zd_w);
}

template <bool OUT_F32>
void dispatch_b(const vd_conv_desc& d, hipStream_t s) {
    int tile = d.tile;
    if (tile <= 0 || tile > 7) tile = 2;
    switch (tile) {
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

int vd_conv_igemm_bf16(const vd_conv_desc* d, int out_f32, void* stream) {
    VD_REQUIRE(d && d->in && d->wp && d->out, "vd_conv_igemm_bf16: null pointer");
    VD_REQUIRE(d->Ci > 0 && d->Ci % 64 == 0, "vd_conv_igemm_bf16: Ci=%d must be a positive multiple of 64", d->Ci);
    VD_REQUIRE(d->T >= 1 && d->T <= VD_MAX_TAPS && d->Kfr >= 1 && d->N % d->Kfr == 0, "vd_conv_igemm_bf16: bad taps");
    VD_REQUIRE(d->out_stride == 1 && d->out_oy == 0 && d->out_ox == 0 && d->Ho == d->Hg && d->Wo == d->Wg,
               "vd_conv_igemm_bf16: forward geometry only");
    VD_REQUIRE(d->ldo >= d->Co && (int64_t)d->N * d->Hi * d->Wi < (1ll << 31) && (int64_t)d->N * d->Hg * d->Wg < (1ll << 31),
               "vd_conv_igemm_bf16: bad sizes");
    VD_REQUIRE(!(d->flags & VD_EPI_RESIDUAL) || (d->residual && d->ldr >= d->Co), "vd_conv_igemm_bf16: residual missing");
    VD_REQUIRE(!d->in_scale && !d->stats_part, "vd_conv_igemm_bf16: in-load transform / fused statistics are fp32-path features");
    static const int probe = getenv("VD_IGEMM_PROBE") ? atoi(getenv("VD_IGEMM_PROBE")) : 0;
    vd_conv_desc dd = *d;
    dd.flags |= (probe & 7) << 8;
    if (out_f32) dispatch_b<true>(dd, (hipStream_t)stream);
    else dispatch_b<false>(dd, (hipStream_t)stream);
    VD_CHECK_LAUNCH("vd_conv_igemm_bf16");
    return VD_OK;
}

int vd_pack_weight_bf16(const float* wp_f32, void* wp_bf16, int Co, int Co_pad, int Ci, int Ci_pad, int T, void* stream) {
    VD_REQUIRE(wp_f32 && wp_bf16 && Co > 0 && Co_pad >= Co && Ci > 0 && Ci_pad >= Ci && T > 0, "vd_pack_weight_bf16: bad args");
    const int64_t total = (int64_t)Co_pad * T * Ci_pad;
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
