// vd_conv_c32_bf16.hip - the two 32-channel 3x3 convs of Darknet-53's first stage on bf16 tensors, as their own kernel.
//
// Reference: _conv2d(64, 3, 1, 2) and the residual block's _conv2d(64, 3, 1, 1) right behind the stem,
// models/definitions/darknet/three_darknet.py:182-183 / :105-107 (nn.Conv2D + BatchNorm + LeakyReLU of
// models/definitions/layers.py:63-70): 32 -> 64 channels at 608 -> 304 (stride 2) and at 304 x 304 (stride 1) for a 608 x 608
// frame.  They are 13 % of the bf16 detect step's conv time at a third of the other layers' rate in k_conv_igemm_bf16: K is
// 288 (five K-steps of two taps), so a tile is mostly prologue and epilogue, every tap gathers its own copy of the activation
// rows, and the maps are too wide (W >= 208) for the halo loop's row-major staging.
//
// This kernel is shaped for exactly that case:
//   * a workgroup owns a 2-D PATCH of output pixels (8 x 32 at stride 1, 4 x 32 at stride 2) and stages the input patch with
//     its one-pixel border ONCE (10 x 34 or 9 x 65 pixels of 64 bytes): 1.33x / 1.14x the patch's own bytes instead of 9x
//     (one gather per tap) - the nine taps are LDS address offsets.  Stride 2: even and odd input columns go to separate
//     planes of a patch row, so a tap's 32 pixels are consecutive LDS pixels whatever the stride;
//   * an LDS pixel is 80 bytes (64 of channels + 16 of pad): the 16-byte operand reads of 16 consecutive pixels fall on the
//     64 banks' sixteen 4-bank groups exactly once;
//   * the whole weight panel (64 x 288 bf16 = 36 KB) lives in REGISTERS as MFMA B fragments (36 fragments of 4 VGPRs per
//     lane), loaded once per workgroup: the grid is persistent (one workgroup per CU walking the patches, the next patch's
//     two patches' requests in flight under the current one's multiply), so neither
//     weights nor per-launch setup are paid per tile, and LDS bandwidth goes to the activation operand alone;
//   * a wave's 32-row MFMA block is one patch row (32 pixels along x), 18 MFMA steps of 16 channels (tap, channel half).
// Epilogue as k_conv_igemm_bf16's: per-wave LDS transpose, folded BatchNorm scale / shift, LeakyReLU, residual, bf16x4 stores.
// Same sums in the same order as the generic kernel would form them?  No: that one walks K as (tap pair, channel), this one
// as (tap, channel half) - fp32 accumulation of the same 288 products in another order, then one bf16 rounding.
#include "vd_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int PXB = 80;      // LDS bytes per staged pixel
constexpr int SLD = 36;      // floats per row of a wave's transpose patch

// what a request outside the image (or past the last tile) reads: requests are never conditional - with a branch around a
// load the compiler cannot count what is in flight where the paths merge, and drains everything before every use
__device__ __attribute__((aligned(64))) float g_zero_page_c32[16];

// developer build (-DVD_C32_STAMP=1, tools/stamp_c32.sh): thread 0 of one mid-grid workgroup records the cycle counter at the
// phase boundaries of its third tile pair
#ifndef VD_C32_STAMP
#define VD_C32_STAMP 0
#endif
#if VD_C32_STAMP
__device__ unsigned long long g_c32_stamps[16];
#define C32_STAMP(i)                                                                                          \
    do {                                                                                                      \
        if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && tile == (int)blockIdx.x + 4 * (int)gridDim.x) \
            g_c32_stamps[i] = __builtin_readcyclecounter();                                                   \
    } while (0)
#else
#define C32_STAMP(i)
#endif

template <int S>
struct C32Geo {
    static constexpr int TW = 32, TH = S == 1 ? 8 : 4, TM = TH / 4;
    static constexpr int PR = S * (TH - 1) + 3;      // input rows of a patch: TH + 2 | 2 TH + 1
    static constexpr int PC = S * (TW - 1) + 3;      // input columns: 34 | 65
    static constexpr int PATCH_B = ((PR * PC * PXB + 15) / 16) * 16;
    static constexpr int LDS_B = PATCH_B + 4 * 2 * 32 * SLD * 4;      // + the waves' transpose patches (two 32 x 32 blocks each)
    static constexpr int LDS_STEM_B = LDS_B + 3 * (PR + 2) * 68 * 4;   // + the STEM form's frame window
};

// STEM form: what the kernel needs to compute its own input patch - the stem's folded Conv+BN+LeakyReLU output - from the frame
struct vd_stem_args {
    const float* x;        // [N][3][H][W] fp32 frames (H = Hi, W = Wi of the conv descriptor)
    const float* w;        // stem weights, packed [32][32] fp32: row = output channel, k = tap * 3 + channel (k >= 27: zero)
    const float* scale;    // folded BatchNorm of the stem
    const float* shift;
    float slope;
};

// n / d for d >= 1 with rcp = 0xFFFFFFFF / d + 1 (exact after one correction)
__device__ __forceinline__ unsigned udiv_rcp_c(unsigned n, unsigned d, unsigned rcp) {
    unsigned q = d == 1u ? n : __umulhi(n, rcp);
    q -= (q * d > n) ? 1u : 0u;
    return q;
}

// where a masked store goes instead of through an exec-mask branch (64 lanes x 8 bytes)
__device__ __attribute__((aligned(64))) unsigned long long g_sink_c32[64];

// RES: residual add (the block's skip input) in the epilogue.  Folded BatchNorm + LeakyReLU are always there (the host takes
// the kernel only for launches that ask for both): one wave per SIMD issues an instruction every four cycles, so the
// epilogue is written for instruction count - no runtime flag, no masked-store branch, nothing per tile that is the same for
// every tile.  (First build: 13.0 k cycles per tile, of which the epilogue 6.1 k and the request addressing 2.2 k - against
// 2.3 k of MFMA; tools/stamp_c32.sh.)
// STATS (the forward of bf16-storage training, vd_conv_desc.stats_part): the raw conv outputs are stored (no epilogue
// transform) and the tile's per-channel sum / sum of squares of the fp32 accumulators go to row `tile` of the partial table
// [tiles][2 * 64], as k_conv_igemm_bf16 writes its rows (vd_bn_sum_partials / _sum_finalize finish them in fp64).
// STEM (stride 2, inference): the input patch is not read from memory - it is the stem's output (3 -> 32 channels, 3x3, folded
// BatchNorm, LeakyReLU, rounded to bf16 exactly as k_stem_fwd<true, true> writes it: bf16 frame values and weights on
// v_mfma_f32_32x32x16_bf16, two steps of k = tap * 3 + channel) computed here from an 11 x 67 x 3 window of the frame, so the
// stem's 32-channel map (757 MB at batch 32 / 608 x 608) is neither written nor read.  Bit-identical to the two launches.
template <int S, bool RES, bool STATS, bool STEM = false>
__global__ __launch_bounds__(256, 1) void k_conv3x3_c32_bf16(const vd_conv_desc p, const int ntx, const int nty, const int ntiles,
                                                             const vd_stem_args sa) {
    static_assert(!STEM || (S == 2 && !RES && !STATS), "the fused stem feeds the stride-2 inference conv");
    using G = C32Geo<S>;
    constexpr int TW = G::TW, TH = G::TH, TM = G::TM, PR = G::PR, PC = G::PC;
    constexpr int NCH = PR * PC * 4;                  // 16-byte chunks of a patch
    constexpr int LPT = (NCH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
    unsigned char* patch = smem_c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* stg = reinterpret_cast<float*>(smem_c + G::PATCH_B) + wave * (2 * 32 * SLD);      // two 32 x 32 blocks per wave
    const __bf16* in = reinterpret_cast<const __bf16*>(p.in);
    const __bf16* wp = reinterpret_cast<const __bf16*>(p.wp);
    const __bf16* res = reinterpret_cast<const __bf16*>(p.residual);
    __bf16* out = reinterpret_cast<__bf16*>(p.out);

    // ---- the weight panel: B fragment (k-step ks, column block ni) = columns ni * 32 + (lane & 31), k = 16 ks + 8 (lane >> 5) ..+7
    // of the packed rows [Co][9 * 32] (tap-major, vd_pack_weight_bf16)
    bf16x8 wreg[18][2];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            wreg[ks][ni] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)(ni * 32 + (lane & 31)) * 288 + ks * 16 + (lane >> 5) * 8);
    // ---- per tap: LDS byte offset of (patch row of output row 0, pixel of output column 0), from the launch's own tap table
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = p.dy[t], dx = p.dx[t];
        const int col = S == 1 ? dx + 1 : (dx == 0 ? (PC + 1) / 2 : (dx + 1) / 2);      // stride 2: even columns first, then odd
        toff[t] = ((dy + 1) * PC + col) * PXB;
    }
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    static_assert(!(STATS && RES), "the training forward stores raw outputs");
    f32x4 sc[2], sh[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        sc[ni] = STATS ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(p.scale + ni * 32 + ec4);
        sh[ni] = STATS ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(p.shift + ni * 32 + ec4);
    }
    const float slope = p.slope;
    const int lane_a = (lane & 31) * PXB + (lane >> 5) * 16;      // this lane's pixel / k-half within an operand read
    const __bf16* zpage = reinterpret_cast<const __bf16*>(g_zero_page_c32);
    __bf16* sink = reinterpret_cast<__bf16*>(g_sink_c32) + lane * 4;
    const unsigned rcp_x = 0xFFFFFFFFu / (unsigned)ntx + 1u, rcp_y = 0xFFFFFFFFu / (unsigned)nty + 1u;
    // ---- what does not depend on the tile, per thread: of its patch chunks the (row, column) in the patch, the element offset
    // from the patch's first pixel and the LDS address; of its epilogue pixels the element offsets from the tile's first pixel
    int ch_rc[LPT], ch_goff[LPT], ch_lds[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        const int idx = tid + j * 256;
        const int pix = idx >> 2, q = idx & 3;
        const int r = pix / PC, c = pix - r * PC;
        const int cc = S == 1 ? c : ((c & 1) ? (PC + 1) / 2 + (c >> 1) : (c >> 1));
        ch_rc[j] = idx < NCH ? (r | (c << 8)) : (0xff | (0x7fffff << 8));    // (past the patch: a column no image has)
        ch_goff[j] = (r * p.Wi + c) * 32 + q * 8;
        ch_lds[j] = idx < NCH ? (r * PC + cc) * PXB + q * 16 : -1;
    }
    int ep_out[4], ep_res[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ep_out[i] = (erow + 8 * i) * p.ldo + ec4;
        ep_res[i] = (erow + 8 * i) * p.ldr + ec4;
    }
    // STEM: the frame window (3 channels x FR rows x FC columns around the patch, zero outside the frame) in LDS behind the
    // transpose patches, pitch FP floats; a thread's LF window elements, and per lane the 16 k-values of its two MFMA operands
    constexpr int FR = PR + 2, FC = PC + 2, FP = 68, NF = 3 * FR * FC, LF = (NF + 255) / 256;
    float* fwin = reinterpret_cast<float*>(smem_c + G::PATCH_B + 4 * 2 * 32 * SLD * 4);
    int f_rc[STEM ? LF : 1], f_goff[STEM ? LF : 1], f_lds[STEM ? LF : 1], koff[STEM ? 16 : 1];
    bf16x8 sb0, sb1;
    float ssc = 1.f, ssh = 0.f;
    if constexpr (STEM) {
#pragma unroll
        for (int j = 0; j < LF; ++j) {
            const int e = tid + j * 256;
            const int ch = e / (FR * FC), rem = e - ch * (FR * FC);
            const int fr = rem / FC, fc = rem - fr * FC;
            f_rc[j] = e < NF ? (fr | (fc << 8)) : (0xff | (0x7fffff << 8));
            f_goff[j] = (ch * p.Hi + fr) * p.Wi + fc;
            f_lds[j] = e < NF ? (ch * FR + fr) * FP + fc : -1;
        }
        const int hh = lane >> 5;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = (e < 8 ? 8 * hh : 16 + 8 * hh) + (e & 7);
            const int tap = k / 3, c = k - 3 * tap;
            koff[e] = k < 27 ? (c * FR + tap / 3) * FP + tap % 3 : -1;      // window offset of tap (dy, dx) = (tap / 3 - 1, tap % 3 - 1)
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sb0[e] = (__bf16)sa.w[(lane & 31) * 32 + 8 * hh + e];
            sb1[e] = (__bf16)sa.w[(lane & 31) * 32 + 16 + 8 * hh + e];
        }
        ssc = sa.scale[lane & 31];
        ssh = sa.shift[lane & 31];
    }

    // ---- the input patch of a tile: every request first (16-byte chunks, pixels outside the image read a zero page - never a
    // branch around a request: the compiler could not count what is in flight where the paths merge); the stores to LDS follow
    // TWO tiles later - the next two patches are in flight while this one is multiplied and stored (one workgroup per CU: 144
    // VGPRs of weights + 64 of accumulators do not fit two waves per SIMD)
    v4i ldA[LPT], ldB[LPT];
    auto request = [&](const int tile, v4i (&ld)[LPT]) {
        const unsigned tr = udiv_rcp_c((unsigned)tile, (unsigned)ntx, rcp_x);
        const int tx = tile - (int)tr * ntx;
        const unsigned n = udiv_rcp_c(tr, (unsigned)nty, rcp_y);
        const int ty = (int)tr - (int)n * nty;
        const int iy0 = S * ty * TH - 1, ix0 = S * tx * TW - 1;
        const bool tv = tile < ntiles;
        if constexpr (STEM) {
            // the frame window: one float per request, window origin (iy0 - 1, ix0 - 1)
            const float* fb = sa.x + (int64_t)(int)n * 3 * p.Hi * p.Wi + (int64_t)(iy0 - 1) * p.Wi + (ix0 - 1);
            const float* zf = reinterpret_cast<const float*>(g_zero_page_c32);
            static_assert(LF <= LPT, "the window's requests reuse the patch's registers");
#pragma unroll
            for (int j = 0; j < LF; ++j) {
                const int iy = iy0 - 1 + (f_rc[j] & 0xff), ix = ix0 - 1 + (f_rc[j] >> 8);
                const bool ok = tv && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                ld[j][0] = __float_as_int(*(ok ? fb + f_goff[j] : zf));
            }
        } else {
            const __bf16* base = in + ((int64_t)((int)n * p.Hi + iy0) * p.Wi + ix0) * 32;
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const int iy = iy0 + (ch_rc[j] & 0xff), ix = ix0 + (ch_rc[j] >> 8);
                const bool ok = tv && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                ld[j] = *reinterpret_cast<const v4i*>(ok ? base + ch_goff[j] : zpage);
            }
        }
    };
    // the residual rows of a tile's epilogue pixels (a lane: 4 pixels x 4 columns per 32 x 32 block), one tile ahead as well
    auto request_rv = [&](const int tile, bf16x4 (&rv)[TM][2][4]) {
        if constexpr (RES) {
            const unsigned tr = udiv_rcp_c((unsigned)tile, (unsigned)ntx, rcp_x);
            const int tx = tile - (int)tr * ntx;
            const unsigned n = udiv_rcp_c(tr, (unsigned)nty, rcp_y);
            const int ty = (int)tr - (int)n * nty;
            const bool tv = tile < ntiles;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int oy = ty * TH + wave * TM + mi;
                const __bf16* rb = res + ((int64_t)((int)n * p.Ho + oy) * p.Wo + tx * TW) * p.ldr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = tv && oy < p.Ho && tx * TW + erow + 8 * i < p.Wo;
                    const __bf16* src = ok ? rb + ep_res[i] : zpage;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) rv[mi][ni][i] = *reinterpret_cast<const bf16x4*>(src + (ok ? ni * 32 : 0));
                }
            }
        }
    };
    auto do_tile = [&](const int tile, v4i (&ld)[LPT], const bf16x4 (&rv)[TM][2][4], bf16x4 (&rv_next)[TM][2][4]) {
        const bool valid = tile < ntiles;                 // (a workgroup's last pair may be one tile: the second half runs masked)
        const unsigned tr = udiv_rcp_c((unsigned)tile, (unsigned)ntx, rcp_x);
        const int tx = tile - (int)tr * ntx;
        const unsigned n = udiv_rcp_c(tr, (unsigned)nty, rcp_y);
        const int ty = (int)tr - (int)n * nty;
        const int oy0 = ty * TH, ox0 = tx * TW;
        C32_STAMP(0);
        __syncthreads();                                  // the previous patch's operand reads are over
        C32_STAMP(1);
        if constexpr (STEM) {
#pragma unroll
            for (int j = 0; j < LF; ++j)
                if (f_lds[j] >= 0) fwin[f_lds[j]] = __int_as_float(ld[j][0]);
            __syncthreads();
            // the stem on the patch's 9 x 65 pixels, 32 at a time: block (row r, column parity) = the 32 pixels of one parity
            // plane of one patch row - the LDS pixels a tap of the stride-2 conv reads are exactly such a run, so a lane's 16
            // results land at compile-time offsets from one address - and one last block for the 33rd even column of the nine
            // rows.  Per block: 16 window reads, two MFMAs, the stem's epilogue per column, 16 two-byte stores.  No branch
            // around an LDS access (a masked read became saveexec + read + wait: sixteen serial round trips per block, 57 k
            // cycles per tile in the first build): k >= 27 reads element 0 and is zeroed by a select.
            const int iy0 = S * oy0 - 1, ix0 = S * ox0 - 1;
            const int hh = lane >> 5, li = lane & 31;
            auto stem_block = [&](const int base, unsigned char* const dst0, const int dstep, const int iy_l, const int iystep,
                                  const int ix_l, const int ixstep, const int nvalid) {
                // element q of the accumulator = block row rr(q) = (q & 3) + 8 (q >> 2) + 4 hh: stored at dst0 + rr * dstep,
                // frame pixel (iy_l + rr * iystep, ix_l + rr * ixstep); rows >= nvalid are not stored (dst0 then points at a dump)
                float gv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) gv[e] = fwin[base + (koff[e] < 0 ? 0 : koff[e])];
                bf16x8 a0, a1;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = koff[e] >= 0 ? gv[e] : 0.f;
                    if (e < 8) a0[e] = (__bf16)v; else a1[e - 8] = (__bf16)v;
                }
                f32x16 sacc;
#pragma unroll
                for (int q = 0; q < 16; ++q) sacc[q] = 0.f;
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, sb0, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, sb1, sacc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int rq = (q & 3) + 8 * (q >> 2);                    // + 4 hh: folded into dst0 / iy_l / ix_l by the caller
                    float t = sacc[q] * ssc + ssh;
                    t = t > 0.f ? t : t * sa.slope;
                    // the conv's zero padding is a padding of the STEM'S OUTPUT: patch pixels outside the frame are zeros
                    const bool inside = (unsigned)(iy_l + rq * iystep) < (unsigned)p.Hi && (unsigned)(ix_l + rq * ixstep) < (unsigned)p.Wi;
                    unsigned char* dst = (rq + 4 * hh < nvalid) ? dst0 + rq * dstep : smem_c + G::PATCH_B + li * 2;
                    *reinterpret_cast<__bf16*>(dst) = (__bf16)(inside ? t : 0.f);
                }
            };
#pragma unroll 1
            for (int blk = wave; blk < 2 * PR; blk += 4) {                    // (rolled: five blocks' registers at once spill)
                const int r = blk >> 1, par = blk & 1;                         // (uniform)
                stem_block(r * FP + 2 * li + par,
                           patch + (r * PC + par * ((PC + 1) / 2) + 4 * hh) * PXB + li * 2, PXB,
                           iy0 + r, 0, ix0 + par + 8 * hh, 2, 32);
            }
            if (wave == 2)                                                     // column 64 (the 33rd even one) of the PR rows
                stem_block((li < PR ? li : PR - 1) * FP + (PC - 1),
                           patch + (4 * hh * PC + (PC - 1) / 2) * PXB + li * 2, PC * PXB,
                           iy0 + 4 * hh, 1, ix0 + PC - 1, 0, PR);
        } else {
#pragma unroll
            for (int j = 0; j < LPT; ++j)
                if (ch_lds[j] >= 0) *reinterpret_cast<v4i*>(patch + ch_lds[j]) = ld[j];
        }
        __syncthreads();
        C32_STAMP(2);
        request_rv(tile + (int)gridDim.x, rv_next);
        request(tile + 2 * (int)gridDim.x, ld);
        C32_STAMP(3);
        // ---- 18 MFMA steps: (tap t, channel half h); a wave's row block mi is output row wave * TM + mi of the patch
        f32x16 acc[TM][2];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        // (the operand reads of step ks + 1 are issued before the MFMAs of step ks: one wave per SIMD, nobody else hides them)
        auto frags = [&](bf16x8 (&fa)[TM], int ks) {
            const int t = ks >> 1, h = ks & 1;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                fa[mi] = *reinterpret_cast<const bf16x8*>(patch + toff[t] + (S * (wave * TM + mi)) * (PC * PXB) + h * 32 + lane_a);
        };
        bf16x8 fa0[TM], fa1[TM];
        frags(fa0, 0);
#pragma unroll
        for (int ks = 0; ks < 18; ks += 2) {
            frags(fa1, ks + 1);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[mi], wreg[ks][ni], acc[mi][ni], 0, 0, 0);
            if (ks + 2 < 18) frags(fa0, ks + 2);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[mi], wreg[ks + 1][ni], acc[mi][ni], 0, 0, 0);
        }
        C32_STAMP(4);
        if constexpr (STATS) {
            // per-channel sum / sum of squares over the tile's pixels inside the image: a lane's column is lane & 31, its 16
            // rows of a block are patch columns (r & 3) + 8 (r >> 2) + 4 (lane >> 5); half-waves by shuffle, waves through LDS
            float* red = reinterpret_cast<float*>(smem_c + G::PATCH_B);      // [4 waves][64][2] (the transpose patches, unused yet)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const bool rowok = valid && oy0 + wave * TM + mi < p.Ho;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int px = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const float v = (rowok && ox0 + px < p.Wo) ? acc[mi][ni][r] : 0.f;
                        s1 += v;
                        s2 += v * v;
                    }
                }
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (lane < 32) {
                    red[(wave * 64 + ni * 32 + lane) * 2 + 0] = s1;
                    red[(wave * 64 + ni * 32 + lane) * 2 + 1] = s2;
                }
            }
            __syncthreads();
            if (tid < 64 && valid) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s1 += red[(w * 64 + tid) * 2 + 0];
                    s2 += red[(w * 64 + tid) * 2 + 1];
                }
                float* dstp = p.stats_part + (int64_t)tile * 128;
                dstp[tid] = s1;
                dstp[64 + tid] = s2;
            }
            __syncthreads();                              // the patches are free for the transposes
        }
        // ---- epilogue: a patch row's two 32 x 32 blocks (64 channels) go through the wave's LDS patch together; a lane then
        // owns 4 columns of 4 pixels in each.  Pixels past the image's edge are stored to a sink, not branched around.
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int oy = oy0 + wave * TM + mi;
            __bf16* ob = out + ((int64_t)((int)n * p.Ho + oy) * p.Wo + ox0) * p.ldo;
            const bool rowok = valid && oy < p.Ho;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stg[ni * (32 * SLD) + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + (lane & 31)] = acc[mi][ni][r];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            f32x4 v[2][4];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[ni][i] = *reinterpret_cast<const f32x4*>(stg + ni * (32 * SLD) + (erow + 8 * i) * SLD + ec4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = rowok && ox0 + erow + 8 * i < p.Wo;
                __bf16* dst = ok ? ob + ep_out[i] : sink;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    bf16x4 o;
                    if constexpr (STATS) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[ni][i][e];
                    } else {
                        const f32x4 tt = v[ni][i] * sc[ni] + sh[ni];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float u = fmaxf(tt[e], tt[e] * slope);        // LeakyReLU, 0 < slope < 1 (checked on the host)
                            if constexpr (RES) u += (float)rv[mi][ni][i][e];
                            o[e] = (__bf16)u;
                        }
                    }
                    *reinterpret_cast<bf16x4*>(dst + (ok ? ni * 32 : 0)) = o;
                }
            }
        }
        C32_STAMP(5);
    };
    bf16x4 rvA[TM][2][4], rvB[TM][2][4];
    request((int)blockIdx.x, ldA);
    request((int)(blockIdx.x + gridDim.x), ldB);
    request_rv((int)blockIdx.x, rvA);
    for (int tile = blockIdx.x; tile < ntiles; tile += 2 * (int)gridDim.x) {
        do_tile(tile, ldA, rvA, rvB);
        do_tile(tile + (int)gridDim.x, ldB, rvB, rvA);
    }
}

int cus_c32() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

template <int S, bool RES, bool STATS, bool STEM = false>
void launch_c32(const vd_conv_desc& d, hipStream_t s, const vd_stem_args& sa = vd_stem_args{}) {
    using G = C32Geo<S>;
    auto kfn = k_conv3x3_c32_bf16<S, RES, STATS, STEM>;
    constexpr int lds = STEM ? G::LDS_STEM_B : G::LDS_B;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const int ntx = (d.Wo + G::TW - 1) / G::TW, nty = (d.Ho + G::TH - 1) / G::TH;
    const int64_t ntiles = (int64_t)d.N * ntx * nty;
    const int64_t grid = ntiles < (int64_t)cus_c32() ? ntiles : (int64_t)cus_c32();      // persistent: one workgroup per CU
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(256), lds, s, d, ntx, nty, (int)ntiles, sa);
}

}  // namespace

// 1 where vd_conv_igemm_bf16 with d->tile == 16 runs this kernel: a 3x3 conv (any tap order within {-1,0,1}^2) of 32 -> 64
// channels, stride 1 or 2, 'same' padding, bf16 output at stride 1, the vector epilogue's alignment; either the inference
// cell's epilogue or the training forward's raw output + statistics; no fused backward reductions / temporal taps
bool vd_conv_c32_bf16_ok(const vd_conv_desc& d, bool out_f32) {
    if (out_f32 || d.Ci != 32 || d.Co != 64 || d.T != 9 || d.Kfr != 1 || (d.in_stride != 1 && d.in_stride != 2)) return false;
    if (d.out_stride != 1 || d.out_oy || d.out_ox || d.Ho != d.Hg || d.Wo != d.Wg || d.bs_part || d.in_scale) return false;
    if (d.Hg != (d.Hi + d.in_stride - 1) / d.in_stride || d.Wg != (d.Wi + d.in_stride - 1) / d.in_stride) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        if (d.dy[t] < -1 || d.dy[t] > 1 || d.dx[t] < -1 || d.dx[t] > 1 || d.dz[t] != 0) return false;
        seen |= 1u << ((d.dy[t] + 1) * 3 + d.dx[t] + 1);
    }
    if (seen != 0x1ffu) return false;
    auto al = [](const void* q, int a) { return (uintptr_t)q % a == 0; };
    if (d.ldo % 4 || !al(d.out, 8) || !al(d.in, 16) || !al(d.wp, 16)) return false;
    if ((d.flags & VD_EPI_RESIDUAL) && (d.ldr % 4 || !al(d.residual, 8))) return false;
    if (d.stats_part) {
        // the training forward: raw outputs + the statistics rows, no epilogue transform
        if (d.flags & (VD_EPI_AFFINE | VD_EPI_LEAKY | VD_EPI_RESIDUAL)) return false;
    } else {
        // the inference cell's epilogue: folded BatchNorm (both vectors) + LeakyReLU with a slope in (0, 1) [+ residual]
        if (!(d.flags & VD_EPI_AFFINE) || !(d.flags & VD_EPI_LEAKY) || !d.scale || !d.shift || !al(d.scale, 16) || !al(d.shift, 16)) return false;
        if (!(d.slope > 0.f && d.slope < 1.f)) return false;
    }
    if ((int64_t)d.Hi * d.Wi * 32 >= (1ll << 31) || d.Wi >= (1 << 22) || d.ldo >= (1 << 20) || d.ldr >= (1 << 20)) return false;     // 32-bit offsets inside a frame
    return (int64_t)d.N * ((d.Wo + 31) / 32) * ((d.Ho + 3) / 4) < (1ll << 30);
}

#if VD_C32_STAMP
extern "C" int vd_debug_c32_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c32_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

// rows of the statistics table a launch of this kernel writes (one per patch)
int64_t vd_conv_c32_bf16_tiles(const vd_conv_desc& d) {
    const int th = d.in_stride == 1 ? 8 : 4;
    return (int64_t)d.N * ((d.Wo + 31) / 32) * ((d.Ho + th - 1) / th);
}

void vd_conv_c32_bf16_launch(const vd_conv_desc& d, hipStream_t s) {
    const bool r = d.flags & VD_EPI_RESIDUAL;
    if (d.stats_part) { if (d.in_stride == 1) launch_c32<1, false, true>(d, s); else launch_c32<2, false, true>(d, s); }
    else if (d.in_stride == 1) { if (r) launch_c32<1, true, false>(d, s); else launch_c32<1, false, false>(d, s); }
    else { if (r) launch_c32<2, true, false>(d, s); else launch_c32<2, false, false>(d, s); }
}

// The stem and the stride-2 conv behind it in ONE launch (bf16 inference): `d` describes the 32 -> 64 channel conv exactly as
// for vd_conv_igemm_bf16 with tile 16 (d->in is not read: the stem's output is computed inside the kernel from the fp32 NCHW
// frames); the stem's epilogue is folded BatchNorm + LeakyReLU.  Replaces vd_stem_conv(..., out_bf16 = 1) + vd_conv_igemm_bf16
// for _conv2d(32, 3, 1, 1) + _conv2d(64, 3, 1, 2), three_darknet.py:163-164,182-183, with bit-identical outputs.
extern "C" int vd_stem_conv_c32_bf16(const float* x_nchw, const float* stem_wp, const float* stem_scale, const float* stem_shift,
                                     float stem_slope, const vd_conv_desc* d, void* stream) {
    VD_REQUIRE(x_nchw && stem_wp && stem_scale && stem_shift && d && d->wp && d->out, "vd_stem_conv_c32_bf16: null pointer");
    vd_conv_desc dd = *d;
    if (!dd.in) dd.in = reinterpret_cast<const float*>(x_nchw);      // (alignment check of a pointer that is never read)
    VD_REQUIRE(dd.in_stride == 2 && !(dd.flags & VD_EPI_RESIDUAL) && !dd.stats_part && vd_conv_c32_bf16_ok(dd, false),
               "vd_stem_conv_c32_bf16: the descriptor is not the first-stage stride-2 conv (3x3, 32 -> 64 channels, bf16, folded BatchNorm + LeakyReLU)");
    VD_REQUIRE((int64_t)3 * dd.Hi * dd.Wi < (1ll << 31) && ((uintptr_t)x_nchw % 4 == 0), "vd_stem_conv_c32_bf16: frame size");
    vd_stem_args sa;
    sa.x = x_nchw; sa.w = stem_wp; sa.scale = stem_scale; sa.shift = stem_shift; sa.slope = stem_slope;
    launch_c32<2, false, false, true>(dd, (hipStream_t)stream, sa);
    VD_CHECK_LAUNCH("vd_stem_conv_c32_bf16");
    return VD_OK;
}
