// vd_wgrad_halo.hip - weight gradient of the 3x3 stride-1 'same' convolutions with the activation operand staged ONCE
// per pixel as a sliding halo ring in LDS (the weight-gradient twin of k_conv_igemm's HALO loop, vd_conv.hip).
//
// Replaces, on the reference side: autograd.backward wrt the nn.Conv2D weight of _conv2d (models/definitions/layers.py:66-67,
// train_yolov3.py:631) for every 3x3 / stride-1 / pad-1 layer with Co >= 128 (Darknet-53 residual bodies, the neck's 3x3 cells).
//
//   dwp[co][t*Ci + ci] = sum_{n,y,x} dout[n,y,x,co] * in[n, y+dy[t], x+dx[t], ci]          (zero outside the image)
//
// k_conv_wgrad (vd_conv.hip) gives a workgroup a BM x 128 tile of the (co, tap*Ci + ci) matrix and gathers BOTH operands once
// per pixel step: a 3x3 layer reads every activation pixel nine times (once per tap tile) and every dout pixel Ci*9/128
// times - 48 KB into the CU per 32-pixel step against 1536 matrix-pipe cycles, 31 B/clk/CU, which is the rate the guide
// measures for L2-resident gathers (MI355X_MICROARCH.md, "Indexed rows"): the kernel sits on the CU's load path.
// Here a workgroup owns BM output channels x (9 taps x NCH chunks of 32 input channels), BM x NCH = 128 x 2 or 256 x 1: the
// dout tile of a pixel step is shared by the nine taps and the chunks, and the activation rows enter LDS once, into a RING
// indexed by pixel position; tap t reads the ring at row offset dy*(W+1) + dx.  24 KB (128 x 2: 16 dout + 8 activation) or
// 36 KB (256 x 1) per 3456 matrix-pipe cycles: 7-10 B/clk/CU.
//
// Borders without masks: the reduction runs over a PADDED position space q = (n*(H+1) + y)*(W+1) + x with one pad column
// per image row and one pad row per image.  Every out-of-image tap read lands on a pad position (x-1 at x = 0 is the
// previous row's pad column, y-1 at y = 0 the previous image's pad row, ...), which the ring holds as zeros, and dout is
// zero at pad positions, so they add nothing.  Cost: (1 + 1/W)(1 + 1/H) of the MFMA work (1.08 at 26x26, 1.04 at 52x52);
// gain: no per-(lane, tap) select in the K loop - a tap's LDS address is one add and one and.
//
// Arithmetic: NPL = 2 is the two-way fp16 split of vd_conv.hip (VD_MATH_F16X2: fp32 tensors, three f16 MFMAs per product
// block, fp32-accurate); BF = bf16-stored operands (VD_STORE_BF16: one bf16 MFMA per block).  Accumulators, slabs and the
// gradient are fp32; pixel ranges ("splits") go to workspace slabs summed in slab order by k_reduce_slabs (deterministic).
#include "vd_common.h"
#include "vd_wgrad_halo.h"
#include <stdlib.h>

namespace {

__device__ __attribute__((aligned(64))) float g_zero_page_wh[64];

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_p;

#ifndef VD_WH_SETPRIO
#define VD_WH_SETPRIO 2        // developer A/B: s_setprio 1 around 0 = nothing, 1 = the whole multiply phase, 2 = every MFMA group
#endif
constexpr int WH_BP = 32;      // padded positions per K-step (64 for the bf16-tensor form where LDS allows)
constexpr int WH_CH = 32;      // input channels per workgroup
constexpr int WH_T = 9;

__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    f32x2 v = {a, b};
    f16x2 r = __builtin_convertvector(v, f16x2);        // v_cvt_pk_f16_f32 (RNE)
    return __builtin_bit_cast(unsigned, r);
}
// x*s = h + l (+ at most 2^-23 |x*s|), h = fp16(x*s), l = fp16(x*s - h): vd_conv.hip split2
__device__ __forceinline__ void split2(const f32x4 v, const float s, v2i& h, v2i& l) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = v[2 * q] * s, x1 = v[2 * q + 1] * s;
        const unsigned hp = pk_f16(x0, x1);
        const f16x2 hv = __builtin_bit_cast(f16x2, hp);
        h[q] = (int)hp;
        l[q] = (int)pk_f16(x0 - (float)hv[0], x1 - (float)hv[1]);
    }
}

template <bool BF>
__device__ __forceinline__ f32x16 mma(const v4i a, const v4i b, const f32x16 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Eight waves.  BM = 256: wave w owns output channels 32 w .. of ONE 32-channel chunk of `in`; BM = 128: waves 0-3 and
// 4-7 own the same 128 output channels and two neighbouring chunks (NCH = 2), so a 128-channel layer still runs two
// waves per SIMD on one dout tile.  Each wave: nine 32x32 accumulators (144 registers).
// LDS: [2 stages][NPL planes][BP positions][BM co] fp16 (dout; 64-B channel chunks XOR-swizzled with position & 3, as
// k_conv_wgrad), then per chunk and plane the activation ring [RING + BP rows][32 ch] fp16, 64 B per row: the 4
// consecutive rows x 64 B a half-wave's ds_read_b64_tr_b16 touches are 256 contiguous bytes (mod the ring) = all 64 banks.
// Inside the multiply phase the operand fragments of item i + 1 (an item = one tap of one 16-position block) are read before
// the MFMAs of item i, and every MFMA group runs at s_setprio 1 (same-box A/B: without it the bf16 form loses 5 %).
// ---------------------------------------------------------------------------------------------------------------------
struct WhCursor {      // a position of the padded space: column, row within the image (H = the pad row), and the
    int x, y, pr;      // real-pixel index of (n, y, 0)
};

// P2: the ring has a power-of-two row count (wrap = one and); otherwise it has exactly the 2 BP + 2 hloa rows the window
// needs and wraps by compare - the form that lets the two-chunk tile of a 104-wide map fit LDS (320 rows instead of 512).
// BP: padded positions per K-step, 32 or 64 (64 halves the barriers and the per-step cursor / request work per MFMA: what the
// bf16-tensor form, with one MFMA per product block, is bound by: +6 %; the fp16-split form measured the same at both).
template <int BM, bool BF, bool P2, int BP>
__global__ __launch_bounds__(512, 2) void k_conv_wgrad_halo(const vd_wgrad_desc p, float* __restrict__ dst, const int64_t q_per_split,
                                                            const int ring_rows, const int hloa, const int64_t zd_in,
                                                            const int64_t zd_do) {
    constexpr int NT = 512;
    constexpr int NPL = BF ? 1 : 2;
    constexpr int NCH = 256 / BM;                        // 32-channel chunks of `in` per workgroup
    constexpr int MW = BM / 32;                          // waves along the output channels
    constexpr int VW = BF ? 8 : 4;                       // channels per 16-byte load
    constexpr int TPP = BM / VW;                         // dout lanes per position
    constexpr int RW = 64 / TPP;                         // positions one wave instruction covers
    constexpr int ARP = NT / TPP;                        // positions per pass of the whole workgroup
    constexpr int APASS = BP / ARP;
    static_assert(RW >= 1 && APASS >= 1 && APASS * ARP == BP, "dout loader");
    constexpr int MIR = BP;                              // ring rows mirrored past the end: one wrapped address serves a lane's BP - 8 rows
    constexpr int XTP = NCH * WH_CH / VW;                // activation lanes per position
    constexpr int XNT = BP * XTP;                        // 16-byte pieces of the BP new positions of a step
    constexpr int XPASS = XNT > NT ? XNT / NT : 1;       // ... per lane
    constexpr int XPP = NT / XTP;                        // positions one pass of the workgroup covers
    static_assert(XPASS * NT == XNT || XNT < NT, "activation loader");
    constexpr int APL = BP * BM * 2;                     // bytes of one dout plane
    using LT = typename vd_select<BF, v4i, f32x4>::type;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As3 = reinterpret_cast<char*>(smem);           // [2][NPL][32][BM] fp16
    char* Xr = As3 + 2 * NPL * APL;                      // [NCH][NPL][RING + 32][64 B]
    const int RING = ring_rows;                          // a multiple of 32
    const int XPL = (RING + MIR) * 64;                  // bytes of one ring plane
    const int RB = RING * 64;                            // bytes of the ring proper
    auto wrapB = [&](int a) -> int {                     // byte offset in (-RB, 2 RB) -> [0, RB)
        if (P2) return a & (RB - 1);
        a += a < 0 ? RB : 0;
        a -= a >= RB ? RB : 0;
        return a;
    };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % MW, wch = wave / MW;          // the wave's 32 output channels / its chunk of `in`
    const int W = p.Wi, H = p.Hi, W1 = W + 1, H1 = H + 1;
    const int64_t Q = (int64_t)p.N * H1 * W1;            // < 2^31, checked on the host

    int sexp_a = 0, sexp_b = 0;
    float scl_a = 1.f, scl_b = 1.f;
    if (!BF) {
        sexp_a = vd_f16_scale_exp(vd_amax_read(p.amax_dout));
        sexp_b = vd_f16_scale_exp(vd_amax_read(p.amax_in));
        scl_a = __uint_as_float((unsigned)(127 + sexp_a) << 23);
        scl_b = __uint_as_float((unsigned)(127 + sexp_b) << 23);
    }

    // block -> (split, channel chunk group, co tile): whole pixel ranges per XCD, so a range's dout / activation panels are
    // fetched into ONE L2 and shared by its workgroups
    const int mtiles = (p.Co + BM - 1) / BM, ctiles = p.Ci / (NCH * WH_CH);
    int b = vd_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = b % mtiles; b /= mtiles;
    const int tile_c = b % ctiles;
    const int split = b / ctiles;
    const int q_begin = (int)((int64_t)split * q_per_split);
    int q_end = (int)((int64_t)q_begin + q_per_split < Q ? (int64_t)q_begin + q_per_split : Q);
    if (q_end < q_begin) q_end = q_begin;
    const int nks = (q_end - q_begin + BP - 1) / BP;
    const int c0 = tile_c * (NCH * WH_CH);

    // tap t reads the ring tapB[t] bytes from the centre row
    int tapB[WH_T];
#pragma unroll
    for (int t = 0; t < WH_T; ++t) tapB[t] = (p.dy[t] * W1 + p.dx[t]) * 64;

    // ---- cursors over the padded space: decoded once with divisions, then advanced 32 positions per step with carries.
    // Written without short-circuit conditions: with them hipcc turned the (wave-uniform) dout cursors into a chain of
    // scalar branches around every request.
    const int step_x = BP % W1, step_y = BP / W1;
    auto decode = [&](int q) -> WhCursor {
        const unsigned qu = (unsigned)q;
        const unsigned row = qu / (unsigned)W1;
        const unsigned n_ = row / (unsigned)H1;
        WhCursor c;
        c.x = (int)(qu - row * (unsigned)W1);
        c.y = (int)(row - n_ * (unsigned)H1);
        c.pr = ((int)n_ * H + c.y) * W;
        return c;
    };
    auto advance = [&](WhCursor& c) {
        int x = c.x + step_x;
        const int cx = x >= W1 ? 1 : 0;
        x -= cx ? W1 : 0;
        const int d = step_y + cx;
        int y = c.y + d, pr = c.pr + d * W;
        const int c2 = y >= H1 ? 1 : 0;                  // past the pad row: next image (the pad row is not a real row)
        y -= c2 ? H1 : 0;
        pr -= c2 ? W : 0;
        const int c3 = y >= H1 ? 1 : 0;                  // tiny maps: a step can cross two images (32 / W1 + 1 <= 2 H1)
        y -= c3 ? H1 : 0;
        pr -= c3 ? W : 0;
        c.x = x; c.y = y; c.pr = pr;
    };

    // ---- dout loader: wave-uniform positions (scalar cursors), lane = co chunk.  Pass i covers positions ARP*i + RW*wave + hf.
    const int hf = lane / TPP;                           // which of the wave instruction's RW positions
    const int alc = (lane % TPP) * VW;
    const int co = tile_m * BM + alc;
    const bool co_ok = co < p.Co;
    WhCursor ac[APASS][RW];
#pragma unroll
    for (int i = 0; i < APASS; ++i)
#pragma unroll
        for (int r = 0; r < RW; ++r) ac[i][r] = decode(q_begin + ARP * i + RW * wave + r);
    int aq = q_begin;                                    // first position of the next dout tile to request
    auto gloadA = [&](LT (&ra)[APASS]) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            int64_t sel = zd_do;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int q = aq + ARP * i + RW * wave + r;
                const WhCursor c = ac[i][r];
                const bool ok = (c.x < W) & (c.y < H) & (q < q_end);            // wave-uniform
                const int64_t o = (int64_t)(c.pr + c.x) * p.ldd + co;
                const int64_t so = (ok & co_ok) ? o : zd_do;
                if (RW == 1 || hf == r) sel = so;
                advance(ac[i][r]);
            }
            if constexpr (BF) ra[i] = *reinterpret_cast<const v4i*>(reinterpret_cast<const __bf16*>(p.dout) + sel);
            else ra[i] = *reinterpret_cast<const f32x4*>(p.dout + sel);
        }
        aq += BP;
    };
    auto lstoreA = [&](int buf, const LT (&ra)[APASS]) {
        char* a3 = As3 + buf * NPL * APL + (alc & 31) * 2;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int px = ARP * i + RW * wave + hf;
            char* r = a3 + px * (BM * 2) + (((alc >> 5) ^ (px & 3)) << 6);
            if constexpr (BF) {
                *reinterpret_cast<v4i*>(r) = ra[i];
            } else {
                v2i h, l;
                split2(ra[i], scl_a, h, l);
                *reinterpret_cast<v2i*>(r) = h;
                *reinterpret_cast<v2i*>(r + APL) = l;
            }
        }
    };

    // ---- activation loader: lanes < XNT, one 16-byte piece each: position xj of the step's 32 new ones, channels xc..
    const int xj = tid / XTP, xc = (tid % XTP) * VW;
    const bool xact = XNT >= NT || tid < XNT;
    char* const xlane = Xr + (xc >> 5) * (NPL * XPL) + (xc & 31) * 2;      // the lane's chunk ring, its channel bytes
    auto xload_cur = [&](const WhCursor c, bool inrange) -> LT {
        const bool ok = inrange & (c.x < W) & (c.y < H);
        const int64_t o = (int64_t)(c.pr + c.x) * p.Ci + c0 + xc;
        const int64_t sel = ok ? o : zd_in;
        if constexpr (BF) return *reinterpret_cast<const v4i*>(reinterpret_cast<const __bf16*>(p.in) + sel);
        else return *reinterpret_cast<const f32x4*>(p.in + sel);
    };
    auto xstore_row = [&](const LT v, int row) {         // ring row `row` (< RING), mirrored when < 32
        char* r = xlane + row * 64;
        if constexpr (BF) {
            *reinterpret_cast<v4i*>(r) = v;
            if (row < MIR) *reinterpret_cast<v4i*>(r + RING * 64) = v;
        } else {
            v2i h, l;
            split2(v, scl_b, h, l);
            *reinterpret_cast<v2i*>(r) = h;
            *reinterpret_cast<v2i*>(r + XPL) = l;
            if (row < MIR) {
                *reinterpret_cast<v2i*>(r + RING * 64) = h;
                *reinterpret_cast<v2i*>(r + RING * 64 + XPL) = l;
            }
        }
    };
    // steady-state cursors of the lane's positions in the stream of new blocks (block s = positions q_begin + BP (s+1) + hloa ..)
    int xq = q_begin + BP + hloa + xj;
    WhCursor xcur[XPASS];
#pragma unroll
    for (int k2 = 0; k2 < XPASS; ++k2) xcur[k2] = decode(xq + k2 * XPP);
    auto gloadX = [&](LT (&xv)[XPASS]) {
#pragma unroll
        for (int k2 = 0; k2 < XPASS; ++k2) {
            xv[k2] = xload_cur(xcur[k2], xact & ((int64_t)(xq + k2 * XPP) < Q));
            advance(xcur[k2]);
        }
        xq += BP;
    };

    f32x16 acc[WH_T];
#pragma unroll
    for (int t = 0; t < WH_T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- MFMA operand addressing (tr_operand's lane map, vd_conv.hip): 16-lane group g, lane 4*qq + pp addresses position
    // row 8*(g>>1) + qq (+4; +16 for the second half), channels 16*(g&1) + 4*pp .., and receives channel (lane & 15)
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int a_off = (8 * (g >> 1) + qq) * (BM * 2) + ((wco ^ qq) << 6) + ((16 * (g & 1) + 4 * pp) << 1);
    const int x_lane = (8 * (g >> 1) + qq) * 64 + ((16 * (g & 1) + 4 * pp) << 1);
    const char* const xwave = Xr + wch * (NPL * XPL);    // the wave's chunk ring
    int xb = hloa * 64 + x_lane;                         // step 0: position q_begin sits at ring row hloa

    auto readA = [&](v4i (&fa)[NPL], const char* a3, int kc) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const char* pa = a3 + q * APL + kc * 16 * (BM * 2);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(pa));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(pa + 4 * (BM * 2)));
            fa[q] = __builtin_bit_cast(v4i, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto readB = [&](v4i (&fb)[NPL], int at, int kc) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const char* pb = xwave + q * XPL + at + kc * 16 * 64;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(pb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(pb + 256));
            fb[q] = __builtin_bit_cast(v4i, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto compute = [&](int buf) {
        const char* a3 = As3 + buf * NPL * APL + a_off;
        auto at = [&](int t) { return wrapB(xb + tapB[t]); };         // tap t's ring address: one add, one and (P2)
        // operand fragments PF items ahead of the MFMAs that use them (an item = one tap of one 16-position half).  One item
        // is enough: PF = 2 on the 128-row tile (8 more VGPRs) measured the same within noise on every layer shape
        // (same-box A/B, tools/wgrad_bench.py) - the LDS round trip is not what the loop waits for.
        constexpr int PF = 1;
        v4i fa[NPL], fb[PF + 1][NPL];
#if VD_WH_SETPRIO == 1
        __builtin_amdgcn_s_setprio(1);
#endif
        readA(fa, a3, 0);
        constexpr int NI = (BP / 16) * WH_T;             // items: (16-position block kc, tap t)
#pragma unroll
        for (int j = 0; j < PF; ++j) readB(fb[j], at(j % WH_T), j / WH_T);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = i % WH_T;
            if (i + PF < NI) readB(fb[(i + PF) % (PF + 1)], at((i + PF) % WH_T), (i + PF) / WH_T);
            if (i > 0 && t == 0) readA(fa, a3, i / WH_T); // (one register set: the next block's dout fragment is read where
#if VD_WH_SETPRIO == 2                                   // it is needed - a second set put the kernel over 256 VGPRs)
            __builtin_amdgcn_s_setprio(1);
#endif
            if constexpr (BF) {
                acc[t] = mma<true>(fa[0], fb[i % (PF + 1)][0], acc[t]);
            } else {                                      // smallest partial products first: al*bh, ah*bl, ah*bh
                acc[t] = mma<false>(fa[1], fb[i % (PF + 1)][0], acc[t]);
                acc[t] = mma<false>(fa[0], fb[i % (PF + 1)][1], acc[t]);
                acc[t] = mma<false>(fa[0], fb[i % (PF + 1)][0], acc[t]);
            }
#if VD_WH_SETPRIO == 2
            __builtin_amdgcn_s_setprio(0);
#endif
        }
#if VD_WH_SETPRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        xb = wrapB(xb + BP * 64);
    };

    // ---- prologue: ring rows [0, BP + 2 hloa) = positions q_begin - hloa ..; dout tile 0 -> stage 0
    LT ra[APASS];
    {
        // four passes of loads in flight per round trip (the ring's first fill is 96 .. 576 positions)
        const int nfill = BP + 2 * hloa;
        constexpr int PPP = NT / XTP;                    // positions per pass with every lane loading
        const int fj = tid / XTP;
        for (int r0 = 0; r0 < nfill; r0 += 4 * PPP) {
            LT v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = r0 + e * PPP + fj;
                const int q = q_begin - hloa + row;
                const bool in = (row < nfill) & (q >= 0) & ((int64_t)q < Q);
                v[e] = xload_cur(decode(in ? q : 0), in);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = r0 + e * PPP + fj;
                if (row < nfill) xstore_row(v[e], row);
            }
        }
    }
    gloadA(ra);
    lstoreA(0, ra);
    __syncthreads();
    int xrow = wrapB((BP + 2 * hloa + xj) * 64) >> 6;    // ring row of the lane's first position in block 0

    // One step: request dout tile s+1 and activation block s, multiply tile s, then split + store both (tile s+1 into the other
    // LDS stage, block s into ring rows nobody reads before the barrier): a whole multiply phase of latency cover with ONE
    // register set per operand.  Requests past the range read the zero page and the rows they land in are never read, so the
    // loop has no guards (a guarded request makes hipcc drain vmcnt(0) where paths merge - vd_conv.hip); an odd step count
    // runs one step on zero dout rows.  (Tried and dropped: the two waves of a SIMD running [multiply] and [split + store] in
    // opposite order, vd_conv.hip's STAGGER - no difference on any layer shape in a same-box A/B, and it needs a second
    // register set for the dout tile; as an if / else of the two orders hipcc spilled 800 registers of accumulators.)
    auto step = [&](int u) {
        gloadA(ra);
        LT xv[XPASS];
        gloadX(xv);
        compute(u);
        lstoreA(u ^ 1, ra);
        if (xact) {
#pragma unroll
            for (int k2 = 0; k2 < XPASS; ++k2) xstore_row(xv[k2], k2 == 0 ? xrow : (wrapB((xrow + k2 * XPP) * 64) >> 6));
        }
        xrow = wrapB((xrow + BP) * 64) >> 6;
        __syncthreads();
    };
    for (int ks = 0; ks < nks; ks += 2) {
        step(0);
        step(1);
    }

    if (!BF) {                          // undo the two operand scales: an exact power of two
        const int de = -(sexp_a + sexp_b);
#pragma unroll
        for (int t = 0; t < WH_T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_ldexpf(acc[t][r], de);
    }
    const int Ktot = WH_T * p.Ci;
    float* out = dst + (int64_t)split * p.Co * Ktot;
#pragma unroll
    for (int t = 0; t < WH_T; ++t) {
        const int jc = t * p.Ci + c0 + wch * WH_CH + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tile_m * BM + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < p.Co) out[(int64_t)row * Ktot + jc] = acc[t][r];
        }
    }
}

// (process-wide caches here and in launch_k: one device per process, include/viddet_hip.h "Conventions")
const void* zero_page_wh() {
    static const void* zp = nullptr;
    if (!zp) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_page_wh)) != hipSuccess) q = nullptr;
        zp = q;
    }
    return zp;
}

// ---- launch configuration: tile (BM x NCH chunks), positions per step, ring rows ----------------------------------------
struct WhCfg {
    int bm, nch, bp, hloa, ring;
    bool p2, ok;
    int64_t lds;
};

int64_t wh_lds_bytes(bool bf, int bm, int bp, int rows) {
    const int npl = bf ? 1 : 2;
    return (int64_t)2 * npl * bp * bm * 2 + (int64_t)(256 / bm) * npl * (rows + bp) * 64;     // dout stages + rings (+ mirror rows)
}

WhCfg wh_config(const vd_wgrad_desc& d) {
    static const int force_bm = getenv("VD_WGRAD_HALO_BM") ? atoi(getenv("VD_WGRAD_HALO_BM")) : 0;
    static const int force_bp = getenv("VD_WGRAD_HALO_BP") ? atoi(getenv("VD_WGRAD_HALO_BP")) : 0;
    constexpr int64_t LDS = 160 * 1024;
    WhCfg c{};
    const bool bf = d.flags & VD_STORE_BF16;
    c.hloa = ((d.Wi + 2 + 31) / 32) * 32;                  // halo either side of a step's positions, rounded up
    // 64 positions per step for bf16 tensors (one MFMA per product block: the step's fixed work is what binds; +6 %), 32 for
    // the fp16 split (three MFMAs per block: 64 measured the same)
    const int bp_first = (bf && force_bp != 32) ? 64 : 32;               // (VD_WGRAD_HALO_BP=32: developer A/B of the bf16 form)
    // 128 output channels x two 32-channel chunks where Ci allows it and the rings fit LDS: 24 KB of operand loads per 32
    // positions (16 dout + 8 activation) against 36 KB for 256 x one chunk, and half the dout re-reads over the grid
    for (int bp = bp_first; bp >= 32 && !c.ok; bp -= 32) {
        const int need = 2 * bp + 2 * c.hloa;              // a step's window (positions + halo either side) + the block being written
        int bm = (d.Ci % (2 * WH_CH) == 0 && wh_lds_bytes(bf, 128, bp, need) <= LDS) ? 128 : 256;
        if (force_bm == 256 && d.Co >= 256) bm = 256;
        if (wh_lds_bytes(bf, bm, bp, need) > LDS) continue;
        int pow2 = 128;
        while (pow2 < need) pow2 *= 2;
        c.bm = bm; c.nch = 256 / bm; c.bp = bp;
        c.ring = wh_lds_bytes(bf, bm, bp, pow2) <= LDS ? pow2 : need;     // power of two (wrap = one and) where it fits
        c.p2 = (c.ring & (c.ring - 1)) == 0;
        c.lds = wh_lds_bytes(bf, bm, bp, c.ring);
        // the position cursors carry over at most two images per step: a map too small for 64 positions retries 32
        c.ok = bp / (d.Wi + 1) < 2 * (d.Hi + 1);
    }
    return c;
}

template <int BM, bool BF, bool P2, int BP>
void launch_k(const vd_wgrad_desc& d, const WhCfg& c, float* dst, int splits, int64_t qps, hipStream_t s) {
    static int attr_lds = 0;
    if ((int)c.lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wgrad_halo<BM, BF, P2, BP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)c.lds);
        attr_lds = (int)c.lds;
    }
    const int64_t tiles = vd_cdiv(d.Co, BM) * (d.Ci / ((256 / BM) * WH_CH));
    int64_t zd_in, zd_do;
    if (BF) {
        const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page_wh());
        zd_in = zp - reinterpret_cast<const __bf16*>(d.in);
        zd_do = zp - reinterpret_cast<const __bf16*>(d.dout);
    } else {
        const float* zp = reinterpret_cast<const float*>(zero_page_wh());
        zd_in = zp - d.in;
        zd_do = zp - d.dout;
    }
    hipLaunchKernelGGL((k_conv_wgrad_halo<BM, BF, P2, BP>), dim3((unsigned)(tiles * splits)), dim3(512), (size_t)c.lds, s, d, dst, qps,
                       c.ring, c.hloa, zd_in, zd_do);
}
template <int BM, bool BF>
void launch(const vd_wgrad_desc& d, const WhCfg& c, float* dst, int splits, int64_t qps, hipStream_t s) {
    if constexpr (BF) {            // (the fp16-split form measured the same at 32 and 64 positions per step: one instantiation)
        if (c.bp == 64) {
            if (c.p2) launch_k<BM, BF, true, 64>(d, c, dst, splits, qps, s);
            else launch_k<BM, BF, false, 64>(d, c, dst, splits, qps, s);
            return;
        }
    }
    {
        if (c.p2) launch_k<BM, BF, true, 32>(d, c, dst, splits, qps, s);
        else launch_k<BM, BF, false, 32>(d, c, dst, splits, qps, s);
    }
}

}   // namespace

bool vd_wgrad_halo_ok(const vd_wgrad_desc& d) {
    if (!(d.flags & VD_WGRAD_HALO)) return false;
    const bool bf = d.flags & VD_STORE_BF16;
    if (!bf && !(d.flags & VD_MATH_F16X2)) return false;
    if (!bf && (!d.amax_in || !d.amax_dout)) return false;
    if (d.in_scale || d.T != WH_T || d.in_stride != 1 || d.Kfr != 1 || d.Hg != d.Hi || d.Wg != d.Wi) return false;
    if (d.Wi < 5 || d.Hi < 2 || d.Ci % WH_CH != 0) return false;
    const int vw = bf ? 8 : 4;                             // channels per 16-byte load: rows and tiles must be whole loads
    if (d.Co % vw != 0 || d.ldd % vw != 0) return false;
    for (int t = 0; t < WH_T; ++t)
        if (d.dz[t] != 0 || d.dy[t] < -1 || d.dy[t] > 1 || d.dx[t] < -1 || d.dx[t] > 1) return false;
    if ((int64_t)d.N * (d.Hi + 1) * (d.Wi + 1) >= (1ll << 31) - (1 << 17)) return false;
    const WhCfg c = wh_config(d);
    return c.ok && d.Co >= c.bm && d.Ci % (c.nch * WH_CH) == 0;          // (no half-empty co tile; the carry bound is wh_config's)
}

// split count: the fullest last round of 256 - VD_WGRAD_HALO_RESERVE one-per-CU workgroups, as wgrad_pick_splits
int vd_wgrad_halo_splits(const vd_wgrad_desc& d) {
    if (d.splits > 0) return d.splits;
    static const int reserve = getenv("VD_WGRAD_HALO_RESERVE") ? atoi(getenv("VD_WGRAD_HALO_RESERVE"))
                               : (getenv("VD_WGRAD_RESERVE") ? atoi(getenv("VD_WGRAD_RESERVE")) : 0);
    const int64_t Q = (int64_t)d.N * (d.Hi + 1) * (d.Wi + 1);
    const WhCfg cfg = wh_config(d);
    const int64_t tiles = vd_cdiv(d.Co, cfg.bm) * (d.Ci / (cfg.nch * WH_CH));
    const int64_t slots = 256 - reserve;
    int64_t s = 1;
    double best = -1.0;
    const int64_t lo = (slots / tiles) > 1 ? (slots / tiles) : 1, hi = vd_cdiv(3 * slots, tiles);
    for (int64_t c = lo; c <= hi; ++c) {
        const double x = (double)(tiles * c) / (double)slots;
        const double fill = x / (double)vd_cdiv(tiles * c, slots);
        if (fill > best + 0.02) { best = fill; s = c; }
    }
    const int64_t maxs = vd_cdiv(Q, 16 * WH_BP);        // >= 512 positions per workgroup (the ring's first fill is 100-480)
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return (int)s;
}

void vd_wgrad_halo_launch(const vd_wgrad_desc& d, float* dst, int splits, hipStream_t s) {
    const WhCfg c = wh_config(d);
    const int64_t Q = (int64_t)d.N * (d.Hi + 1) * (d.Wi + 1);
    const int64_t qps = vd_cdiv(vd_cdiv(Q, splits), 2 * c.bp) * (2 * c.bp);      // even step counts
    const bool bf = d.flags & VD_STORE_BF16;
    if (c.bm == 256) {
        if (bf) launch<256, true>(d, c, dst, splits, qps, s);
        else launch<256, false>(d, c, dst, splits, qps, s);
    } else {
        if (bf) launch<128, true>(d, c, dst, splits, qps, s);
        else launch<128, false>(d, c, dst, splits, qps, s);
    }
}
