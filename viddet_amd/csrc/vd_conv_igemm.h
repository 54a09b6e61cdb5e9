// vd_conv_igemm.h - the tap-list implicit-GEMM kernel template (k_conv_igemm), its operand-split helpers and its launcher,
// shared by the translation units that instantiate it: vd_conv.hip (one workgroup per tile) and vd_conv_sk.hip (the
// persistent stream-K form).  Internal to the library (not part of the C-ABI).
#pragma once
#include "vd_common.h"
#include "vd_wgrad_halo.h"
#include <stdlib.h>

namespace {


#ifndef VD_WG_REMAP
#define VD_WG_REMAP 1
#endif
#ifndef VD_KORDER
#define VD_KORDER 1
#endif
#ifndef VD_SETPRIO
#define VD_SETPRIO 1
#endif
#ifndef VD_PROBE
#define VD_PROBE 0
#endif
#ifndef VD_PD
#define VD_PD 3
#endif
#ifndef VD_BNT
#define VD_BNT 0
#endif
#ifndef VD_KROT
#define VD_KROT 0
#endif
#ifndef VD_HALO_STAGGER
#define VD_HALO_STAGGER 0
#endif
constexpr int BK = 32;
constexpr int LDS_LD = 36;

// a resident page of zeros: the load target of padded / out-of-window rows (see the tap_off select in k_conv_igemm)
__device__ __attribute__((aligned(64))) float g_zero_page[64];

// developer build (-DVD_STAMP=1, tools/stamp_conv.py): wave 0 of one mid-grid workgroup records s_memtime at the
// phase boundaries of k_conv_igemm
#ifndef VD_STAMP
#define VD_STAMP 0
#endif
#if VD_STAMP
__device__ unsigned long long g_stamps_f32[16];
#define STAMP(i)                                                                          \
    do {                                                                                  \
        if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_stamps_f32[i] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define STAMP(i)
#endif

// intra-wave LDS hand-off: LDS ops of one wave execute in order, so only the compiler must be kept from
// reordering the accesses (no instruction is generated)
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));      // one 16-byte MFMA operand (8 bf16 or 8 fp16), type-agnostic

// ---- split-operand fp32 math (VD_MATH_SPLIT) -------------------------------------------------------------
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)  (round-to-nearest, the subtractions are
// exact): three bf16 pieces carry 24 significand bits, |m| <= 2^-8 |x|, |l| <= 2^-16 |x|.  A product a*b is then
// accumulated in fp32 from the six partial products ah*bh, ah*bm, am*bh, ah*bl, al*bh, am*bm on the bf16 matrix
// pipe (16x the fp32 MFMA rate); the three dropped terms are below 2^-23 |a*b| and of either sign - the size of
// one fp32 rounding.  LDS rows hold the three planes back to back, [h: 32 bf16][m: 32][l: 32] = 192 B, and the four
// 16-B slots of a plane are XOR-swizzled with bits 2..3 of the row: the 16 rows of a ds_read_b128 lane group then
// fall on 16 distinct slots of the 64 banks, and the two rows a ds_write_b64 lane group stores (8 lanes x 8 B each)
// fall on the two halves of the 32 write banks (192 B = 16 dwords mod 32) - both conflict-free without padding.

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);       // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ void split3(const f32x4 v, uint2& h, uint2& m, uint2& l) {
    unsigned hh[2], mm[2], ll[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = v[2 * q], x1 = v[2 * q + 1];
        const unsigned hp = pk_bf16(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, hp << 16);
        const float r1 = x1 - __builtin_bit_cast(float, hp & 0xffff0000u);
        const unsigned mp = pk_bf16(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, mp << 16);
        const float s1 = r1 - __builtin_bit_cast(float, mp & 0xffff0000u);
        hh[q] = hp; mm[q] = mp; ll[q] = pk_bf16(s0, s1);
    }
    h = make_uint2(hh[0], hh[1]); m = make_uint2(mm[0], mm[1]); l = make_uint2(ll[0], ll[1]);
}

// ---- two-way fp16 split (VD_MATH_F16X2) ---------------------------------------------------------------------
// With s the tensor's power-of-two scale (amax*s in [2^14, 2^15)): h = fp16(x*s), l = fp16(x*s - h), both round-to-
// nearest; x*s - h is exact in fp32.  A product is accumulated from al*bh, ah*bl, ah*bh (f16 MFMAs, fp32 accumulate);
// al*bl < 2^-22 |a*b| is dropped.  LDS rows hold the two planes back to back, [h: 32 fp16][l: 32] = 128 B = eight 16-B
// slots s = 4*plane + chunk, stored at slot s ^ key(row) (f16x2_key below): the 16 rows
// of every ds_read_b128 lane group (both MFMA operand maps) then fall on 16 distinct slots of the 64 banks, and the two
// rows of a ds_write_b64 lane group on the two halves of the 32 write banks - conflict-free without padding.
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    f32x2 v = {a, b};
    f16x2 r = __builtin_convertvector(v, f16x2);        // v_cvt_pk_f16_f32 (RNE) on gfx950
    return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ void split2(const f32x4 v, const float s, uint2& h, uint2& l) {
    unsigned hh[2], ll[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = v[2 * q] * s, x1 = v[2 * q + 1] * s;
        const unsigned hp = pk_f16(x0, x1);
        const f16x2 hv = __builtin_bit_cast(f16x2, hp);
        hh[q] = hp;
        ll[q] = pk_f16(x0 - (float)hv[0], x1 - (float)hv[1]);
    }
    h = make_uint2(hh[0], hh[1]); l = make_uint2(ll[0], ll[1]);
}

// M16 (16x16x32 operand map: a ds_read_b128 lane group takes 16 consecutive rows but TWO k-chunks, c for 8 of them and
// c + 1 for the other 8) needs a key that leaves slot bit 0 alone, or the halo loop's reads at shifted rows collide 2-way for
// 12 of the 16 row alignments (PMC: SQ_LDS_BANK_CONFLICT 1.2e7 of 1.6e7 LDS cycles on the first halo build); searched by
// brute force over lane groups x row offsets x planes, both keys are conflict-free for their map at ANY start row and for
// the ds_write_b64 stores.
template <bool M16>
__device__ __forceinline__ int f16x2_key(int row) {
    return M16 ? ((((row >> 1) & 3) << 1) ^ ((row & 1) << 2)) : (((row >> 1) & 7) ^ ((row & 1) << 2));
}

template <int NPL>
__device__ __forceinline__ f32x16 mfma32(const v4i a, const v4i b, const f32x16 c) {
    if (NPL == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int NPL>
__device__ __forceinline__ f32x4 mfma16(const v4i a, const v4i b, const f32x4 c) {
    if (NPL == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// partial products of one block, smallest first: plane indices of the A and B operands
template <int NPL> struct Terms;
template <> struct Terms<3> { static constexpr int N = 6; static constexpr int QA[6] = {1, 2, 0, 1, 0, 0}, QB[6] = {1, 0, 2, 0, 1, 0}; };
template <> struct Terms<2> { static constexpr int N = 3; static constexpr int QA[6] = {1, 0, 0, 0, 0, 0}, QB[6] = {0, 1, 0, 0, 0, 0}; };
template <> struct Terms<1> { static constexpr int N = 1; static constexpr int QA[6] = {0, 0, 0, 0, 0, 0}, QB[6] = {0, 0, 0, 0, 0, 0}; };

// n / d for d >= 1 with rcp = 0xFFFFFFFF / d + 1: the multiply-high overshoots the quotient by at most one
__device__ __forceinline__ unsigned udiv_rcp(unsigned n, unsigned d, unsigned rcp) {
    unsigned q = d == 1u ? n : __umulhi(n, rcp);
    q -= (q * d > n) ? 1u : 0u;
    return q;
}

struct RowInfo {
    int64_t off;     // element offset of (pixel of tap (0,0,0), channel lc4) in `in`
    unsigned mask;   // bit t set <=> tap t of this row lies inside the image / temporal window
};

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel
//   zd_in / zd_w: element offsets (relative to p.in / p.wp) of a resident page of zeros.  Rows that
//   fall outside the image, the temporal window, M or Co read that page, so every load of a K-step
//   is unconditional (one v_cndmask on the offset, no exec-masked branch per load) and the loop
//   body is one straight-line stream.
// ---------------------------------------------------------------------------------------------
// M16 (split math only): v_mfma_f32_16x16x32_bf16 instead of 32x32x16 - the same cycles per FLOP, but the chip holds a
// higher clock on it under load (MI355X_MICROARCH.md, DVFS item 7).  A 32x32 accumulator region is then four 16x16
// tiles kept in one f32x16 as [4*(2*sm+sn) + r]: rows 16*sm + 4*(lane>>4) + r, column 16*sn + (lane&15).
// BS: instantiation with the fused BatchNorm-backward reductions in the epilogue (their accumulators would cost the
// other instantiations registers, and the 8-wave fp32-MFMA tiles sit at the 128-VGPR occupancy edge)
// (second launch bound: the 8-wave fp32-MFMA tiles run two workgroups per CU = 4 waves per SIMD = 128 VGPRs)
// NPL: bf16 planes per operand in the split arithmetic - 3 = exact split, 6 partial products (fp32-accurate);
// 1 = only the leading piece, ONE bf16 MFMA per product block (VD_MATH_BF16: bf16-rounded operands, fp32 accumulate).
// HALO (NPL == 2, 3x3 stride-1 'same' geometry, forward and data gradient): the activation operand is staged ONCE per
// 32-channel chunk as a pixel halo tile - the BM output pixels of the tile plus W+1 pixels either side, split into fp16
// planes - and the nine taps read it at shifted rows, instead of gathering and splitting the same pixels nine times.
// Measured on the generic loop (timing probes, tools/conv_probe.py): the activation gather's vector-memory instructions
// cost 30 % of a 3x3 launch wherever they hit (L1, L2 or HBM) - the CU's load path, not the memory system, is the limit -
// so the fix is fewer bytes INTO the CU: (BM + 2W + 2) / (9 BM) of them.
// SK (NPL == 2): the persistent stream-K form (vd_conv_sk.hip).  One workgroup per CU slot owns a contiguous run of
// (tile, K-unit) units of its XCD group's tiles, equal in length to every other workgroup's run, so the chip finishes
// together whatever the tile count (676 tiles on 256 CUs were 3 rounds for 2.64 rounds of work).  A run cuts at most two
// tiles: the K-PREFIX of its last tile is computed FIRST and its raw accumulators handed to the next workgroup through
// global memory (write-through stores, a counter), the K-SUFFIX of its first tile LAST, starting from the accumulators the
// previous workgroup left - the same MFMA chain in the same order as an uncut tile, so every output is bit-identical to the
// one-workgroup-per-tile kernel's (nothing is summed across the seam).  The hand-off is needed a whole run after it was
// published; a bounded poll that gives up recomputes the prefix itself, so no dispatch order can hang the grid.
// PAR (NPL == 2, generic loop; vd_conv_par.hip): the data gradient of a 3x3 / stride-2 / pad-1 convolution as ONE launch
// (VD_CONV_PARITY4, include/viddet_hip.h).  GEMM rows are the positions q of the incoming gradient's grid, GEMM columns
// the four output parity classes x Cin (column block c = class (py, px) = (c >> 1, c & 1), output pixel (2 qy + py,
// 2 qx + px)), taps the four offsets {0, 1}^2 of a 2x2 window: the nine (class, kernel tap) pairs of the four parity
// launches it replaces are nine of the sixteen (offset, class) weight blocks, the other seven are zero and their MFMAs are
// skipped (wave-uniform test per 32-column block and K-step).  One gather and one LDS stage of dz instead of four.
template <int WM, int WN, int TM, int TN, bool XF, bool SP, bool M16 = false, bool BS = false, int NPL = 3, bool HALO = false, bool SK = false,
          bool PAR = false>
__global__ __launch_bounds__(WM * WN * 64, ((!SP && WM * WN == 8) ? 4 : ((SP && WM * WN == 4) ? 2 : 1))) void k_conv_igemm(const vd_conv_desc p, const int64_t zd_in, const int64_t zd_w, const int sk_lds_flag, const int sk_timeout) {
    static_assert(NPL == 3 || ((NPL == 1 || NPL == 2) && SP), "planes");
    static_assert(!SK || (NPL == 2 && VD_KORDER && !VD_KROT && !XF), "stream-K: the fp16-split tiles, taps-innermost K order");
    static_assert(!PAR || (NPL == 2 && !HALO && !SK && VD_KORDER && !VD_KROT), "parity-fused data gradient: fp16-split tiles, generic loop");
    static_assert(NPL != 2 || !XF, "the fp16 split needs the max-abs of the operand it splits: no in-load transform");
    static_assert(!HALO || (NPL == 2 && WM * WN == 8 && WN * TN * 32 >= 64), "the halo loop exists for the 8-wave fp16-split tiles");
    constexpr int SP_ROWB = NPL * 64;          // LDS row of the split arithmetic: NPL planes of 32 bf16 (fp16 for NPL == 2)
    constexpr int NTERM = Terms<NPL>::N;
    static_assert(!M16 || SP, "the 16x16x32 shape exists for the bf16 operands of the split arithmetic");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NT = WM * WN * 64;          // 4 or 8 waves
    constexpr int RPP = NT / 8;               // tile rows one pass of float4 lanes covers
    // BN < RPP (the 32-column split tile: 8 waves, 64 loader rows per pass): one weight pass in which only the lanes of
    // rows < BN take part - the others request the zero page and skip the LDS store (BHALF)
    constexpr bool BHALF = BN < RPP;
    constexpr int AP = BM / RPP, BP = BHALF ? 1 : BN / RPP;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    static_assert(BM % RPP == 0 && (BHALF || BN % RPP == 0), "tile vs loader");
    static_assert(!BHALF || SP, "the half-populated weight pass exists for the split-math tiles");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                        // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;      // [2][BN][LDS_LD]
    char* As3 = reinterpret_cast<char*>(smem);                  // SP: [2][BM][SP_ROWB]
    char* Bs3 = As3 + 2 * BM * SP_ROWB;                         // SP: [2][BN][SP_ROWB]

    const int64_t M = (int64_t)p.N * p.Hg * p.Wg;
    const int ntile = (p.Co + BN - 1) / BN;
    const int Ktot = p.T * p.Ci;
    // ---- work items of this workgroup.  !SK: one whole tile.  SK: [K-prefix of the run's last tile (published)], the whole
    // tiles, [K-suffix of the run's first tile (continues the previous workgroup's accumulators)]
    constexpr int KU = HALO ? 9 : 1;                       // K-steps per schedulable unit (HALO: one 32-channel chunk)
    const int upt = p.T * (p.Ci / BK) / KU;                // units per tile
    int sk_nitems = 1, sk_t0 = 0, sk_nwhole = 0, sk_brem = 0, sk_erem = 0, sk_tlast = 0, sk_tfirst = 0, sk_me = 0;
    if (SK) {
        // XCD groups (blocks b and b + 8 share an XCD): group g owns a contiguous range of whole tiles, its workgroups
        // equal shares of that range's units - every seam stays inside one group (one L2), and no chain of hand-offs
        // crosses groups
        const int G = (int)gridDim.x, g = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int gq = G >> 3, gr = G & 7;
        const int gs = gq + (g < gr ? 1 : 0);                              // workgroups of this group
        const int gbase = g * gq + (g < gr ? g : gr);                      // index of its first workgroup (seam ids)
        const int64_t ntiles_all = vd_cdiv(M, BM) * ntile;
        const int tq = (int)(ntiles_all >> 3), tr = (int)(ntiles_all & 7);
        const int tg0 = g * tq + (g < tr ? g : tr), tgn = tq + (g < tr ? 1 : 0);   // the group's tiles
        const int64_t U = (int64_t)tgn * upt;
        const int64_t ub = U * j / gs, ue = U * (j + 1) / gs;
        sk_brem = (int)(ub % upt);
        sk_erem = (int)(ue % upt);
        const int tb = (int)(ub / upt), te = (int)(ue / upt);              // first tile touched, tile holding the end
        sk_tfirst = tg0 + tb;
        sk_t0 = sk_tfirst + (sk_brem ? 1 : 0);
        sk_nwhole = te - (tb + (sk_brem ? 1 : 0));
        if (sk_nwhole < 0) sk_nwhole = 0;                                  // (a run inside one tile: the host rules it out)
        sk_tlast = tg0 + te;
        sk_nitems = (sk_erem ? 1 : 0) + sk_nwhole + (sk_brem ? 1 : 0);
        sk_me = gbase + j;
    }
    float amx = 0.f;                                        // max-abs of what this lane stores (p.amax_out)

    // NPL == 2: power-of-two operand scales from the tensors' max-abs (vd_common.h); requested first, used by the first
    // LDS store, so the two loads hide behind the row geometry
    int sexp_a = 0, sexp_b = 0;
    float scl_a = 1.f, scl_b = 1.f;
    if (NPL == 2) {
        sexp_a = vd_f16_scale_exp(vd_amax_read(p.amax_in));
        sexp_b = vd_f16_scale_exp(vd_amax_read(p.amax_w));
        scl_a = __uint_as_float((unsigned)(127 + sexp_a) << 23);
        scl_b = __uint_as_float((unsigned)(127 + sexp_b) << 23);
    }

    STAMP(0);
    // Row geometry (M < 2^31, checked on the host).  The tap table sits in lane registers (lane t = tap t, read back
    // with v_readlane), so neither the mask loop here nor the K loop pays a scalar-memory round trip per tap, and the
    // two divisions per row are a multiply-high by a reciprocal computed once (udiv_rcp).  The per-tap loop used to
    // cost 12k cycles of a workgroup's life: a quarter of the whole 32->64 3x3 layers.
    const int tlane = (int)(threadIdx.x & 63) < p.T ? (int)(threadIdx.x & 63) : 0;
    const int tap_dy = p.dy[tlane], tap_dx = p.dx[tlane], tap_dz = p.dz[tlane];
    const unsigned rcp_w = 0xFFFFFFFFu / (unsigned)p.Wg + 1u, rcp_h = 0xFFFFFFFFu / (unsigned)p.Hg + 1u;

    for (int sk_it = 0; sk_it < sk_nitems; ++sk_it) {
    // (SK: the thread id is made opaque once per item - otherwise everything the prologue and the epilogue derive from it is
    // loop-invariant, gets hoisted out of the item loop, lives across the K loop and is spilled: 1.1 KB of scratch per lane
    // and ~150 scratch accesses per item on the first build, 11 us per tile)
    int tid = threadIdx.x;
    if (SK) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3;
    const int lc4 = (tid & 7) * 4;
    int lid = SK ? 0 : vd_xcd_remap(blockIdx.x, gridDim.x);
    int kb = 0, ke = upt;                                   // unit range of this item
    int sk_mode = 0;                                        // 0 whole tile, 1 publish the prefix, 2 continue a prefix
    if (SK) {
        __syncthreads();                                    // the previous item's epilogue is done with LDS
        const bool hasA = sk_erem != 0;
        if (hasA && sk_it == 0) { lid = sk_tlast; ke = sk_erem; sk_mode = 1; }
        else if (sk_it - (hasA ? 1 : 0) < sk_nwhole) lid = sk_t0 + sk_it - (hasA ? 1 : 0);
        else { lid = sk_tfirst; kb = sk_brem; sk_mode = 2; }
    }
    // Tile order.  One workgroup per tile: column tiles fastest, so that the workgroups an XCD runs together (consecutive ids
    // after the XCD remap) share activation rows AND weight columns in its L2.  SK: a workgroup walks its run of tiles alone
    // and its XCD neighbours are a whole run apart, so the order that shares operands is the other one - row tiles
    // fastest: the workgroups of a group then sit in the same one or two column tiles and stream ONE weight panel through
    // their L2 (the first build, column tiles fastest, fetched 570 MB per launch of 256 -> 512 @26 where the one-tile
    // form fetches 162: PMC, profiles/r04_pmc_mfma_streamk.txt).
    const int mt_all = (int)vd_cdiv(M, BM);
    const int tile_n = SK ? lid / mt_all : lid % ntile, tile_m = SK ? lid % mt_all : lid / ntile;
    RowInfo ri[AP];
    {
        int riy[AP], rix[AP], rfz[AP];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int64_t m = (int64_t)tile_m * BM + lrow + RPP * i;
            const unsigned mu = m < M ? (unsigned)m : 0u;
            const unsigned t = udiv_rcp(mu, (unsigned)p.Wg, rcp_w);
            const int gx = (int)(mu - t * (unsigned)p.Wg);
            const unsigned n_ = udiv_rcp(t, (unsigned)p.Hg, rcp_h);
            const int gy = (int)(t - n_ * (unsigned)p.Hg);
            const int n = (int)n_;
            riy[i] = gy * p.in_stride;
            rix[i] = gx * p.in_stride;
            rfz[i] = p.Kfr == 1 ? 0 : n % p.Kfr;
            ri[i].off = (int64_t)((n * p.Hi + riy[i]) * p.Wi + rix[i]) * p.Ci + lc4;
            ri[i].mask = 0u;
        }
        for (int t2 = 0; t2 < p.T; ++t2) {
            const int dy = __builtin_amdgcn_readlane(tap_dy, t2), dx = __builtin_amdgcn_readlane(tap_dx, t2),
                      dz = __builtin_amdgcn_readlane(tap_dz, t2);
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const bool ok = (unsigned)(riy[i] + dy) < (unsigned)p.Hi && (unsigned)(rix[i] + dx) < (unsigned)p.Wi &&
                                (unsigned)(rfz[i] + dz) < (unsigned)p.Kfr;
                ri[i].mask |= ok ? (1u << t2) : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < AP; ++i)
            if ((int64_t)tile_m * BM + lrow + RPP * i >= M) ri[i].mask = 0u;
    }
    int64_t boff[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = tile_n * BN + lrow + RPP * i;
        boff[i] = (n < p.Co && (!BHALF || lrow < BN)) ? (int64_t)n * Ktot + lc4 : (int64_t)-1;   // -1: reads zeros
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    if (SK) {
        if (sk_mode == 2) {
            // the K-suffix of a tile whose prefix the previous workgroup of this group computed at the START of its run:
            // one lane polls that seam's counter (relaxed, bounded), one agent-scope acquire drops this CU's stale lines,
            // then every lane loads its accumulators as they were stored.  A poll that gives up (the producer is not
            // resident yet: another kernel holds its CU) computes the whole tile instead - the same chain, the same bits.
            unsigned* cnt = reinterpret_cast<unsigned*>(p.sk_ws);
            unsigned* seen = cnt + VD_SK_MAX_WG;
            int* lds_flag = reinterpret_cast<int*>(reinterpret_cast<char*>(smem) + sk_lds_flag);
            if (tid == 0) {
                const unsigned target = __hip_atomic_load(seen + sk_me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                int ok = 0;
                for (;;) {
                    const unsigned c = __hip_atomic_load(cnt + sk_me - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(c - target) >= 0 && sk_timeout > 0) { ok = 1; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)sk_timeout) break;
                    __builtin_amdgcn_s_sleep(4);
                }
                __hip_atomic_store(seen + sk_me, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!ok) __hip_atomic_fetch_add(cnt + VD_SK_MAX_WG - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // diagnostics: polls given up
                *lds_flag = ok;
                if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int ok = *lds_flag;
            if (ok) {
                const f32x4* slot = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.sk_ws) + VD_SK_HEADER_BYTES +
                                                                   (int64_t)(sk_me - 1) * (BM * BN * 4));
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = slot[((mi * TN + ni) * 4 + q) * NT + tid];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mi][ni][4 * q + e] = v[e];
                        }
            } else {
                kb = 0;
                sk_mode = 0;
            }
        }
    }

    // register sets = K-steps of global-load latency cover (the 4-wave 128x128 tile has no VGPRs left for a third)
    constexpr int PD = (VD_PD > 2 && ((WM * WN == 4 && TM * TN == 4) || (M16 && TM * TN == 4))) ? 2 : VD_PD;
    f32x4 ra[PD][AP], rb[PD][BP];
    // k-step cursor of the NEXT tile to load.  VD_KROT: every M tile starts the channel-chunk loop at a different chunk (and
    // wraps), so that at any moment the workgroups of the chip read different 128-B pieces of the 4*Ci-byte pixel records
    // instead of all the same one (memory-channel balance); a tile's sum order is rotated, not its value set
    int t_tap = 0, c0 = VD_KROT ? (int)((unsigned)tile_m % (unsigned)(p.Ci / BK)) * BK : 0;
    if (SK && !HALO) {                                     // taps innermost: K-step ks = (chunk ks / T, tap ks % T)
        c0 = (kb / p.T) * BK;
        t_tap = kb - (kb / p.T) * p.T;
    }

    // tap part of the wave-uniform source offset: refreshed only when the tap changes (every Ci/32 K-steps), so the
    // scalar loads of dy/dx/dz and their s_waitcnt leave the per-step critical path
    const int64_t tap_eo = (int64_t)((tap_dz * p.Hi + tap_dy) * p.Wi + tap_dx) * p.Ci;   // lane t: tap t
    const int tap_eo_lo = (int)(tap_eo & 0xffffffffll), tap_eo_hi = (int)(tap_eo >> 32);
    auto tap_off = [&](int t) -> int64_t {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(tap_eo_lo, t);
        const int hi = __builtin_amdgcn_readlane(tap_eo_hi, t);
        return ((int64_t)hi << 32) | (int64_t)lo;
    };
    int64_t tap_soff = tap_off(t_tap);
    auto gload = [&](f32x4 (&ra)[AP], f32x4 (&rb)[BP]) {
        const int64_t soff = tap_soff + c0;     // wave-uniform
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (XF) {
            sc = *reinterpret_cast<const f32x4*>(p.in_scale + c0 + lc4);
            sh = *reinterpret_cast<const f32x4*>(p.in_shift + c0 + lc4);
        }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool ok = (ri[i].mask >> t_tap) & 1u;
            const int64_t o = ri[i].off + soff;
            const int64_t sel = (ok && !(VD_PROBE & 8) && !((VD_PROBE & 32) && t_tap != 0)) ? o : zd_in;   // probe bit 3: every A request hits the zero page; bit 5: all but tap 0's
            f32x4 v = *reinterpret_cast<const f32x4*>(p.in + sel);
            if (XF) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[e] * sc[e] + sh[e];
                    t = t > 0.f ? t : t * p.in_slope;
                    v[e] = ok ? t : 0.f;
                }
            }
            ra[i] = v;
        }
        const int64_t koff = (int64_t)t_tap * p.Ci + c0;
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int64_t o = boff[i] + koff;
            const int64_t sel = (boff[i] >= 0 && !(VD_PROBE & 16)) ? o : zd_w;   // probe bit 4: same for the weights
#if VD_BNT
            rb[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.wp + sel));   // weights bypass the CU's L1
#else
            rb[i] = *reinterpret_cast<const f32x4*>(p.wp + sel);
#endif
        }
#if VD_KORDER
        // taps innermost: the T taps of one 32-channel chunk touch (almost) the same pixels, shifted
        ++t_tap;
        if (t_tap >= p.T) {
            t_tap = 0;
            c0 += BK;
            if (VD_KROT && c0 >= p.Ci) c0 = 0;
        }
        tap_soff = tap_off(t_tap);
#else
        c0 += BK;
        if (c0 >= p.Ci) {
            c0 = 0;
            ++t_tap;
            if (t_tap < p.T) tap_soff = tap_off(t_tap);
        }
#endif
    };
    auto lstore = [&](int buf, const f32x4 (&ra)[AP], const f32x4 (&rb)[BP]) {
        if (SP && NPL == 2) {
            const int key = f16x2_key<M16>(lrow);              // lrow + RPP * i has the same low four bits (RPP = 32 or 64)
            const int c = (tid & 7) >> 1, half = (tid & 1) << 3;
            const int oh = ((c ^ key) << 4) + half, ol = (((4 + c) ^ key) << 4) + half;
            char* a3 = As3 + buf * BM * SP_ROWB;
            char* b3 = Bs3 + buf * BN * SP_ROWB;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                uint2 h, l;
                split2(ra[i], scl_a, h, l);
                char* r = a3 + (lrow + RPP * i) * SP_ROWB;
                *reinterpret_cast<uint2*>(r + oh) = h;
                *reinterpret_cast<uint2*>(r + ol) = l;
            }
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                if (BHALF && lrow >= BN) break;           // wave-uniform: these lanes hold no weight row
                uint2 h, l;
                split2(rb[i], scl_b, h, l);
                char* r = b3 + (lrow + RPP * i) * SP_ROWB;
                *reinterpret_cast<uint2*>(r + oh) = h;
                *reinterpret_cast<uint2*>(r + ol) = l;
            }
            return;
        }
        if (SP) {
            // swizzled slot + half; key = row bits 2..3 for the 32-row operand map, 2 * row bit 3 for the 16-row one
            // (both make the 16 rows of every ds_read_b128 lane group hit 16 distinct slots)
            const int wkey = M16 ? 2 * ((lrow >> 3) & 1) : ((lrow >> 2) & 3);
            const int wsl = ((((tid & 7) >> 1) ^ wkey) << 4) + ((tid & 1) << 3);
            char* a3 = As3 + buf * BM * SP_ROWB + wsl;
            char* b3 = Bs3 + buf * BN * SP_ROWB + wsl;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                uint2 h, m, l;
                char* r = a3 + (lrow + RPP * i) * SP_ROWB;
                if (NPL == 3) {
                    split3(ra[i], h, m, l);
                    *reinterpret_cast<uint2*>(r + 64) = m;
                    *reinterpret_cast<uint2*>(r + 128) = l;
                } else {
                    h = make_uint2(pk_bf16(ra[i][0], ra[i][1]), pk_bf16(ra[i][2], ra[i][3]));
                }
                *reinterpret_cast<uint2*>(r) = h;
            }
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                if (BHALF && lrow >= BN) break;           // wave-uniform: these lanes hold no weight row
                uint2 h, m, l;
                char* r = b3 + (lrow + RPP * i) * SP_ROWB;
                if (NPL == 3) {
                    split3(rb[i], h, m, l);
                    *reinterpret_cast<uint2*>(r + 64) = m;
                    *reinterpret_cast<uint2*>(r + 128) = l;
                } else {
                    h = make_uint2(pk_bf16(rb[i][0], rb[i][1]), pk_bf16(rb[i][2], rb[i][3]));
                }
                *reinterpret_cast<uint2*>(r) = h;
            }
            return;
        }
        float* a = As + buf * BM * LDS_LD;
        float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < AP; ++i)
            *reinterpret_cast<f32x4*>(a + (lrow + RPP * i) * LDS_LD + lc4) = ra[i];
#pragma unroll
        for (int i = 0; i < BP; ++i)
            *reinterpret_cast<f32x4*>(b + (lrow + RPP * i) * LDS_LD + lc4) = rb[i];
    };
    // PAR: tap (offset) of the K-step about to be multiplied, and the 16 (offset, class) bits of the nonzero weight blocks
    int ctap = 0;
    const unsigned par_mask = PAR ? (unsigned)p.par_mask : 0xffffu;
    const int par_col0 = tile_n * BN + wn * TN * 32;              // first column of this wave
    auto par_nz = [&](int col) -> bool {                          // is the block of column `col` nonzero for tap ctap?
        return !PAR || ((par_mask >> (ctap * 4 + col / p.par_cin)) & 1u);
    };
    auto compute = [&](int buf) {
        if (PAR) {
            bool any = false;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) any = any || par_nz(par_col0 + ni * 32);
            if (!any) {                                           // every block of this wave's columns is zero for this offset
                if (++ctap >= 4) ctap = 0;
                return;
            }
        }
        if (SP && M16) {
            const int r16 = lane & 15, ch = lane >> 4;              // operand row within a 16-row block, 8-k chunk
            const int rkey = NPL == 2 ? f16x2_key<M16>(r16) : 2 * ((r16 >> 3) & 1);
            // byte offset of plane q's chunk `ch` inside a row
            auto slot = [&](int q) { return NPL == 2 ? (((4 * q + ch) ^ rkey) << 4) : (q * 64 + ((ch ^ rkey) << 4)); };
            const char* a3 = As3 + (buf * BM + wm * TM * 32 + r16) * SP_ROWB;
            const char* b3 = Bs3 + (buf * BN + wn * TN * 32 + r16) * SP_ROWB;
            v4i fa[2 * TM][NPL];
#pragma unroll
            for (int mb = 0; mb < 2 * TM; ++mb)
#pragma unroll
                for (int q = 0; q < NPL; ++q)
                    fa[mb][q] = *reinterpret_cast<const v4i*>(a3 + mb * 16 * SP_ROWB + slot(q));
#pragma unroll
            for (int nb = 0; nb < 2 * TN; ++nb) {
                if (PAR && !par_nz(par_col0 + nb * 16)) continue;
                v4i fb[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) fb[q] = *reinterpret_cast<const v4i*>(b3 + nb * 16 * SP_ROWB + slot(q));
#if VD_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int mb = 0; mb < 2 * TM; ++mb) {
                    f32x16& A_ = acc[mb >> 1][nb >> 1];
                    const int e0 = 4 * (2 * (mb & 1) + (nb & 1));
                    f32x4 c = {A_[e0], A_[e0 + 1], A_[e0 + 2], A_[e0 + 3]};
#pragma unroll
                    for (int t = 0; t < NTERM; ++t)
                        c = mfma16<NPL>(fa[mb][Terms<NPL>::QA[t]], fb[Terms<NPL>::QB[t]], c);
                    A_[e0] = c[0]; A_[e0 + 1] = c[1]; A_[e0 + 2] = c[2]; A_[e0 + 3] = c[3];
                }
#if VD_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
            if (PAR) { if (++ctap >= 4) ctap = 0; }
            return;
        }
        if (SP) {
            const char* a3 = As3 + (buf * BM + wm * TM * 32 + (lane & 31)) * SP_ROWB;
            const char* b3 = Bs3 + (buf * BN + wn * TN * 32 + (lane & 31)) * SP_ROWB;
            const int swz = NPL == 2 ? f16x2_key<M16>(lane & 31) : ((lane >> 2) & 3), hh = lane >> 5;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                v4i fa[TM][NPL], fb[TN][NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int so = NPL == 2 ? (((4 * q + kc * 2 + hh) ^ swz) << 4) : (q * 64 + (((kc * 2 + hh) ^ swz) << 4));
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        fa[mi][q] = *reinterpret_cast<const v4i*>(a3 + mi * 32 * SP_ROWB + so);
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        fb[ni][q] = *reinterpret_cast<const v4i*>(b3 + ni * 32 * SP_ROWB + so);
                }
                // smallest partial products first (Terms<NPL>)
#if VD_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int t = 0; t < NTERM; ++t)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni) {
                            if (PAR && !par_nz(par_col0 + ni * 32)) continue;
                            acc[mi][ni] = mfma32<NPL>(fa[mi][Terms<NPL>::QA[t]], fb[ni][Terms<NPL>::QB[t]], acc[mi][ni]);
                        }
#if VD_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
            if (PAR) { if (++ctap >= 4) ctap = 0; }
            return;
        }
        const float* a = As + buf * BM * LDS_LD + (wm * TM * 32 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
        const float* b = Bs + buf * BN * LDS_LD + (wn * TN * 32 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                fa[mi] = *reinterpret_cast<const f32x4*>(a + mi * 32 * LDS_LD + kc * 8);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                fb[ni] = *reinterpret_cast<const f32x4*>(b + ni * 32 * LDS_LD + kc * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][j], fb[ni][j],
                                                                           acc[mi][ni], 0, 0, 0);
        }
    };

    const int nks = SK ? (ke - kb) * KU : p.T * (p.Ci / BK);     // K-steps of this item
    STAMP(1);
    if constexpr (HALO) {
        // ---- LDS: two halo buffers [R + 2 rows][128 B] (row R = zeros for taps outside the image, row R + 1 = a sink for
        // the stream's idle slots), then the ring of three weight stages [BN][128 B]
        constexpr int HT = 9;                                  // taps (3x3), checked on the host
        const int W = p.Wi;
        const int R = BM + 2 * (W + 1);
        const int ZROW = R, DROW = R + 1;
        const int ABUF = (R + 2) * 128;
        char* Ah = As3;
        char* Bh = As3 + 2 * ABUF;
        const int nchunk = SK ? ke : p.Ci / BK;               // (end of this item's chunk range)
        const int cbeg = SK ? kb : 0;                          // first chunk of this item
        const int64_t m0 = (int64_t)tile_m * BM;
        const int64_t Mtot = (int64_t)p.N * p.Hi * p.Wi;       // == M in this geometry
        if (tid < 16) *reinterpret_cast<f32x4*>(Ah + (tid >> 3) * ABUF + ZROW * 128 + (tid & 7) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NRB = M16 ? 2 * TM : TM;                 // operand row blocks per wave (16 or 32 rows each)
        constexpr int RBS = M16 ? 16 : 32;
        int jbase[NRB];                                      // filled in the prologue, under the first loads' latency
        unsigned amask[NRB];
        const int tap_ro = tap_dy * W + tap_dx;                // lane t: halo-row offset of tap t
        // ---- halo stream: item (chunk c, slot s) = rows lrow + 64 s of chunk c's halo, one float4 per thread.  Everything
        // that depends only on the thread is hoisted: the slot's validity (9 bits), the element offset of slot 0, the LDS
        // byte offset of slot 0 and the swizzle key (64 s leaves the low four row bits alone; the sink row takes whatever
        // slot the key gives it)
        const int hcs = (tid & 7) >> 1, hhalf = (tid & 1) << 3;
        unsigned hvalid = 0u;
#pragma unroll
        for (int s2 = 0; s2 < HT; ++s2) {
            const int j = lrow + 64 * s2;
            const int64_t pin = m0 - (W + 1) + j;
            hvalid |= (j < R && (uint64_t)pin < (uint64_t)Mtot) ? (1u << s2) : 0u;
        }
        const int64_t hoff0 = (m0 - (W + 1) + lrow) * (int64_t)p.Ci + lc4;      // element offset of (slot 0, chunk 0)
        const int64_t hslot = 64ll * p.Ci;                                      // elements between slots
        const int hkey = f16x2_key<M16>(lrow);
        const int hl0 = lrow * 128 + hhalf + ((hcs ^ hkey) << 4);               // LDS byte offset of the h piece, slot 0
        const int hl1 = lrow * 128 + hhalf + (((4 + hcs) ^ hkey) << 4);         // ... of the l piece
        const int hsink = (DROW - lrow) * 128;                                  // hl0 + hsink lies in the sink row
        auto hload = [&](int c, int s) -> f32x4 {
            const bool ok = ((hvalid >> s) & 1u) && c < nchunk;
            const int64_t sel = ok ? hoff0 + (int64_t)s * hslot + (int64_t)c * BK : zd_in;
            return *reinterpret_cast<const f32x4*>(p.in + sel);
        };
        auto hstore = [&](const f32x4 v, int c, int s) {
            const int ro_ = (c & 1) * ABUF + (((hvalid >> s) & 1u) ? s * 8192 : hsink);
            uint2 h, l;
            split2(v, scl_a, h, l);
            *reinterpret_cast<uint2*>(Ah + ro_ + hl0) = h;
            *reinterpret_cast<uint2*>(Ah + ro_ + hl1) = l;
        };
        // ---- weights: the generic loop's requests / stores, weight operand only
        int bt = 0, bc0 = cbeg * BK;
        auto gloadB = [&](f32x4 (&rb)[BP]) {
            const int64_t koff = (int64_t)bt * p.Ci + bc0;
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int64_t sel = boff[i] >= 0 ? boff[i] + koff : zd_w;
                rb[i] = *reinterpret_cast<const f32x4*>(p.wp + sel);
            }
            if (++bt >= HT) { bt = 0; bc0 += BK; }
        };
        auto lstoreB = [&](int buf, const f32x4 (&rb)[BP]) {
            const int key = f16x2_key<M16>(lrow);
            const int oh = ((hcs ^ key) << 4) + hhalf, ol = (((4 + hcs) ^ key) << 4) + hhalf;
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                uint2 h, l;
                split2(rb[i], scl_b, h, l);
                char* r = Bh + (buf * BN + lrow + RPP * i) * 128;
                *reinterpret_cast<uint2*>(r + oh) = h;
                *reinterpret_cast<uint2*>(r + ol) = l;
            }
        };
        // byte offsets (within Ah) of the lane's operand rows for one (halo buffer, tap), slot key folded in; computed one
        // K-step ahead so that a step opens with its LDS reads, not with their address arithmetic
        const int lsel = M16 ? (lane >> 4) : (lane >> 5);
        auto hrows = [&](int (&av)[NRB], int abuf, int tap) {
            const int ro = __builtin_amdgcn_readlane(tap_ro, tap);
#pragma unroll
            for (int b = 0; b < NRB; ++b) {
                const int j = ((amask[b] >> tap) & 1u) ? jbase[b] + ro : ZROW;
                av[b] = abuf * ABUF + j * 128 + ((f16x2_key<M16>(j) ^ lsel) << 4);
            }
        };
        // ---- MFMA operand fragments, software-pipelined: a K-step never opens with exposed LDS latency.  The measured
        // ceiling of the [8 reads -> wait -> 12 MFMAs] x 2 step (a bare read + MFMA loop) was ~50 % of the matrix pipe: the
        // two waves of a SIMD leave the barrier together and wait for their reads together.  Here the reads of half-step
        // h + 1 are issued before the MFMAs of half-step h (for the second half of a step that is the NEXT K-step's first
        // half: its weight tile sits in the third ring slot, stored a step ago and published by the last barrier).
        const int swz = f16x2_key<M16>(lane & (RBS - 1));
        auto fragsA = [&](v4i (&fa)[NRB][2], const int (&av)[NRB], int kc) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int b2 = 0; b2 < NRB; ++b2)
                    fa[b2][q] = *reinterpret_cast<const v4i*>(Ah + (av[b2] ^ ((4 * q + (M16 ? 0 : 2 * kc)) << 4)));
        };
        // non-M16: the TN 32-column blocks of k-half kc; M16: ONE 16-column block nb (all 32 k)
        auto fragsB = [&](v4i (&fb)[M16 ? 1 : TN][2], int slot, int sub) {
            const char* b3 = Bh + (slot * BN + wn * TN * 32 + (lane & (RBS - 1))) * 128 + (M16 ? sub * 16 * 128 : 0);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int ni = 0; ni < (M16 ? 1 : TN); ++ni)
                    fb[ni][q] = *reinterpret_cast<const v4i*>(b3 + ni * 32 * 128 + (((4 * q + (M16 ? 0 : 2 * sub) + lsel) ^ swz) << 4));
        };
        auto mm32 = [&](const v4i (&fa)[NRB][2], const v4i (&fb)[M16 ? 1 : TN][2]) {
#if VD_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = mfma32<2>(fa[mi][Terms<2>::QA[t]], fb[M16 ? 0 : ni][Terms<2>::QB[t]], acc[mi][ni]);
#if VD_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        auto mm16 = [&](const v4i (&fa)[NRB][2], const v4i (&fb)[M16 ? 1 : TN][2], int nb) {
#if VD_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int mb = 0; mb < 2 * TM; ++mb) {
                f32x16& A_ = acc[mb >> 1][nb >> 1];
                const int e0 = 4 * (2 * (mb & 1) + (nb & 1));
                f32x4 c = {A_[e0], A_[e0 + 1], A_[e0 + 2], A_[e0 + 3]};
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    c = mfma16<2>(fa[M16 ? mb : 0][Terms<2>::QA[t]], fb[0][Terms<2>::QB[t]], c);
                A_[e0] = c[0]; A_[e0 + 1] = c[1]; A_[e0 + 2] = c[2]; A_[e0 + 3] = c[3];
            }
#if VD_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        // ---- prologue: the whole halo of chunk 0 and the first two weight tiles are requested TOGETHER, the operand-row
        // geometry below is computed under their latency, then everything is split and stored (one exposed round trip per
        // tile instead of three; the K-sweep's fixed cost did not move with it - 39 us vs 37 us at 512 @26 - so the
        // launch's fixed part is cold first touches and the tail, not these round trips)
        f32x4 t9[HT];
#pragma unroll
        for (int s2 = 0; s2 < HT; ++s2) t9[s2] = hload(cbeg, s2);
        gloadB(rb[0]);
        gloadB(rb[1]);
        // ---- per MFMA operand row: halo row of the centre tap and the 9-bit mask of taps inside the image
#pragma unroll
        for (int b = 0; b < NRB; ++b) {
            const int il = wm * TM * 32 + b * RBS + (lane & (RBS - 1));       // output pixel within the tile
            const int64_t m = m0 + il;
            const unsigned mu = m < M ? (unsigned)m : 0u;
            const unsigned t = udiv_rcp(mu, (unsigned)p.Wg, rcp_w);
            const int gx = (int)(mu - t * (unsigned)p.Wg);
            const unsigned n_ = udiv_rcp(t, (unsigned)p.Hg, rcp_h);
            const int gy = (int)(t - n_ * (unsigned)p.Hg);
            unsigned mk = 0u;
            for (int t2 = 0; t2 < HT; ++t2) {
                const int dy = __builtin_amdgcn_readlane(tap_dy, t2), dx = __builtin_amdgcn_readlane(tap_dx, t2);
                mk |= ((unsigned)(gy + dy) < (unsigned)p.Hi && (unsigned)(gx + dx) < (unsigned)p.Wi) ? (1u << t2) : 0u;
            }
            amask[b] = m < M ? mk : 0u;
            jbase[b] = il + W + 1;
        }
#pragma unroll
        for (int s2 = 0; s2 < HT; ++s2) hstore(t9[s2], cbeg, s2);
        lstoreB(0, rb[0]);
        lstoreB(1, rb[1]);
        __syncthreads();
        constexpr int HD = 3;                                  // halo items in flight (one request and one store per K-step)
        f32x4 hv[HD];
        int hc = cbeg + 1, hs = 0;                             // next item to request (chunk, slot)
        int sc = cbeg + 1, ss = 0;                             // next item to store
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            hv[d] = hload(hc, hs);
            if (++hs >= HT) { hs = 0; ++hc; }
        }
        // weight tile t travels: requested at step t - PD - 1 (set t % PD), stored at step t - 2 (ring slot t % 3), read from
        // step t - 1 on
#pragma unroll
        for (int d = 2; d <= PD; ++d)
            if (d < nks) gloadB(rb[d % PD]);
        constexpr int UNH = 6;                                 // = lcm(3 ring slots, PD in {2, 3}, HD)
        static_assert(UNH % PD == 0 && UNH % HD == 0 && UNH % 3 == 0, "unroll vs register sets / ring");
        int cc = cbeg, ct = 0;                                 // chunk / tap of the step being multiplied
        int avc[NRB];
        hrows(avc, cbeg & 1, 0);
        v4i FA0[NRB][2], FA1[NRB][2], FB0[M16 ? 1 : TN][2], FB1[M16 ? 1 : TN][2];
        fragsA(FA0, avc, 0);
        fragsB(FB0, 0, 0);
        // The two waves that share a SIMD (w, w + 4) run a step's two phases in opposite order: the "late" wave stores the
        // weight tile and its halo item and renews its requests FIRST and multiplies second, so its VALU / LDS phase overlaps
        // the partner's MFMAs instead of both leaving the barrier into their MFMAs together and both idling the pipe while
        // they split and store.  Built with -DVD_HALO_STAGGER=1 only: measured +-2 % against the lockstep form on every tile
        // (gpurun_out r2h), so the default keeps the smaller loop body.
        const bool late = VD_HALO_STAGGER && (__builtin_amdgcn_readfirstlane(wave) & 4);
        auto hstores = [&](int u, bool do_store) {
            if (do_store) lstoreB((u + 2) % 3, rb[(u + 2) % PD]);
            // the halo stream: store the item requested HD steps ago (into the NEXT chunk's buffer: nobody reads it before the
            // barrier that ends step 7 of this chunk; slot 8 is always a sink slot because R <= 512), then reuse its registers
            // for a new request.  Items past the last chunk read the zero page and land in the sink row: no conditional
            // request or store in the stream.
            hstore(hv[u % HD], sc, ss);
            if (++ss >= HT) { ss = 0; ++sc; }
            hv[u % HD] = hload(hc, hs);
            if (++hs >= HT) { hs = 0; ++hc; }
        };
        auto hstep = [&](int u, bool has_next, bool do_req, bool do_store) {
            if (late) hstores(u, do_store);
            if (!M16) {
                fragsA(FA1, avc, 1);                            // (this step, k-half 1)
                fragsB(FB1, u % 3, 1);
                if (do_req) gloadB(rb[(u + 1) % PD]);
                mm32(FA0, FB0);
                if (++ct >= HT) { ct = 0; ++cc; }
                hrows(avc, cc & 1, ct);
                if (has_next) {                                 // (next step, k-half 0)
                    fragsA(FA0, avc, 0);
                    fragsB(FB0, (u + 1) % 3, 0);
                }
                mm32(FA1, FB1);
            } else {
                // 16x16x32: a step is 2 TN column blocks of 16; the A fragments of the whole step are FA0, the next step's
                // are read into FA1 under the second block and swapped by name below
                fragsB(FB1, u % 3, 1);
                if (do_req) gloadB(rb[(u + 1) % PD]);
                mm16(FA0, FB0, 0);
                if (++ct >= HT) { ct = 0; ++cc; }
                int avn[NRB];
                hrows(avn, cc & 1, ct);
#pragma unroll
                for (int nb = 1; nb < 2 * TN; ++nb) {
                    if (nb + 1 < 2 * TN) fragsB((nb & 1) ? FB0 : FB1, u % 3, nb + 1);
                    else if (has_next) fragsB((nb & 1) ? FB0 : FB1, (u + 1) % 3, 0);
                    if (nb == 1 && has_next) fragsA(FA1, avn, 0);
                    mm16(FA0, (nb & 1) ? FB1 : FB0, nb);
                }
#pragma unroll
                for (int b2 = 0; b2 < NRB; ++b2) {
                    FA0[b2][0] = FA1[b2][0];
                    FA0[b2][1] = FA1[b2][1];
                    avc[b2] = avn[b2];
                }
            }
            if (!late) hstores(u, do_store);
            __syncthreads();
        };
        int ks = 0;
        for (; ks + UNH + PD + 1 <= nks; ks += UNH) {
#pragma unroll
            for (int u = 0; u < UNH; ++u) hstep(u, true, true, true);
        }
        for (; ks < nks; ks += UNH) {
#pragma unroll
            for (int u = 0; u < UNH; ++u)
                if (ks + u < nks) hstep(u, ks + u + 1 < nks, ks + u + PD + 1 < nks, ks + u + 2 < nks);
        }
    } else {
    gload(ra[0], rb[0]);
    STAMP(2);
    lstore(0, ra[0], rb[0]);
    __syncthreads();
    STAMP(3);
    // timing probes (compile with -DVD_PROBE=bits; results are garbage): bit0 skip the global loads, bit1 skip the
    // LDS stores (and the operand split), bit2 skip the per-step barrier, bit3 / bit4 every activation / weight request
    // reads the resident zero page (the requests are issued, the memory system is not exercised)
    constexpr bool ld = !(VD_PROBE & 1), st = !(VD_PROBE & 2), bar = !(VD_PROBE & 4);
    // PD K-steps of global-load latency cover with PD register sets (tile t lives in set t % PD): tile ks+PD is
    // requested while tile ks is multiplied, and the registers of tile ks+1 (requested PD-1 steps earlier, so the
    // wait is a counted vmcnt that leaves the newer requests in flight) are written to LDS at the end of the
    // step.  One K-step of cover (~1.9 us of MFMA work) left HBM latency exposed under load.
    // The steady-state loop is branch-free: with a guard on any request or store the compiler cannot prove which
    // requests are still in flight where paths merge, and drains them all (vmcnt(0)) before every new request.
    constexpr int UN = (PD % 2 == 0) ? PD : 2 * PD;        // unroll so that set and LDS-buffer indices are static
    constexpr bool STAGGER = SP && (WM * WN == 8);
    const bool stag = (wave >> 2) & 1;
#pragma unroll
    for (int d = 1; d < PD; ++d)
        if (d < nks && ld) gload(ra[d], rb[d]);
    int ks = 0;
    if (nks >= PD) {                                       // (else the prologue requests above were partial)
        for (; ks + UN + PD <= nks; ks += UN) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (ld) gload(ra[u % PD], rb[u % PD]);
                // the two waves that share a SIMD (w, w+4) run the step's two phases in opposite order, so one
                // wave's operand split + LDS stores overlap the other's MFMAs instead of idling the matrix pipe
                if (STAGGER && stag) {
                    if (st) lstore((u + 1) & 1, ra[(u + 1) % PD], rb[(u + 1) % PD]);
                    compute(u & 1);
                } else {
                    compute(u & 1);
                    if (st) lstore((u + 1) & 1, ra[(u + 1) % PD], rb[(u + 1) % PD]);
                }
                if (bar) __syncthreads();
            }
        }
    }
    for (; ks < nks; ks += UN) {                           // the last < UN + PD steps, guarded
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (ks + u < nks) {
                if (ks + u + PD < nks && ld) gload(ra[u % PD], rb[u % PD]);
                compute(u & 1);
                if (ks + u + 1 < nks && st) lstore((u + 1) & 1, ra[(u + 1) % PD], rb[(u + 1) % PD]);
                if (bar) __syncthreads();
            }
        }
    }

    }   // !HALO
    STAMP(4);
    if (SK) {
        if (sk_mode == 1) {
            // publish the K-prefix: raw accumulators (the operand scales are undone by whoever finishes the tile), 16 bytes
            // per lane and store, write-through (sc1) so that no release fence has to write back this XCD's L2; every
            // storing wave drains its stores, the workgroup meets, one lane bumps the seam's counter
            char* slot = reinterpret_cast<char*>(p.sk_ws) + VD_SK_HEADER_BYTES + (int64_t)sk_me * (BM * BN * 4);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slot, 0, BM * BN * 4, 0x00027000);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), rs, (((mi * TN + ni) * 4 + q) * NT + tid) * 16, 0, 16);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p.sk_ws) + sk_me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
    }
    if (NPL == 2) {                    // undo the two operand scales: an exact power of two
        const int de = -(sexp_a + sexp_b);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __builtin_ldexpf(acc[mi][ni][r], de);
    }
    // ---- epilogue -----------------------------------------------------------------------
    // The C/D layout has a column per lane and rows across registers: stored directly, one instruction writes two
    // 128-B row segments of 4 B per lane.  Each wave instead transposes one 32x32 accumulator tile at a time through
    // a private LDS patch (the operand tiles are dead after the loop's last barrier; LDS ops of one wave execute in
    // order, so no block barrier) and then owns 4 consecutive columns of 4 rows: scale/shift/LeakyReLU/residual and
    // the store run on float4s, 8 full 128-B row segments per instruction.  The thin early layers (K = 32..288,
    // outputs of 0.7 GB) were store-issue bound in the epilogue.
    const bool direct = (p.out_stride == 1 && p.out_oy == 0 && p.out_ox == 0 && p.Ho == p.Hg &&
                         p.Wo == p.Wg);
    float* stg = smem + wave * (32 * LDS_LD);
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    const bool bstat = BS && p.bs_part != nullptr;      // fused BatchNorm backward reductions (see viddet_hip.h)
    float bs1[TN][4], bs2[TN][4];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) bs1[ni][e] = bs2[ni][e] = 0.f;
    // float4 path: rows 16-B aligned (wave-uniform; every tensor of the model qualifies, odd pitches fall back)
    const bool vec_ok = (p.ldo % 4 == 0) && ((uintptr_t)p.out % 16 == 0) &&
                        (!(p.flags & VD_EPI_RESIDUAL) || ((p.ldr % 4 == 0) && ((uintptr_t)p.residual % 16 == 0)));
    const bool has_aff = p.flags & VD_EPI_AFFINE, has_leaky = p.flags & VD_EPI_LEAKY, has_res = p.flags & VD_EPI_RESIDUAL;
    bool fast = vec_ok && (p.Co % 4 == 0) && (!has_aff || (((uintptr_t)p.scale | (uintptr_t)p.shift) % 16 == 0));
    if (BS) fast = fast && (!bstat || (((uintptr_t)p.bs_z | (uintptr_t)p.bs_scale | (uintptr_t)p.bs_shift |
                                        (uintptr_t)p.bs_mean | (uintptr_t)p.bs_invstd) % 16 == 0));
    // (SK: the host launches the stream-K form only where this path applies - streamk_grid() - and the element-wise path is
    // not instantiated: its loop-invariant address arithmetic would be hoisted out of the item loop and spilled)
    if (SK || fast) {
        // Straight-line path (every launch of the network).  Per-column constants are loaded once up front; the
        // residual / BatchNorm-input rows of block b+1 are requested before block b goes through the LDS patch, so
        // their latency hides behind it; nothing ever waits for a store.  (The element-wise path below waited for
        // every load and store round trip in turn: 8-20k cycles per workgroup.)
        constexpr int NB = TM * TN;
        f32x4 sc[TN], sh[TN], qsc[BS ? TN : 1], qsh[BS ? TN : 1], qmu[BS ? TN : 1], qis[BS ? TN : 1];
        int colv[TN];
        const f32x4 ones = {1.f, 1.f, 1.f, 1.f}, zeros = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = tile_n * BN + wn * TN * 32 + ni * 32 + ec4;
            // (PAR: GEMM column = class * Cin + channel; everything below addresses by the channel)
            const int chn = PAR ? col % p.par_cin : col;
            colv[ni] = col < p.Co ? chn : -1;
            const int cc = col < p.Co ? chn : 0;
            sc[ni] = (has_aff && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + cc) : ones;
            sh[ni] = (has_aff && p.shift) ? *reinterpret_cast<const f32x4*>(p.shift + cc) : zeros;
            if (BS) {
                qsc[ni] = bstat ? *reinterpret_cast<const f32x4*>(p.bs_scale + cc) : zeros;
                qsh[ni] = bstat ? *reinterpret_cast<const f32x4*>(p.bs_shift + cc) : zeros;
                qmu[ni] = bstat ? *reinterpret_cast<const f32x4*>(p.bs_mean + cc) : zeros;
                qis[ni] = bstat ? *reinterpret_cast<const f32x4*>(p.bs_invstd + cc) : zeros;
            }
        }
        // two register slots (block b+1 in flight under block b) except on the 8-wave fp32-MFMA tiles, which live
        // inside 128 VGPRs: there block b's rows are requested just before its own pass through LDS
        constexpr int NSL = (!SP && WM * WN == 8) ? 1 : 2;
        f32x4 rres[NSL][4], rz[BS ? NSL : 1][4];
        int64_t ropix[NSL][4];
        auto issue = [&](int mi, int ni, f32x4 (&rr)[4], f32x4 (&zz)[4], int64_t (&op)[4]) {
            const int cc = colv[ni] < 0 ? 0 : colv[ni];
            // PAR: the parity class of this 32-column block picks the output pixel of a row
            const int pcls = PAR ? (tile_n * BN + wn * TN * 32 + ni * 32) / p.par_cin : 0;
            const int o_y = PAR ? (pcls >> 1) : p.out_oy, o_x = PAR ? (pcls & 1) : p.out_ox;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + erow + 8 * i;
                m = m < M ? m : M - 1;
                int64_t opix = m;
                if (PAR || !direct) {
                    const unsigned mu = (unsigned)m;
                    const unsigned t = udiv_rcp(mu, (unsigned)p.Wg, rcp_w);
                    const int gx = (int)(mu - t * (unsigned)p.Wg);
                    const unsigned n = udiv_rcp(t, (unsigned)p.Hg, rcp_h);
                    const int gy = (int)(t - n * (unsigned)p.Hg);
                    opix = ((int64_t)n * p.Ho + (gy * p.out_stride + o_y)) * p.Wo + (gx * p.out_stride + o_x);
                }
                op[i] = opix;
                if (has_res) rr[i] = *reinterpret_cast<const f32x4*>(p.residual + opix * p.ldr + cc);
                if (BS && bstat) zz[i] = *reinterpret_cast<const f32x4*>(p.bs_z + opix * p.ldo + cc);
            }
        };
        if (NSL == 2) issue(0, 0, rres[0], rz[0], ropix[0]);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int mi = b % TM, ni = b / TM;
            constexpr int S1 = NSL - 1;
            if (NSL == 1) issue(mi, ni, rres[0], rz[0], ropix[0]);
            else if (b + 1 < NB)
                issue((b + 1) % TM, (b + 1) / TM, rres[(b + 1) & S1], rz[BS ? ((b + 1) & S1) : 0], ropix[(b + 1) & S1]);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = M16 ? 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int scol = M16 ? 16 * ((r >> 2) & 1) + (lane & 15) : (lane & 31);
                stg[srow * LDS_LD + scol] = acc[mi][ni][r];
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(stg + (erow + 8 * i) * LDS_LD + ec4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + erow + 8 * i;
                const bool ok = m < M && colv[ni] >= 0;
                f32x4 t = v[i];
                if (has_aff) t = t * sc[ni] + sh[ni];
                if (has_leaky) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = t[e] > 0.f ? t[e] : t[e] * p.slope;
                }
                if (has_res) t += rres[b & S1][i];
                if (ok && !(VD_PROBE & 64)) {             // probe bit 6: no output stores
                    *reinterpret_cast<f32x4*>(p.out + ropix[b & S1][i] * p.ldo + colv[ni]) = t;
                    amx = fmaxf(amx, fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fabsf(t[2]), fabsf(t[3]))));
                }
                if (BS && bstat) {
                    const f32x4 z = rz[BS ? (b & S1) : 0][i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = z[e] * qsc[ni][e] + qsh[ni][e];
                        float g = u > 0.f ? t[e] : t[e] * p.bs_slope;
                        g = ok ? g : 0.f;
                        bs1[ni][e] += g;
                        bs2[ni][e] += g * (z[e] - qmu[ni][e]) * qis[ni][e];
                    }
                }
            }
        }
    } else if constexpr (!SK) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int col = tile_n * BN + wn * TN * 32 + ni * 32 + ec4;
        const int nvalid = p.Co - col;                    // columns col .. col+3 that exist (<= 0: none)
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.flags & VD_EPI_AFFINE) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nvalid) {
                    if (p.scale) sc[e] = p.scale[col + e];
                    if (p.shift) sh[e] = p.shift[col + e];
                }
        }
        float qsc[4], qsh[4], qmu[4], qis[4];             // BatchNorm constants of the layer whose dy this is
        if (bstat) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = e < nvalid;
                qsc[e] = ok ? p.bs_scale[col + e] : 0.f;
                qsh[e] = ok ? p.bs_shift[col + e] : 0.f;
                qmu[e] = ok ? p.bs_mean[col + e] : 0.f;
                qis[e] = ok ? p.bs_invstd[col + e] : 0.f;
            }
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            WAVE_SYNC();                                  // the previous tile's reads are done before it is overwritten
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = M16 ? 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int scol = M16 ? 16 * ((r >> 2) & 1) + (lane & 15) : (lane & 31);
                stg[srow * LDS_LD + scol] = acc[mi][ni][r];
            }
            WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = erow + 8 * i;
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * LDS_LD + ec4);
                const int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + row;
                if (nvalid <= 0 || m >= M) continue;
                int64_t opix = m;
                if (!direct) {
                    const unsigned mu = (unsigned)m;
                    const unsigned t = mu / (unsigned)p.Wg;
                    const int gx = (int)(mu - t * (unsigned)p.Wg);
                    const int64_t n = t / (unsigned)p.Hg;
                    const int gy = (int)(t - (unsigned)n * (unsigned)p.Hg);
                    opix = (n * p.Ho + (gy * p.out_stride + p.out_oy)) * p.Wo + (gx * p.out_stride + p.out_ox);
                }
                if (p.flags & VD_EPI_AFFINE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
                }
                if (p.flags & VD_EPI_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
                }
                float* dst = p.out + opix * p.ldo + col;
                if (nvalid >= 4 && vec_ok) {
                    if (p.flags & VD_EPI_RESIDUAL) v += *reinterpret_cast<const f32x4*>(p.residual + opix * p.ldr + col);
                    *reinterpret_cast<f32x4*>(dst) = v;
                    amx = fmaxf(amx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    if (bstat) {
                        const f32x4 z = *reinterpret_cast<const f32x4*>(p.bs_z + opix * p.ldo + col);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float u = z[e] * qsc[e] + qsh[e];
                            const float g = u > 0.f ? v[e] : v[e] * p.bs_slope;
                            bs1[ni][e] += g;
                            bs2[ni][e] += g * (z[e] - qmu[e]) * qis[e];
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < nvalid) {
                            float t = v[e];
                            if (p.flags & VD_EPI_RESIDUAL) t += p.residual[opix * p.ldr + col + e];
                            dst[e] = t;
                            amx = fmaxf(amx, fabsf(t));
                            if (bstat) {
                                const float z = p.bs_z[opix * p.ldo + col + e];
                                const float u = z * qsc[e] + qsh[e];
                                const float g = u > 0.f ? t : t * p.bs_slope;
                                bs1[ni][e] += g;
                                bs2[ni][e] += g * (z - qmu[e]) * qis[e];
                            }
                        }
                }
            }
        }
    }

    }
    STAMP(5);
    // ---- fused BatchNorm backward reductions: a lane holds 4 columns x (4 rows x TM tiles); fold the 8 row groups of
    // the wave (lane bits 3..5), then the WM waves that share the columns, one writer per (tile_m, column)
    if (bstat) {
        __syncthreads();        // every wave is done with its staging patch
        float* red = smem;      // [WM][BN][2]
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = bs1[ni][e], b = bs2[ni][e];
                a += __shfl_xor(a, 8);  b += __shfl_xor(b, 8);
                a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                if (lane < 8) {
                    const int c = wn * TN * 32 + ni * 32 + ec4 + e;
                    red[(wm * BN + c) * 2 + 0] = a;
                    red[(wm * BN + c) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (PAR) {
            // columns of different parity classes are the same CHANNEL: one table row [2 Cin] per (tile_m, tile_n), the
            // classes a tile holds folded into it; rows of tiles narrower than Cin fill their own channels only (the host
            // zeroes the table first).  vd_conv_igemm_mtiles() counts these rows.
            const int cin = p.par_cin, ncls = BN > cin ? BN / cin : 1, nch = BN < cin ? BN : cin;
            float* dstp = p.bs_part + ((int64_t)tile_m * ntile + tile_n) * 2 * cin;
            for (int c = tid; c < nch; c += NT) {
                if (tile_n * BN + c < p.Co) {
                    float a = 0.f, b = 0.f;
                    for (int j = 0; j < ncls; ++j)
#pragma unroll
                        for (int w = 0; w < WM; ++w) {
                            a += red[(w * BN + c + j * cin) * 2 + 0];
                            b += red[(w * BN + c + j * cin) * 2 + 1];
                        }
                    const int ch = (tile_n * BN + c) % cin;
                    dstp[ch] = a;
                    dstp[cin + ch] = b;
                }
            }
        } else {
        for (int c = tid; c < BN; c += NT) {
            const int colc = tile_n * BN + c;
            if (colc < p.Co) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    a += red[(w * BN + c) * 2 + 0];
                    b += red[(w * BN + c) * 2 + 1];
                }
                float* dstp = p.bs_part + (int64_t)tile_m * 2 * p.Co;
                dstp[colc] = a;
                dstp[p.Co + colc] = b;
            }
        }
        }
    }

    // ---- fused BatchNorm statistics (training forward): per-column sum / sum of squares of this block's raw
    // conv outputs, written as one row of the partial table [tile_m][2*Co] (no atomics: each (tile_m, column) has
    // exactly one writer; vd_bn_sum_partials finishes the reduction in fp64 in a fixed order).  The C/D layout
    // puts a column on a lane, so the sums are lane-local over the 16*TM rows, then folded across the two
    // half-waves and the WM waves that share the column.
    if (p.stats_part) {
        __syncthreads();        // every wave is done with its staging patch
        float* red = smem;      // [WM][BN][2] : the operand tiles are dead after the last barrier of the K loop
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            if (M16) {
                // two columns per lane (sn = 0, 1), rows spread over the four 16-lane groups
                float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3);
                        const float v = (m < M) ? acc[mi][ni][r] : 0.f;
                        s1[(r >> 2) & 1] += v;
                        s2[(r >> 2) & 1] += v * v;
                    }
#pragma unroll
                for (int sn = 0; sn < 2; ++sn) {
                    s1[sn] += __shfl_xor(s1[sn], 16); s2[sn] += __shfl_xor(s2[sn], 16);
                    s1[sn] += __shfl_xor(s1[sn], 32); s2[sn] += __shfl_xor(s2[sn], 32);
                    if (lane < 16) {
                        const int c = wn * TN * 32 + ni * 32 + 16 * sn + lane;
                        red[(wm * BN + c) * 2 + 0] = s1[sn];
                        red[(wm * BN + c) * 2 + 1] = s2[sn];
                    }
                }
                continue;
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + (r & 3) + 8 * (r >> 2) +
                                      4 * (lane >> 5);
                    const float v = (m < M) ? acc[mi][ni][r] : 0.f;
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lane < 32) {
                const int c = wn * TN * 32 + ni * 32 + lane;
                red[(wm * BN + c) * 2 + 0] = s1;
                red[(wm * BN + c) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            const int col = tile_n * BN + c;
            if (col < p.Co) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    s1 += red[(w * BN + c) * 2 + 0];
                    s2 += red[(w * BN + c) * 2 + 1];
                }
                float* dstp = p.stats_part + (int64_t)tile_m * 2 * p.Co;
                dstp[col] = s1;
                dstp[p.Co + col] = s2;
            }
        }
    }
    }   // work items
    if (p.amax_out) vd_amax_publish(p.amax_out, amx);
    STAMP(6);
}

const float* zero_page() {
    static const float* zp = nullptr;
    if (!zp) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_page)) != hipSuccess) q = nullptr;
        zp = (const float*)q;
    }
    return zp;
}

template <int WM, int WN, int TM, int TN, bool XF, bool SP, bool M16, bool BS, int NPL, bool HALO = false, bool SK = false, bool PAR = false>
int launch_igemm_bs(const vd_conv_desc& d, hipStream_t s);

// LDS bytes of the halo loop for a BM x BN tile on a map of width W: two halo buffers of BM + 2 (W + 1) rows (+ zero row
// + sink row) and a ring of three weight stages, 128 B per row
inline int64_t halo_lds_bytes(int BM, int BN, int W) { return 2ll * (BM + 2 * (W + 1) + 2) * 128 + 3ll * BN * 128; }

// 3x3 stride-1 'same' geometry (forward, or the data gradient of such a conv), halo within the LDS and the 9 x 64-row
// slots of the halo stream
inline bool halo_ok(const vd_conv_desc& d, int BM, int BN) {
    if (!(d.flags & VD_MATH_F16X2) || (d.flags & VD_MATH_NOHALO) || d.in_scale || d.T != 9 || d.in_stride != 1 || d.Kfr != 1 ||
        d.Hg != d.Hi || d.Wg != d.Wi || BN < 64)
        return false;
    for (int t = 0; t < 9; ++t)
        if (d.dy[t] < -1 || d.dy[t] > 1 || d.dx[t] < -1 || d.dx[t] > 1 || d.dz[t] != 0) return false;
    return BM + 2 * (d.Wi + 1) <= 8 * 64 && halo_lds_bytes(BM, BN, d.Wi) <= 160 * 1024;     // 8 stream slots + 1 idle (see hstep)
}

template <int WM, int WN, int TM, int TN, bool XF, bool SP = false, bool M16 = false>
int launch_igemm(const vd_conv_desc& d, hipStream_t s) {
    // the in-load transform (XF) and the backward reductions never meet: one is a forward feature, one a dgrad one
    if (SP && (d.flags & VD_MATH_BF16)) {          // one-plane arithmetic (training in bf16 products); no XF variant
        if (d.bs_part) return launch_igemm_bs<WM, WN, TM, TN, false, SP, M16, true, SP ? 1 : 3>(d, s);
        return launch_igemm_bs<WM, WN, TM, TN, false, SP, M16, false, SP ? 1 : 3>(d, s);
    }
    if (SP && (d.flags & VD_MATH_F16X2)) {         // two fp16 planes, three MFMAs per product block; no XF variant
        constexpr bool HT_ = SP && WM * WN == 8 && WN * TN * 32 >= 64;     // tiles the halo loop is instantiated for
        if (HT_ && halo_ok(d, WM * TM * 32, WN * TN * 32)) {
            if (d.bs_part) return launch_igemm_bs<WM, WN, TM, TN, false, SP, M16, true, SP ? 2 : 3, HT_>(d, s);
            return launch_igemm_bs<WM, WN, TM, TN, false, SP, M16, false, SP ? 2 : 3, HT_>(d, s);
        }
        if (d.bs_part) return launch_igemm_bs<WM, WN, TM, TN, false, SP, M16, true, SP ? 2 : 3>(d, s);
        return launch_igemm_bs<WM, WN, TM, TN, false, SP, M16, false, SP ? 2 : 3>(d, s);
    }
    if (!XF && d.bs_part) return launch_igemm_bs<WM, WN, TM, TN, false, SP, M16, true, 3>(d, s);
    return launch_igemm_bs<WM, WN, TM, TN, XF, SP, M16, false, 3>(d, s);
}

// CUs of the device (one device per process, include/viddet_hip.h "Conventions")
inline int device_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// Persistent stream-K grid of a BM x BN tile with `wg_per_cu` co-resident workgroups per CU, or 0 where the form does not
// apply: every XCD group must own at least as many tiles as workgroups (a run then cuts at most two tiles and no tile
// has more than two parts), the workspace must hold one accumulator tile per workgroup
inline int streamk_grid(const vd_conv_desc& d, int BM, int BN, int wg_per_cu) {
    if (!(d.flags & VD_CONV_STREAMK) || !d.sk_ws) return 0;
    // the float4 epilogue path only (every tensor of the network qualifies): 16-byte aligned rows and per-column vectors
    auto al16 = [](const void* q) { return (uintptr_t)q % 16 == 0; };
    if (d.ldo % 4 || d.Co % 4 || !al16(d.out) || ((d.flags & VD_EPI_RESIDUAL) && (d.ldr % 4 || !al16(d.residual))) ||
        ((d.flags & VD_EPI_AFFINE) && (!al16(d.scale) || !al16(d.shift))) ||
        (d.bs_part && !(al16(d.bs_z) && al16(d.bs_scale) && al16(d.bs_shift) && al16(d.bs_mean) && al16(d.bs_invstd))))
        return 0;
    const int64_t ntiles = vd_cdiv((int64_t)d.N * d.Hg * d.Wg, BM) * vd_cdiv(d.Co, BN);
    const int G = device_cus() * wg_per_cu;
    if (G > VD_SK_MAX_WG - 2 || G < 8 || (ntiles >> 3) < (G + 7) / 8) return 0;
    if (ntiles <= G + G / 16) return 0;                    // (nearly) one tile per workgroup: nothing to balance
    if (d.sk_ws_bytes < (int64_t)VD_SK_HEADER_BYTES + (int64_t)G * BM * BN * 4) return 0;
    return G;
}

template <int WM, int WN, int TM, int TN, bool XF, bool SP, bool M16, bool BS, int NPL, bool HALO, bool SK, bool PAR>
int launch_igemm_bs(const vd_conv_desc& d, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int lds0 = SP ? 2 * (BM + BN) * NPL * 64 : 2 * (BM + BN) * LDS_LD * (int)sizeof(float);
    constexpr int lds_epi = WM * WN * 32 * LDS_LD * 4 > WM * BN * 2 * 4 ? WM * WN * 32 * LDS_LD * 4 : WM * BN * 2 * 4;
    constexpr int lds_fixed = lds0 > lds_epi ? lds0 : lds_epi;       // operand stages, or the epilogue patches if larger
    static_assert(lds_fixed + (SK ? 16 : 0) <= 160 * 1024, "LDS budget");
    int lds = lds_fixed;
    if (HALO) {
        const int64_t hb = halo_lds_bytes(BM, BN, d.Wi);
        lds = hb > lds_epi ? (int)hb : lds_epi;
    }
    int64_t nblk = vd_cdiv((int64_t)d.N * d.Hg * d.Wg, BM) * vd_cdiv(d.Co, BN);
    int sk_flag_off = 0;
    if (SK) {
        // one 16-byte word behind the operand stages: the poll result one lane hands to the workgroup
        const int G = streamk_grid(d, BM, BN, (SP && WM * WN == 4) ? 2 : 1);
        lds = (lds + 15) & ~15;
        if (G == 0 || lds + 16 > 160 * 1024) return 1;     // not a stream-K launch: the caller takes the one-tile-per-workgroup form
        sk_flag_off = lds;
        lds += 16;
        nblk = G;
    }
    static bool attr_done = false;
    auto kfn = k_conv_igemm<WM, WN, TM, TN, XF, SP, M16, BS, NPL, HALO, SK, PAR>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  HALO ? 160 * 1024 : lds_fixed + (SK ? 16 : 0));
        attr_done = true;
    }
    const float* zp = zero_page();
    const int64_t zd_in = zp - d.in, zd_w = zp - d.wp;      // element deltas (all pointers are float-aligned)
    // (developer / test switch VD_SK_TIMEOUT_TICKS: 0 makes every consumer give up at once and recompute its prefix)
    static const int sk_timeout = getenv("VD_SK_TIMEOUT_TICKS") ? atoi(getenv("VD_SK_TIMEOUT_TICKS")) : VD_SK_TIMEOUT_TICKS;
    hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(WM * WN * 64), lds, s, d, zd_in, zd_w, sk_flag_off, sk_timeout);
    return 0;
}

}  // namespace
