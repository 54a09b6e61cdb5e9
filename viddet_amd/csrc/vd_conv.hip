// vd_conv.hip — tap-list implicit-GEMM convolution for gfx950 on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), plus its weight-gradient twin and the weight (un)packers.
//
// Replaces, on the reference side (paths under /root/reference):
//   nn.Conv2D inside _conv2d                     models/definitions/layers.py:66-67
//   nn.Conv2D prediction conv (with bias)        models/definitions/yolo/yolo3.py:62
//   nn.Conv3D inside _conv3d / _conv21d          models/definitions/layers.py:73-89
//   their autograd backward                      train_yolov3.py:631
//
// Data layout: activations NHWC fp32 (channel-contiguous => the GEMM K run of one tap is one
// contiguous 128-B segment per pixel), weights packed [Co][T*Ci] (K-contiguous).
//
// GEMM view (forward / dgrad):  M = N*Hg*Wg pixels, N = Co, K = T*Ci.
//   LDS tiles [rows][32+4] fp32: the +4 pad makes 16 rows that are distinct mod 16 hit 16
//   distinct 4-bank slots, so every ds_read_b128 lane group is conflict free.
//   MFMA operand map (guide §3): lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31].  Each lane
//   reads 4 consecutive k (one ds_read_b128) at k0 = 8*kc + 4*(l>>5) and feeds them to 4 MFMAs;
//   over (kc, half, j) every k of the 32-chunk is summed exactly once.
//   C/D map: acc[r] -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31  => a store instruction writes two
//   128-B row segments (channel-contiguous NHWC).
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
template <int WM, int WN, int TM, int TN, bool XF, bool SP, bool M16 = false, bool BS = false, int NPL = 3, bool HALO = false>
__global__ __launch_bounds__(WM * WN * 64, ((!SP && WM * WN == 8) ? 4 : ((SP && WM * WN == 4) ? 2 : 1))) void k_conv_igemm(const vd_conv_desc p, const int64_t zd_in, const int64_t zd_w) {
    static_assert(NPL == 3 || ((NPL == 1 || NPL == 2) && SP), "planes");
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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t M = (int64_t)p.N * p.Hg * p.Wg;
    const int ntile = (p.Co + BN - 1) / BN;
    const int lid = vd_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % ntile, tile_m = lid / ntile;
    const int Ktot = p.T * p.Ci;

    const int lrow = tid >> 3;
    const int lc4 = (tid & 7) * 4;

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
    const int tlane = lane < p.T ? lane : 0;
    const int tap_dy = p.dy[tlane], tap_dx = p.dx[tlane], tap_dz = p.dz[tlane];
    const unsigned rcp_w = 0xFFFFFFFFu / (unsigned)p.Wg + 1u, rcp_h = 0xFFFFFFFFu / (unsigned)p.Hg + 1u;
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

    // register sets = K-steps of global-load latency cover (the 4-wave 128x128 tile has no VGPRs left for a third)
    constexpr int PD = (VD_PD > 2 && ((WM * WN == 4 && TM * TN == 4) || (M16 && TM * TN == 4))) ? 2 : VD_PD;
    f32x4 ra[PD][AP], rb[PD][BP];
    // k-step cursor of the NEXT tile to load.  VD_KROT: every M tile starts the channel-chunk loop at a different chunk (and
    // wraps), so that at any moment the workgroups of the chip read different 128-B pieces of the 4*Ci-byte pixel records
    // instead of all the same one (memory-channel balance); a tile's sum order is rotated, not its value set
    int t_tap = 0, c0 = VD_KROT ? (int)((unsigned)tile_m % (unsigned)(p.Ci / BK)) * BK : 0;

    // tap part of the wave-uniform source offset: refreshed only when the tap changes (every Ci/32 K-steps), so the
    // scalar loads of dy/dx/dz and their s_waitcnt leave the per-step critical path
    const int64_t tap_eo = (int64_t)((tap_dz * p.Hi + tap_dy) * p.Wi + tap_dx) * p.Ci;   // lane t: tap t
    const int tap_eo_lo = (int)(tap_eo & 0xffffffffll), tap_eo_hi = (int)(tap_eo >> 32);
    auto tap_off = [&](int t) -> int64_t {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(tap_eo_lo, t);
        const int hi = __builtin_amdgcn_readlane(tap_eo_hi, t);
        return ((int64_t)hi << 32) | (int64_t)lo;
    };
    int64_t tap_soff = tap_off(0);
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
    auto compute = [&](int buf) {
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
                        for (int ni = 0; ni < TN; ++ni)
                            acc[mi][ni] = mfma32<NPL>(fa[mi][Terms<NPL>::QA[t]], fb[ni][Terms<NPL>::QB[t]], acc[mi][ni]);
#if VD_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
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

    const int nks = p.T * (p.Ci / BK);
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
        const int nchunk = p.Ci / BK;
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
        int bt = 0, bc0 = 0;
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
        for (int s2 = 0; s2 < HT; ++s2) t9[s2] = hload(0, s2);
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
        for (int s2 = 0; s2 < HT; ++s2) hstore(t9[s2], 0, s2);
        lstoreB(0, rb[0]);
        lstoreB(1, rb[1]);
        __syncthreads();
        constexpr int HD = 3;                                  // halo items in flight (one request and one store per K-step)
        f32x4 hv[HD];
        int hc = 1, hs = 0;                                    // next item to request (chunk, slot)
        int sc = 1, ss = 0;                                    // next item to store
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
        int cc = 0, ct = 0;                                    // chunk / tap of the step being multiplied
        int avc[NRB];
        hrows(avc, 0, 0);
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
    float amx = 0.f;                                    // max-abs of what this lane stores (p.amax_out)
    // float4 path: rows 16-B aligned (wave-uniform; every tensor of the model qualifies, odd pitches fall back)
    const bool vec_ok = (p.ldo % 4 == 0) && ((uintptr_t)p.out % 16 == 0) &&
                        (!(p.flags & VD_EPI_RESIDUAL) || ((p.ldr % 4 == 0) && ((uintptr_t)p.residual % 16 == 0)));
    const bool has_aff = p.flags & VD_EPI_AFFINE, has_leaky = p.flags & VD_EPI_LEAKY, has_res = p.flags & VD_EPI_RESIDUAL;
    bool fast = vec_ok && (p.Co % 4 == 0) && (!has_aff || (((uintptr_t)p.scale | (uintptr_t)p.shift) % 16 == 0));
    if (BS) fast = fast && (!bstat || (((uintptr_t)p.bs_z | (uintptr_t)p.bs_scale | (uintptr_t)p.bs_shift |
                                        (uintptr_t)p.bs_mean | (uintptr_t)p.bs_invstd) % 16 == 0));
    if (fast) {
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
            colv[ni] = col < p.Co ? col : -1;
            const int cc = col < p.Co ? col : 0;
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
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + erow + 8 * i;
                m = m < M ? m : M - 1;
                int64_t opix = m;
                if (!direct) {
                    const unsigned mu = (unsigned)m;
                    const unsigned t = udiv_rcp(mu, (unsigned)p.Wg, rcp_w);
                    const int gx = (int)(mu - t * (unsigned)p.Wg);
                    const unsigned n = udiv_rcp(t, (unsigned)p.Hg, rcp_h);
                    const int gy = (int)(t - n * (unsigned)p.Hg);
                    opix = ((int64_t)n * p.Ho + (gy * p.out_stride + p.out_oy)) * p.Wo + (gx * p.out_stride + p.out_ox);
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
    } else {
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
    if (p.amax_out) vd_amax_publish(p.amax_out, amx);
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

template <int WM, int WN, int TM, int TN, bool XF, bool SP, bool M16, bool BS, int NPL, bool HALO = false>
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

template <int WM, int WN, int TM, int TN, bool XF, bool SP, bool M16, bool BS, int NPL, bool HALO>
int launch_igemm_bs(const vd_conv_desc& d, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int lds0 = SP ? 2 * (BM + BN) * NPL * 64 : 2 * (BM + BN) * LDS_LD * (int)sizeof(float);
    constexpr int lds_epi = WM * WN * 32 * LDS_LD * 4 > WM * BN * 2 * 4 ? WM * WN * 32 * LDS_LD * 4 : WM * BN * 2 * 4;
    constexpr int lds_fixed = lds0 > lds_epi ? lds0 : lds_epi;       // operand stages, or the epilogue patches if larger
    static_assert(lds_fixed <= 160 * 1024, "LDS budget");
    int lds = lds_fixed;
    if (HALO) {
        const int64_t hb = halo_lds_bytes(BM, BN, d.Wi);
        lds = hb > lds_epi ? (int)hb : lds_epi;
    }
    static bool attr_done = false;
    auto kfn = k_conv_igemm<WM, WN, TM, TN, XF, SP, M16, BS, NPL, HALO>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  HALO ? 160 * 1024 : lds_fixed);
        attr_done = true;
    }
    const int64_t M = (int64_t)d.N * d.Hg * d.Wg;
    const int64_t nblk = vd_cdiv(M, BM) * vd_cdiv(d.Co, BN);
    const float* zp = zero_page();
    const int64_t zd_in = zp - d.in, zd_w = zp - d.wp;      // element deltas (all pointers are float-aligned)
    hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(WM * WN * 64), lds, s, d, zd_in, zd_w);
    return 0;
}

// Tile variants (vd_conv_desc.tile; 0 = heuristic).  The host autotunes per launch record at plan-build
// time (viddet_amd/ops.py autotune_conv): round quantisation (n blocks over 512 slots), how fast a lone
// block runs and the K depth interact, and no closed-form rule picked the winner for every layer.
int igemm_resolve_tile(const vd_conv_desc& d) {
    int tile = d.tile;
    if (tile <= 0 || tile > 8) {
        if (d.Co <= 32) tile = 7;
        else if (d.Co <= 64) tile = 6;
        else tile = 2;
    }
    return tile;
}

int igemm_tile_bm(int tile) { return (tile == 4 || tile == 5) ? 64 : 128; }

// split-operand math (flags & VD_MATH_SPLIT): 8 waves, one workgroup per CU (the three bf16 planes of a
// double-buffered 256x128 stage fill the 160 KB of LDS)
int igemm_split_resolve_tile(const vd_conv_desc& d) {
    int tile = d.tile;
    if (tile <= 0 || tile > 16) tile = d.Co <= 32 ? 9 : (d.Co <= 64 ? 3 : 1);
    return tile;
}
int igemm_split_tile_bm(int tile) { return (tile >= 13 ? 256 : (tile >= 11 ? 128 : ((tile >= 9 || ((tile - 1) & 3) == 0 || ((tile - 1) & 3) == 2) ? 256 : 128))); }

template <bool XF>
int dispatch_igemm_split(const vd_conv_desc& d, hipStream_t s) {
    switch (igemm_split_resolve_tile(d)) {
        case 1: return launch_igemm<4, 2, 2, 2, XF, true>(d, s);    // 256 x 128, 8 waves of 64x64
        case 2: return launch_igemm<4, 2, 1, 2, XF, true>(d, s);    // 128 x 128, 8 waves of 32x64
        case 3: return launch_igemm<4, 2, 2, 1, XF, true>(d, s);    // 256 x  64, 8 waves of 64x32
        case 4: return launch_igemm<4, 2, 1, 1, XF, true>(d, s);    // 128 x  64, 8 waves of 32x32
        // 5..8: the same tiles on the 16x16x32 MFMA shape
        case 5: return launch_igemm<4, 2, 2, 2, XF, true, true>(d, s);
        case 6: return launch_igemm<4, 2, 1, 2, XF, true, true>(d, s);
        case 7: return launch_igemm<4, 2, 2, 1, XF, true, true>(d, s);
        case 8: return launch_igemm<4, 2, 1, 1, XF, true, true>(d, s);
        // 9, 10: 256 x 32 (8 waves of 32x32) for the 32-channel outputs of the first stage, both MFMA shapes
        case 9: return launch_igemm<8, 1, 1, 1, XF, true>(d, s);
        case 10: return launch_igemm<8, 1, 1, 1, XF, true, true>(d, s);
        // 11, 12: 128 x 128 as FOUR waves of 64x64 - half the LDS and threads of tile 1 at the same per-wave shape, so two
        // workgroups share a CU and one's prologue / epilogue runs under the other's K loop (short-K 1x1 and stride-2 layers)
        case 11: return launch_igemm<2, 2, 2, 2, XF, true>(d, s);
        case 12: return launch_igemm<2, 2, 2, 2, XF, true, true>(d, s);
        // 13, 14: 256 x 32 as four waves of 64x32 (two workgroups per CU): twice the MFMAs per wave, K-step and barrier of
        // tiles 9 / 10 for the 32-channel outputs of the first stage
        case 13: return launch_igemm<4, 1, 2, 1, XF, true>(d, s);
        case 14: return launch_igemm<4, 1, 2, 1, XF, true, true>(d, s);
        // 15, 16: 256 x 64 as four waves of 64x64 (the 128 x 64 four-wave form measured no better than tiles 4 / 8)
        // (256 x 128 as four waves of 128x64 - one wave per SIMD, 24 LDS operand reads per 48 MFMAs instead of 16 per 24 - was
        // measured too: 210 TF against 219 for tile 1's generic loop at 256->512 @26; fewer LDS reads do not pay for losing the
        // second wave of a SIMD)
        case 15: return launch_igemm<4, 1, 2, 2, XF, true>(d, s);
        default: return launch_igemm<4, 1, 2, 2, XF, true, true>(d, s);
    }
}

template <bool XF>
int dispatch_igemm(const vd_conv_desc& d, hipStream_t s) {
    if (d.flags & (VD_MATH_SPLIT | VD_MATH_BF16 | VD_MATH_F16X2)) return dispatch_igemm_split<XF>(d, s);
    const int tile = igemm_resolve_tile(d);
    switch (tile) {
        case 1: return launch_igemm<2, 2, 2, 2, XF>(d, s);   // 128 x 128, 4 waves of 64x64
        case 2: return launch_igemm<4, 2, 1, 2, XF>(d, s);   // 128 x 128, 8 waves of 32x64
        case 3: return launch_igemm<2, 4, 2, 1, XF>(d, s);   // 128 x 128, 8 waves of 64x32
        case 4: return launch_igemm<2, 2, 1, 2, XF>(d, s);   //  64 x 128, 4 waves of 32x64
        case 5: return launch_igemm<2, 4, 1, 1, XF>(d, s);   //  64 x 128, 8 waves of 32x32
        case 6: return launch_igemm<2, 2, 2, 1, XF>(d, s);   // 128 x  64, 4 waves of 64x32
        case 7: return launch_igemm<4, 1, 1, 1, XF>(d, s);   // 128 x  32, 4 waves of 32x32
        default: return launch_igemm<4, 2, 1, 1, XF>(d, s);  // 128 x  64, 8 waves of 32x32
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  D[co][j] = sum_pix dout[pix][co] * in[pix shifted by tap(j)][c(j)],  j = t*Ci + c
//   GEMM rows = co, cols = the flat (tap, channel) index j (exactly the packed weight layout), reduction
//   = pixels.  Both operands are channel-contiguous per pixel, so the LDS tiles are [pixel][BM|128] and
//   the MFMA operands are fetched with conflict-free ds_read_b32 (lane -> channel, half-wave -> pixel
//   parity).  A thread's 4-column chunk never straddles a tap (Ci % 4 == 0), so each thread owns one
//   (tap, channel) for the whole reduction: narrow layers (Ci = 32/64) fill the 128-wide tile with 4 / 2
//   taps instead of wasting it, and the dout tile is shared by those taps.
//   Tiles <WM,WN,TM,TN>: 128x128 (2,2,2,2), 64x128 (2,2,1,2), 32x128 (1,4,1,1) by Co.
// ---------------------------------------------------------------------------------------------
constexpr int WG_BN = 128, WG_BP = 32;
constexpr int WG_BP_BF = 64;      // pixels per K-step of the bf16-tensor form (VD_STORE_BF16)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// One 32-channel x 16-pixel bf16 MFMA operand out of a pixel-major LDS plane [pixel][channels] (wgrad, split
// math): two ds_read_b64_tr_b16 hardware-transposed reads.  Per 16-lane group, lane 4q+p supplies the address of
// pixel row q, channels 4p..4p+3, and lane i receives channel i of the 4 pixels - exactly the operand map (lane =
// channel, elements = 8 consecutive k) without a transposing store.  `base` points at (pixel 16*kc, channel c0) of
// the plane; 64-B channel chunks are XOR-swizzled with (pixel & 3) so that the 4 pixel rows of a half-wave fall on
// the 4 quarters of the 64 banks (pitch a multiple of 256 B); 128-B rows (64 channels) use (pixel >> 1) & 1, which
// does the same with their two chunks.
__device__ __forceinline__ int sp_key(int px, int pitch) { return pitch >= 256 ? (px & 3) : ((px >> 1) & 1); }

__device__ __forceinline__ bf16x8 tr_operand(const char* plane, int pitch, int c0, int kc, int lane) {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
    const int row0 = 16 * kc + 8 * (g >> 1) + qq;
    const int key = (pitch >= 256) ? qq : (qq >> 1);          // = sp_key(row0) = sp_key(row0 + 4)
    const int colb = (((c0 >> 5) ^ key) << 6) + ((16 * (g & 1) + 4 * pp) << 1);
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(plane + row0 * pitch + colb));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(plane + (row0 + 4) * pitch + colb));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// second launch bound = waves per SIMD the kernel must fit: two workgroups per CU for the fp32-MFMA tiles (the
// 8-wave ones must stay within 128 VGPRs), one 8-wave workgroup for the split-math tiles
// BF (VD_STORE_BF16, bf16-storage training): `in` and `dout` are bf16 tensors - a lane's EIGHT channels are one 16-byte
// load that goes to the (single) LDS plane untouched, and a K-step is BP = 64 pixels (the same bytes per step and twice the
// MFMAs per barrier of the fp32-tensor form: with one MFMA term per product the 32-pixel step was barrier-bound, 378 TF);
// the weight gradient itself stays fp32.
template <int WM, int WN, int TM, int TN, bool XF, bool SP, int NPL = 3, bool BF = false, int BP = WG_BP>
__global__ __launch_bounds__(WM * WN * 64, (SP ? (TM * TN == 1 ? 4 : 2) : WM * WN / 2)) void k_conv_wgrad(const vd_wgrad_desc p, float* __restrict__ dst,
                                                             int splits, int64_t pix_per_split, const int64_t zd_in,
                                                             const int64_t zd_do) {
    static_assert(!BF || (SP && NPL == 1 && !XF), "bf16-stored operands: one plane, no in-load transform");
    using LT = typename vd_select<BF, v4i, f32x4>::type;        // what a lane holds of one (pixel, VW channels): ext vectors (a
                                                                // struct type such as uint4 sent the register sets to scratch)
    constexpr int VW = BF ? 8 : 4;            // channels per lane and load
    constexpr int BM = WM * TM * 32;
    constexpr int NT = WM * WN * 64;          // 4 or 8 waves
    static_assert(WN * TN * 32 == WG_BN && (WM * WN == 4 || WM * WN == 8), "tile");
    constexpr int AROWS = NT * VW / BM;       // pixel rows of the dout tile one pass of NT 16-byte lanes covers
    constexpr int APASS = BP / AROWS;
    constexpr int BROWS = NT * VW / WG_BN;
    constexpr int BPASS = BP / BROWS;
    static_assert(APASS >= 1 && BPASS >= 1, "loader");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][WG_BP][BM]      dout
    float* Bs = smem + 2 * BP * BM;           // [2][BP][WG_BN]   in
    // SP: bf16 planes h/m/l, pixel-major, [2 stages][3][WG_BP][BM] then [2][3][WG_BP][WG_BN]
    constexpr int APL = BP * BM * 2, BPL = BP * WG_BN * 2;            // bytes of one plane
    char* As3 = reinterpret_cast<char*>(smem);
    char* Bs3 = As3 + 2 * NPL * APL;
    constexpr int NTERM = Terms<NPL>::N;
    static_assert(NPL != 2 || !XF, "no in-load transform in the fp16 split");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int sexp_a = 0, sexp_b = 0;               // NPL == 2: operand scales from the tensors' max-abs (A = dout, B = in)
    float scl_a = 1.f, scl_b = 1.f;
    if (NPL == 2) {
        sexp_a = vd_f16_scale_exp(vd_amax_read(p.amax_dout));
        sexp_b = vd_f16_scale_exp(vd_amax_read(p.amax_in));
        scl_a = __uint_as_float((unsigned)(127 + sexp_a) << 23);
        scl_b = __uint_as_float((unsigned)(127 + sexp_b) << 23);
    }
    const int Ktot = p.T * p.Ci;
    const int jtiles = (Ktot + WG_BN - 1) / WG_BN;
    const int mtiles = (p.Co + BM - 1) / BM;
    // block -> (split, jtile, mtile); XCD-aware: a contiguous run of logical ids per XCD, i.e. whole pixel ranges
    // (splits), so the dout / in panels of a split are fetched into ONE L2 and shared by its mtiles*jtiles blocks
#if VD_WG_REMAP
    int b = vd_xcd_remap(blockIdx.x, gridDim.x);
#else
    int b = blockIdx.x;
#endif
    const int tile_m = b % mtiles; b /= mtiles;
    const int tile_j = b % jtiles;
    const int split = b / jtiles;
    const int64_t P = (int64_t)p.N * p.Hg * p.Wg;
    const int64_t p_begin = (int64_t)split * pix_per_split;
    int64_t p_end = p_begin + pix_per_split;
    if (p_end > P) p_end = P;

    // B operand: this thread's column chunk -> (tap, channel), fixed for the whole reduction
    const int blpix = tid / (WG_BN / VW);
    const int blc = (tid % (WG_BN / VW)) * VW;
    const int j = tile_j * WG_BN + blc;
    const bool j_ok = j < Ktot;
    const int tap = j_ok ? j / p.Ci : 0;
    const int ci = j_ok ? j - tap * p.Ci : 0;
    const int dy = p.dy[tap], dx = p.dx[tap], dz = p.dz[tap];
    const int64_t boff = (int64_t)((dz * p.Hi + dy) * p.Wi + dx) * p.Ci + ci;
    // A operand (dout)
    const int alpix = tid / (BM / VW);
    const int alc = (tid % (BM / VW)) * VW;
    const int co = tile_m * BM + alc;
    const bool co_ok = co < p.Co;             // Co, Ci multiples of 4 => whole float4 in or out
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (XF && j_ok) {
        sc = *reinterpret_cast<const f32x4*>(p.in_scale + ci);
        sh = *reinterpret_cast<const f32x4*>(p.in_shift + ci);
    }

    // pixel cursor of each B pass: decoded once with divisions, then advanced by WG_BP pixels per K-step with
    // carries (the per-step divisions were ~40 % of the loop's VALU work)
    int cgx[BPASS], cgy[BPASS], cn[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int64_t pix = p_begin + blpix + BROWS * i;
        const unsigned pu = (unsigned)(pix < P ? pix : 0);
        const unsigned t = pu / (unsigned)p.Wg;
        cgx[i] = (int)(pu - t * (unsigned)p.Wg);
        const unsigned n_ = t / (unsigned)p.Hg;
        cgy[i] = (int)(t - n_ * (unsigned)p.Hg);
        cn[i] = (int)n_;
    }
    const int step_x = BP % p.Wg, step_y = BP / p.Wg;
    const bool one_wrap = (step_y + 1) <= p.Hg;     // at most one image boundary per step (true unless the map is tiny)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // register sets (k_conv_igemm has the rationale); the 64x64-per-wave tiles have no VGPRs left for a third
    constexpr int PD = (SP && TM * TN == 2) ? 3 : 2;
    LT ra[PD][APASS], rb[PD][BPASS];
    int64_t next_p = p_begin;                 // first pixel of the next tile to request
    auto gload = [&](LT (&ra)[APASS], LT (&rb)[BPASS]) {
        const int64_t pbase = next_p;
        next_p += BP;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int64_t pix = pbase + alpix + AROWS * i;
            const bool ok = pix < p_end && co_ok;
            const int64_t sel = ok ? pix * p.ldd + co : zd_do;
            if constexpr (BF) ra[i] = *reinterpret_cast<const v4i*>(reinterpret_cast<const __bf16*>(p.dout) + sel);
            else ra[i] = *reinterpret_cast<const f32x4*>(p.dout + sel);
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int64_t pix = pbase + blpix + BROWS * i;
            const int gx = cgx[i], gy = cgy[i], n = cn[i];
            const int iy = gy * p.in_stride + dy, ix = gx * p.in_stride + dx;
            const int fz = (n % p.Kfr) + dz;
            const bool bok = pix < p_end && j_ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi &&
                             (unsigned)fz < (unsigned)p.Kfr;
            const int64_t o = (int64_t)((n * p.Hi + gy * p.in_stride) * p.Wi + gx * p.in_stride) * p.Ci + boff;
            const int64_t sel = bok ? o : zd_in;
            if constexpr (BF) {
                rb[i] = *reinterpret_cast<const v4i*>(reinterpret_cast<const __bf16*>(p.in) + sel);
            } else {
                f32x4 vb = *reinterpret_cast<const f32x4*>(p.in + sel);
                if (XF) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tt = vb[e] * sc[e] + sh[e];
                        tt = tt > 0.f ? tt : tt * p.in_slope;
                        vb[e] = bok ? tt : 0.f;
                    }
                }
                rb[i] = vb;
            }
            // advance the cursor by BP pixels
            int nx = gx + step_x, ny = gy + step_y;
            if (nx >= p.Wg) { nx -= p.Wg; ++ny; }
            int nn = n;
            if (one_wrap) {
                if (ny >= p.Hg) { ny -= p.Hg; ++nn; }
            } else {
                nn += ny / p.Hg;
                ny = ny % p.Hg;
            }
            cgx[i] = nx; cgy[i] = ny; cn[i] = nn;
        }
    };
    auto lstore = [&](int buf, const LT (&ra)[APASS], const LT (&rb)[BPASS]) {
        if constexpr (BF) {
            char* a3 = As3 + buf * NPL * APL + (alc & 31) * 2;
            char* b3 = Bs3 + buf * NPL * BPL + (blc & 31) * 2;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const int px = alpix + AROWS * i;
                *reinterpret_cast<v4i*>(a3 + px * (BM * 2) + (((alc >> 5) ^ sp_key(px, BM * 2)) << 6)) = ra[i];
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const int px = blpix + BROWS * i;
                *reinterpret_cast<v4i*>(b3 + px * (WG_BN * 2) + (((blc >> 5) ^ sp_key(px, WG_BN * 2)) << 6)) = rb[i];
            }
            return;
        } else if constexpr (SP) {
            char* a3 = As3 + buf * NPL * APL + (alc & 31) * 2;
            char* b3 = Bs3 + buf * NPL * BPL + (blc & 31) * 2;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                uint2 h, m, l;
                const int px = alpix + AROWS * i;
                char* r = a3 + px * (BM * 2) + (((alc >> 5) ^ sp_key(px, BM * 2)) << 6);
                if (NPL == 3) {
                    split3(ra[i], h, m, l);
                    *reinterpret_cast<uint2*>(r + APL) = m;
                    *reinterpret_cast<uint2*>(r + 2 * APL) = l;
                } else if (NPL == 2) {
                    split2(ra[i], scl_a, h, l);
                    *reinterpret_cast<uint2*>(r + APL) = l;
                } else {
                    h = make_uint2(pk_bf16(ra[i][0], ra[i][1]), pk_bf16(ra[i][2], ra[i][3]));
                }
                *reinterpret_cast<uint2*>(r) = h;
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                uint2 h, m, l;
                const int px = blpix + BROWS * i;
                char* r = b3 + px * (WG_BN * 2) + (((blc >> 5) ^ sp_key(px, WG_BN * 2)) << 6);
                if (NPL == 3) {
                    split3(rb[i], h, m, l);
                    *reinterpret_cast<uint2*>(r + BPL) = m;
                    *reinterpret_cast<uint2*>(r + 2 * BPL) = l;
                } else if (NPL == 2) {
                    split2(rb[i], scl_b, h, l);
                    *reinterpret_cast<uint2*>(r + BPL) = l;
                } else {
                    h = make_uint2(pk_bf16(rb[i][0], rb[i][1]), pk_bf16(rb[i][2], rb[i][3]));
                }
                *reinterpret_cast<uint2*>(r) = h;
            }
            return;
        } else {
            float* a = As + buf * BP * BM;
            float* bb = Bs + buf * BP * WG_BN;
#pragma unroll
            for (int i = 0; i < APASS; ++i) *reinterpret_cast<f32x4*>(a + (alpix + AROWS * i) * BM + alc) = ra[i];
#pragma unroll
            for (int i = 0; i < BPASS; ++i) *reinterpret_cast<f32x4*>(bb + (blpix + BROWS * i) * WG_BN + blc) = rb[i];
        }
    };
    auto compute = [&](int buf) {
        if (SP) {
            const char* a3 = As3 + buf * NPL * APL;
            const char* b3 = Bs3 + buf * NPL * BPL;
#pragma unroll
            for (int kc = 0; kc < BP / 16; ++kc) {
                v4i fa[TM][NPL], fb[TN][NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        fa[mi][q] = __builtin_bit_cast(v4i, tr_operand(a3 + q * APL, BM * 2, wm * TM * 32 + mi * 32, kc, lane));
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        fb[ni][q] = __builtin_bit_cast(v4i, tr_operand(b3 + q * BPL, WG_BN * 2, wn * TN * 32 + ni * 32, kc, lane));
                }
#pragma unroll
                for (int t = 0; t < NTERM; ++t)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni)
                            acc[mi][ni] = mfma32<NPL>(fa[mi][Terms<NPL>::QA[t]], fb[ni][Terms<NPL>::QB[t]], acc[mi][ni]);
            }
            return;
        }
        const float* a = As + buf * BP * BM + (lane >> 5) * BM + wm * TM * 32 + (lane & 31);
        const float* bb = Bs + buf * BP * WG_BN + (lane >> 5) * WG_BN + wn * TN * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BP / 2; ++kk) {
            float fa[TM], fb[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) fa[mi] = a[kk * 2 * BM + mi * 32];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) fb[ni] = bb[kk * 2 * WG_BN + ni * 32];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    // same pipeline as k_conv_igemm: PD register sets, branch-free steady state, guarded tail
    const int nks = (p_end > p_begin) ? (int)vd_cdiv(p_end - p_begin, BP) : 0;
    constexpr int UN = (PD % 2 == 0) ? PD : 2 * PD;
    if (nks > 0) {
        gload(ra[0], rb[0]);
        lstore(0, ra[0], rb[0]);
    }
    __syncthreads();
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

    if (NPL == 2) {                    // undo the two operand scales: an exact power of two
        const int de = -(sexp_a + sexp_b);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __builtin_ldexpf(acc[mi][ni][r], de);
    }
    float* out = dst + (int64_t)split * p.Co * Ktot;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int jc = tile_j * WG_BN + wn * TN * 32 + ni * 32 + (lane & 31);
        if (jc >= Ktot) continue;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tile_m * BM + wm * TM * 32 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.Co) out[(int64_t)row * Ktot + jc] = acc[mi][ni][r];
            }
        }
    }
}

// Deterministic sum of the split-K slabs: dst[i] = ((ws[0][i] + ws[1][i]) + ws[2][i]) + ...  The additions stay in
// slab order (bit-reproducible), but the loads of 8 slabs are issued together: thin layers have 100+ slabs of a
// few thousand floats, and a one-load-at-a-time chain made the kernel latency bound (48 us average).
__global__ void k_reduce_slabs(const float* __restrict__ ws, float* __restrict__ dst, int64_t n4, int splits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(ws) + i;
    f32x4 a = src[0];
    int s = 1;
    for (; s + 8 <= splits; s += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(s + u) * n4];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; s < splits; ++s) a += src[(int64_t)s * n4];
    reinterpret_cast<f32x4*>(dst)[i] = a;
}

bool wgrad_split_math(const vd_wgrad_desc& d) {
    return (d.flags & VD_STORE_BF16) || ((d.flags & (VD_MATH_SPLIT | VD_MATH_BF16 | VD_MATH_F16X2)) && d.Co >= 64);
}
int wgrad_bm(const vd_wgrad_desc& d) {
    if (wgrad_split_math(d)) return d.Co >= 256 ? 256 : (d.Co >= 128 ? 128 : 64);      // (bf16 storage, Co = 32: half a 64-row tile)
    return d.Co <= 32 ? 32 : (d.Co <= 64 ? 64 : 128);
}
// workgroups resident at once: 2 per CU for the fp32-MFMA tiles, 1 per CU for the split-math tiles (LDS)
// A weight-gradient workgroup lives for its whole pixel range (0.2-1.3 ms) and, with 8 waves x 256 VGPRs, shares its CU
// with nothing: a grid that fills all 256 CUs starves the critical-path stream it runs beside - the rocprofv3 timeline
// showed a 44-block partial-sum reduction crawling through ONE free CU for 0.62 ms.  Round 1 therefore sized the grid for
// 240 CUs (VD_WGRAD_RESERVE = 16 withheld: +1.0 % on the training step then, 688.7 vs 682.0 frames/s).  With the BatchNorm
// reductions fused into the conv epilogues and the 3x3 layers on the halo-ring kernel the same A/B now favours the full
// chip (round 3, two boxes: reserve 0 / 0 1040-1042 and 1008-1013 frames/s against 1031-1035 and 1001-1004 for 16 / 16).
int wgrad_slots(const vd_wgrad_desc& d) {
    static const int reserve = getenv("VD_WGRAD_RESERVE") ? atoi(getenv("VD_WGRAD_RESERVE")) : 0;
    return (wgrad_split_math(d) && d.Co >= 128) ? 256 - reserve : 512 - 2 * reserve;
}

int wgrad_pick_splits(const vd_wgrad_desc& d) {
    if (d.splits > 0) return d.splits;
    const int64_t P = (int64_t)d.N * d.Hg * d.Wg;
    const int64_t tiles = vd_cdiv(d.Co, wgrad_bm(d)) * vd_cdiv((int64_t)d.T * d.Ci, WG_BN);
    // blocks = tiles * splits run 512 at a time (2 per CU).  Pick the split count whose last round is fullest,
    // searching from one round up to three; fewer splits win ties (every split adds a slab to write and re-read:
    // a 128x256 output split 512 ways moved 134 MB for an 11 GFLOP layer).
    int64_t s = 1;
    double best = -1.0;
    const int64_t slots = wgrad_slots(d);
    const int64_t lo = (slots / tiles) > 1 ? (slots / tiles) : 1, hi = vd_cdiv(3 * slots, tiles);
    for (int64_t c = lo; c <= hi; ++c) {
        const double x = (double)(tiles * c) / (double)slots;
        const double fill = x / (double)vd_cdiv(tiles * c, slots);
        if (fill > best + 0.02) { best = fill; s = c; }
    }
    const int64_t maxs = vd_cdiv(P, 8 * WG_BP);        // >= 8 k-steps per block
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return (int)s;
}

template <int WM, int WN, int TM, int TN>
void launch_wgrad_bf(const vd_wgrad_desc& d, float* dst, int splits, int64_t pps, hipStream_t s) {
    constexpr int BM = WM * TM * 32;
    constexpr int lds = 2 * WG_BP_BF * (BM + WG_BN) * 2;
    auto kfn = k_conv_wgrad<WM, WN, TM, TN, false, true, 1, true, WG_BP_BF>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const int64_t tiles = vd_cdiv(d.Co, BM) * vd_cdiv((int64_t)d.T * d.Ci, WG_BN);
    // the zero page, as bf16 element offsets from the two (bf16) tensors
    const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page());
    const int64_t zd_in = zp - reinterpret_cast<const __bf16*>(d.in), zd_do = zp - reinterpret_cast<const __bf16*>(d.dout);
    hipLaunchKernelGGL(kfn, dim3((unsigned)(tiles * splits)), dim3(WM * WN * 64), lds, s, d, dst, splits, pps, zd_in, zd_do);
}

template <int WM, int WN, int TM, int TN, bool SP, int NPL>
void launch_wgrad_n(const vd_wgrad_desc& d, float* dst, int splits, int64_t pps, hipStream_t s) {
    constexpr int BM = WM * TM * 32;
    constexpr int lds = SP ? 2 * NPL * WG_BP * (BM + WG_BN) * 2 : 2 * WG_BP * (BM + WG_BN) * (int)sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(!SP || BM >= 64, "the 64-B chunk swizzle of the split planes needs rows of >= 128 B");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wgrad<WM, WN, TM, TN, false, SP, NPL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wgrad<WM, WN, TM, TN, (NPL != 2), SP, NPL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const int64_t tiles = vd_cdiv(d.Co, BM) * vd_cdiv((int64_t)d.T * d.Ci, WG_BN);
    const float* zp = zero_page();
    const int64_t zd_in = zp - d.in, zd_do = zp - d.dout;
    if (d.in_scale && NPL != 2)
        hipLaunchKernelGGL((k_conv_wgrad<WM, WN, TM, TN, (NPL != 2), SP, NPL>), dim3((unsigned)(tiles * splits)), dim3(WM * WN * 64), lds,
                           s, d, dst, splits, pps, zd_in, zd_do);
    else
        hipLaunchKernelGGL((k_conv_wgrad<WM, WN, TM, TN, false, SP, NPL>), dim3((unsigned)(tiles * splits)), dim3(WM * WN * 64), lds,
                           s, d, dst, splits, pps, zd_in, zd_do);
}

template <int WM, int WN, int TM, int TN, bool SP = false>
void launch_wgrad(const vd_wgrad_desc& d, float* dst, int splits, int64_t pps, hipStream_t s) {
    if (SP && (d.flags & VD_MATH_BF16)) launch_wgrad_n<WM, WN, TM, TN, SP, SP ? 1 : 3>(d, dst, splits, pps, s);
    else if (SP && (d.flags & VD_MATH_F16X2)) launch_wgrad_n<WM, WN, TM, TN, SP, SP ? 2 : 3>(d, dst, splits, pps, s);
    else launch_wgrad_n<WM, WN, TM, TN, SP, 3>(d, dst, splits, pps, s);
}

// ---------------------------------------------------------------------------------------------
// weight packers
// ---------------------------------------------------------------------------------------------
__global__ void k_pack_fwd(const float* __restrict__ w, float* __restrict__ wp, int Co, int Co_pad,
                           int Ci, int T) {
    // wp[co][t*Ci + ci] = w[co][ci][t]   (w is OIHW / OIDHW, taps contiguous last)
    const int64_t total = (int64_t)Co_pad * T * Ci;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci);
        const int64_t r = i / Ci;
        const int t = (int)(r % T);
        const int co = (int)(r / T);
        wp[i] = (co < Co) ? w[((int64_t)co * Ci + ci) * T + t] : 0.f;
    }
}

struct TapList { int v[VD_MAX_TAPS]; };

template <typename OT>      // float, or __bf16 (the images of bf16-storage training)
__global__ void k_pack_dgrad(const float* __restrict__ w, OT* __restrict__ wp, int Co, int Co_pad,
                             int Ci, int T, const TapList taps, int ntaps, int src_packed) {
    // wp[ci][j*Co_pad + co] = w[co][ci][taps[j]]   (w OIHW, or fwd-packed [co][t*Ci+ci] if src_packed)
    const int64_t total = (int64_t)Ci * ntaps * Co_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co_pad);
        const int64_t r = i / Co_pad;
        const int j = (int)(r % ntaps);
        const int ci = (int)(r / ntaps);
        float v = 0.f;
        if (co < Co)
            v = src_packed ? w[((int64_t)co * T + taps.v[j]) * Ci + ci] : w[((int64_t)co * Ci + ci) * T + taps.v[j]];
        wp[i] = (OT)v;
    }
}

__global__ void k_unpack_wgrad(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int T) {
    const int64_t total = (int64_t)Co * Ci * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const int64_t r = i / T;
        const int ci = (int)(r % Ci);
        const int co = (int)(r / Ci);
        dw[i] = dwp[((int64_t)co * T + t) * Ci + ci];
    }
}

// stem im2col: in [N,H,W,3] (nchw=0) or [N,3,H,W] (nchw=1) -> col [N,H,W,32],
// col[.., (ky*3+kx)*3 + c] = in[y+ky-1][x+kx-1][c], zero padded, entries 27..31 = 0
__global__ void k_stem_im2col(const float* __restrict__ in, float* __restrict__ col, int N, int H, int W,
                              int nchw) {
    const int64_t total = (int64_t)N * H * W * 8;   // one float4 of the 32-wide row per thread
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 7);
        const int64_t pix = i >> 3;
        const int x = (int)(pix % W);
        const int64_t t = pix / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = q * 4 + e;
            float val = 0.f;
            if (k < 27) {
                const int c = k % 3, tap = k / 3;
                const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    val = nchw ? in[(((int64_t)n * 3 + c) * H + iy) * W + ix]
                               : in[(((int64_t)n * H + iy) * W + ix) * 3 + c];
                }
            }
            v[e] = val;
        }
        reinterpret_cast<f32x4*>(col)[i] = v;
    }
}

int check_taps(int T, const int32_t* dy, const int32_t* dx, int Hi, int Wi) {
    if (T < 1 || T > VD_MAX_TAPS) return 0;
    for (int t = 0; t < T; ++t)
        if (dy[t] < -64 || dy[t] > 64 || dx[t] < -64 || dx[t] > 64) return 0;
    return Hi < 4096 && Wi < 4096;
}

}  // namespace

extern "C" {

#if VD_STAMP
int vd_debug_stamps_f32(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_f32), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

int vd_conv_igemm(const vd_conv_desc* d, void* stream) {
    VD_REQUIRE(d && d->in && d->wp && d->out, "vd_conv_igemm: null pointer");
    VD_REQUIRE(d->Ci > 0 && d->Ci % 32 == 0, "vd_conv_igemm: Ci=%d must be a positive multiple of 32", d->Ci);
    VD_REQUIRE(d->N > 0 && d->Hg > 0 && d->Wg > 0 && d->Co > 0, "vd_conv_igemm: bad shape");
    VD_REQUIRE(check_taps(d->T, d->dy, d->dx, d->Hi, d->Wi), "vd_conv_igemm: bad tap list (T=%d)", d->T);
    VD_REQUIRE(d->Kfr >= 1 && d->Kfr < 128 && d->N % d->Kfr == 0, "vd_conv_igemm: bad Kfr=%d", d->Kfr);
    VD_REQUIRE(d->ldo >= d->Co, "vd_conv_igemm: ldo < Co");
    VD_REQUIRE((int64_t)d->N * d->Hi * d->Wi < (1ll << 31), "vd_conv_igemm: input pixel count overflows int32");
    VD_REQUIRE((int64_t)d->N * d->Hg * d->Wg < (1ll << 31), "vd_conv_igemm: GEMM row count overflows int32");
    VD_REQUIRE((d->Hg - 1) * d->out_stride + d->out_oy < d->Ho && (d->Wg - 1) * d->out_stride + d->out_ox < d->Wo,
               "vd_conv_igemm: output grid exceeds output tensor");
    VD_REQUIRE(!(d->flags & VD_EPI_RESIDUAL) || (d->residual && d->ldr >= d->Co), "vd_conv_igemm: residual missing");
    VD_REQUIRE((d->in_scale == nullptr) == (d->in_shift == nullptr), "vd_conv_igemm: in_scale/in_shift mismatch");
    VD_REQUIRE(!d->stats_part || (d->flags & 7) == 0, "vd_conv_igemm: fused BN statistics need a raw (epilogue-free) output");
    // (any output geometry: a stride-2 data gradient is four parity launches, each adding its own rows to the partial table)
    VD_REQUIRE(!d->bs_part || (!d->stats_part && !d->in_scale && d->bs_z && d->bs_scale && d->bs_shift && d->bs_mean && d->bs_invstd),
               "vd_conv_igemm: fused BN backward reductions need all five bs_* inputs (and no forward statistics / in-load transform)");
    VD_REQUIRE(!(d->flags & VD_MATH_F16X2) || (d->amax_in && d->amax_w && !d->in_scale),
               "vd_conv_igemm: VD_MATH_F16X2 needs amax_in and amax_w (and no in-load transform)");
    hipStream_t s = (hipStream_t)stream;
    if (d->in_scale) dispatch_igemm<true>(*d, s);
    else dispatch_igemm<false>(*d, s);
    VD_CHECK_LAUNCH("vd_conv_igemm");
    return VD_OK;
}

int vd_conv_igemm_mtiles(const vd_conv_desc* d) {
    if (!d) return 0;
    const int64_t M = (int64_t)d->N * d->Hg * d->Wg;
    if (d->flags & (VD_MATH_SPLIT | VD_MATH_BF16 | VD_MATH_F16X2)) return (int)vd_cdiv(M, igemm_split_tile_bm(igemm_split_resolve_tile(*d)));
    return (int)vd_cdiv(M, igemm_tile_bm(igemm_resolve_tile(*d)));
}

int64_t vd_conv_wgrad_ws_bytes(const vd_wgrad_desc* d) {
    if (!d) return 0;
    const int s = vd_wgrad_halo_ok(*d) ? vd_wgrad_halo_splits(*d) : wgrad_pick_splits(*d);
    return (s > 1) ? (int64_t)s * d->Co * d->T * d->Ci * (int64_t)sizeof(float) : 0;
}

int vd_conv_wgrad_uses_halo(const vd_wgrad_desc* d) { return (d && vd_wgrad_halo_ok(*d)) ? 1 : 0; }

int vd_conv_wgrad(const vd_wgrad_desc* d, void* ws, int64_t ws_bytes, void* stream) {
    VD_REQUIRE(d && d->in && d->dout && d->dwp, "vd_conv_wgrad: null pointer");
    VD_REQUIRE(d->Ci > 0 && d->Ci % 4 == 0 && d->Co % 4 == 0, "vd_conv_wgrad: Ci=%d Co=%d must be multiples of 4", d->Ci, d->Co);
    VD_REQUIRE(check_taps(d->T, d->dy, d->dx, d->Hi, d->Wi), "vd_conv_wgrad: bad tap list");
    VD_REQUIRE(d->Kfr >= 1 && d->Kfr < 128 && d->N % d->Kfr == 0, "vd_conv_wgrad: bad Kfr");
    VD_REQUIRE(d->ldd >= d->Co && d->ldd % 4 == 0, "vd_conv_wgrad: bad ldd");
    VD_REQUIRE((d->in_scale == nullptr) == (d->in_shift == nullptr), "vd_conv_wgrad: in_scale/in_shift mismatch");
    VD_REQUIRE(!(d->flags & VD_STORE_BF16) || !d->in_scale, "vd_conv_wgrad: no in-load transform on bf16-stored operands");
    // bf16-stored operands move as 16-byte loads of EIGHT channels: a group may not straddle a tap, a row end or Co
    VD_REQUIRE(!(d->flags & VD_STORE_BF16) || (d->Ci % 8 == 0 && d->Co % 8 == 0 && d->ldd % 8 == 0 &&
                                               ((uintptr_t)d->in | (uintptr_t)d->dout) % 16 == 0),
               "vd_conv_wgrad: VD_STORE_BF16 needs Ci=%d, Co=%d, ldd=%d multiples of 8 and 16-byte aligned in / dout", d->Ci, d->Co, d->ldd);
    VD_REQUIRE(!(d->flags & VD_MATH_F16X2) || d->Co < 64 || (d->amax_in && d->amax_dout && !d->in_scale),
               "vd_conv_wgrad: VD_MATH_F16X2 needs amax_in and amax_dout (and no in-load transform)");
    VD_REQUIRE((int64_t)d->N * d->Hg * d->Wg < (1ll << 31) && (int64_t)d->N * d->Hi * d->Wi < (1ll << 31),
               "vd_conv_wgrad: pixel count overflows int32");
    const bool halo = vd_wgrad_halo_ok(*d);           // VD_WGRAD_HALO and a geometry the halo-ring kernel serves
    const int splits = halo ? vd_wgrad_halo_splits(*d) : wgrad_pick_splits(*d);
    const int64_t need = (splits > 1) ? (int64_t)splits * d->Co * d->T * d->Ci * (int64_t)sizeof(float) : 0;
    if (need > ws_bytes || (need > 0 && !ws)) {
        vd_set_error("vd_conv_wgrad: workspace %lld < %lld", (long long)ws_bytes, (long long)need);
        return VD_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (halo) {
        vd_wgrad_halo_launch(*d, (splits > 1) ? (float*)ws : d->dwp, splits, s);
        VD_CHECK_LAUNCH("vd_conv_wgrad/halo");
        if (splits > 1) {
            const int64_t n4 = (int64_t)d->Co * d->T * d->Ci / 4;
            hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)vd_cdiv(n4, 64)), dim3(64), 0, s, (const float*)ws, d->dwp, n4, splits);
            VD_CHECK_LAUNCH("vd_conv_wgrad/reduce");
        }
        return VD_OK;
    }
    const int64_t P = (int64_t)d->N * d->Hg * d->Wg;
    const int bp = (d->flags & VD_STORE_BF16) ? WG_BP_BF : WG_BP;
    int64_t pps = vd_cdiv(vd_cdiv(P, splits), bp) * bp;
    float* dst = (splits > 1) ? (float*)ws : d->dwp;
    const int bm = wgrad_bm(*d);
    if (d->flags & VD_STORE_BF16) {
        if (bm == 256) launch_wgrad_bf<4, 2, 2, 2>(*d, dst, splits, pps, s);
        else if (bm == 128) launch_wgrad_bf<2, 4, 2, 1>(*d, dst, splits, pps, s);
        else launch_wgrad_bf<2, 4, 1, 1>(*d, dst, splits, pps, s);
    } else if (wgrad_split_math(*d)) {
        if (bm == 256) launch_wgrad<4, 2, 2, 2, true>(*d, dst, splits, pps, s);        // 256x128, 8 waves of 64x64
        else if (bm == 128) launch_wgrad<2, 4, 2, 1, true>(*d, dst, splits, pps, s);   // 128x128, 8 waves of 64x32
        else launch_wgrad<2, 4, 1, 1, true>(*d, dst, splits, pps, s);                  //  64x128, 8 waves of 32x32
    } else if (bm == 32) launch_wgrad<1, 4, 1, 1>(*d, dst, splits, pps, s);
    else if (bm == 64) launch_wgrad<2, 2, 1, 2>(*d, dst, splits, pps, s);
    else {
        static const int wv = getenv("VD_WGRAD_VARIANT") ? atoi(getenv("VD_WGRAD_VARIANT")) : 2;
        if (wv == 0) launch_wgrad<2, 2, 2, 2>(*d, dst, splits, pps, s);        // 128x128, 4 waves of 64x64
        else if (wv == 1) launch_wgrad<4, 2, 1, 2>(*d, dst, splits, pps, s);   // 128x128, 8 waves of 32x64
        else launch_wgrad<2, 4, 2, 1>(*d, dst, splits, pps, s);                // 128x128, 8 waves of 64x32
    }
    VD_CHECK_LAUNCH("vd_conv_wgrad");
    if (splits > 1) {
        const int64_t n4 = (int64_t)d->Co * d->T * d->Ci / 4;
        hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)vd_cdiv(n4, 64)), dim3(64), 0, s, (const float*)ws, d->dwp, n4, splits);
        VD_CHECK_LAUNCH("vd_conv_wgrad/reduce");
    }
    return VD_OK;
}

int vd_pack_weight_fwd(const float* w, float* wp, int Co, int Co_pad, int Ci, int kd, int kh, int kw, void* stream) {
    VD_REQUIRE(w && wp && Co > 0 && Co_pad >= Co && Ci > 0, "vd_pack_weight_fwd: bad args");
    const int T = kd * kh * kw;
    const int64_t total = (int64_t)Co_pad * T * Ci;
    const int nb = (int)(vd_cdiv(total, 256) < 4096 ? vd_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(k_pack_fwd, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wp, Co, Co_pad, Ci, T);
    VD_CHECK_LAUNCH("vd_pack_weight_fwd");
    return VD_OK;
}

static int pack_dgrad_any(const float* w, void* wp, int out_bf16, int Co, int Co_pad, int Ci, int kd, int kh, int kw,
                          const int32_t* taps, int ntaps, int src_packed, void* stream) {
    VD_REQUIRE(w && wp && taps && ntaps > 0 && ntaps <= VD_MAX_TAPS && Co_pad >= Co, "vd_pack_weight_dgrad: bad args");
    const int T = kd * kh * kw;
    TapList tl;
    for (int j = 0; j < VD_MAX_TAPS; ++j) tl.v[j] = (j < ntaps) ? taps[j] : 0;
    for (int j = 0; j < ntaps; ++j) VD_REQUIRE(taps[j] >= 0 && taps[j] < T, "vd_pack_weight_dgrad: tap index out of range");
    const int64_t total = (int64_t)Ci * ntaps * Co_pad;
    const int nb = (int)(vd_cdiv(total, 256) < 4096 ? vd_cdiv(total, 256) : 4096);
    if (out_bf16)
        hipLaunchKernelGGL(k_pack_dgrad<__bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wp, Co, Co_pad, Ci, T, tl, ntaps,
                           src_packed);
    else
        hipLaunchKernelGGL(k_pack_dgrad<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, (float*)wp, Co, Co_pad, Ci, T, tl, ntaps,
                           src_packed);
    VD_CHECK_LAUNCH("vd_pack_weight_dgrad");
    return VD_OK;
}

int vd_pack_weight_dgrad(const float* w, float* wp, int Co, int Co_pad, int Ci, int kd, int kh, int kw,
                         const int32_t* taps, int ntaps, int src_packed, void* stream) {
    return pack_dgrad_any(w, wp, 0, Co, Co_pad, Ci, kd, kh, kw, taps, ntaps, src_packed, stream);
}

/* the same layout as bf16 [Ci][ntaps * Co_pad] (bf16-storage training: Co_pad = the K pitch of the data gradient) */
int vd_pack_weight_dgrad_bf16(const float* w, void* wp_bf16, int Co, int Co_pad, int Ci, int kd, int kh, int kw,
                              const int32_t* taps, int ntaps, int src_packed, void* stream) {
    return pack_dgrad_any(w, wp_bf16, 1, Co, Co_pad, Ci, kd, kh, kw, taps, ntaps, src_packed, stream);
}

int vd_unpack_wgrad(const float* dwp, float* dw, int Co, int Ci, int kd, int kh, int kw, void* stream) {
    VD_REQUIRE(dwp && dw, "vd_unpack_wgrad: null");
    const int T = kd * kh * kw;
    const int64_t total = (int64_t)Co * Ci * T;
    const int nb = (int)(vd_cdiv(total, 256) < 4096 ? vd_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(k_unpack_wgrad, dim3(nb), dim3(256), 0, (hipStream_t)stream, dwp, dw, Co, Ci, T);
    VD_CHECK_LAUNCH("vd_unpack_wgrad");
    return VD_OK;
}

int vd_stem_im2col(const float* in, float* col, int N, int H, int W, int nchw, void* stream) {
    VD_REQUIRE(in && col && N > 0 && H > 0 && W > 0, "vd_stem_im2col: bad args");
    const int64_t total = (int64_t)N * H * W * 8;
    const int nb = (int)(vd_cdiv(total, 256) < 8192 ? vd_cdiv(total, 256) : 8192);
    hipLaunchKernelGGL(k_stem_im2col, dim3(nb), dim3(256), 0, (hipStream_t)stream, in, col, N, H, W, nchw);
    VD_CHECK_LAUNCH("vd_stem_im2col");
    return VD_OK;
}

}  // extern "C"
