// vd_conv_igemm_bf16.h - the bf16-storage / bf16-MFMA implicit-GEMM kernel template (k_conv_igemm_bf16) and its launcher,
// shared by vd_conv_bf16.hip (one workgroup per tile) and vd_conv_bf16_sk.hip (persistent stream-K form).  Internal.
#pragma once
#include "vd_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int ROW_B = 144;              // LDS row: 128 B of K + 16 B pad
constexpr int KCH = 64;                 // bf16 channels per K-step

__device__ __attribute__((aligned(64))) float g_zero_page_b[64];

// developer build (-DVD_STAMP=1, tools/stamp_conv.py): wave 0 of one mid-grid workgroup records s_memtime at the
// phase boundaries of the kernel
#ifndef VD_STAMP
#define VD_STAMP 0
#endif
#if VD_STAMP
__device__ unsigned long long g_stamps[16];
#define STAMP(i)                                                                          \
    do {                                                                                  \
        if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_stamps[i] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define STAMP(i)
#endif

// n / d for d >= 1 with rcp = 0xFFFFFFFF / d + 1: the multiply-high overshoots the quotient by at most one
__device__ __forceinline__ unsigned udiv_rcp(unsigned n, unsigned d, unsigned rcp) {
    unsigned q = d == 1u ? n : __umulhi(n, rcp);
    q -= (q * d > n) ? 1u : 0u;
    return q;
}

struct RowInfoB {
    int64_t off;     // element offset (bf16) of (pixel of tap (0,0,0), channel 8*(tid&7))
    unsigned mask;
};

// PAIR (Ci == 32, the two first-stage 3x3 convs): a K-step is still one 128-byte LDS row, made of TWO taps x 32
// channels - the packed weight row [T][32] is already contiguous that way, and each lane's 16-byte chunk picks its
// tap (chunk >> 2) - so 32-channel activations are stored unpadded and a 9-tap conv takes 5 K-steps instead of 9
// half-empty ones.
// HALO (3x3 stride-1 'same' geometry, 8-wave tiles): the activation operand is staged once per 64-channel chunk - the tile's
// BM output pixels plus W + 1 pixels either side, 128 B per pixel - and the nine taps read it at row offsets dy * W + dx
// (rows of taps outside the image read a zero row): (BM + 2 W + 2) / (9 BM) of the gather bytes.  Same scheme as the fp16-split
// halo loop of vd_conv.hip (there with the rationale and the measurements); here the stream moves bf16 rows untouched.
// BS (bf16-storage training, data gradients): the fused BatchNorm-backward reductions of vd_conv_igemm (vd_conv_desc.bs_*)
// in the epilogue - sum g and sum g * xhat of the layer whose dy this launch completes, from the fp32 values before they
// are rounded to bf16; z (p.bs_z) is a bf16 tensor of the output's geometry.
// SK: the persistent stream-K form (vd_conv_bf16_sk.hip) - equal runs of (tile, K-unit) units per workgroup, the tile a run
// boundary cuts finished by the next workgroup FROM the first one's accumulators (same MFMA chain, bit-identical outputs).
// The scheme, its hand-off protocol and its rationale are k_conv_igemm's (vd_conv_igemm.h); only the storage type differs.
// SK with sk_split = S >= 2: the SPLIT-K form for launches with too few tiles to fill the chip (batch-1 detection: 24 tiles of
// 128 x 128 at 19 x 19).  Workgroup b computes K-units [upt part / S, upt (part + 1) / S) of tile b / S (part = b % S),
// publishes its raw accumulators (the stream-K hand-off: write-through stores, drain, barrier, one relaxed agent-scope
// atomic), and the workgroup whose atomic finds S - 1 earlier arrivals resets the tile's counter, acquires, sums the S
// partials IN PART ORDER (its own read back like the others: one code path, a fixed association whoever arrives last) and
// runs the epilogue; the others exit.  Nobody waits for anybody: no poll, no co-residency requirement.  Deterministic, but
// NOT the association of the one-tile launch (S chains of K / S products summed, instead of one chain of K).
template <int WM, int WN, int TM, int TN, bool OUT_F32, bool PAIR, bool HALO = false, bool BS = false, bool SK = false>
__global__ __launch_bounds__(WM * WN * 64) void k_conv_igemm_bf16(const vd_conv_desc p, const int64_t zd_in,
                                                                  const int64_t zd_w, const int sk_lds_flag, const int sk_timeout, const int sk_split) {
    static_assert(!HALO || (WM * WN == 8 && !PAIR), "the halo loop exists for the 8-wave tiles");
    static_assert(!SK || !PAIR, "stream-K: not for the two-taps-per-step first-stage tiles");
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

    const int64_t M = (int64_t)p.N * p.Hg * p.Wg;
    const int ntile = (p.Co + BN - 1) / BN;
    const int Ktot = p.T * p.Ci;
    // ---- work items (see k_conv_igemm): !SK one whole tile; SK [K-prefix of the run's last tile, published], the whole
    // tiles, [K-suffix of the run's first tile, continuing the previous workgroup's accumulators]
    constexpr int KU = HALO ? 9 : 1;                       // K-steps per schedulable unit (HALO: one 64-channel chunk)
    const int upt = PAIR ? (p.T + 1) / 2 : p.T * (p.Ci / KCH) / KU;
    int sk_nitems = 1, sk_t0 = 0, sk_nwhole = 0, sk_brem = 0, sk_erem = 0, sk_tlast = 0, sk_tfirst = 0, sk_me = 0;
    if (SK && sk_split == 0) {
        const int G = (int)gridDim.x, g = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int gq = G >> 3, gr = G & 7;
        const int gs = gq + (g < gr ? 1 : 0);
        const int gbase = g * gq + (g < gr ? g : gr);
        const int64_t ntiles_all = vd_cdiv(M, BM) * ntile;
        const int tq = (int)(ntiles_all >> 3), tr = (int)(ntiles_all & 7);
        const int tg0 = g * tq + (g < tr ? g : tr), tgn = tq + (g < tr ? 1 : 0);
        const int64_t U = (int64_t)tgn * upt;
        const int64_t ub = U * j / gs, ue = U * (j + 1) / gs;
        sk_brem = (int)(ub % upt);
        sk_erem = (int)(ue % upt);
        const int tb = (int)(ub / upt), te = (int)(ue / upt);
        sk_tfirst = tg0 + tb;
        sk_t0 = sk_tfirst + (sk_brem ? 1 : 0);
        sk_nwhole = te - (tb + (sk_brem ? 1 : 0));
        if (sk_nwhole < 0) sk_nwhole = 0;
        sk_tlast = tg0 + te;
        sk_nitems = (sk_erem ? 1 : 0) + sk_nwhole + (sk_brem ? 1 : 0);
        sk_me = gbase + j;
    }

    STAMP(0);
    // Row geometry.  The tap table sits in lane registers (lane t = tap t, read back with v_readlane) so the mask
    // loop has no scalar-memory round trip per tap, and the two divisions per row are multiply-high by a reciprocal
    // computed once (exact after one correction: see udiv_rcp).
    const int tlane = (int)(threadIdx.x & 63) < p.T ? (int)(threadIdx.x & 63) : 0;
    const int tap_dy = p.dy[tlane], tap_dx = p.dx[tlane], tap_dz = p.dz[tlane];
    const unsigned rcp_w = 0xFFFFFFFFu / (unsigned)p.Wg + 1u, rcp_h = 0xFFFFFFFFu / (unsigned)p.Hg + 1u;

    for (int sk_it = 0; sk_it < sk_nitems; ++sk_it) {
    int tid = threadIdx.x;
    if (SK) asm volatile("" : "+v"(tid));     // (opaque per item: nothing derived from it is hoisted across the K loop)
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3;
    const int lc8 = (tid & 7) * 8;            // bf16 element offset of this lane's 16-byte chunk (within the K-step)
    const int lca = PAIR ? (tid & 3) * 8 : lc8;   // ... within its pixel's channel run
    const int ptap = PAIR ? ((tid >> 2) & 1) : 0; // PAIR: which tap of the K-step's pair this lane gathers
    int lid = SK ? 0 : vd_xcd_remap(blockIdx.x, gridDim.x);
    int kb = 0, ke = upt;
    int sk_mode = 0;                          // 0 whole tile, 1 publish the prefix, 2 continue a prefix
    if (SK) {
        __syncthreads();                      // the previous item's epilogue is done with LDS
        const bool hasA = sk_erem != 0;
        if (sk_split > 0) {                   // split-K: one (tile, part) per workgroup
            const int part = (int)blockIdx.x % sk_split;
            lid = (int)blockIdx.x / sk_split;
            kb = upt * part / sk_split;
            ke = upt * (part + 1) / sk_split;
            sk_me = (int)blockIdx.x;
            sk_mode = 3;
        }
        else if (hasA && sk_it == 0) { lid = sk_tlast; ke = sk_erem; sk_mode = 1; }
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
    RowInfoB ri[AP];
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
        ri[i].off = (int64_t)((n * p.Hi + riy[i]) * p.Wi + rix[i]) * p.Ci + lca;
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
    if (SK) {
        if (sk_mode == 2) {                   // take over the previous workgroup's accumulators (k_conv_igemm has the protocol)
            unsigned* cnt = reinterpret_cast<unsigned*>(p.sk_ws);
            unsigned* seen = cnt + VD_SK_MAX_WG;
            int* lds_flag = reinterpret_cast<int*>(smem_b + sk_lds_flag);
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
                if (!ok) __hip_atomic_fetch_add(cnt + VD_SK_MAX_WG - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

    // PD register sets of gathered tiles + a branch-free steady-state loop: see k_conv_igemm (vd_conv.hip)
    constexpr int PD = ((WM * WN == 4 && TM * TN == 4) || TM * TN == 8 || AP + BP > 8) ? 2 : 3;
    f32x4 ra[PD][AP], rb[PD][BP];
    int t_tap = 0, c0 = 0;
    if (SK && !HALO) {                        // taps innermost: K-step ks = (chunk ks / T, tap ks % T)
        c0 = (kb / p.T) * KCH;
        t_tap = kb - (kb / p.T) * p.T;
    }

    const int64_t tap_eo = (int64_t)((tap_dz * p.Hi + tap_dy) * p.Wi + tap_dx) * p.Ci;   // lane t: tap t
    const int tap_eo_lo = (int)(tap_eo & 0xffffffffll), tap_eo_hi = (int)(tap_eo >> 32);
    auto tap_off = [&](int t) -> int64_t {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(tap_eo_lo, t);
        const int hi = __builtin_amdgcn_readlane(tap_eo_hi, t);
        return ((int64_t)hi << 32) | (int64_t)lo;
    };
    int64_t tap_soff = tap_off(t_tap);
    auto gload = [&](f32x4 (&ra)[AP], f32x4 (&rb)[BP]) {
        if constexpr (PAIR) {
            // t_tap counts K-steps; taps 2*t_tap and 2*t_tap+1 (bit T of every mask is clear: an odd tail is zeros)
            const int t0 = 2 * t_tap;
            const int64_t so0 = tap_off(t0), so1 = tap_off(t0 + 1 < p.T ? t0 + 1 : t0);
            const int64_t soff = ptap ? so1 : so0;
            const int tl = t0 + ptap;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const bool ok = (ri[i].mask >> tl) & 1u;
                const int64_t sel = ok ? ri[i].off + soff : zd_in;
                ra[i] = *reinterpret_cast<const f32x4*>(in + sel);
            }
            const int koff = t_tap * KCH;
            const bool kin = koff + lc8 < Ktot;
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int64_t sel = (boff[i] >= 0 && kin) ? boff[i] + koff : zd_w;
                rb[i] = *reinterpret_cast<const f32x4*>(wp + sel);
            }
            ++t_tap;
            return;
        }
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

    const int nks = SK ? (ke - kb) * KU : (PAIR ? (p.T + 1) / 2 : p.T * (p.Ci / KCH));     // K-steps of this item
    STAMP(1);
    if constexpr (HALO) {
        constexpr int HT = 9;
        const int W = p.Wi;
        const int R = BM + 2 * (W + 1);
        const int ZROW = R, DROW = R + 1;                     // a zero row (taps outside the image), a sink row (idle stream slots)
        const int ABUF = (R + 2) * ROW_B;
        unsigned char* Ah = smem_b;                           // [2][R + 2][ROW_B]
        unsigned char* Bh = smem_b + 2 * ABUF;                // [2][BN][ROW_B]
        const int nchunk = SK ? ke : p.Ci / KCH;              // (end of this item's chunk range)
        const int cbeg = SK ? kb : 0;
        const int64_t m0 = (int64_t)tile_m * BM;
        const int64_t Mtot = (int64_t)p.N * p.Hi * p.Wi;
        if (tid < 16) *reinterpret_cast<f32x4*>(Ah + (tid >> 3) * ABUF + ZROW * ROW_B + (tid & 7) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        // halo stream: item (chunk c, slot s) = rows lrow + 64 s of chunk c's halo, 16 B per thread
        unsigned hvalid = 0u;
#pragma unroll
        for (int s2 = 0; s2 < HT; ++s2) {
            const int j = lrow + 64 * s2;
            const int64_t pin = m0 - (W + 1) + j;
            hvalid |= (j < R && (uint64_t)pin < (uint64_t)Mtot) ? (1u << s2) : 0u;
        }
        const int64_t hoff0 = (m0 - (W + 1) + lrow) * (int64_t)p.Ci + lc8;
        const int64_t hslot = 64ll * p.Ci;
        const int hl0 = lrow * ROW_B + (tid & 7) * 16;
        const int hsink = (DROW - lrow) * ROW_B;
        auto hload = [&](int c, int s) -> f32x4 {
            const bool ok = ((hvalid >> s) & 1u) && c < nchunk;
            const int64_t sel = ok ? hoff0 + (int64_t)s * hslot + (int64_t)c * KCH : zd_in;
            return *reinterpret_cast<const f32x4*>(in + sel);
        };
        auto hstore = [&](const f32x4 v, int c, int s) {
            const int ro_ = (c & 1) * ABUF + (((hvalid >> s) & 1u) ? s * 64 * ROW_B : hsink);
            *reinterpret_cast<f32x4*>(Ah + ro_ + hl0) = v;
        };
        int bt = 0, bc0 = cbeg * KCH;
        auto gloadB = [&](f32x4 (&rb)[BP]) {
            const int64_t koff = (int64_t)bt * p.Ci + bc0;
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int64_t sel = boff[i] >= 0 ? boff[i] + koff : zd_w;
                rb[i] = *reinterpret_cast<const f32x4*>(wp + sel);
            }
            if (++bt >= HT) { bt = 0; bc0 += KCH; }
        };
        auto lstoreB = [&](int buf, const f32x4 (&rb)[BP]) {
#pragma unroll
            for (int i = 0; i < BP; ++i)
                *reinterpret_cast<f32x4*>(Bh + (buf * BN + lrow + RPP * i) * ROW_B + (tid & 7) * 16) = rb[i];
        };
        // prologue: the whole halo of chunk 0 and the first weight tile requested together, the operand-row geometry under
        // their latency
        f32x4 t9[HT];
#pragma unroll
        for (int s2 = 0; s2 < HT; ++s2) t9[s2] = hload(cbeg, s2);
        gloadB(rb[0]);
        int jbase[TM];
        unsigned amask[TM];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int il = wm * TM * 32 + mi * 32 + (lane & 31);
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
            amask[mi] = m < M ? mk : 0u;
            jbase[mi] = il + W + 1;
        }
        const int tap_ro = tap_dy * W + tap_dx;               // lane t: halo-row offset of tap t
#pragma unroll
        for (int s2 = 0; s2 < HT; ++s2) hstore(t9[s2], cbeg, s2);
        lstoreB(0, rb[0]);
        __syncthreads();
        constexpr int HD = 3;                                 // halo items in flight: one request and one store per K-step
        f32x4 hv[HD];
        int hc = cbeg + 1, hs = 0, sc = cbeg + 1, ss = 0;     // next item to request / to store (chunk, slot)
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            hv[d] = hload(hc, hs);
            if (++hs >= HT) { hs = 0; ++hc; }
        }
#pragma unroll
        for (int d = 1; d < PD; ++d)
            if (d < nks) gloadB(rb[d]);
        int cc = cbeg, ct = 0;                                // chunk / tap of the step being multiplied
        auto computeH = [&](int wbuf) {
            const int ro = __builtin_amdgcn_readlane(tap_ro, ct);
            const unsigned char* a[TM];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int j = ((amask[mi] >> ct) & 1u) ? jbase[mi] + ro : ZROW;
                a[mi] = Ah + (cc & 1) * ABUF + j * ROW_B + 16 * (lane >> 5);
            }
            const unsigned char* b = Bh + wbuf * BN * ROW_B + (wn * TN * 32 + (lane & 31)) * ROW_B + 16 * (lane >> 5);
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(a[mi] + kc * 32);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8*>(b + ni * 32 * ROW_B + kc * 32);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
            }
            if (++ct >= HT) { ct = 0; ++cc; }
        };
        auto hstream = [&](int u) {
            // store the item requested HD steps ago into the NEXT chunk's buffer (nobody reads it before the barrier that ends
            // this chunk's last step), then reuse its registers for a new request; items past the last chunk read the zero
            // page and land in the sink row or in a buffer nobody reads any more: no conditional memory operation
            hstore(hv[u % HD], sc, ss);
            if (++ss >= HT) { ss = 0; ++sc; }
            hv[u % HD] = hload(hc, hs);
            if (++hs >= HT) { hs = 0; ++hc; }
        };
        constexpr int UNH = 6;                                // = lcm(2 weight stages, PD in {2, 3}, HD)
        static_assert(UNH % PD == 0 && UNH % HD == 0, "unroll vs register sets");
        int ks = 0;
        for (; ks + UNH + PD <= nks; ks += UNH) {
#pragma unroll
            for (int u = 0; u < UNH; ++u) {
                gloadB(rb[u % PD]);
                computeH(u & 1);
                lstoreB((u + 1) & 1, rb[(u + 1) % PD]);
                hstream(u);
                __syncthreads();
            }
        }
        for (; ks < nks; ks += UNH) {
#pragma unroll
            for (int u = 0; u < UNH; ++u) {
                if (ks + u < nks) {
                    if (ks + u + PD < nks) gloadB(rb[u % PD]);
                    computeH(u & 1);
                    if (ks + u + 1 < nks) lstoreB((u + 1) & 1, rb[(u + 1) % PD]);
                    hstream(u);
                    __syncthreads();
                }
            }
        }
    } else {
    gload(ra[0], rb[0]);
    STAMP(2);
    lstore(0, ra[0], rb[0]);
    __syncthreads();
    STAMP(3);
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
    }   // !HALO

    STAMP(4);
    if (SK) {
        if (sk_mode == 1 || sk_mode == 3) {   // publish raw accumulators (a K-prefix, or a split's part): write-through, drained
            char* slot = reinterpret_cast<char*>(p.sk_ws) + VD_SK_HEADER_BYTES + (int64_t)sk_me * (BM * BN * 4);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slot, 0, BM * BN * 4, 0x00027000);
            typedef int v4i_ __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_, v), rs, (((mi * TN + ni) * 4 + q) * NT + tid) * 16, 0, 16);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (sk_mode == 1) {               // stream-K: one counter bump, the next workgroup of the group takes it from here
                if (tid == 0)
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p.sk_ws) + sk_me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            // split-K: the last part to arrive finishes the tile
            int* lds_flag = reinterpret_cast<int*>(smem_b + sk_lds_flag);
            if (tid == 0) {
                unsigned* tcnt = reinterpret_cast<unsigned*>(p.sk_ws) + VD_SK_SPLIT_CNT_OFF + lid;
                const unsigned old = __hip_atomic_fetch_add(tcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == (unsigned)(sk_split - 1);
                if (last) {
                    __hip_atomic_store(tcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                *lds_flag = last;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!*lds_flag) continue;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            const f32x4* part0 = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.sk_ws) + VD_SK_HEADER_BYTES +
                                                                (int64_t)lid * sk_split * (BM * BN * 4));
            for (int sp = 0; sp < sk_split; ++sp) {
                const f32x4* ps = part0 + (int64_t)sp * (BM * BN / 4);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = ps[((mi * TN + ni) * 4 + q) * NT + tid];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mi][ni][4 * q + e] += v[e];
                        }
            }
        }
    }
    // ---- epilogue (fp32 math).  Per wave, one 32x32
    // accumulator tile at a time is transposed through a private LDS patch so that each lane owns 4 consecutive
    // columns of 4 rows: scale/shift/LeakyReLU/residual on 4-vectors and 8-byte (bf16 x4) or 16-byte (fp32 heads)
    // stores instead of 2 bytes per lane.
    const __bf16* res = reinterpret_cast<const __bf16*>(p.residual);
    constexpr int SLD = 36;
    float* stg = reinterpret_cast<float*>(smem_b) + wave * (32 * SLD);
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    const bool has_aff = p.flags & VD_EPI_AFFINE, has_res = p.flags & VD_EPI_RESIDUAL, has_leaky = p.flags & VD_EPI_LEAKY;
    // output pixel of GEMM row m: the row itself (forward, stride-1 data gradients) or the strided / offset pixel of a
    // stride-2 data gradient's parity class (vd_conv.hip has the same map)
    const bool direct = (p.out_stride == 1 && p.out_oy == 0 && p.out_ox == 0 && p.Ho == p.Hg && p.Wo == p.Wg);
    auto out_pix = [&](int64_t m) -> int64_t {
        if (direct) return m;
        const unsigned mu = (unsigned)m;
        const unsigned t = udiv_rcp(mu, (unsigned)p.Wg, rcp_w);
        const int gx = (int)(mu - t * (unsigned)p.Wg);
        const unsigned n = udiv_rcp(t, (unsigned)p.Hg, rcp_h);
        const int gy = (int)(t - n * (unsigned)p.Hg);
        return ((int64_t)n * p.Ho + (gy * p.out_stride + p.out_oy)) * p.Wo + (gx * p.out_stride + p.out_ox);
    };
    float bs_acc1[BS ? TN : 1][4], bs_acc2[BS ? TN : 1][4];      // BS: sum g / sum g * xhat of this lane's columns
#pragma unroll
    for (int ni = 0; ni < (BS ? TN : 1); ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) bs_acc1[ni][e] = bs_acc2[ni][e] = 0.f;
    const bool vec_ok = (p.ldo % 4 == 0) && ((uintptr_t)p.out % 16 == 0) && (p.Co % 4 == 0) &&
                        (!has_res || ((p.ldr % 4 == 0) && ((uintptr_t)p.residual % 8 == 0))) &&
                        (!has_aff || (((uintptr_t)p.scale | (uintptr_t)p.shift) % 16 == 0));
    if (SK || vec_ok) {                       // (SK: the host launches the form only where this path applies)
        // Straight-line path (every layer of the network): all scale/shift and residual loads are issued first, then
        // each 32x32 accumulator block goes through the wave's LDS patch (LDS is in order within a wave, so the
        // compiler barrier is all the synchronisation a block needs) and is stored without ever waiting on a store.
        f32x4 sc[TN], sh[TN];
        int colv[TN];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = tile_n * BN + wn * TN * 32 + ni * 32 + ec4;
            colv[ni] = col < p.Co ? col : -1;
            const int cc = col < p.Co ? col : 0;
            sc[ni] = (has_aff && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + cc) : f32x4{1.f, 1.f, 1.f, 1.f};
            sh[ni] = (has_aff && p.shift) ? *reinterpret_cast<const f32x4*>(p.shift + cc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const bool bstat = BS && p.bs_part != nullptr;
        const __bf16* bz = reinterpret_cast<const __bf16*>(p.bs_z);
        f32x4 qsc[BS ? TN : 1], qsh[BS ? TN : 1], qmu[BS ? TN : 1], qis[BS ? TN : 1];
        bf16x4 zv[BS ? TM : 1][BS ? TN : 1][4];
        if (BS) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int cc = colv[ni] < 0 ? 0 : colv[ni];
                const f32x4 z0 = {0.f, 0.f, 0.f, 0.f};
                qsc[ni] = bstat ? *reinterpret_cast<const f32x4*>(p.bs_scale + cc) : z0;
                qsh[ni] = bstat ? *reinterpret_cast<const f32x4*>(p.bs_shift + cc) : z0;
                qmu[ni] = bstat ? *reinterpret_cast<const f32x4*>(p.bs_mean + cc) : z0;
                qis[ni] = bstat ? *reinterpret_cast<const f32x4*>(p.bs_invstd + cc) : z0;
            }
            if (bstat) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + erow + 8 * i;
                            m = m < M ? m : M - 1;
                            zv[mi][ni][i] = *reinterpret_cast<const bf16x4*>(bz + out_pix(m) * p.ldo + (colv[ni] < 0 ? 0 : colv[ni]));
                        }
            }
        }
        bf16x4 rv[TM][TN][4];
        if (has_res) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + erow + 8 * i;
                        m = m < M ? m : M - 1;
                        rv[mi][ni][i] = *reinterpret_cast<const bf16x4*>(res + out_pix(m) * p.ldr + (colv[ni] < 0 ? 0 : colv[ni]));
                    }
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + (lane & 31)] = acc[mi][ni][r];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                f32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(stg + (erow + 8 * i) * SLD + ec4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + erow + 8 * i;
                    f32x4 t = v[i];
                    if (has_aff) t = t * sc[ni] + sh[ni];
                    if (has_leaky) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = t[e] > 0.f ? t[e] : t[e] * p.slope;
                    }
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] += (float)rv[mi][ni][i][e];
                    }
                    if (BS && bstat) {
                        const bool ok = m < M && colv[ni] >= 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float z = (float)zv[mi][ni][i][e];
                            const float u = z * qsc[ni][e] + qsh[ni][e];
                            float g = u > 0.f ? t[e] : t[e] * p.bs_slope;
                            g = ok ? g : 0.f;
                            bs_acc1[ni][e] += g;
                            bs_acc2[ni][e] += g * (z - qmu[ni][e]) * qis[ni][e];
                        }
                    }
                    if (m < M && colv[ni] >= 0) {
                        const int64_t op = out_pix(m);
                        if (OUT_F32) *reinterpret_cast<f32x4*>(p.out + op * p.ldo + colv[ni]) = t;
                        else {
                            bf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (__bf16)t[e];
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + op * p.ldo + colv[ni]) = o;
                        }
                    }
                }
            }
    } else if constexpr (!SK) {
        // general path (odd leading dimensions / unaligned pointers): element-wise tails
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = tile_n * BN + wn * TN * 32 + ni * 32 + ec4;
            const int nvalid = p.Co - col;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
            if (has_aff) {
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
                    const int64_t op = out_pix(m);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < nvalid) {
                            float t = v[e];
                            if (has_aff) t = t * sc[e] + sh[e];
                            if (has_leaky) t = t > 0.f ? t : t * p.slope;
                            if (has_res) t += (float)res[op * p.ldr + col + e];
                            if (OUT_F32) p.out[op * p.ldo + col + e] = t;
                            else reinterpret_cast<__bf16*>(p.out)[op * p.ldo + col + e] = (__bf16)t;
                        }
                }
            }
        }
    }
    STAMP(5);
    // ---- fused BatchNorm backward reductions: one row of the partial table [tile_m][2 * Co] per M tile (as k_conv_igemm)
    if constexpr (BS) {
        if (p.bs_part != nullptr) {
            __syncthreads();        // every wave is done with its staging patch
            float* red = reinterpret_cast<float*>(smem_b);      // [WM][BN][2]
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = bs_acc1[ni][e], b = bs_acc2[ni][e];
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
    }
    // ---- fused BatchNorm statistics (bf16-storage training forward): per-column sum / sum of squares of this block's raw
    // conv outputs, from the fp32 ACCUMULATORS (before they are rounded to bf16), one row of the partial table
    // [tile_m][2 * Co] per M tile - no atomics, vd_bn_sum_partials finishes in fp64 in a fixed order (as k_conv_igemm)
    if (p.stats_part) {
        __syncthreads();        // every wave is done with its staging patch
        float* red = reinterpret_cast<float*>(smem_b);      // [WM][BN][2]
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = (int64_t)tile_m * BM + wm * TM * 32 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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
#if VD_STAMP
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_stamps[6] = wall_clock64();
#endif
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

// LDS bytes of the halo loop: two halo buffers of BM + 2 (W + 1) rows (+ zero row + sink row) and two weight stages
inline int64_t halo_lds_b(int BM, int BN, int W) { return 2ll * (BM + 2 * (W + 1) + 2) * ROW_B + 2ll * BN * ROW_B; }

inline bool halo_ok_b(const vd_conv_desc& d, int BM, int BN) {
    if ((d.flags & VD_MATH_NOHALO) || d.T != 9 || d.in_stride != 1 || d.Kfr != 1 || d.Hg != d.Hi || d.Wg != d.Wi || d.Ci % KCH)
        return false;
    for (int t = 0; t < 9; ++t)
        if (d.dy[t] < -1 || d.dy[t] > 1 || d.dx[t] < -1 || d.dx[t] > 1 || d.dz[t] != 0) return false;
    // nine stream slots of 64 rows; the epilogue's staging patches (36 KB) reuse the same LDS
    return BM + 2 * (d.Wi + 1) <= 9 * 64 && halo_lds_b(BM, BN, d.Wi) <= 160 * 1024;
}

inline int device_cus_b() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// persistent grid of the stream-K form, or 0 where it does not apply (k_conv_igemm's rules: streamk_grid in vd_conv_igemm.h)
inline int streamk_grid_b(const vd_conv_desc& d, int BM, int BN, int wg_per_cu) {
    if (!(d.flags & VD_CONV_STREAMK) || !d.sk_ws || wg_per_cu < 1) return 0;
    auto al = [](const void* q, int a) { return (uintptr_t)q % a == 0; };
    if (d.ldo % 4 || d.Co % 4 || !al(d.out, 16) || ((d.flags & VD_EPI_RESIDUAL) && (d.ldr % 4 || !al(d.residual, 8))) ||
        ((d.flags & VD_EPI_AFFINE) && (!al(d.scale, 16) || !al(d.shift, 16))))
        return 0;
    const int64_t ntiles = vd_cdiv((int64_t)d.N * d.Hg * d.Wg, BM) * vd_cdiv(d.Co, BN);
    const int G = device_cus_b() * wg_per_cu;
    if (G > VD_SK_MAX_WG - 2 || G < 8 || (ntiles >> 3) < (G + 7) / 8) return 0;
    if (ntiles <= G + G / 16) return 0;
    if (d.sk_ws_bytes < (int64_t)VD_SK_HEADER_BYTES + (int64_t)G * BM * BN * 4) return 0;
    return G;
}

// split-K factor of a launch whose tiles cannot fill the chip (see the kernel's header comment), or 0: at least two parts, at
// most 8 (the finishing workgroup reads S x BM x BN x 4 bytes alone), at least `min_units` K-units per part, no more
// workgroups than the chip holds at once (so that the parts of a tile run together), a counter and a slot per (tile, part)
inline int splitk_factor_b(const vd_conv_desc& d, int BM, int BN, int wg_per_cu, int upt, int min_units) {
    if (!(d.flags & VD_CONV_SPLITK) || !d.sk_ws || wg_per_cu < 1) return 0;
    static const int off = getenv("VD_SPLITK") ? !atoi(getenv("VD_SPLITK")) : 0;
    if (off) return 0;
    auto al = [](const void* q, int a) { return (uintptr_t)q % a == 0; };
    if (d.ldo % 4 || d.Co % 4 || !al(d.out, 16) || ((d.flags & VD_EPI_RESIDUAL) && (d.ldr % 4 || !al(d.residual, 8))) ||
        ((d.flags & VD_EPI_AFFINE) && (!al(d.scale, 16) || !al(d.shift, 16))))
        return 0;
    const int64_t ntiles = vd_cdiv((int64_t)d.N * d.Hg * d.Wg, BM) * vd_cdiv(d.Co, BN);
    const int64_t slots = (int64_t)device_cus_b() * wg_per_cu;
    if (ntiles > VD_SK_SPLIT_MAX_TILES || ntiles * 2 > slots) return 0;
    int64_t S = slots / ntiles;
    if (S > upt / min_units) S = upt / min_units;
    if (S > 8) S = 8;
    if (S < 2) return 0;
    if (d.sk_ws_bytes < (int64_t)VD_SK_HEADER_BYTES + ntiles * S * BM * BN * 4) return 0;
    return (int)S;
}

// returns 0 when launched; 1 when SK was asked for and does not apply (the caller launches the classic form); query_only:
// 0 = would launch as a stream-K grid, 2 = as a split-K grid, 1 = neither
template <int WM, int WN, int TM, int TN, bool OUT_F32, bool PAIR = false, bool HALO = false, bool BS = false, bool SK = false>
int launch_b2(const vd_conv_desc& d, hipStream_t s, bool query_only = false) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int lds_gen = 2 * (BM + BN) * ROW_B;
    int lds = HALO ? (int)(halo_lds_b(BM, BN, d.Wi) > 8 * 32 * 36 * 4 ? halo_lds_b(BM, BN, d.Wi) : 8 * 32 * 36 * 4) : lds_gen;
    auto kfn = k_conv_igemm_bf16<WM, WN, TM, TN, OUT_F32, PAIR, HALO, BS, SK>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  HALO ? 160 * 1024 : lds_gen + (SK ? 16 : 0));
        attr_done = true;
    }
    const int64_t M = (int64_t)d.N * d.Hg * d.Wg;
    int64_t nblk = vd_cdiv(M, BM) * vd_cdiv(d.Co, BN);
    int sk_flag_off = 0, sk_split = 0;
    if (SK) {
        lds = (lds + 15) & ~15;
        if (lds + 16 > 160 * 1024) return 1;
        // co-resident workgroups per CU of THIS instantiation at this LDS size (registers, LDS, waves)
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kfn), WM * WN * 64, lds + 16) != hipSuccess) per_cu = 0;
        const int G = streamk_grid_b(d, BM, BN, per_cu > 2 ? 2 : per_cu);       // (at most two per CU: more only shrink the runs)
        if (G == 0) {
            // too few tiles for a stream-K grid: the split-K form, where its conditions hold (K-units: a 64-channel chunk of
            // nine steps in the halo loop - at least one per part; at least four K-steps per part otherwise)
            const int upt = d.T * (d.Ci / KCH) / (HALO ? 9 : 1);
            sk_split = BS ? 0 : splitk_factor_b(d, BM, BN, per_cu > 4 ? 4 : per_cu, upt, HALO ? 1 : 4);
            if (sk_split == 0) return 1;
            nblk *= sk_split;
        } else {
            nblk = G;
        }
        if (query_only) return sk_split ? 2 : 0;
        sk_flag_off = lds;
        lds += 16;
    }
    const __bf16* zp = reinterpret_cast<const __bf16*>(zero_page_b());
    const int64_t zd_in = zp - reinterpret_cast<const __bf16*>(d.in);
    const int64_t zd_w = zp - reinterpret_cast<const __bf16*>(d.wp);
    static const int sk_timeout = getenv("VD_SK_TIMEOUT_TICKS") ? atoi(getenv("VD_SK_TIMEOUT_TICKS")) : VD_SK_TIMEOUT_TICKS;
    hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(WM * WN * 64), lds, s, d, zd_in, zd_w, sk_flag_off, sk_timeout, sk_split);
    return 0;
}

template <int WM, int WN, int TM, int TN, bool OUT_F32, bool PAIR = false>
void launch_b(const vd_conv_desc& d, hipStream_t s) {
    if constexpr (!OUT_F32 && TM * TN <= 4) {
        if (d.bs_part) {              // fused BatchNorm-backward reductions (the entry point keeps the other tiles out)
            if constexpr (WM * WN == 8 && !PAIR) {
                if (halo_ok_b(d, WM * TM * 32, WN * TN * 32)) { launch_b2<WM, WN, TM, TN, OUT_F32, PAIR, true, true>(d, s); return; }
            }
            { launch_b2<WM, WN, TM, TN, OUT_F32, PAIR, false, true>(d, s); return; }
        }
    }
    if constexpr (WM * WN == 8 && !PAIR && TM * TN <= 4) {
        if (halo_ok_b(d, WM * TM * 32, WN * TN * 32)) { launch_b2<WM, WN, TM, TN, OUT_F32, PAIR, true>(d, s); return; }
    }
    launch_b2<WM, WN, TM, TN, OUT_F32, PAIR, false>(d, s);
}

}  // namespace
