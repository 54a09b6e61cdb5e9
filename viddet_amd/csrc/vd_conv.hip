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

#include "vd_conv_igemm.h"

int vd_igemm_sk_dispatch(const vd_conv_desc& d, int tile, hipStream_t s);     // vd_conv_sk.hip
int vd_igemm_sk_applies(const vd_conv_desc& d, int tile);
int vd_igemm_par_dispatch(const vd_conv_desc& d, int tile, hipStream_t s);    // vd_conv_par.hip
int vd_igemm_par_tile(int tile);
int vd_igemm_par_bm(int tile);

namespace {

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
    // VD_CONV_STREAMK: the persistent stream-K grid of the same tile (vd_conv_sk.hip) where it applies
    if (!XF && (d.flags & VD_CONV_STREAMK) && vd_igemm_sk_dispatch(d, igemm_split_resolve_tile(d), s) == 0) return 0;
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
    VD_REQUIRE((d->flags & VD_CONV_PARITY4) || d->ldo >= d->Co, "vd_conv_igemm: ldo < Co");
    VD_REQUIRE((int64_t)d->N * d->Hi * d->Wi < (1ll << 31), "vd_conv_igemm: input pixel count overflows int32");
    VD_REQUIRE((int64_t)d->N * d->Hg * d->Wg < (1ll << 31), "vd_conv_igemm: GEMM row count overflows int32");
    VD_REQUIRE((d->Hg - 1) * d->out_stride + d->out_oy < d->Ho && (d->Wg - 1) * d->out_stride + d->out_ox < d->Wo,
               "vd_conv_igemm: output grid exceeds output tensor");
    VD_REQUIRE(!(d->flags & VD_EPI_RESIDUAL) || (d->residual && (d->ldr >= d->Co || (d->flags & VD_CONV_PARITY4))), "vd_conv_igemm: residual missing");
    VD_REQUIRE((d->in_scale == nullptr) == (d->in_shift == nullptr), "vd_conv_igemm: in_scale/in_shift mismatch");
    VD_REQUIRE(!d->stats_part || (d->flags & 7) == 0, "vd_conv_igemm: fused BN statistics need a raw (epilogue-free) output");
    // (any output geometry: a stride-2 data gradient is four parity launches, each adding its own rows to the partial table)
    VD_REQUIRE(!d->bs_part || (!d->stats_part && !d->in_scale && d->bs_z && d->bs_scale && d->bs_shift && d->bs_mean && d->bs_invstd),
               "vd_conv_igemm: fused BN backward reductions need all five bs_* inputs (and no forward statistics / in-load transform)");
    VD_REQUIRE(!(d->flags & VD_CONV_STREAMK) || !d->sk_ws || ((uintptr_t)d->sk_ws % 16 == 0 && d->sk_ws_bytes >= VD_SK_HEADER_BYTES),
               "vd_conv_igemm: VD_CONV_STREAMK workspace must be 16-byte aligned and hold its header");
    VD_REQUIRE(!(d->flags & VD_MATH_F16X2) || (d->amax_in && d->amax_w && !d->in_scale),
               "vd_conv_igemm: VD_MATH_F16X2 needs amax_in and amax_w (and no in-load transform)");
    hipStream_t s = (hipStream_t)stream;
    if (d->flags & VD_CONV_PARITY4) {
        VD_REQUIRE((d->flags & VD_MATH_F16X2) && !(d->flags & (VD_EPI_AFFINE | VD_EPI_LEAKY)) && !d->stats_part && !d->in_scale,
                   "vd_conv_igemm: VD_CONV_PARITY4 runs in VD_MATH_F16X2 with a residual / bs_* epilogue only");
        VD_REQUIRE(d->T == 4 && d->dy[0] == 0 && d->dx[0] == 0 && d->dy[1] == 0 && d->dx[1] == 1 && d->dy[2] == 1 && d->dx[2] == 0 &&
                   d->dy[3] == 1 && d->dx[3] == 1 && d->in_stride == 1 && d->Hg == d->Hi && d->Wg == d->Wi && d->Kfr == 1,
                   "vd_conv_igemm: VD_CONV_PARITY4 taps are the offsets (0,0) (0,1) (1,0) (1,1) on the gradient's own grid");
        VD_REQUIRE(d->out_stride == 2 && d->Ho == 2 * d->Hg && d->Wo == 2 * d->Wg && d->par_cin >= 32 && d->par_cin % 32 == 0 &&
                   d->Co == 4 * d->par_cin && d->ldo >= d->par_cin && d->ldo % 4 == 0 && (uintptr_t)d->out % 16 == 0 &&
                   (!(d->flags & VD_EPI_RESIDUAL) || (d->ldr % 4 == 0 && (uintptr_t)d->residual % 16 == 0)),
                   "vd_conv_igemm: VD_CONV_PARITY4 geometry (even output, Co = 4 par_cin, aligned rows)");
        VD_REQUIRE(!d->bs_part || (((uintptr_t)d->bs_z | (uintptr_t)d->bs_scale | (uintptr_t)d->bs_shift | (uintptr_t)d->bs_mean |
                                    (uintptr_t)d->bs_invstd) % 16 == 0), "vd_conv_igemm: VD_CONV_PARITY4 bs_* alignment");
        vd_igemm_par_dispatch(*d, vd_igemm_par_tile(d->tile), s);
        VD_CHECK_LAUNCH("vd_conv_igemm/parity4");
        return VD_OK;
    }
    if (d->in_scale) dispatch_igemm<true>(*d, s);
    else dispatch_igemm<false>(*d, s);
    VD_CHECK_LAUNCH("vd_conv_igemm");
    return VD_OK;
}

int vd_conv_igemm_streamk(const vd_conv_desc* d) {
    if (!d || !(d->flags & VD_CONV_STREAMK) || !(d->flags & VD_MATH_F16X2)) return 0;
    return vd_igemm_sk_applies(*d, igemm_split_resolve_tile(*d));
}

int64_t vd_conv_igemm_streamk_ws_bytes(void) {
    // the largest accumulator tile per slot: the 256 x 256 tiles of k_conv_igemm_bf16 at one workgroup per CU
    return (int64_t)VD_SK_HEADER_BYTES + (int64_t)device_cus() * 256 * 256 * 4;
}

int vd_conv_igemm_mtiles(const vd_conv_desc* d) {
    if (!d) return 0;
    const int64_t M = (int64_t)d->N * d->Hg * d->Wg;
    if (d->flags & VD_CONV_PARITY4)           // one table row per (M tile, column tile): columns of 128
        return (int)(vd_cdiv(M, vd_igemm_par_bm(vd_igemm_par_tile(d->tile))) * vd_cdiv(d->Co, 128));
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
