// vd_stem.hip — the first convolution of Darknet-53 (3x3, 3 -> 32 channels, stride 1, pad 1) straight from the
// NCHW frame batch, forward and weight gradient.
//
// Reference: darknet stem `_conv2d(32, 3, 1, 1)` models/definitions/three_darknet.py:163-164 on the batch the
// transforms produce as (B,3,H,W) fp32 (models/definitions/yolo/transforms.py:239-245).
//
// Cin = 3 gives K = 27: as a tile of the generic implicit GEMM 90 % of the MFMA K lanes would be padding, and the
// 32-wide im2col that feeds it costs a 1.4 GB round trip per pass at batch 64 / 416x416.  Both directions are
// HBM-bound here (forward writes B*H*W*32 outputs, the weight gradient reads as many gradients), so:
//   forward   : one lane = one pixel x 32 channels on the fp32 VALU (864 FMAs against weights broadcast from LDS),
//               27 coalesced input loads per lane, output transposed through LDS so a wave
//               stores 1 KB contiguous; optional BN fold + LeakyReLU (inference), bf16 output (bf16 inference) or raw
//               output + per-block BatchNorm partial sums (training).
//   wgrad     : D[32 co][27 k] = sum_px dz[px][co] * x[px + tap(k)][c(k)] on v_mfma_f32_32x32x2_f32 with BOTH operands
//               straight from global memory - lane (co, pixel parity) reads dz coalesced, lane (k, pixel parity)
//               gathers its frame value - no LDS, no im2col; per-block partial tiles + a fixed-order reduction.
#include "vd_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int SC = 32;          // output channels
constexpr int SK = 27;          // 3 * 3 * 3 real taps x channels; packed weight rows are 32 wide (k >= 27 zero)
constexpr int SPX = 256;        // pixels per workgroup (one per lane)

// w: fwd-packed [32][32] with k = (ky*3 + kx)*3 + c  (vd_pack_weight_fwd of the OIHW stem weight; k >= 27 are zero)
// MF (bf16 outputs: bf16 inference and bf16-storage training, where every conv operand is bf16-rounded anyway): the 864 FMAs
// per pixel go to the matrix pipe instead - per 32 pixels two v_mfma_f32_32x32x16_bf16 (K = 27 padded to 32) on the frame
// values and weights rounded to bf16 in registers, fp32 accumulation; the gathers, the LDS tile, the statistics and the
// stores are the VALU form's.  (The VALU form was compute-bound at 0.43-0.47 ms for 11 M pixels; fp32 tensors keep it:
// it is exact fp32.)
template <bool OUT_BF16, bool MF = false>
__global__ __launch_bounds__(SPX) void k_stem_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                  void* __restrict__ out, int ldo, int N, int H, int W,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  float slope, int flags, float* __restrict__ stats_part) {
    __shared__ float tile[SPX][SC + 1];           // +1: the transposing reads walk a column
    if constexpr (MF) {
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int64_t HW = (int64_t)H * W;
        const int64_t P = (int64_t)N * HW;
        const int hh = lane >> 5;                 // operand k-half: k = 8*hh .. +7 (first MFMA), 16 + 8*hh .. (second)
        // B operand: lane (co = lane & 31, hh) holds w[co][k] for its 2 x 8 k values (k >= 27 are zero in the packed rows)
        bf16x8_t b0, b1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            b0[e] = (__bf16)w[(lane & 31) * 32 + 8 * hh + e];
            b1[e] = (__bf16)w[(lane & 31) * 32 + 16 + 8 * hh + e];
        }
#pragma unroll
        for (int gsel = 0; gsel < 2; ++gsel) {    // a wave owns 64 of the workgroup's 256 pixels: two 32-pixel MFMA row blocks
            const int prow = wave * 64 + gsel * 32 + (lane & 31);
            const int64_t pix = (int64_t)blockIdx.x * SPX + prow;
            bf16x8_t a0, a1;
            {
                const bool pin = pix < P;
                const int64_t pp = pin ? pix : 0;
                const int n = (int)(pp / HW);
                const int r = (int)(pp - (int64_t)n * HW);
                const int y = r / W, xx = r - y * W;
                const float* xn = x + (int64_t)n * 3 * HW;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = (e < 8 ? 8 * hh : 16 + 8 * hh) + (e & 7);          // this lane's e-th k
                    const int tap = k / 3, c = k - 3 * tap;
                    const int iy = y + tap / 3 - 1, ix = xx + tap % 3 - 1;
                    const bool ok = pin && k < SK && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    const float t = xn[ok ? (int64_t)c * HW + (int64_t)iy * W + ix : 0];
                    v[e] = ok ? t : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0[e] = (__bf16)v[e]; a1[e] = (__bf16)v[8 + e]; }
            }
            f32x16 acc;
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) acc[r2] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            // C/D map: acc[r] = C[row (r & 3) + 8 (r >> 2) + 4 hh][col lane & 31]; epilogue per column, then into the tile
            const int co = lane & 31;
            const float sc = (flags & VD_EPI_AFFINE) ? scale[co] : 1.f, sh = (flags & VD_EPI_AFFINE) ? shift[co] : 0.f;
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                float t = acc[r2];
                if (flags & VD_EPI_AFFINE) t = t * sc + sh;
                if (flags & VD_EPI_LEAKY) t = t > 0.f ? t : t * slope;
                tile[wave * 64 + gsel * 32 + (r2 & 3) + 8 * (r2 >> 2) + 4 * hh][co] = t;
            }
        }
        __syncthreads();
    } else {
    // weights transposed to [k][co] in LDS: for one k the 32 output-channel weights are 8 broadcast ds_read_b128
    // (every lane the same address).  Kept in SGPRs they spilled: 864 values against 100 scalar registers.
    __shared__ __attribute__((aligned(16))) float wT[SK][SC];
    const int tid = threadIdx.x;
    for (int i = tid; i < SK * SC; i += SPX) wT[i / SC][i % SC] = w[(i % SC) * 32 + i / SC];
    __syncthreads();
    const int64_t HW = (int64_t)H * W;
    const int64_t P = (int64_t)N * HW;
    const int64_t pix = (int64_t)blockIdx.x * SPX + tid;
    float acc[SC];
#pragma unroll
    for (int c = 0; c < SC; ++c) acc[c] = 0.f;
    if (pix < P) {
        const int n = (int)(pix / HW);
        const int r = (int)(pix - (int64_t)n * HW);
        const int y = r / W, xx = r - y * W;
        const float* xn = x + (int64_t)n * 3 * HW;
        // the 27 frame values first (branch-free: out-of-frame taps read pixel 0 and are zeroed), then a pure
        // LDS-read + FMA stream per k - with the loads inside that stream the compiler parked all 864 weights in
        // registers and scratch while waiting for them
        float xv[SK];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = xx + kx - 1;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int64_t o = ok ? (int64_t)iy * W + ix : 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float t = xn[(int64_t)c * HW + o];
                    xv[(ky * 3 + kx) * 3 + c] = ok ? t : 0.f;
                }
            }
#pragma unroll
        for (int k = 0; k < SK; ++k) {
#pragma unroll
            for (int c4 = 0; c4 < SC; c4 += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&wT[k][c4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c4 + e] = fmaf(xv[k], wv[e], acc[c4 + e]);
            }
            __builtin_amdgcn_sched_barrier(0);        // keep each k's weight reads next to its FMAs
        }
    }
#pragma unroll
    for (int c = 0; c < SC; ++c) {
        float v = acc[c];
        if (flags & VD_EPI_AFFINE) v = v * scale[c] + shift[c];
        if (flags & VD_EPI_LEAKY) v = v > 0.f ? v : v * slope;
        tile[tid][c] = v;
    }
    __syncthreads();
    }   // !MF
    const int tid = threadIdx.x;
    const int64_t P = (int64_t)N * H * W;
    // fused BatchNorm statistics of the raw outputs (training; flags == 0, so the tile holds them): column sums out of
    // the LDS tile - 8 row segments x 32 columns per pass, then 64 lanes fold the 8 partials (rows past the last pixel
    // hold zeros).  One table row per workgroup.  (Per-column wave shuffles cost 384 cross-lane ops per wave.)
    if (stats_part) {
        __shared__ float red[8][2 * SC];
        const int c = tid & 31, seg = tid >> 5;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int r = seg * 32; r < seg * 32 + 32; ++r) {
            const float v = tile[r][c];
            s1 += v;
            s2 += v * v;
        }
        red[seg][c] = s1;
        red[seg][SC + c] = s2;
        __syncthreads();
        if (tid < 2 * SC) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q][tid];
            stats_part[(int64_t)blockIdx.x * 2 * SC + tid] = t;
        }
    }
    // 256 pixels x 32 channels = 2048 float4: lane i writes float4 (pixel i/8 + 32*j, channels 4*(i%8)..)
    const int64_t p0 = (int64_t)blockIdx.x * SPX;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pl = (tid >> 3) + 32 * j, c4 = (tid & 7) * 4;
        if (p0 + pl < P) {
            const f32x4 v = {tile[pl][c4], tile[pl][c4 + 1], tile[pl][c4 + 2], tile[pl][c4 + 3]};
            if (OUT_BF16) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + (p0 + pl) * ldo + c4) = o;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (p0 + pl) * ldo + c4) = v;
            }
        }
    }
}

// Weight gradient.  One wave owns a run of pixels; per MFMA (K = 2 pixels): A[co = lane&31][px = lane>>5] = dz, coalesced
// 128-B rows; B[px][k = lane&31] = x[n][c(k)][y + dy(k)][x + dx(k)] (zero outside the frame and for k >= 27).
// part: [gridDim.x * 4 waves][32 co][32 k] partial tiles, summed in wave order by k_stem_wgrad_reduce.
template <typename T>      // storage type of dz (float, or __bf16 in bf16-storage training)
__global__ __launch_bounds__(256) void k_stem_wgrad(const float* __restrict__ x, const T* __restrict__ dz, int ldd,
                                                    float* __restrict__ part, int N, int H, int W, int64_t px_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t HW = (int64_t)H * W, P = (int64_t)N * HW;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p_begin = gw * px_per_wave;
    int64_t p_end = p_begin + px_per_wave;
    if (p_end > P) p_end = P;
    const int k = lane & 31, par = lane >> 5;
    const bool k_ok = k < SK;
    const int c = k_ok ? k % 3 : 0, tap = k_ok ? k / 3 : 0;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // pixel cursor of this lane (p_begin + par), advanced by 2 per step
    int64_t p = p_begin + par;
    int n = 0, y = 0, xx = 0;
    if (p < P) {
        n = (int)(p / HW);
        const int r = (int)(p - (int64_t)n * HW);
        y = r / W;
        xx = r - y * W;
    }
    // batches of UNR steps: all 2*UNR loads of a batch are issued before its MFMAs (one step at a time left a single
    // pair of loads in flight per wave: latency bound at 1.4 TB/s)
    constexpr int UNR = 8;
    for (; p - par < p_end; p += 2 * UNR) {           // wave-uniform trip count (p - par is the batch's first pixel)
        float a[UNR], b[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t pu = p + 2 * u;
            const bool in_run = pu < p_end;
            a[u] = in_run ? (float)dz[pu * ldd + k] : 0.f;                   // k doubles as co for the A operand
            const int iy = y + dy, ix = xx + dx;
            const bool ok = in_run && k_ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            b[u] = ok ? x[((int64_t)n * 3 + c) * HW + (int64_t)iy * W + ix] : 0.f;
            xx += 2;
            while (xx >= W) { xx -= W; ++y; }
            if (y >= H) { y -= H; ++n; }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    float* dst = part + gw * (SC * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[co * 32 + (lane & 31)] = acc[r];
    }
}

// Weight gradient, second form: one workgroup = FOUR image rows (a wave each), the frame window they read - three channels
// x six rows (y0 - 1 .. y0 + 4) x (W + 2) columns, zero outside the frame - staged ONCE in LDS.  The first form's B operand
// is a gather of 27 different (channel, row, column) addresses per half-wave and MFMA step, i.e. 27 cache lines through the
// CU's address path for 256 bytes of dz: the kernel ran at 2.5 TB/s of its 1.4 GB.  Here the B operand is one ds_read_b32 per
// step at a per-lane base plus an immediate (row pitch = 4 mod 32 floats: the 27 lanes of a half-wave fall on 27 different
// banks), the dz rows stream as before, and the four waves' tiles are summed in LDS (wave order) into ONE partial tile per
// workgroup.  part: [N * ceil(H / 4)][32 co][32 k].
constexpr int SROWS = 4;
template <typename T>
__global__ __launch_bounds__(256) void k_stem_wgrad_rows(const float* __restrict__ x, const T* __restrict__ dz, int ldd,
                                                         float* __restrict__ part, int N, int H, int W, int P) {
    extern __shared__ float win[];                    // [3][SROWS + 2][P]; afterwards [4][32 * 32] for the wave tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int groups = (H + SROWS - 1) / SROWS;
    const int n = blockIdx.x / groups, y0 = (blockIdx.x % groups) * SROWS;
    const int64_t HW = (int64_t)H * W;
    const float* xn = x + (int64_t)n * 3 * HW;
    for (int i = tid; i < 3 * (SROWS + 2) * P; i += 256) {
        const int col = i % P - 1, rr = i / P;        // rr = c * (SROWS + 2) + r
        const int c = rr / (SROWS + 2), iy = y0 - 1 + rr % (SROWS + 2);
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)col < (unsigned)W;
        win[i] = ok ? xn[(int64_t)c * HW + (int64_t)iy * W + col] : 0.f;
    }
    __syncthreads();
    const int k = lane & 31, par = lane >> 5;
    const bool k_ok = k < SK;
    const int c = k_ok ? k % 3 : 0, tap = k_ok ? k / 3 : 0;
    const int y = y0 + wave;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (y < H) {                                       // wave-uniform
        // window index of (channel c, row y + dy, column x + dx) for x = par: + 2 per step
        const float* b0 = win + (c * (SROWS + 2) + wave + tap / 3) * P + (tap % 3) + par;
        const T* a0 = dz + ((int64_t)n * HW + (int64_t)y * W + par) * ldd + k;          // k doubles as co for the A operand
        constexpr int UNR = 8;
        for (int x0 = 0; x0 < W; x0 += 2 * UNR) {
            float a[UNR], b[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool in = x0 + 2 * u + par < W;
                a[u] = in ? (float)a0[(int64_t)(x0 + 2 * u) * ldd] : 0.f;
                b[u] = (in && k_ok) ? b0[x0 + 2 * u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        }
    }
    __syncthreads();                                   // the window is dead: its memory takes the four wave tiles
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        win[wave * (SC * 32) + co * 32 + (lane & 31)] = acc[r];
    }
    __syncthreads();
    float* dst = part + (int64_t)blockIdx.x * (SC * 32);
    for (int i = tid; i < SC * 32; i += 256)
        dst[i] = ((win[i] + win[SC * 32 + i]) + win[2 * SC * 32 + i]) + win[3 * SC * 32 + i];
}

// dst[chunk][i] = sum of part[q][i] over the chunk's slabs, in slab order; gridDim.y chunks (one for the final pass)
__global__ void k_stem_wgrad_reduce(const float* __restrict__ part, int nparts, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // 32 x 32 outputs
    if (i >= SC * 32) return;
    const int per = (nparts + gridDim.y - 1) / gridDim.y;
    int q = blockIdx.y * per;
    int q1 = q + per;
    if (q1 > nparts) q1 = nparts;
    float s = 0.f;
    for (; q + 8 <= q1; q += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(q + u) * (SC * 32) + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; q < q1; ++q) s += part[(int64_t)q * (SC * 32) + i];
    dst[(int64_t)blockIdx.y * (SC * 32) + i] = s;
}

constexpr int STEM_RED_CHUNKS = 64;          // first reduction level: 64 chunks of slabs, then one pass over the 64 rows
constexpr int64_t STEM_WG_WAVES = 8192;      // partial tiles: 8192 x 4 KB = 32 MB of workspace at most

}  // namespace

extern "C" {

int vd_stem_conv_blocks(int N, int H, int W) { return (int)vd_cdiv((int64_t)N * H * W, SPX); }

int vd_stem_conv(const float* x_nchw, const float* wp, void* out, int ldo, int N, int H, int W, const float* scale,
                 const float* shift, float slope, int flags, int out_bf16, float* stats_part, void* stream) {
    VD_REQUIRE(x_nchw && wp && out && N > 0 && H > 0 && W > 0 && ldo >= SC && ldo % 4 == 0, "vd_stem_conv: bad args");
    VD_REQUIRE(!(flags & VD_EPI_AFFINE) || (scale && shift), "vd_stem_conv: affine epilogue needs scale and shift");
    VD_REQUIRE(!stats_part || flags == 0, "vd_stem_conv: fused statistics need the raw output (they come from the fp32 values, bf16 output or not)");
    VD_REQUIRE((int64_t)N * H * W < (1ll << 31), "vd_stem_conv: pixel count overflows int32");
    const int nb = vd_stem_conv_blocks(N, H, W);
    static const int use_mf = getenv("VD_STEM_MFMA") ? atoi(getenv("VD_STEM_MFMA")) : 1;      // developer A/B switch
    if (out_bf16 && use_mf)
        hipLaunchKernelGGL((k_stem_fwd<true, true>), dim3(nb), dim3(SPX), 0, (hipStream_t)stream, x_nchw, wp, out, ldo, N, H, W,
                           scale, shift, slope, flags, stats_part);
    else if (out_bf16)
        hipLaunchKernelGGL(k_stem_fwd<true>, dim3(nb), dim3(SPX), 0, (hipStream_t)stream, x_nchw, wp, out, ldo, N, H, W,
                           scale, shift, slope, flags, stats_part);
    else
        hipLaunchKernelGGL(k_stem_fwd<false>, dim3(nb), dim3(SPX), 0, (hipStream_t)stream, x_nchw, wp, out, ldo, N, H, W,
                           scale, shift, slope, flags, stats_part);
    VD_CHECK_LAUNCH("vd_stem_conv");
    return VD_OK;
}

// row pitch of the LDS window: >= W + 2 and = 4 mod 32 floats (bank spread of the 27 operand lanes)
static int stem_rows_pitch(int W) {
    int p = W + 2;
    while (p % 32 != 4) ++p;
    return p;
}
static int64_t stem_rows_lds(int W) {
    const int64_t win = (int64_t)3 * (SROWS + 2) * stem_rows_pitch(W) * 4, tiles = (int64_t)4 * SC * 32 * 4;
    return win > tiles ? win : tiles;
}
// the row form (k_stem_wgrad_rows) where its window fits 64 KB of LDS (W <= 900); VD_STEM_WGRAD_ROWS=0: developer A/B
static bool stem_rows_ok(int W) {
    static const int on = getenv("VD_STEM_WGRAD_ROWS") ? atoi(getenv("VD_STEM_WGRAD_ROWS")) : 1;
    return on && stem_rows_lds(W) <= 64 * 1024;
}
static int64_t stem_run_waves(int N, int H, int W) {
    const int64_t P = (int64_t)N * H * W;
    int64_t waves = vd_cdiv(P, 256);                 // >= 128 MFMA steps per wave
    if (waves > STEM_WG_WAVES) waves = STEM_WG_WAVES;
    return vd_cdiv(waves, 4) * 4;
}

int64_t vd_stem_wgrad_ws_bytes(int N, int H, int W) {
    const int64_t tiles_rows = (int64_t)N * vd_cdiv(H, SROWS), tiles_run = stem_run_waves(N, H, W);
    const int64_t tiles = tiles_rows > tiles_run ? tiles_rows : tiles_run;       // either form's partial tiles
    return (tiles + STEM_RED_CHUNKS) * SC * 32 * (int64_t)sizeof(float);
}

static int stem_wgrad_any(const float* x_nchw, const void* dz, int dz_bf16, int ldd, float* dwp, int N, int H, int W, void* ws,
                         int64_t ws_bytes, void* stream) {
    VD_REQUIRE(x_nchw && dz && dwp && N > 0 && H > 0 && W > 0 && ldd >= SC, "vd_stem_wgrad: bad args");
    const int64_t need = vd_stem_wgrad_ws_bytes(N, H, W);
    if (!ws || ws_bytes < need) {
        vd_set_error("vd_stem_wgrad: workspace %lld < %lld", (long long)ws_bytes, (long long)need);
        return VD_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t tiles;
    if (stem_rows_ok(W)) {
        tiles = (int64_t)N * vd_cdiv(H, SROWS);
        const int P = stem_rows_pitch(W), lds = (int)stem_rows_lds(W);
        if (dz_bf16)
            hipLaunchKernelGGL(k_stem_wgrad_rows<__bf16>, dim3((unsigned)tiles), dim3(256), lds, s, x_nchw, (const __bf16*)dz, ldd,
                               (float*)ws, N, H, W, P);
        else
            hipLaunchKernelGGL(k_stem_wgrad_rows<float>, dim3((unsigned)tiles), dim3(256), lds, s, x_nchw, (const float*)dz, ldd,
                               (float*)ws, N, H, W, P);
    } else {
        const int64_t P = (int64_t)N * H * W;
        tiles = stem_run_waves(N, H, W);
        int64_t ppw = vd_cdiv(P, tiles);
        ppw = vd_cdiv(ppw, 2) * 2;                       // even: a step is two pixels
        if (dz_bf16)
            hipLaunchKernelGGL(k_stem_wgrad<__bf16>, dim3((unsigned)(tiles / 4)), dim3(256), 0, s, x_nchw, (const __bf16*)dz, ldd, (float*)ws,
                               N, H, W, ppw);
        else
            hipLaunchKernelGGL(k_stem_wgrad<float>, dim3((unsigned)(tiles / 4)), dim3(256), 0, s, x_nchw, (const float*)dz, ldd, (float*)ws,
                               N, H, W, ppw);
    }
    VD_CHECK_LAUNCH("vd_stem_wgrad");
    float* lvl1 = (float*)ws + tiles * (SC * 32);
    hipLaunchKernelGGL(k_stem_wgrad_reduce, dim3(4, STEM_RED_CHUNKS), dim3(256), 0, s, (const float*)ws, (int)tiles, lvl1);
    VD_CHECK_LAUNCH("vd_stem_wgrad/reduce1");
    hipLaunchKernelGGL(k_stem_wgrad_reduce, dim3(4, 1), dim3(256), 0, s, (const float*)lvl1, STEM_RED_CHUNKS, dwp);
    VD_CHECK_LAUNCH("vd_stem_wgrad/reduce2");
    return VD_OK;
}

int vd_stem_wgrad(const float* x_nchw, const float* dz, int ldd, float* dwp, int N, int H, int W, void* ws, int64_t ws_bytes,
                  void* stream) {
    return stem_wgrad_any(x_nchw, dz, 0, ldd, dwp, N, H, W, ws, ws_bytes, stream);
}

/* dz stored as bf16 (bf16-storage training); frames and the weight gradient stay fp32 */
int vd_stem_wgrad_bf16(const float* x_nchw, const void* dz, int ldd, float* dwp, int N, int H, int W, void* ws, int64_t ws_bytes,
                       void* stream) {
    return stem_wgrad_any(x_nchw, dz, 1, ldd, dwp, N, H, W, ws, ws_bytes, stream);
}

}  // extern "C"
