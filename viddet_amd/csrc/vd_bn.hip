// vd_bn.hip — BatchNorm (training statistics, apply, backward) + LeakyReLU on NHWC [M, C] fp32.
//
// Replaces  norm_layer(epsilon=1e-5, momentum=0.9) + nn.LeakyReLU(0.1)   models/definitions/layers.py:68-69
// (mxnet BatchNorm / gluon.contrib SyncBatchNorm, train_yolov3.py:347-354) and their backward.
//
// All kernels are HBM-bound streaming passes: float4 (16 B/lane) accesses, channel-contiguous, so a
// wave covers 1 KiB of one pixel row run.  Per-channel reductions are two-level and deterministic:
// fp32 per-thread partials over short runs -> LDS -> per-block rows in a workspace -> fp64 finalize.
// The fp64 [2C] sums are the SyncBN exchange unit (one all-reduce per layer per direction).
#include "vd_common.h"
#include <stdlib.h>

namespace {

constexpr int RED_THREADS = 256;

// MODE 0: a = x, b = x*x          (forward statistics)
// MODE 1: a = g, b = g*xhat       (backward reductions), g = dy * leaky'(x*scale+shift)
// T: storage type of x / dy (float, or __bf16 for bf16-storage training: fp32 arithmetic on the widened values)
template <int MODE, typename T = float>
__global__ __launch_bounds__(RED_THREADS) void k_bn_partial(const T* __restrict__ x,
                                                            const T* __restrict__ dy,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int64_t M,
                                                            int C, float slope, float* __restrict__ part) {
    __shared__ f32x4 sa[RED_THREADS], sb[RED_THREADS];
    const int cvec = C >> 2;
    const int cblk = cvec < RED_THREADS ? cvec : RED_THREADS;   // float4 columns handled per sweep
    const int rl = RED_THREADS / cblk;                           // row lanes
    const int tcol = threadIdx.x % cblk, trow = threadIdx.x / cblk;
    const int64_t rows_per_blk = vd_cdiv(M, gridDim.x);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > M) r1 = M;
    for (int cb = 0; cb < cvec; cb += cblk) {
        const int col = cb + tcol;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (trow < rl && col < cvec) {
            f32x4 sc, sh, mu, is;
            if (MODE == 1) {
                sc = reinterpret_cast<const f32x4*>(scale)[col];
                sh = reinterpret_cast<const f32x4*>(shift)[col];
                mu = reinterpret_cast<const f32x4*>(mean)[col];
                is = reinterpret_cast<const f32x4*>(invstd)[col];
            }
            for (int64_t r = r0 + trow; r < r1; r += rl) {
                const f32x4 v = vd_ld4(x, r * cvec + col);
                if (MODE == 0) {
                    a += v;
                    b += v * v;
                } else {
                    const f32x4 d = vd_ld4(dy, r * cvec + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = v[e] * sc[e] + sh[e];
                        const float g = u > 0.f ? d[e] : d[e] * slope;
                        a[e] += g;
                        b[e] += g * (v[e] - mu[e]) * is[e];
                    }
                }
            }
        }
        sa[threadIdx.x] = a;
        sb[threadIdx.x] = b;
        __syncthreads();
        if (trow == 0 && col < cvec) {
            for (int j = 1; j < rl; ++j) {
                a += sa[j * cblk + tcol];
                b += sb[j * cblk + tcol];
            }
            float* dst = part + (int64_t)blockIdx.x * 2 * C;
            reinterpret_cast<f32x4*>(dst)[col] = a;
            reinterpret_cast<f32x4*>(dst + C)[col] = b;
        }
        __syncthreads();
    }
}

// bf16 tensors, 8 channels (16 bytes) per thread and load: the 4-channel form above moved 8 bytes per lane on bf16 data and
// ran at 2.1 TB/s (the stand-alone backward reductions of bf16-storage training: 4.9 ms of a 35 ms step)
template <int MODE>
__global__ __launch_bounds__(RED_THREADS) void k_bn_partial_bf16x8(const __bf16* __restrict__ x, const __bf16* __restrict__ dy,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   int64_t M, int C, float slope, float* __restrict__ part) {
    __shared__ f32x8 sa[RED_THREADS], sb[RED_THREADS];
    const int cvec = C >> 3;
    const int cblk = cvec < RED_THREADS ? cvec : RED_THREADS;
    const int rl = RED_THREADS / cblk;
    const int tcol = threadIdx.x % cblk, trow = threadIdx.x / cblk;
    const int64_t rows_per_blk = vd_cdiv(M, gridDim.x);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > M) r1 = M;
    for (int cb = 0; cb < cvec; cb += cblk) {
        const int col = cb + tcol;
        f32x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = b[e] = 0.f;
        if (trow < rl && col < cvec) {
            f32x8 sc, sh, mu, is;
            if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sc[e] = scale[col * 8 + e]; sh[e] = shift[col * 8 + e];
                    mu[e] = mean[col * 8 + e]; is[e] = invstd[col * 8 + e];
                }
            }
            // two rows in flight per thread: the loads of row r + rl are issued before row r is consumed
            for (int64_t r = r0 + trow; r < r1; r += 2 * rl) {
                const bool two = r + rl < r1;
                const f32x8 v0 = vd_ld8(x, r * cvec + col);
                const f32x8 v1 = two ? vd_ld8(x, (r + rl) * cvec + col) : f32x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (MODE == 0) {
                    a += v0 + v1;
                    b += v0 * v0 + v1 * v1;
                } else {
                    const f32x8 d0 = vd_ld8(dy, r * cvec + col);
                    const f32x8 d1 = two ? vd_ld8(dy, (r + rl) * cvec + col) : f32x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float u0 = v0[e] * sc[e] + sh[e], u1 = v1[e] * sc[e] + sh[e];
                        const float g0 = u0 > 0.f ? d0[e] : d0[e] * slope, g1 = u1 > 0.f ? d1[e] : d1[e] * slope;
                        a[e] += g0 + g1;
                        b[e] += g0 * (v0[e] - mu[e]) * is[e] + (two ? g1 * (v1[e] - mu[e]) * is[e] : 0.f);
                    }
                }
            }
        }
        sa[threadIdx.x] = a;
        sb[threadIdx.x] = b;
        __syncthreads();
        if (trow == 0 && col < cvec) {
            for (int j = 1; j < rl; ++j) {
                a += sa[j * cblk + tcol];
                b += sb[j * cblk + tcol];
            }
            float* dst = part + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dst[col * 8 + e] = a[e];
                dst[C + col * 8 + e] = b[e];
            }
        }
        __syncthreads();
    }
}

// 64 columns x 16 row-slices per block: each thread sums every 16th partial row (coalesced across
// the 64 columns), then the 16 slices are combined in a fixed order -> deterministic, and 16x the
// memory-level parallelism of one thread per column (that form was latency-bound: 170 us per call).
__global__ __launch_bounds__(1024) void k_bn_sum_partials(const float* __restrict__ part, int nblk, int C2,
                                                          double* __restrict__ sums) {
    __shared__ double sh[16][64];
    const int cx = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cx;
    double s = 0.0;
    if (i < C2) {
        int b = sl;
        for (; b + 48 < nblk; b += 64) {          // four rows in flight, added in row order
            const float p0 = part[(int64_t)b * C2 + i], p1 = part[(int64_t)(b + 16) * C2 + i];
            const float p2 = part[(int64_t)(b + 32) * C2 + i], p3 = part[(int64_t)(b + 48) * C2 + i];
            s += (double)p0; s += (double)p1; s += (double)p2; s += (double)p3;
        }
        for (; b < nblk; b += 16) s += (double)part[(int64_t)b * C2 + i];
    }
    sh[sl][cx] = s;
    __syncthreads();
    if (sl == 0 && i < C2) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += sh[j][cx];
        sums[i] = t;
    }
}

// tables taller than SUMP_DIRECT rows go through two levels with one chunk per ~SUMP_DIRECT rows (at most SUMP_CHUNKS):
// a direct pass over 2048 rows of a 64-column table is 2 workgroups and 150 us; 64 chunks for a 300-row table is 26 us
// of mostly idle launches
constexpr int SUMP_DIRECT = 256, SUMP_CHUNKS = 64;
inline int sump_chunks(int nblk) {
    int c = (nblk + SUMP_DIRECT - 1) / SUMP_DIRECT;
    return c > SUMP_CHUNKS ? SUMP_CHUNKS : c;
}

// level 1 of the tall-table reduction: block (cb, r) sums rows [r*chunk, (r+1)*chunk) of 64 columns
__global__ __launch_bounds__(1024) void k_bn_sum_partials_l1(const float* __restrict__ part, int nblk, int C2,
                                                             double* __restrict__ out) {
    __shared__ double sh[16][64];
    const int cx = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cx;
    const int chunk = (nblk + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * chunk;
    int r1 = r0 + chunk;
    if (r1 > nblk) r1 = nblk;
    double s = 0.0;
    if (i < C2) {
        int b = r0 + sl;
        for (; b + 48 < r1; b += 64) {            // four rows in flight, added in row order
            const float p0 = part[(int64_t)b * C2 + i], p1 = part[(int64_t)(b + 16) * C2 + i];
            const float p2 = part[(int64_t)(b + 32) * C2 + i], p3 = part[(int64_t)(b + 48) * C2 + i];
            s += (double)p0; s += (double)p1; s += (double)p2; s += (double)p3;
        }
        for (; b < r1; b += 16) s += (double)part[(int64_t)b * C2 + i];
    }
    sh[sl][cx] = s;
    __syncthreads();
    if (sl == 0 && i < C2) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += sh[j][cx];
        out[(int64_t)blockIdx.y * C2 + i] = t;
    }
}

__global__ void k_bn_sum_partials_l2(const double* __restrict__ in, int rows, int C2, double* __restrict__ sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C2) return;
    double s = 0.0;
    int r = 0;
    for (; r + 8 <= rows; r += 8) {            // eight rows in flight, added in row order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = in[(int64_t)(r + u) * C2 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < rows; ++r) s += in[(int64_t)r * C2 + i];
    sums[i] = s;
}

__global__ void k_bn_finalize(const double* __restrict__ sums, double count, int C,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                              float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                              float* __restrict__ scale, float* __restrict__ shift,
                              float* __restrict__ smean, float* __restrict__ sinv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mean = sums[c] / count;
    double var = sums[C + c] / count - mean * mean;   // biased, as the reference normalises with
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma[c], b = beta[c];
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (smean) smean[c] = (float)mean;
    if (sinv) sinv[c] = invstd;
    if (rmean) rmean[c] = rmean[c] * momentum + (float)mean * (1.f - momentum);
    if (rvar) rvar[c] = rvar[c] * momentum + (float)var * (1.f - momentum);
}

// Fused tail of the BatchNorm statistics for tables of up to SUMP_FUSED rows: a workgroup owns 32 channels, sums BOTH of
// their columns (c and C + c) of the partial table in fp64 - 16 row slices, fixed order, as k_bn_sum_partials - and
// finishes in place.  MODE 0 (forward): sums -> scale/shift/mean/invstd + running statistics (k_bn_finalize).
// MODE 1 (backward): sums2 -> dgamma/dbeta (k_bn_param_grads).  The fp64 sums are written too (SyncBN-free path only,
// but bn_bwd_apply and tests read them).  Replaces 2-4 tiny launches on the critical path of every conv by one.
template <int MODE>
__global__ __launch_bounds__(1024) void k_bn_sum_fused(const float* __restrict__ part, int nblk, int C, double* __restrict__ sums,
                                                       double count, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float momentum,
                                                       float* __restrict__ rmean, float* __restrict__ rvar,
                                                       float* __restrict__ scale, float* __restrict__ shift,
                                                       float* __restrict__ smean, float* __restrict__ sinv,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sh[16][64];
    const int cx = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int ch = blockIdx.x * 32 + (cx & 31);               // channel; cx >> 5 selects the sum / sum-of-squares column
    const int col = (cx >> 5) * C + ch;
    double s = 0.0;
    if (ch < C) {
        const int C2 = 2 * C;
        int b = sl;
        for (; b + 48 < nblk; b += 64) {
            const float p0 = part[(int64_t)b * C2 + col], p1 = part[(int64_t)(b + 16) * C2 + col];
            const float p2 = part[(int64_t)(b + 32) * C2 + col], p3 = part[(int64_t)(b + 48) * C2 + col];
            s += (double)p0; s += (double)p1; s += (double)p2; s += (double)p3;
        }
        for (; b < nblk; b += 16) s += (double)part[(int64_t)b * C2 + col];
    }
    sh[sl][cx] = s;
    __syncthreads();
    if (sl == 0 && ch < C) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += sh[j][cx];
        sums[col] = t;
        sh[0][cx] = t;
    }
    __syncthreads();
    if (threadIdx.x < 32 && ch < C) {
        const double a = sh[0][threadIdx.x], bq = sh[0][32 + threadIdx.x];     // column c, column C + c
        if (MODE == 0) {
            const double mean = a / count;
            double var = bq / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = gamma[ch] * invstd;
            scale[ch] = sc;
            shift[ch] = beta[ch] - (float)mean * sc;
            if (smean) smean[ch] = (float)mean;
            if (sinv) sinv[ch] = invstd;
            if (rmean) rmean[ch] = rmean[ch] * momentum + (float)mean * (1.f - momentum);
            if (rvar) rvar[ch] = rvar[ch] * momentum + (float)var * (1.f - momentum);
        } else {
            dbeta[ch] = (float)a;
            dgamma[ch] = (float)bq;
        }
    }
}

constexpr int SUMP_FUSED = 1024;          // rows up to which the fused single-launch tail is used

__global__ void k_bn_fold_eval(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                               int C, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

// Streaming kernels below: the launch makes gridDim.x * 256 a multiple of cvec (fixed_col_blocks), so a thread's float4
// column col = i % cvec never changes along its grid-stride loop - the per-channel constants are loaded (and the fp64
// sums converted) once per thread instead of once per element, and the 64-bit modulo leaves the loop.
template <typename T = float>
__global__ void k_bn_apply_leaky(const T* __restrict__ x, const float* __restrict__ scale,
                                 const float* __restrict__ shift, const T* __restrict__ res,
                                 T* __restrict__ y, int64_t n4, int cvec, float slope, float* __restrict__ amax) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int col = (int)(i0 % cvec);
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[col];
    const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[col];
    float amx = 0.f;
    for (int64_t i = i0; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = vd_ld4(x, i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = v[e] * sc[e] + sh[e];
            o[e] = u > 0.f ? u : u * slope;
        }
        if (res) o += vd_ld4(res, i);
        vd_st4(y, i, o);
        amx = fmaxf(amx, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
    if (amax) vd_amax_publish(amax, amx);       // max-abs of the tensor for its consumers' fp16 operand scale
}

template <typename T = float>
__global__ void k_bn_bwd_apply(const T* __restrict__ x, const T* __restrict__ dy,
                               const float* __restrict__ scale, const float* __restrict__ shift,
                               const float* __restrict__ mean, const float* __restrict__ invstd,
                               const double* __restrict__ sums2, double count, int64_t n4, int C,
                               float slope, T* __restrict__ dx, float* __restrict__ amax) {
    const int cvec = C >> 2;
    const float inv_count = (float)(1.0 / count);
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int col = (int)(i0 % cvec);                      // fixed along the loop (see k_bn_apply_leaky)
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[col];
    const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[col];
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[col];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[col];
    f32x4 mg, mgx;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        mg[e] = (float)sums2[col * 4 + e] * inv_count;
        mgx[e] = (float)sums2[C + col * 4 + e] * inv_count;
    }
    float amx = 0.f;
    for (int64_t i = i0; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = vd_ld4(x, i);
        const f32x4 d = vd_ld4(dy, i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = v[e] * sc[e] + sh[e];
            const float g = u > 0.f ? d[e] : d[e] * slope;
            const float xh = (v[e] - mu[e]) * is[e];
            o[e] = sc[e] * (g - mg[e] - xh * mgx[e]);
        }
        vd_st4(dx, i, o);
        amx = fmaxf(amx, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
    if (amax) vd_amax_publish(amax, amx);
}

__global__ void k_bn_param_grads(const double* __restrict__ sums2, int C, float* __restrict__ dgamma,
                                 float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)sums2[c];
    dgamma[c] = (float)sums2[C + c];
}

// developer A/B switch: VD_BN_BF16X8=0 keeps the 4-channel reductions on bf16 tensors
bool bf16x8_enabled() {
    static const int v = getenv("VD_BN_BF16X8") ? atoi(getenv("VD_BN_BF16X8")) : 1;
    return v != 0;
}

int red_blocks(int64_t M) {
    int64_t nb = vd_cdiv(M, 64);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}

int stream_blocks(int64_t n) {
    int64_t nb = vd_cdiv(n, 256);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// block count with (blocks * 256) % cvec == 0, so that every thread of a grid-stride loop keeps its channel column
int fixed_col_blocks(int64_t n4, int cvec) {
    int a = 256, b = cvec;
    while (b) { const int t = a % b; a = b; b = t; }          // a = gcd(256, cvec)
    const int64_t step = cvec / a;
    int64_t nb = stream_blocks(n4);
    nb = vd_cdiv(nb, step) * step;
    return (int)nb;
}

// ---- operand-range guard of the fp16-split arithmetic (VD_MATH_F16X2, include/viddet_hip.h vd_range_guard) ------------
// One workgroup per BatchNorm layer: the spread of the per-channel magnitudes of what the layer's cell writes,
//   y channel c ~ |gamma_c| + |beta_c|   (y = leaky(gamma xhat + beta), xhat normalised over the batch)
//   dz channel c ~ |scale_c|             (dz = scale_c (g - mean g - xhat mean g xhat), scale = gamma * invstd)
// as ratio = min_c / max_c.  The fp16 split scales a TENSOR by one power of two: a channel 2^17 below the largest one is
// staged with fewer than 22 significant bits, and an output that reads mostly that channel inherits the loss.
__global__ void k_range_guard(const vd_guard_item* __restrict__ items, float thresh, float* __restrict__ ratios,
                              int32_t* __restrict__ flags) {
    const vd_guard_item it = items[blockIdx.x];
    float lo[2] = {3.0e38f, 3.0e38f}, hi[2] = {0.f, 0.f};
    for (int c = threadIdx.x; c < it.C; c += blockDim.x) {
        const float u = fabsf(it.gamma[c]) + fabsf(it.beta[c]);
        const float v = it.scale ? fabsf(it.scale[c]) : 1.f;
        lo[0] = fminf(lo[0], u); hi[0] = fmaxf(hi[0], u);
        lo[1] = fminf(lo[1], v); hi[1] = fmaxf(hi[1], v);
    }
    __shared__ float red[4][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = lo[0]; red[wave][1] = hi[0]; red[wave][2] = lo[1]; red[wave][3] = hi[1]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        float a = red[0][0], b = red[0][1], c2 = red[0][2], d = red[0][3];
        for (int w = 1; w < nw; ++w) { a = fminf(a, red[w][0]); b = fmaxf(b, red[w][1]); c2 = fminf(c2, red[w][2]); d = fmaxf(d, red[w][3]); }
        const float r0 = b > 0.f ? a / b : 1.f, r1 = d > 0.f ? c2 / d : 1.f;
        ratios[2 * blockIdx.x] = r0;
        ratios[2 * blockIdx.x + 1] = r1;
        flags[2 * blockIdx.x] = r0 < thresh ? 1 : 0;
        flags[2 * blockIdx.x + 1] = r1 < thresh ? 1 : 0;
    }
}

}  // namespace

extern "C" {

int64_t vd_bn_stats_ws_bytes(int64_t M, int C) { return (int64_t)red_blocks(M) * 2 * C * (int64_t)sizeof(float); }

int vd_bn_stats(const float* x, int64_t M, int C, double* sums, void* ws, int64_t ws_bytes, void* stream) {
    VD_REQUIRE(x && sums && ws && M > 0 && C > 0 && C % 4 == 0, "vd_bn_stats: bad args (C=%d)", C);
    const int nb = red_blocks(M);
    if (ws_bytes < (int64_t)nb * 2 * C * (int64_t)sizeof(float)) {
        vd_set_error("vd_bn_stats: workspace too small");
        return VD_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((k_bn_partial<0, float>), dim3(nb), dim3(RED_THREADS), 0, s, x, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, C, 0.f, (float*)ws);
    VD_CHECK_LAUNCH("vd_bn_stats");
    hipLaunchKernelGGL(k_bn_sum_partials, dim3((unsigned)vd_cdiv(2 * C, 64)), dim3(1024), 0, s, (const float*)ws, nb,
                       2 * C, sums);
    VD_CHECK_LAUNCH("vd_bn_stats/sum");
    return VD_OK;
}

int64_t vd_bn_sum_partials_ws_bytes(int nblk, int C) {
    return nblk > SUMP_DIRECT ? (int64_t)SUMP_CHUNKS * 2 * C * (int64_t)sizeof(double) : 0;
}

int vd_bn_sum_partials(const float* part, int nblk, int C, double* sums, void* ws, int64_t ws_bytes, void* stream) {
    VD_REQUIRE(part && sums && nblk > 0 && C > 0, "vd_bn_sum_partials: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (nblk <= SUMP_DIRECT) {
        hipLaunchKernelGGL(k_bn_sum_partials, dim3((unsigned)vd_cdiv(2 * C, 64)), dim3(1024), 0, s, part, nblk, 2 * C, sums);
        VD_CHECK_LAUNCH("vd_bn_sum_partials");
        return VD_OK;
    }
    // tall tables (the fused conv statistics of the 416x416 layers have ~10^5 rows): two fixed-order levels
    if (!ws || ws_bytes < vd_bn_sum_partials_ws_bytes(nblk, C)) {
        vd_set_error("vd_bn_sum_partials: workspace too small");
        return VD_EWORKSPACE;
    }
    const int chunks = sump_chunks(nblk);
    hipLaunchKernelGGL(k_bn_sum_partials_l1, dim3((unsigned)vd_cdiv(2 * C, 64), chunks), dim3(1024), 0, s, part, nblk, 2 * C,
                       (double*)ws);
    VD_CHECK_LAUNCH("vd_bn_sum_partials/l1");
    hipLaunchKernelGGL(k_bn_sum_partials_l2, dim3((unsigned)vd_cdiv(2 * C, 64)), dim3(64), 0, s, (const double*)ws, chunks,
                       2 * C, sums);
    VD_CHECK_LAUNCH("vd_bn_sum_partials/l2");
    return VD_OK;
}

int vd_bn_finalize(const double* sums, double count, int C, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                   float* save_mean, float* save_invstd, void* stream) {
    VD_REQUIRE(sums && gamma && beta && scale && shift && count > 0 && C > 0, "vd_bn_finalize: bad args");
    hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)vd_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, count,
                       C, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd);
    VD_CHECK_LAUNCH("vd_bn_finalize");
    return VD_OK;
}

int vd_bn_sum_finalize(const float* part, int nblk, int C, double* sums, double count, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                       float* scale, float* shift, float* save_mean, float* save_invstd, void* ws, int64_t ws_bytes,
                       void* stream) {
    VD_REQUIRE(part && sums && gamma && beta && scale && shift && nblk > 0 && C > 0 && count > 0, "vd_bn_sum_finalize: bad args");
    if (nblk <= SUMP_FUSED) {
        hipLaunchKernelGGL(k_bn_sum_fused<0>, dim3((unsigned)vd_cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream, part, nblk, C, sums,
                           count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd,
                           (float*)nullptr, (float*)nullptr);
        VD_CHECK_LAUNCH("vd_bn_sum_finalize");
        return VD_OK;
    }
    const int rc = vd_bn_sum_partials(part, nblk, C, sums, ws, ws_bytes, stream);
    if (rc != VD_OK) return rc;
    return vd_bn_finalize(sums, count, C, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean,
                          save_invstd, stream);
}

int vd_bn_sum_param_grads(const float* part, int nblk, int C, double* sums2, float* dgamma, float* dbeta, void* ws,
                          int64_t ws_bytes, void* stream) {
    VD_REQUIRE(part && sums2 && dgamma && dbeta && nblk > 0 && C > 0, "vd_bn_sum_param_grads: bad args");
    if (nblk <= SUMP_FUSED) {
        hipLaunchKernelGGL(k_bn_sum_fused<1>, dim3((unsigned)vd_cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream, part, nblk, C, sums2,
                           1.0, (const float*)nullptr, (const float*)nullptr, 0.f, 0.f, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, dgamma, dbeta);
        VD_CHECK_LAUNCH("vd_bn_sum_param_grads");
        return VD_OK;
    }
    const int rc = vd_bn_sum_partials(part, nblk, C, sums2, ws, ws_bytes, stream);
    if (rc != VD_OK) return rc;
    return vd_bn_param_grads(sums2, C, dgamma, dbeta, stream);
}

int vd_range_guard(const vd_guard_item* items, int n, float thresh, float* ratios, int32_t* flags, void* stream) {
    VD_REQUIRE(items && ratios && flags && n > 0 && thresh > 0.f && thresh < 1.f, "vd_range_guard: bad args");
    hipLaunchKernelGGL(k_range_guard, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, items, thresh, ratios, flags);
    VD_CHECK_LAUNCH("vd_range_guard");
    return VD_OK;
}

int vd_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                    float eps, int C, float* scale, float* shift, void* stream) {
    VD_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "vd_bn_fold_eval: bad args");
    hipLaunchKernelGGL(k_bn_fold_eval, dim3((unsigned)vd_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    VD_CHECK_LAUNCH("vd_bn_fold_eval");
    return VD_OK;
}

int vd_bn_apply_leaky(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                      int64_t M, int C, float slope, float* amax_out, void* stream) {
    VD_REQUIRE(x && scale && shift && y && M > 0 && C > 0 && C % 4 == 0, "vd_bn_apply_leaky: bad args");
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(k_bn_apply_leaky<float>, dim3(fixed_col_blocks(n4, C / 4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                       residual, y, n4, C / 4, slope, amax_out);
    VD_CHECK_LAUNCH("vd_bn_apply_leaky");
    return VD_OK;
}

int vd_bn_bwd_reduce(const float* x, const float* dy, const float* scale, const float* shift,
                     const float* save_mean, const float* save_invstd, int64_t M, int C, float slope,
                     double* sums2, void* ws, int64_t ws_bytes, void* stream) {
    VD_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && sums2 && ws && M > 0 && C % 4 == 0,
               "vd_bn_bwd_reduce: bad args");
    const int nb = red_blocks(M);
    if (ws_bytes < (int64_t)nb * 2 * C * (int64_t)sizeof(float)) {
        vd_set_error("vd_bn_bwd_reduce: workspace too small");
        return VD_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((k_bn_partial<1, float>), dim3(nb), dim3(RED_THREADS), 0, s, x, dy, scale, shift, save_mean, save_invstd,
                       M, C, slope, (float*)ws);
    VD_CHECK_LAUNCH("vd_bn_bwd_reduce");
    hipLaunchKernelGGL(k_bn_sum_partials, dim3((unsigned)vd_cdiv(2 * C, 64)), dim3(1024), 0, s, (const float*)ws, nb,
                       2 * C, sums2);
    VD_CHECK_LAUNCH("vd_bn_bwd_reduce/sum");
    return VD_OK;
}

int vd_bn_param_grads(const double* sums2, int C, float* dgamma, float* dbeta, void* stream) {
    VD_REQUIRE(sums2 && dgamma && dbeta && C > 0, "vd_bn_param_grads: bad args");
    hipLaunchKernelGGL(k_bn_param_grads, dim3((unsigned)vd_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums2, C,
                       dgamma, dbeta);
    VD_CHECK_LAUNCH("vd_bn_param_grads");
    return VD_OK;
}

int vd_bn_bwd_apply(const float* x, const float* dy, const float* scale, const float* shift, const float* save_mean,
                    const float* save_invstd, const double* sums2, double count, int64_t M, int C, float slope,
                    float* dx, float* amax_out, void* stream) {
    VD_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && sums2 && dx && count > 0 && C % 4 == 0,
               "vd_bn_bwd_apply: bad args");
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(fixed_col_blocks(n4, C / 4)), dim3(256), 0, (hipStream_t)stream, x, dy, scale, shift,
                       save_mean, save_invstd, sums2, count, n4, C, slope, dx, amax_out);
    VD_CHECK_LAUNCH("vd_bn_bwd_apply");
    return VD_OK;
}

// ---- bf16-storage training (BASELINE configs[4]): the same passes on bf16 tensors - fp32 arithmetic on the widened values,
// fp32 per-channel vectors, fp64 [2C] sums; x / dy / y / dx are bf16 [M, C]
int vd_bn_stats_bf16(const void* x, int64_t M, int C, double* sums, void* ws, int64_t ws_bytes, void* stream) {
    VD_REQUIRE(x && sums && ws && M > 0 && C > 0 && C % 4 == 0, "vd_bn_stats_bf16: bad args (C=%d)", C);
    const int nb = red_blocks(M);
    if (ws_bytes < (int64_t)nb * 2 * C * (int64_t)sizeof(float)) {
        vd_set_error("vd_bn_stats_bf16: workspace too small");
        return VD_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (bf16x8_enabled() && C % 8 == 0 && (uintptr_t)x % 16 == 0)
        hipLaunchKernelGGL(k_bn_partial_bf16x8<0>, dim3(nb), dim3(RED_THREADS), 0, s, (const __bf16*)x, (const __bf16*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, C, 0.f, (float*)ws);
    else
        hipLaunchKernelGGL((k_bn_partial<0, __bf16>), dim3(nb), dim3(RED_THREADS), 0, s, (const __bf16*)x, (const __bf16*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, C, 0.f, (float*)ws);
    VD_CHECK_LAUNCH("vd_bn_stats_bf16");
    hipLaunchKernelGGL(k_bn_sum_partials, dim3((unsigned)vd_cdiv(2 * C, 64)), dim3(1024), 0, s, (const float*)ws, nb, 2 * C, sums);
    VD_CHECK_LAUNCH("vd_bn_stats_bf16/sum");
    return VD_OK;
}

int vd_bn_apply_leaky_bf16(const void* x, const float* scale, const float* shift, const void* residual, void* y,
                           int64_t M, int C, float slope, void* stream) {
    VD_REQUIRE(x && scale && shift && y && M > 0 && C > 0 && C % 4 == 0, "vd_bn_apply_leaky_bf16: bad args");
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(k_bn_apply_leaky<__bf16>, dim3(fixed_col_blocks(n4, C / 4)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x,
                       scale, shift, (const __bf16*)residual, (__bf16*)y, n4, C / 4, slope, (float*)nullptr);
    VD_CHECK_LAUNCH("vd_bn_apply_leaky_bf16");
    return VD_OK;
}

int vd_bn_bwd_reduce_bf16(const void* x, const void* dy, const float* scale, const float* shift, const float* save_mean,
                          const float* save_invstd, int64_t M, int C, float slope, double* sums2, void* ws, int64_t ws_bytes,
                          void* stream) {
    VD_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && sums2 && ws && M > 0 && C % 4 == 0,
               "vd_bn_bwd_reduce_bf16: bad args");
    const int nb = red_blocks(M);
    if (ws_bytes < (int64_t)nb * 2 * C * (int64_t)sizeof(float)) {
        vd_set_error("vd_bn_bwd_reduce_bf16: workspace too small");
        return VD_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (bf16x8_enabled() && C % 8 == 0 && ((uintptr_t)x | (uintptr_t)dy) % 16 == 0)
        hipLaunchKernelGGL(k_bn_partial_bf16x8<1>, dim3(nb), dim3(RED_THREADS), 0, s, (const __bf16*)x, (const __bf16*)dy, scale, shift,
                           save_mean, save_invstd, M, C, slope, (float*)ws);
    else
        hipLaunchKernelGGL((k_bn_partial<1, __bf16>), dim3(nb), dim3(RED_THREADS), 0, s, (const __bf16*)x, (const __bf16*)dy, scale, shift,
                           save_mean, save_invstd, M, C, slope, (float*)ws);
    VD_CHECK_LAUNCH("vd_bn_bwd_reduce_bf16");
    hipLaunchKernelGGL(k_bn_sum_partials, dim3((unsigned)vd_cdiv(2 * C, 64)), dim3(1024), 0, s, (const float*)ws, nb, 2 * C, sums2);
    VD_CHECK_LAUNCH("vd_bn_bwd_reduce_bf16/sum");
    return VD_OK;
}

int vd_bn_bwd_apply_bf16(const void* x, const void* dy, const float* scale, const float* shift, const float* save_mean,
                         const float* save_invstd, const double* sums2, double count, int64_t M, int C, float slope,
                         void* dx, void* stream) {
    VD_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && sums2 && dx && count > 0 && C % 4 == 0,
               "vd_bn_bwd_apply_bf16: bad args");
    const int64_t n4 = M * (C / 4);
    hipLaunchKernelGGL(k_bn_bwd_apply<__bf16>, dim3(fixed_col_blocks(n4, C / 4)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x,
                       (const __bf16*)dy, scale, shift, save_mean, save_invstd, sums2, count, n4, C, slope, (__bf16*)dx,
                       (float*)nullptr);
    VD_CHECK_LAUNCH("vd_bn_bwd_apply_bf16");
    return VD_OK;
}

}  // extern "C"
