// vd_yolo.hip — YOLOv3 head kernels for gfx950: anchor decode, score filter, top-k + per-class NMS,
// and the fused IoU-matching loss (decode -> dynamic ignore mask -> target merge -> 4 losses + grads).
//
// Reference call sites (under /root/reference):
//   YOLOOutputV3.hybrid_forward (decode, per-class rows)   models/definitions/yolo/yolo3.py:132-199
//   concat + F.contrib.box_nms + slice_axis                models/definitions/yolo/yolo3.py:1195-1206
//   YOLOV3DynamicTargetGeneratorSimple / TargetMerger      models/definitions/yolo/yolo_target.py:173-281
//   gluoncv.loss.YOLOV3Loss (not vendored)  models/definitions/yolo/yolo3.py:994,1187
//
// Design: these are HBM-bound.  The reference materialises a (B, C*P, 6) tensor (20-44 MB/img) and
// sorts it; here one wave owns one grid cell (3 anchors x (5+C) contiguous floats = one coalesced
// run), stages it in LDS, and emits only the (score,row) pairs that pass valid_thresh with a
// ballot/popcount wave-aggregated append (one atomic per wave-iteration).  Boxes are re-decoded from
// the head tensor for the <=topk survivors only.  NMS runs one workgroup per image: radix-select ->
// bitonic sort (LDS) -> 64-bit suppression bitmask matrix (LDS) -> wave sweep.
#include "vd_common.h"

namespace {

constexpr int TOPK_MAX = 512;
constexpr int SORT_N = 1024;
constexpr int NMS_THREADS = 1024;

struct Box { float x1, y1, x2, y2; };

// intra-wave LDS hand-off: LDS ops of one wave execute in order, so only the compiler must be kept
// from reordering the accesses (no instruction is generated)
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

__device__ __forceinline__ Box decode_box(const float* raw, int x, int y, float stride, float aw, float ah) {
    // yolo3.py:172-177: centers = (sigmoid(raw_xy)+offset)*stride ; scales = exp(raw_wh)*anchor ; corners
    const float cx = (vd_sigmoid(raw[0]) + (float)x) * stride;
    const float cy = (vd_sigmoid(raw[1]) + (float)y) * stride;
    const float w = expf(raw[2]) * aw;
    const float h = expf(raw[3]) * ah;
    const float hw = w / 2.0f, hh = h / 2.0f;
    return Box{cx - hw, cy - hh, cx + hw, cy + hh};
}

// ------------------------------------------------------------------------------------------
// inference: decode + valid_thresh filter
// ------------------------------------------------------------------------------------------
// Candidates are collected per workgroup in LDS and published with ONE global atomic per workgroup: a returning
// atomic on the 32 per-image counters for every (cell, anchor, class chunk) that passes was a latency chain per wave
// and ~7500 contended atomics per counter (1.1 ms at batch 32, 608x608 for 0.25 GB of input).
constexpr int DF_LCAP = 2048;            // LDS candidate buffer per workgroup; beyond it, appends go straight to global

__global__ __launch_bounds__(256) void k_decode_filter(const vd_head_desc h, float thresh,
                                                       float* __restrict__ cand_score,
                                                       int32_t* __restrict__ cand_row, int cap,
                                                       int32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float lscore[DF_LCAP];
    __shared__ int32_t lrow[DF_LCAP];
    __shared__ int lcount, lfill, gbase;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int npred = 5 + h.C, A = 3 * npred;
    float* row = smem + wave * (((A + 3) & ~3) + 4);       // 16-byte aligned rows
    const int R0 = h.g[0] * h.g[0], R1 = h.g[1] * h.g[1], R2 = h.g[2] * h.g[2];
    const int R = R0 + R1 + R2;
    if (threadIdx.x == 0) {
        lcount = 0;
        lfill = 1 << 30;                      // base of the first reservation that did not fit (none yet)
    }
    __syncthreads();
    // Row staging: one 16-byte load per lane covers a whole 256-float head row, and the NEXT row of this wave is requested
    // before the current one is decoded (round 2's form - 4 B per lane, one row in flight per wave - ran at 1.55 TB/s on
    // 248 MB).  vec: rows start 16-byte aligned and hold a whole number of float4s (every plan of the network).
    constexpr int NV = 4;                    // float4s per lane: rows of up to 1024 floats (C <= 336)
    const int RW = (A + 3) & ~3;
    const bool vec = (h.ldh % 4 == 0) && RW <= NV * 256 &&
                     ((((uintptr_t)h.head[0] | (uintptr_t)h.head[1] | (uintptr_t)h.head[2]) & 15) == 0);
    auto row_src = [&](int r) -> const float* {
        int s, pix;
        if (r < R0) { s = 0; pix = r; }
        else if (r < R0 + R1) { s = 1; pix = r - R0; }
        else { s = 2; pix = r - R0 - R1; }
        return h.head[s] + ((int64_t)b * h.g[s] * h.g[s] + pix) * h.ldh;
    };
    f32x4 nx[NV];
    auto fetch = [&](int r) {
        const float* src = row_src(r);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = 4 * lane + 256 * v;
            nx[v] = e < RW ? *reinterpret_cast<const f32x4*>(src + e) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int rstep = gridDim.x * 4;
    if (vec && (int)(blockIdx.x * 4 + wave) < R) fetch(blockIdx.x * 4 + wave);
    for (int r = blockIdx.x * 4 + wave; r < R; r += rstep) {
        int s, pix, rowbase;
        if (r < R0) { s = 0; pix = r; rowbase = 0; }
        else if (r < R0 + R1) { s = 1; pix = r - R0; rowbase = h.C * 3 * R0; }
        else { s = 2; pix = r - R0 - R1; rowbase = h.C * 3 * (R0 + R1); }
        const int g = h.g[s];
        if (vec) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int e = 4 * lane + 256 * v;
                if (e < RW) *reinterpret_cast<f32x4*>(row + e) = nx[v];
            }
            if (r + rstep < R) fetch(r + rstep);
        } else {
            const float* src = row_src(r);
            for (int e = lane; e < A; e += 64) row[e] = src[e];
        }
        WAVE_SYNC();
        const float o0 = vd_sigmoid(row[4]), o1 = vd_sigmoid(row[npred + 4]), o2 = vd_sigmoid(row[2 * npred + 4]);
        // sigmoid(cls) < 1  =>  score < obj : an anchor whose objectness fails the threshold has no
        // passing class, so its exps are skipped (exact, not an approximation)
        if (o0 > thresh || o1 > thresh || o2 > thresh) {
            const int gg3 = g * g * 3;
            for (int a = 0; a < 3; ++a) {
                const float obj = a == 0 ? o0 : (a == 1 ? o1 : o2);
                if (!(obj > thresh)) continue;
                for (int c0 = 0; c0 < h.C; c0 += 64) {
                    const int c = c0 + lane;
                    float score = 0.f;
                    bool pass = false;
                    if (c < h.C) {
                        score = vd_sigmoid(row[a * npred + 5 + c]) * obj;
                        pass = score > thresh;
                    }
                    const unsigned long long m = __ballot(pass);
                    if (m) {
                        const int n = (int)__popcll(m);
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&lcount, n);                  // LDS atomic
                        base = __shfl(base, 0);
                        const int rowid = rowbase + c * gg3 + pix * 3 + a;
                        if (base + n <= DF_LCAP) {
                            if (pass) {
                                const int slot = base + (int)__popcll(m & ((1ull << lane) - 1ull));
                                lscore[slot] = score;
                                lrow[slot] = rowid;
                            }
                        } else {                                                        // LDS buffer full: publish directly
                            int gb = 0;
                            if (lane == 0) {
                                atomicMin(&lfill, base);
                                gb = atomicAdd(&counts[b], n);
                            }
                            gb = __shfl(gb, 0);
                            if (pass) {
                                const int slot = gb + (int)__popcll(m & ((1ull << lane) - 1ull));
                                if (slot < cap) {
                                    cand_score[(int64_t)b * cap + slot] = score;
                                    cand_row[(int64_t)b * cap + slot] = rowid;
                                }
                            }
                        }
                    }
                }
            }
        }
        WAVE_SYNC();
    }
    __syncthreads();
    // Reservations are handed out in order from 0: every one before the first that did not fit filled its slots, and
    // every one after it starts past DF_LCAP and went to global - the LDS part is exactly [0, min(lcount, lfill)).
    const int nl = lcount < lfill ? lcount : lfill;
    if (threadIdx.x == 0) gbase = atomicAdd(&counts[b], nl);
    __syncthreads();
    for (int i = threadIdx.x; i < nl; i += 256) {
        const int slot = gbase + i;
        if (slot < cap) {
            cand_score[(int64_t)b * cap + slot] = lscore[i];
            cand_row[(int64_t)b * cap + slot] = lrow[i];
        }
    }
}

// Objectness-first form (round 4; the default).  A class score is sigmoid(cls) * sigmoid(obj) < sigmoid(obj), so an
// (pixel, anchor) whose objectness fails valid_thresh has no passing class (exact) - and with a trained network (or
// bench.py's calibrated random one) that is ~95 % of them.  k_decode_filter above streams every 1 KB head row through LDS to
// find that out; here a LANE owns one (pixel, anchor), reads its ONE objectness logit (a 4-byte gather: three of the
// row's eight 128-byte lines), and only the entries that pass have their class vector read - by the whole wave, 64
// classes per load.  At batch 32 / 608x608 / 80 classes the kernel touches ~40 % of the head bytes and has no LDS
// staging at all (16 KB of candidate buffer per workgroup: eight workgroups per CU hide the gather's latency).  Same
// candidate set and row ids as k_decode_filter; the append order is arbitrary in both (vd_nms_topk keys by (score, row)).
__global__ __launch_bounds__(256) void k_decode_filter_obj(const vd_head_desc h, float thresh, float* __restrict__ cand_score,
                                                           int32_t* __restrict__ cand_row, int cap, int32_t* __restrict__ counts) {
    __shared__ float lscore[DF_LCAP];
    __shared__ int32_t lrow[DF_LCAP];
    __shared__ int lcount, lfill, gbase;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int npred = 5 + h.C;
    const int R0 = h.g[0] * h.g[0], R1 = h.g[1] * h.g[1], R2 = h.g[2] * h.g[2];
    const int E = 3 * (R0 + R1 + R2);                 // (pixel, anchor) entries of one image
    if (threadIdx.x == 0) {
        lcount = 0;
        lfill = 1 << 30;
    }
    __syncthreads();
    const int estep = gridDim.x * 256;
    // wave-uniform trip count: the class sweep below is a whole-wave affair
    for (int e0 = blockIdx.x * 256 + (threadIdx.x & ~63); e0 < E; e0 += estep) {
        const int e = e0 + lane;
        const bool in = e < E;
        const int r = in ? e / 3 : 0, a = in ? e - 3 * r : 0;
        int s, pix, rowbase;
        if (r < R0) { s = 0; pix = r; rowbase = 0; }
        else if (r < R0 + R1) { s = 1; pix = r - R0; rowbase = h.C * 3 * R0; }
        else { s = 2; pix = r - R0 - R1; rowbase = h.C * 3 * (R0 + R1); }
        const int g = h.g[s];
        const float* src = h.head[s] + ((int64_t)b * g * g + pix) * h.ldh + a * npred;
        const float obj = in ? vd_sigmoid(src[4]) : 0.f;
        unsigned long long todo = __ballot(in && obj > thresh);
        while (todo) {
            const int l = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            // lane l's entry, by every lane of the wave
            const float* csrc = reinterpret_cast<const float*>(
                ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)(uintptr_t)src >> 32), l) << 32) |
                (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)(uintptr_t)src & 0xffffffffull), l));
            const float o = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, obj), l));
            const int rb = __builtin_amdgcn_readlane(rowbase, l), px = __builtin_amdgcn_readlane(pix, l),
                      an = __builtin_amdgcn_readlane(a, l), gg3 = __builtin_amdgcn_readlane(g * g * 3, l);
            for (int c0 = 0; c0 < h.C; c0 += 64) {
                const int c = c0 + lane;
                float score = 0.f;
                bool pass = false;
                if (c < h.C) {
                    score = vd_sigmoid(csrc[5 + c]) * o;
                    pass = score > thresh;
                }
                const unsigned long long m = __ballot(pass);
                if (m) {
                    const int n = (int)__popcll(m);
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&lcount, n);                  // LDS atomic
                    base = __shfl(base, 0);
                    const int rowid = rb + c * gg3 + px * 3 + an;
                    if (base + n <= DF_LCAP) {
                        if (pass) {
                            const int slot = base + (int)__popcll(m & ((1ull << lane) - 1ull));
                            lscore[slot] = score;
                            lrow[slot] = rowid;
                        }
                    } else {                                                        // LDS buffer full: publish directly
                        int gb = 0;
                        if (lane == 0) {
                            atomicMin(&lfill, base);
                            gb = atomicAdd(&counts[b], n);
                        }
                        gb = __shfl(gb, 0);
                        if (pass) {
                            const int slot = gb + (int)__popcll(m & ((1ull << lane) - 1ull));
                            if (slot < cap) {
                                cand_score[(int64_t)b * cap + slot] = score;
                                cand_row[(int64_t)b * cap + slot] = rowid;
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const int nl = lcount < lfill ? lcount : lfill;      // (see k_decode_filter)
    if (threadIdx.x == 0) gbase = atomicAdd(&counts[b], nl);
    __syncthreads();
    for (int i = threadIdx.x; i < nl; i += 256) {
        const int slot = gbase + i;
        if (slot < cap) {
            cand_score[(int64_t)b * cap + slot] = lscore[i];
            cand_row[(int64_t)b * cap + slot] = lrow[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// inference: top-k + per-class NMS, one workgroup per image
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NMS_THREADS) void k_nms(const vd_head_desc h, const float* __restrict__ cand_score,
                                                     const int32_t* __restrict__ cand_row, int cap,
                                                     const int32_t* __restrict__ counts, float nms_thresh,
                                                     int topk, int post_nms, float* __restrict__ out_ids,
                                                     float* __restrict__ out_scores, float* __restrict__ out_boxes,
                                                     int32_t* __restrict__ out_rows, int32_t* __restrict__ overflow) {
    __shared__ unsigned long long skey[SORT_N];
    __shared__ float bx1[TOPK_MAX], by1[TOPK_MAX], bx2[TOPK_MAX], by2[TOPK_MAX];
    __shared__ int bcls[TOPK_MAX];
    __shared__ unsigned long long smask[TOPK_MAX * (TOPK_MAX / 64)];
    __shared__ unsigned int hist[256];
    __shared__ unsigned int s_prefix_hi, s_prefix_lo, s_kth, s_cnt;
    __shared__ int keep[TOPK_MAX];
    __shared__ int s_nkeep;

    const int tid = threadIdx.x, b = blockIdx.x;
    int n = counts[b];
    if (n > cap) {
        if (tid == 0 && overflow) overflow[b] = n;
        n = cap;
    } else if (tid == 0 && overflow) overflow[b] = 0;
    const float* cs = cand_score + (int64_t)b * cap;
    const int32_t* cr = cand_row + (int64_t)b * cap;

    skey[tid] = ~0ull;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (n <= SORT_N) {
        if (tid < n) skey[tid] = ((unsigned long long)(~__float_as_uint(cs[tid])) << 32) | (unsigned int)cr[tid];
    } else {
        // Exact top-k of the candidate list, independent of the order the filter's atomics appended it in: radix
        // select (8 digits of 8 bits, most significant first) of the topk-th largest 64-bit key
        //     key = score bits << 32 | (0xFFFFFFFF - row)      (scores > 0 => uint order == float order)
        // i.e. score descending, then ORIGINAL ROW ascending - the order a stable sort of the reference's
        // (B, C*P, 6) tensor gives.  Rows are unique, so keys are unique and exactly `topk` keys are
        // >= the selected one: rows tied at the threshold score are kept by row order, never by arrival order.
        if (tid == 0) { s_prefix_hi = 0; s_prefix_lo = 0; s_kth = (unsigned)topk; }
        // The candidates' score bits are read ONCE, NLOC per thread, all requests in flight together (lists of up to
        // NLOC x 1024 = 40960 candidates: 37 k per image at batch 32 / 608x608 with the calibrated 2 % pass rate); the nine
        // sweeps below then run out of registers.  Reading them again from global memory in every sweep - one dependent L2
        // round trip per 1024 candidates and sweep behind the LDS atomics - was ~90 us of this kernel's 220.  Longer lists
        // keep the streaming form.
        constexpr int NLOC = 40;
        const bool loc = n <= NLOC * NMS_THREADS;
        unsigned kreg[NLOC];
#pragma unroll
        for (int j = 0; j < NLOC; ++j) {
            const int i = tid + j * NMS_THREADS;
            kreg[j] = (loc && i < n) ? __float_as_uint(cs[i]) : 0u;
        }
        for (int pass = 7; pass >= 0; --pass) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned phi = s_prefix_hi, plo = s_prefix_lo;
            auto digit = [&](int i, unsigned k) {
                if (pass >= 4) {                       // score digits: prefix = the digits above this one
                    const int hs = 8 * (pass - 3);
                    if (pass == 7 || (k >> hs) == phi) atomicAdd(&hist[(k >> (8 * (pass - 4))) & 255u], 1u);
                } else if (k == phi) {                 // row digits, only among rows tied at the threshold score
                    const unsigned r = 0xFFFFFFFFu - (unsigned)cr[i];
                    const int hs = 8 * (pass + 1);
                    if (pass == 3 || (r >> hs) == plo) atomicAdd(&hist[(r >> (8 * pass)) & 255u], 1u);
                }
            };
            // (wave-aggregated atomics for the leading digit - four or five distinct values - measured no faster: 0.165 vs 0.157 ms)
            if (loc) {
#pragma unroll
                for (int j = 0; j < NLOC; ++j) {
                    const int i = tid + j * NMS_THREADS;
                    if (i < n) digit(i, kreg[j]);
                }
            } else {
                for (int i = tid; i < n; i += NMS_THREADS) digit(i, __float_as_uint(cs[i]));
            }
            __syncthreads();
            // the digit of the kth-largest key: the largest d >= 1 whose suffix count S(d) = sum_{b >= d} hist[b] reaches kth
            // (else 0), by ONE wave - four bins per lane, a suffix scan over the lanes.  (One thread walking the 256 bins was a
            // chain of 256 dependent LDS reads per sweep: ~8 us x 8 sweeps of a kernel whose whole budget is 200.)
            if (tid < 64) {
                const unsigned kth = s_kth;
                const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                const unsigned t = h0 + h1 + h2 + h3;
                unsigned suf = t;                                   // inclusive suffix sum over lanes >= this one
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_down(suf, off);
                    if (tid + off < 64) suf += o;
                }
                const unsigned S3 = suf - t + h3, S2 = S3 + h2, S1 = S2 + h1, S0 = S1 + h0;
                int dl = -1;
                unsigned above = 0;                                 // S(dl + 1)
                if (S3 >= kth) { dl = 4 * tid + 3; above = suf - t; }
                else if (S2 >= kth) { dl = 4 * tid + 2; above = S3; }
                else if (S1 >= kth) { dl = 4 * tid + 1; above = S2; }
                else if (S0 >= kth && tid > 0) { dl = 4 * tid; above = S1; }
                const unsigned long long has = __ballot(dl >= 1);
                int d = 0;
                unsigned cum;
                if (has) {
                    const int top = 63 - (int)__builtin_clzll(has);
                    d = __builtin_amdgcn_readlane(dl, top);
                    cum = (unsigned)__builtin_amdgcn_readlane((int)above, top);
                } else {
                    cum = (unsigned)__builtin_amdgcn_readlane((int)S1, 0);        // nothing reaches kth above bin 0: S(1)
                }
                if (tid == 0) {
                    if (pass >= 4) s_prefix_hi = (phi << 8) | (unsigned)d;
                    else s_prefix_lo = (plo << 8) | (unsigned)d;
                    s_kth = kth - cum;
                }
            }
            __syncthreads();
        }
        const unsigned Ts = s_prefix_hi, Tr = s_prefix_lo;
        auto take = [&](int i, unsigned k) {
            if (k < Ts) return;
            const unsigned row = (unsigned)cr[i];
            if (k > Ts || (0xFFFFFFFFu - row) >= Tr) {
                const unsigned slot = atomicAdd(&s_cnt, 1u);       // exactly topk <= SORT_N of them; the sort orders them
                if (slot < SORT_N) skey[slot] = ((unsigned long long)(~k) << 32) | row;
            }
        };
        if (loc) {
#pragma unroll
            for (int j = 0; j < NLOC; ++j) {
                const int i = tid + j * NMS_THREADS;
                if (i < n) take(i, kreg[j]);
            }
        } else {
            for (int i = tid; i < n; i += NMS_THREADS) take(i, __float_as_uint(cs[i]));
        }
    }
    __syncthreads();
    // bitonic sort ascending: (~score, row) => score descending, original row ascending on ties
    for (int k = 2; k <= SORT_N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int ixj = tid ^ j;
            if (ixj > tid) {
                const unsigned long long a = skey[tid], c = skey[ixj];
                const bool up = ((tid & k) == 0);
                if ((a > c) == up) { skey[tid] = c; skey[ixj] = a; }
            }
            __syncthreads();
        }
    }
    int nsel = n < topk ? n : topk;
    if (nsel > SORT_N) nsel = SORT_N;
    // re-decode the survivors' boxes from the head tensor
    const int R0 = h.g[0] * h.g[0], R1 = h.g[1] * h.g[1];
    const int base1 = h.C * 3 * R0, base2 = h.C * 3 * (R0 + R1);
    const int npred = 5 + h.C;
    if (tid < nsel) {
        const int row = (int)(unsigned int)(skey[tid] & 0xffffffffull);
        int s, rr;
        if (row < base1) { s = 0; rr = row; }
        else if (row < base2) { s = 1; rr = row - base1; }
        else { s = 2; rr = row - base2; }
        const int g = h.g[s], gg3 = g * g * 3;
        const int c = rr / gg3, rem = rr - c * gg3;
        const int pix = rem / 3, a = rem - pix * 3;
        const float* src = h.head[s] + ((int64_t)b * g * g + pix) * h.ldh + a * npred;
        float raw[4] = {src[0], src[1], src[2], src[3]};
        const Box bb = decode_box(raw, pix % g, pix / g, h.stride[s], h.anchors[s][2 * a], h.anchors[s][2 * a + 1]);
        bx1[tid] = bb.x1; by1[tid] = bb.y1; bx2[tid] = bb.x2; by2[tid] = bb.y2;
        bcls[tid] = c;
    }
    __syncthreads();
    const int nw = (nsel + 63) >> 6;
    for (int idx = tid; idx < nsel * nw; idx += NMS_THREADS) {
        const int i = idx / nw, w = idx - i * nw;
        const float ax1 = bx1[i], ay1 = by1[i], ax2 = bx2[i], ay2 = by2[i];
        const float aarea = (ax2 - ax1) * (ay2 - ay1);
        const int ac = bcls[i];
        unsigned long long bits = 0;
        for (int jj = 0; jj < 64; ++jj) {
            const int j = w * 64 + jj;
            if (j <= i || j >= nsel || bcls[j] != ac) continue;
            const float iw = fmaxf(0.f, fminf(ax2, bx2[j]) - fmaxf(ax1, bx1[j]));
            const float ih = fmaxf(0.f, fminf(ay2, by2[j]) - fmaxf(ay1, by1[j]));
            const float inter = iw * ih;
            const float uni = aarea + (bx2[j] - bx1[j]) * (by2[j] - by1[j]) - inter;
            const float iou = uni <= 0.f ? 0.f : inter / uni;
            if (iou > nms_thresh) bits |= (1ull << jj);
        }
        smask[i * (TOPK_MAX / 64) + w] = bits;
    }
    __syncthreads();
    // greedy sweep by one wave: lane w owns removed-word w
    if (tid < 64) {
        unsigned long long removed = 0;
        int nkeep = 0;
        for (int i = 0; i < nsel; ++i) {
            const unsigned long long wordi = __shfl(removed, i >> 6);
            if (!((wordi >> (i & 63)) & 1ull)) {
                if (tid == 0) keep[nkeep] = i;
                ++nkeep;
                if (tid < nw) removed |= smask[i * (TOPK_MAX / 64) + tid];
            }
        }
        if (tid == 0) s_nkeep = nkeep;
    }
    __syncthreads();
    const int nkeep = s_nkeep;
    for (int j = tid; j < post_nms; j += NMS_THREADS) {
        const int64_t o = (int64_t)b * post_nms + j;
        if (j < nkeep) {
            const int i = keep[j];
            const unsigned long long k = skey[i];
            out_ids[o] = (float)bcls[i];
            out_scores[o] = __uint_as_float(~(unsigned int)(k >> 32));
            out_boxes[o * 4 + 0] = bx1[i]; out_boxes[o * 4 + 1] = by1[i];
            out_boxes[o * 4 + 2] = bx2[i]; out_boxes[o * 4 + 3] = by2[i];
            out_rows[o] = (int)(unsigned int)(k & 0xffffffffull);
        } else {
            out_ids[o] = -1.f; out_scores[o] = -1.f;
            out_boxes[o * 4 + 0] = -1.f; out_boxes[o * 4 + 1] = -1.f;
            out_boxes[o * 4 + 2] = -1.f; out_boxes[o * 4 + 3] = -1.f;
            out_rows[o] = -1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// training: fused decode + dynamic ignore + target merge + YOLOV3Loss forward/backward
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bce_logits(float x, float z) {
    // gluon SigmoidBinaryCrossEntropyLoss(from_sigmoid=False): relu(x) - x*z + softrelu(-|x|)
    return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

constexpr int LOSS_MAX_GT = 256;

// One wave per head row (one pixel: 3 anchors x (5 + C) logits).  The row is staged in LDS with 16-B loads, the three
// anchors' boxes are decoded and matched against the gt boxes by 3 x 16 lanes (lane = anchor * 16 + gt index mod 16, a
// 16-lane max), and the gradient row - padding included - leaves with 16-B stores: one pass over head, targets and
// gradient.  (Round 2's form read and wrote 4 B per lane at unaligned anchor offsets and ran decode + the gt loop on
// lanes 0..2 of every wave: 1.18 TB/s.)  VEC = 16-B accesses (ldh % 4 == 0, 16-B aligned tensors: every plan of the
// network); the scalar form stays for odd pitches.
// GT = storage type of the gradient rows dh* (float, or __bf16 in bf16-storage training: the head logits stay fp32)
template <bool VEC, typename GT = float>
__global__ __launch_bounds__(256) void k_yolo_loss(const vd_head_desc h, const float* __restrict__ gt, int M,
                                                   const float* __restrict__ obj_t,
                                                   const float* __restrict__ center_t,
                                                   const float* __restrict__ scale_t,
                                                   const float* __restrict__ weight_t,
                                                   const float* __restrict__ class_t, float ignore_thresh,
                                                   int label_smooth, GT* dh0, GT* dh1, GT* dh2,
                                                   float* __restrict__ box_out, float* __restrict__ part,
                                                   float* am0, float* am1, float* am2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float sgt[LOSS_MAX_GT * 4];
    __shared__ float sred[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int C = h.C, npred = 5 + C, A = 3 * npred;
    const int RW = (A + 3) & ~3;                       // staged row, a whole number of float4s (<= ldh when VEC)
    float* row = smem + wave * (RW + 32);
    float* aux = row + RW;  // [3][8]: 5 grads, mask flag, objness
    const int R0 = h.g[0] * h.g[0], R1 = h.g[1] * h.g[1], R2 = h.g[2] * h.g[2];
    const int R = R0 + R1 + R2, P = 3 * R;
    for (int i = threadIdx.x; i < M * 4; i += blockDim.x) sgt[i] = gt[(int64_t)b * M * 4 + i];
    __syncthreads();
    float l_obj = 0.f, l_ctr = 0.f, l_scl = 0.f, l_cls = 0.f;
    float amx0 = 0.f, amx1 = 0.f, amx2 = 0.f;   // max-abs of the gradients this lane writes, per scale
    const float sw = fminf(1.0f / (float)C, 1.0f / 40.0f);
    const int la = lane >> 4, lm = lane & 15;   // anchor / gt slot of this lane in the matching phase
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        int s, pix, pbase;
        if (r < R0) { s = 0; pix = r; pbase = 0; }
        else if (r < R0 + R1) { s = 1; pix = r - R0; pbase = 3 * R0; }
        else { s = 2; pix = r - R0 - R1; pbase = 3 * (R0 + R1); }
        const int g = h.g[s];
        const int64_t hoff = ((int64_t)b * g * g + pix) * h.ldh;
        const float* src = h.head[s] + hoff;
        GT* dst = (s == 0 ? dh0 : (s == 1 ? dh1 : dh2)) + hoff;
        if (VEC) {
            for (int e = 4 * lane; e < RW; e += 256) *reinterpret_cast<f32x4*>(row + e) = *reinterpret_cast<const f32x4*>(src + e);
        } else {
            for (int e = lane; e < A; e += 64) row[e] = src[e];
        }
        WAVE_SYNC();
        if (la < 3) {
            const int a = la;
            const float* raw = row + a * npred;
            const int64_t p = (int64_t)b * P + pbase + pix * 3 + a;
            const Box bb = decode_box(raw, pix % g, pix / g, h.stride[s], h.anchors[s][2 * a], h.anchors[s][2 * a + 1]);
            // yolo_target.py:202-204 + gluoncv BBoxBatchIOU (offset 0, eps 1e-15, clip at 6.5504e4): this lane's share of
            // the gt boxes, then the maximum over the anchor's 16 lanes (an IoU is never negative: -1 = "none seen")
            float ioumax = -1.f;
            const float parea = (bb.x2 - bb.x1) * (bb.y2 - bb.y1);
            for (int m = lm; m < M; m += 16) {
                const float gx1 = sgt[m * 4], gy1 = sgt[m * 4 + 1], gx2 = sgt[m * 4 + 2], gy2 = sgt[m * 4 + 3];
                const float iw = fminf(fmaxf(fminf(bb.x2, gx2) - fmaxf(bb.x1, gx1), 0.f), 6.5504e4f);
                const float ih = fminf(fmaxf(fminf(bb.y2, gy2) - fmaxf(bb.y1, gy1), 0.f), 6.5504e4f);
                const float inter = iw * ih;
                const float iou = inter / (parea + (gx2 - gx1) * (gy2 - gy1) - inter + 1e-15f);
                ioumax = fmaxf(ioumax, iou);
            }
            ioumax = fmaxf(ioumax, __shfl_xor(ioumax, 1));
            ioumax = fmaxf(ioumax, __shfl_xor(ioumax, 2));
            ioumax = fmaxf(ioumax, __shfl_xor(ioumax, 4));
            ioumax = fmaxf(ioumax, __shfl_xor(ioumax, 8));
            if (lm == 0) {
                if (box_out) {
                    box_out[p * 4 + 0] = bb.x1; box_out[p * 4 + 1] = bb.y1;
                    box_out[p * 4 + 2] = bb.x2; box_out[p * 4 + 3] = bb.y2;
                }
                const float dyn = (M > 0 && ioumax > ignore_thresh) ? -1.f : 0.f;
                const float of = obj_t[p];
                const bool mask = of > 0.f;                                 // yolo_target.py:264
                const float objness = mask ? of : dyn;
                // YOLOV3Loss
                const float hard = objness > 0.f ? 1.f : objness;
                const float omask = objness > 0.f ? objness : (objness >= 0.f ? 1.f : 0.f);
                const float xo = raw[4];
                l_obj += bce_logits(xo, hard) * omask;
                aux[a * 8 + 4] = (vd_sigmoid(xo) - hard) * omask;
                float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
                if (mask) {
                    const float w0 = weight_t[p * 2] * objness, w1 = weight_t[p * 2 + 1] * objness;
                    const float c0 = center_t[p * 2], c1 = center_t[p * 2 + 1];
                    const float s0 = scale_t[p * 2], s1 = scale_t[p * 2 + 1];
                    l_ctr += bce_logits(raw[0], c0) * w0 + bce_logits(raw[1], c1) * w1;
                    g0 = (vd_sigmoid(raw[0]) - c0) * w0;
                    g1 = (vd_sigmoid(raw[1]) - c1) * w1;
                    const float d2 = raw[2] - s0, d3 = raw[3] - s1;
                    l_scl += fabsf(d2) * w0 + fabsf(d3) * w1;
                    g2 = (d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f)) * w0;
                    g3 = (d3 > 0.f ? 1.f : (d3 < 0.f ? -1.f : 0.f)) * w1;
                }
                aux[a * 8 + 0] = g0; aux[a * 8 + 1] = g1; aux[a * 8 + 2] = g2; aux[a * 8 + 3] = g3;
                aux[a * 8 + 5] = mask ? 1.f : 0.f;
                aux[a * 8 + 6] = objness;
            }
        }
        WAVE_SYNC();
        float amx_r = 0.f;
        // gradient of element e = a * npred + j of the row (0 beyond A: the padding channels)
        auto grad_of = [&](int e) -> float {
            if (e >= A) return 0.f;
            const int a = e >= 2 * npred ? 2 : (e >= npred ? 1 : 0);
            const int j = e - a * npred;
            if (j < 5) return aux[a * 8 + j];
            if (!(aux[a * 8 + 5] > 0.f)) return 0.f;
            const float objness = aux[a * 8 + 6];
            const int64_t p = (int64_t)b * P + pbase + pix * 3 + a;
            float t = class_t[p * C + (j - 5)];
            if (label_smooth) {                                   // yolo_target.py:271-278
                if (t > 0.5f) t -= sw;
                if (!(t < -0.5f || t > 0.5f)) t = sw;
            }
            const float cm = (t >= 0.f ? 1.f : 0.f) * objness;    // class_mask * objness_t
            const float x = row[e];
            l_cls += bce_logits(x, t) * cm;
            return (vd_sigmoid(x) - t) * cm;
        };
        if (VEC) {
            for (int e = 4 * lane; e < h.ldh; e += 256) {
                f32x4 gv;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gv[q] = grad_of(e + q);
                    amx_r = fmaxf(amx_r, fabsf(gv[q]));
                }
                vd_st4(dst, e >> 2, gv);
            }
        } else {
            for (int e = lane; e < h.ldh; e += 64) {
                const float gv = grad_of(e);
                dst[e] = (GT)gv;
                amx_r = fmaxf(amx_r, fabsf(gv));
            }
        }
        if (s == 0) amx0 = fmaxf(amx0, amx_r);
        else if (s == 1) amx1 = fmaxf(amx1, amx_r);
        else amx2 = fmaxf(amx2, amx_r);
        WAVE_SYNC();
    }
    if (am0) {
        vd_amax_publish(am0, amx0);
        vd_amax_publish(am1, amx1);
        vd_amax_publish(am2, amx2);
    }
    // block reduction of the four partial losses (fixed order => deterministic)
    float v[4] = {l_obj, l_ctr, l_scl, l_cls};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if (lane == 0) sred[wave][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        part[((int64_t)b * gridDim.x + blockIdx.x) * 4 + k] = sred[0][k] + sred[1][k] + sred[2][k] + sred[3][k];
    }
}

__global__ void k_loss_finalize(const float* __restrict__ part, int nblk, float* __restrict__ losses, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 4) return;
    const int b = i >> 2, k = i & 3;
    double s = 0.0;
    for (int j = 0; j < nblk; ++j) s += (double)part[((int64_t)b * nblk + j) * 4 + k];
    losses[i] = (float)s;
}

int head_ok(const vd_head_desc* h) {
    if (!h || h->B <= 0 || h->C <= 0) return 0;
    for (int s = 0; s < 3; ++s)
        if (!h->head[s] || h->g[s] <= 0 || h->g[s] > 2048) return 0;
    return h->ldh >= 3 * (5 + h->C);
}

int loss_blocks(const vd_head_desc* h) {
    const int R = h->g[0] * h->g[0] + h->g[1] * h->g[1] + h->g[2] * h->g[2];
    int nb = (R + 3) / 4;
    const int cap = 2048 / (h->B < 1 ? 1 : h->B) + 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return nb;
}

}  // namespace

extern "C" {

int vd_yolo_decode_filter(const vd_head_desc* h, float valid_thresh, float* cand_score, int32_t* cand_row,
                          int32_t cap, int32_t* counts, void* stream) {
    VD_REQUIRE(head_ok(h), "vd_yolo_decode_filter: bad head descriptor");
    VD_REQUIRE(cand_score && cand_row && counts && cap > 0, "vd_yolo_decode_filter: bad args");
    VD_REQUIRE(valid_thresh >= 0.f, "vd_yolo_decode_filter: valid_thresh must be >= 0 (scores are keyed as positive floats)");
    VD_REQUIRE((int64_t)h->C * 3 * (h->g[0] * h->g[0] + h->g[1] * h->g[1] + h->g[2] * h->g[2]) < (1ll << 31),
               "vd_yolo_decode_filter: row index overflows int32");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * h->B, s) != hipSuccess) {
        vd_set_error("vd_yolo_decode_filter: memset failed");
        return VD_ELAUNCH;
    }
    // VD_DECODE_ROWS=1: the row-streaming form of rounds 1-3 (developer A/B); default: objectness first
    static const bool rows_form = getenv("VD_DECODE_ROWS") && atoi(getenv("VD_DECODE_ROWS")) != 0;
    if (!rows_form) {
        const int64_t E = 3ll * (h->g[0] * h->g[0] + h->g[1] * h->g[1] + h->g[2] * h->g[2]);
        int nbo = (int)vd_cdiv(E, 256 * 4);               // four entries per lane
        if (nbo > 2048) nbo = 2048;
        if (nbo < 1) nbo = 1;
        hipLaunchKernelGGL(k_decode_filter_obj, dim3(nbo, h->B), dim3(256), 0, s, *h, valid_thresh, cand_score, cand_row, cap, counts);
        VD_CHECK_LAUNCH("vd_yolo_decode_filter");
        return VD_OK;
    }
    const int A = 3 * (5 + h->C);
    const int lds = 4 * (((A + 3) & ~3) + 4) * (int)sizeof(float);
    VD_REQUIRE(lds <= 64 * 1024, "vd_yolo_decode_filter: too many classes for the LDS row stage");
    const int nb = loss_blocks(h);
    hipLaunchKernelGGL(k_decode_filter, dim3(nb, h->B), dim3(256), lds, s, *h, valid_thresh, cand_score, cand_row, cap,
                       counts);
    VD_CHECK_LAUNCH("vd_yolo_decode_filter");
    return VD_OK;
}

int64_t vd_nms_ws_bytes(int B, int cap, int topk) {
    (void)cap; (void)topk;
    return (int64_t)B * (int64_t)sizeof(int32_t);   // overflow flags
}

int vd_nms_topk(const vd_head_desc* h, const float* cand_score, const int32_t* cand_row, int32_t cap,
                const int32_t* counts, float nms_thresh, int topk, int post_nms, float* out_ids, float* out_scores,
                float* out_boxes, int32_t* out_rows, void* ws, int64_t ws_bytes, void* stream) {
    VD_REQUIRE(head_ok(h), "vd_nms_topk: bad head descriptor");
    VD_REQUIRE(cand_score && cand_row && counts && out_ids && out_scores && out_boxes && out_rows, "vd_nms_topk: null");
    VD_REQUIRE(topk > 0 && topk <= TOPK_MAX, "vd_nms_topk: topk=%d outside (0,%d]", topk, TOPK_MAX);
    VD_REQUIRE(post_nms > 0, "vd_nms_topk: post_nms must be > 0");
    VD_REQUIRE(nms_thresh > 0.f && nms_thresh < 1.f, "vd_nms_topk: nms_thresh must be in (0,1)");
    int32_t* overflow = (ws && ws_bytes >= (int64_t)h->B * (int64_t)sizeof(int32_t)) ? (int32_t*)ws : nullptr;
    hipLaunchKernelGGL(k_nms, dim3(h->B), dim3(NMS_THREADS), 0, (hipStream_t)stream, *h, cand_score, cand_row, cap, counts,
                       nms_thresh, topk, post_nms, out_ids, out_scores, out_boxes, out_rows, overflow);
    VD_CHECK_LAUNCH("vd_nms_topk");
    return VD_OK;
}

int64_t vd_yolo_loss_ws_bytes(const vd_head_desc* h) {
    if (!head_ok(h)) return 0;
    return (int64_t)h->B * loss_blocks(h) * 4 * (int64_t)sizeof(float);
}

static int yolo_loss_any(const vd_head_desc* h, const float* gt, int M, const float* obj_t, const float* center_t,
                         const float* scale_t, const float* weight_t, const float* class_t, float ignore_thresh,
                         int label_smooth, float* losses, void* const dhead[3], int dhead_bf16, float* box_out,
                         float* const dhead_amax[3], void* ws, int64_t ws_bytes, void* stream) {
    VD_REQUIRE(head_ok(h), "vd_yolo_loss_fwd_bwd: bad head descriptor");
    VD_REQUIRE(obj_t && center_t && scale_t && weight_t && class_t && losses && dhead && dhead[0] && dhead[1] && dhead[2],
               "vd_yolo_loss_fwd_bwd: null pointer");
    VD_REQUIRE(M >= 0 && M <= LOSS_MAX_GT && (M == 0 || gt), "vd_yolo_loss_fwd_bwd: M=%d outside [0,%d]", M, LOSS_MAX_GT);
    VD_REQUIRE(!dhead_amax || (dhead_amax[0] && dhead_amax[1] && dhead_amax[2]), "vd_yolo_loss_fwd_bwd: dhead_amax needs three slots");
    const int nb = loss_blocks(h);
    const int64_t need = (int64_t)h->B * nb * 4 * (int64_t)sizeof(float);
    if (!ws || ws_bytes < need) {
        vd_set_error("vd_yolo_loss_fwd_bwd: workspace %lld < %lld", (long long)ws_bytes, (long long)need);
        return VD_EWORKSPACE;
    }
    const int A = 3 * (5 + h->C);
    const int lds = 4 * (((A + 3) & ~3) + 32) * (int)sizeof(float);
    VD_REQUIRE(lds <= 48 * 1024, "vd_yolo_loss_fwd_bwd: too many classes for the LDS row stage");
    hipStream_t s = (hipStream_t)stream;
    // 16-B accesses when every head / gradient row starts 16-B aligned and holds a whole number of float4s
    uintptr_t al = 0;
    for (int i = 0; i < 3; ++i) al |= (uintptr_t)h->head[i] | (uintptr_t)dhead[i];
    const bool vec = (h->ldh % 4 == 0) && (al % 16 == 0);
    float* a0 = dhead_amax ? dhead_amax[0] : nullptr, *a1 = dhead_amax ? dhead_amax[1] : nullptr, *a2 = dhead_amax ? dhead_amax[2] : nullptr;
    if (dhead_bf16) {
        auto kfn = vec ? k_yolo_loss<true, __bf16> : k_yolo_loss<false, __bf16>;
        hipLaunchKernelGGL(kfn, dim3(nb, h->B), dim3(256), lds, s, *h, gt, M, obj_t, center_t, scale_t, weight_t, class_t, ignore_thresh,
                           label_smooth, (__bf16*)dhead[0], (__bf16*)dhead[1], (__bf16*)dhead[2], box_out, (float*)ws, a0, a1, a2);
    } else {
        auto kfn = vec ? k_yolo_loss<true, float> : k_yolo_loss<false, float>;
        hipLaunchKernelGGL(kfn, dim3(nb, h->B), dim3(256), lds, s, *h, gt, M, obj_t, center_t, scale_t, weight_t, class_t, ignore_thresh,
                           label_smooth, (float*)dhead[0], (float*)dhead[1], (float*)dhead[2], box_out, (float*)ws, a0, a1, a2);
    }
    VD_CHECK_LAUNCH("vd_yolo_loss_fwd_bwd");
    hipLaunchKernelGGL(k_loss_finalize, dim3((unsigned)vd_cdiv(h->B * 4, 64)), dim3(64), 0, s, (const float*)ws, nb, losses,
                       h->B);
    VD_CHECK_LAUNCH("vd_yolo_loss_fwd_bwd/finalize");
    return VD_OK;
}

int vd_yolo_loss_fwd_bwd(const vd_head_desc* h, const float* gt, int M, const float* obj_t, const float* center_t,
                         const float* scale_t, const float* weight_t, const float* class_t, float ignore_thresh,
                         int label_smooth, float* losses, float* const dhead[3], float* box_out,
                         float* const dhead_amax[3], void* ws, int64_t ws_bytes, void* stream) {
    return yolo_loss_any(h, gt, M, obj_t, center_t, scale_t, weight_t, class_t, ignore_thresh, label_smooth, losses,
                         (void* const*)dhead, 0, box_out, dhead_amax, ws, ws_bytes, stream);
}

/* bf16-storage training: the head logits (vd_head_desc.head) stay fp32, the gradient rows dhead[s] are bf16 [.., ldh] */
int vd_yolo_loss_fwd_bwd_bf16(const vd_head_desc* h, const float* gt, int M, const float* obj_t, const float* center_t,
                              const float* scale_t, const float* weight_t, const float* class_t, float ignore_thresh,
                              int label_smooth, float* losses, void* const dhead[3], float* box_out, void* ws, int64_t ws_bytes,
                              void* stream) {
    return yolo_loss_any(h, gt, M, obj_t, center_t, scale_t, weight_t, class_t, ignore_thresh, label_smooth, losses, dhead, 1,
                         box_out, nullptr, ws, ws_bytes, stream);
}

}  // extern "C"
