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
// developer timing probe (tools/nms_probe.sh): -DVD_NMS_PROBE=n ends k_nms after phase n (results are garbage)
#ifndef VD_NMS_PROBE
#define VD_NMS_PROBE 0
#endif
// (the exit stores something that depends on the phase's LDS results, or the compiler removes the phase)
#define NMS_PROBE_EXIT(n, expr) do { if (VD_NMS_PROBE == (n)) { out_rows[(int64_t)blockIdx.x * post_nms + (threadIdx.x % post_nms)] = (int)(expr); return; } } while (0)

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
    __shared__ unsigned int wtot[NMS_THREADS / 64];
    __shared__ unsigned int s_prefix_hi, s_prefix_lo, s_kth, s_cnt, s_rows_done;
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
        if (tid == 0) { s_prefix_hi = 0; s_prefix_lo = 0; s_kth = (unsigned)topk; s_rows_done = 0; }
        // The candidates' score bits are read ONCE, NLOC per thread, all requests in flight together (lists of up to
        // NLOC x 1024 = 40960 candidates: 37 k per image at batch 32 / 608x608 with the calibrated 2 % pass rate); the
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
        // Digits of 12, 12 and 8 bits, most significant first: three sweeps for the score, then - ONLY when the rows tied at
        // the threshold score are not all taken - three for the row.  (Eight 8-bit digits were 76 us of this kernel's 153:
        // scores of one image share their top byte but for four or five values, so the first sweep's 37 k LDS atomics fell on
        // as many addresses and ran one lane at a time; twelve bits reach three mantissa bits - ~50 bins - and the second
        // digit is spread over all 4096.  The histogram borrows the IoU table's LDS, unused until the select is over.)
        unsigned* hist = reinterpret_cast<unsigned*>(smask);
        for (int pass = 0; pass < 6; ++pass) {
            const int sub = pass < 3 ? pass : pass - 3;
            const int sh = sub == 0 ? 20 : (sub == 1 ? 8 : 0), wd = sub == 2 ? 8 : 12;
            const int nb = 1 << wd;
            for (int i = tid; i < nb; i += NMS_THREADS) hist[i] = 0;
            __syncthreads();
            if (s_rows_done) break;                     // (uniform: written two barriers ago)
            const unsigned phi = s_prefix_hi, plo = s_prefix_lo;
            auto digit = [&](int i, unsigned k) {
                if (pass < 3) {                         // score digits: prefix = the digits above this one
                    if (pass == 0 || (k >> (sh + wd)) == phi) atomicAdd(&hist[(k >> sh) & (unsigned)(nb - 1)], 1u);
                } else if (k == phi) {                  // row digits, only among rows tied at the threshold score
                    const unsigned r = 0xFFFFFFFFu - (unsigned)cr[i];
                    if (pass == 3 || (r >> (sh + wd)) == plo) atomicAdd(&hist[(r >> sh) & (unsigned)(nb - 1)], 1u);
                }
            };
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
            // the digit of the kth-largest key: the bin d with S(d) >= kth > S(d + 1), S(d) = sum_{b >= d} hist[b] (bin 0 when
            // nothing reaches kth).  Four bins per thread (one 16-byte read), a suffix scan over the lanes, the waves' totals
            // through LDS; the one thread that holds the boundary publishes it.
            {
                const unsigned kth = s_kth;
                const int lane_ = tid & 63, wave_ = tid >> 6;
                uint4 hv = make_uint4(0u, 0u, 0u, 0u);
                if (4 * tid < nb) hv = *reinterpret_cast<const uint4*>(hist + 4 * tid);
                const unsigned t = hv.x + hv.y + hv.z + hv.w;
                unsigned suf = t;                                   // inclusive suffix sum over the lanes >= this one
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_down(suf, off);
                    if (lane_ + off < 64) suf += o;
                }
                if (lane_ == 0) wtot[wave_] = suf;
                __syncthreads();
                unsigned above = suf - t;                           // S(4 tid + 4)
                for (int w2 = wave_ + 1; w2 < NMS_THREADS / 64; ++w2) above += wtot[w2];
                const unsigned S3 = above + hv.w, S2 = S3 + hv.z, S1 = S2 + hv.y, S0 = S1 + hv.x;
                int d = -1;
                unsigned cum = 0, tied = 0;
                if (4 * tid < nb) {
                    if (S3 >= kth && above < kth) { d = 4 * tid + 3; cum = above; tied = hv.w; }
                    else if (S2 >= kth && S3 < kth) { d = 4 * tid + 2; cum = S3; tied = hv.z; }
                    else if (S1 >= kth && S2 < kth) { d = 4 * tid + 1; cum = S2; tied = hv.y; }
                    else if ((S0 >= kth || tid == 0) && S1 < kth) { d = 4 * tid; cum = S1; tied = hv.x; }
                }
                if (d >= 0) {
                    if (pass < 3) s_prefix_hi = (phi << wd) | (unsigned)d;
                    else s_prefix_lo = (plo << wd) | (unsigned)d;
                    s_kth = kth - cum;
                    // the score is complete and every row tied at it is wanted: no row digits (Tr = 0 takes them all)
                    if (pass == 2 && kth - cum == tied) s_rows_done = 1;
                }
            }
            __syncthreads();
        }
        __syncthreads();
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
    NMS_PROBE_EXIT(1, skey[tid]);
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
    NMS_PROBE_EXIT(2, skey[tid]);
    int nsel = n < topk ? n : topk;
    if (nsel > SORT_N) nsel = SORT_N;
    // re-decode the survivors' boxes from the head tensor
    const int R0 = h.g[0] * h.g[0], R1 = h.g[1] * h.g[1];
    const int base1 = h.C * 3 * R0, base2 = h.C * 3 * (R0 + R1);
    const int npred = 5 + h.C;
    for (int idx = tid; idx < TOPK_MAX * (TOPK_MAX / 64); idx += NMS_THREADS) smask[idx] = 0ull;   // (was the select's histogram)
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
    NMS_PROBE_EXIT(3, bx1[tid & (TOPK_MAX - 1)] + (float)bcls[tid & (TOPK_MAX - 1)]);
    const int nw = (nsel + 63) >> 6;
    // IoU bitmask, word w of row i = the columns j in [64 w, 64 w + 64) that row i suppresses (j > i, same class, IoU above
    // the threshold).  A lane owns ONE column of the word (its box in registers), a wave walks the rows that can have bits in
    // that word (i < 64 w + 64) and a ballot IS the word: no inner loop over columns, the row's box is a broadcast LDS read.
    // (One thread per (row, word) looping over its 64 columns was 33 us of this kernel; words below the diagonal were
    // zeroed with the boxes above.)
    {
        const int lane_ = tid & 63, wave_ = tid >> 6;
        for (int w = 0; w < nw; ++w) {
            const int j = w * 64 + lane_;
            const bool jv = j < nsel;
            const int jc = jv ? j : 0;
            const float cx1 = bx1[jc], cy1 = by1[jc], cx2 = bx2[jc], cy2 = by2[jc];
            const int ccls = bcls[jc];
            const int iend = nsel < (w + 1) * 64 ? nsel : (w + 1) * 64;
            for (int i = wave_; i < iend; i += NMS_THREADS / 64) {
                // columns of this word the row can suppress at all: behind it in the order, same class.  One workgroup is
                // one CU: the phase is bound by its VALU, so a row with no such column (most rows, with tens of classes
                // among the top-k) costs a compare and a ballot, and the test itself is inter > thresh * union - the
                // same predicate as inter / union > thresh without the division's dozen instructions
                const unsigned long long cand = __ballot(jv && j > i && ccls == bcls[i]);
                if (cand == 0ull) continue;                      // (the table was zeroed above)
                const float ax1 = bx1[i], ay1 = by1[i], ax2 = bx2[i], ay2 = by2[i];
                const float aarea = (ax2 - ax1) * (ay2 - ay1);
                const float iw = fmaxf(0.f, fminf(ax2, cx2) - fmaxf(ax1, cx1));
                const float ih = fmaxf(0.f, fminf(ay2, cy2) - fmaxf(ay1, cy1));
                const float inter = iw * ih;
                const float uni = aarea + (cx2 - cx1) * (cy2 - cy1) - inter;
                const bool bit = ((cand >> lane_) & 1ull) && uni > 0.f && inter > nms_thresh * uni;
                const unsigned long long m = __ballot(bit);
                if (lane_ == 0) smask[i * (TOPK_MAX / 64) + w] = m;
            }
        }
    }
    __syncthreads();
    NMS_PROBE_EXIT(4, smask[tid]);
    // greedy sweep by one wave, 64 rows at a time.  Inside a chunk the decisions depend on each other only through the
    // chunk's OWN word of its rows' masks: lane l holds that word of row 64 c + l and a scalar loop walks the 64 bits
    // (v_readlane with a constant lane, no LDS round trip per row); the kept rows' other words are then OR-ed into the removed set - independent
    // LDS reads, lane w owning word w.  (One dependent shuffle + LDS read per row: 38 us for 400 rows.)
    if (tid < 64) {
        unsigned long long removed = 0;
        int nkeep = 0;
        for (int c = 0; c < nw; ++c) {
            const int i0 = c * 64;
            const int cnt = nsel - i0 < 64 ? nsel - i0 : 64;
            const unsigned long long diag = tid < cnt ? smask[(i0 + tid) * (TOPK_MAX / 64) + c] : 0ull;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            const unsigned rlo = (unsigned)removed, rhi = (unsigned)(removed >> 32);
            unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)rhi, c) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)rlo, c);
            // (rows past the chunk's count start out "removed"; straight-line and branch-free: one wave issues an instruction
            // every four or five cycles and pays ~16 for a taken branch, which made the looped form 110 cycles per row)
            if (cnt < 64) cur |= ~0ull << cnt;
            unsigned long long keepbits = 0;
#pragma unroll
            for (int l = 0; l < 64; ++l) {
                const unsigned long long alive = ((cur >> l) & 1ull) ^ 1ull;
                const unsigned long long dl = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, l) << 32) |
                                              (unsigned)__builtin_amdgcn_readlane((int)dlo, l);
                keepbits |= alive << l;
                cur |= dl & (0ull - alive);
            }
            if ((keepbits >> tid) & 1ull) keep[nkeep + __popcll(keepbits & ((1ull << tid) - 1ull))] = i0 + tid;
            // (eight rows' words requested together, then masked by the rows' keep bits: one LDS round trip per eight rows
            // instead of one per kept row)
            unsigned long long acc = 0;
            const int wl = tid < nw ? tid : 0;
            for (int l0 = 0; l0 < cnt; l0 += 8) {
                unsigned long long v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {                  // (rows past the chunk's count: their keep bit is clear)
                    const int rq = i0 + l0 + q < TOPK_MAX ? i0 + l0 + q : TOPK_MAX - 1;
                    v[q] = smask[rq * (TOPK_MAX / 64) + wl];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc |= ((keepbits >> (l0 + q)) & 1ull) ? v[q] : 0ull;
            }
            removed |= acc;
            nkeep += __popcll(keepbits);
        }
        if (tid == 0) s_nkeep = nkeep;
    }
    __syncthreads();
    NMS_PROBE_EXIT(5, keep[tid & (TOPK_MAX - 1)] + s_nkeep);
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
