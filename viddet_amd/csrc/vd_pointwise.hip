// vd_pointwise.hip — HBM-bound streaming kernels around the convolutions: residual fan-in add,
// nearest-x2 upsample + channel concat (and backward), layout/normalisation of the input frames,
// temporal pooling over the K frames of a window, and the SGD-momentum update.
//
// Reference call sites (under /root/reference):
//   _upsample                      models/definitions/layers.py:11-20
//   slice_like + concat            models/definitions/yolo/yolo3.py:1170-1177
//   to_tensor / normalize          models/definitions/yolo/transforms.py:229-245 (mean/std :167-168)
//   TemporalPooling                models/definitions/layers.py:161-205
//   gluon.Trainer('sgd').step      train_yolov3.py:527-530,634
//
// All kernels move 16 B per lane (float4) with channel-contiguous NHWC addressing and a capped
// grid-stride launch (<= 4096 blocks) so the 256 CUs stay saturated without launch overhead.
#include "vd_common.h"

namespace {

inline int sblocks(int64_t n) {
    int64_t nb = vd_cdiv(n, 256);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return (int)nb;
}

#define GRID_STRIDE(i, n)                                                              \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);          \
         i += (int64_t)gridDim.x * blockDim.x)

__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n4,
                      int64_t n) {
    GRID_STRIDE(i, n4) {
        reinterpret_cast<f32x4*>(o)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
    }
    // tail (n not a multiple of 4)
    const int64_t t = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) o[t] = a[t] + b[t];
}

// bf16 tensors, fp32 sums (bf16-storage training): n8 groups of 8 elements
__global__ void k_add_bf16(const __bf16* __restrict__ a, const __bf16* __restrict__ b, __bf16* __restrict__ o, int64_t n8) {
    GRID_STRIDE(i, n8) { vd_st8(o, i, vd_ld8(a, i) + vd_ld8(b, i)); }
}

// max-abs of x[0..n) into the tensor's amax sub-slots (vd_common.h); float4 body + scalar tail
__device__ __forceinline__ float amax_range(const float* __restrict__ x, int64_t n, int64_t first, int64_t stride) {
    float m = 0.f;
    const int64_t n4 = ((uintptr_t)x % 16 == 0) ? n / 4 : 0;
    for (int64_t i = first; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    for (int64_t i = n4 * 4 + first; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
    return m;
}

__global__ void k_amax(const float* __restrict__ x, int64_t n, float* __restrict__ amax) {
    vd_amax_publish(amax, amax_range(x, n, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x));
}

// blockIdx.y = segment; seg[2*s] = element offset, seg[2*s+1] = element count
__global__ void k_amax_segments(const float* __restrict__ base, const int64_t* __restrict__ seg, float* __restrict__ amax) {
    const int sgm = blockIdx.y;
    const float* x = base + seg[2 * sgm];
    vd_amax_publish(amax + (size_t)sgm * VD_AMAX_FLOATS,
                    amax_range(x, seg[2 * sgm + 1], (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x));
}

__global__ void k_amax_merge(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    const int i = threadIdx.x * VD_AMAX_STRIDE;
    out[i] = b ? fmaxf(a[i], b[i]) : a[i];
}

__global__ void k_fill(float* __restrict__ o, float v, int64_t n) {
    GRID_STRIDE(i, n) o[i] = v;
}

__global__ void k_upcat(const float* __restrict__ up, const float* __restrict__ route, float* __restrict__ out,
                        int N, int Ho, int Wo, int Cu4, int Cr4) {
    const int Ct4 = Cu4 + Cr4;
    const int64_t total = (int64_t)N * Ho * Wo * Ct4;
    const int Hu = Ho >> 1, Wu = Wo >> 1;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Ct4);
        const int64_t pix = i / Ct4;
        f32x4 v;
        if (c < Cu4) {
            const int x = (int)(pix % Wo);
            const int64_t t = pix / Wo;
            const int y = (int)(t % Ho);
            const int64_t n = t / Ho;
            v = reinterpret_cast<const f32x4*>(up)[((n * Hu + (y >> 1)) * Wu + (x >> 1)) * Cu4 + c];
        } else {
            v = reinterpret_cast<const f32x4*>(route)[pix * Cr4 + (c - Cu4)];
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

__global__ void k_upcat_bwd_up(const float* __restrict__ dout, float* __restrict__ dup, int N, int Ho, int Wo,
                               int Cu4, int Ct4) {
    const int Hu = Ho >> 1, Wu = Wo >> 1;
    const int64_t total = (int64_t)N * Hu * Wu * Cu4;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cu4);
        const int64_t pix = i / Cu4;
        const int x = (int)(pix % Wu);
        const int64_t t = pix / Wu;
        const int y = (int)(t % Hu);
        const int64_t n = t / Hu;
        const int64_t base = ((n * Ho + 2 * y) * Wo + 2 * x);
        const f32x4* d = reinterpret_cast<const f32x4*>(dout);
        f32x4 s = d[base * Ct4 + c];
        s += d[(base + 1) * Ct4 + c];
        s += d[(base + Wo) * Ct4 + c];
        s += d[(base + Wo + 1) * Ct4 + c];
        reinterpret_cast<f32x4*>(dup)[i] = s;
    }
}

// the same on bf16 tensors (the four gradients are summed in fp32 and rounded once); C*8 = channel groups of 8
__global__ void k_upcat_bwd_up_bf16(const __bf16* __restrict__ dout, __bf16* __restrict__ dup, int N, int Ho, int Wo,
                                    int Cu8, int Ct8) {
    const int Hu = Ho >> 1, Wu = Wo >> 1;
    const int64_t total = (int64_t)N * Hu * Wu * Cu8;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cu8);
        const int64_t pix = i / Cu8;
        const int x = (int)(pix % Wu);
        const int64_t t = pix / Wu;
        const int y = (int)(t % Hu);
        const int64_t n = t / Hu;
        const int64_t base = ((n * Ho + 2 * y) * Wo + 2 * x);
        f32x8 s = vd_ld8(dout, base * Ct8 + c);
        s += vd_ld8(dout, (base + 1) * Ct8 + c);
        s += vd_ld8(dout, (base + Wo) * Ct8 + c);
        s += vd_ld8(dout, (base + Wo + 1) * Ct8 + c);
        vd_st8(dup, i, s);
    }
}

__global__ void k_upcat_bwd_route(const float* __restrict__ dout, float* __restrict__ droute, int64_t npix,
                                  int Cu4, int Cr4) {
    const int Ct4 = Cu4 + Cr4;
    const int64_t total = npix * Cr4;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cr4);
        const int64_t pix = i / Cr4;
        reinterpret_cast<f32x4*>(droute)[i] = reinterpret_cast<const f32x4*>(dout)[pix * Ct4 + Cu4 + c];
    }
}

__global__ void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int N, int C, int HW) {
    const int64_t total = (int64_t)N * HW * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int64_t t = i / C;
        const int hw = (int)(t % HW);
        const int64_t n = t / HW;
        out[i] = in[(n * C + c) * HW + hw];
    }
}

__global__ void k_preprocess_u8(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    // mean/std of transforms.py:167-168; x/255 first (to_tensor), then (x-mean)/std (normalize)
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    GRID_STRIDE(i, n) {
        const int c = (int)(i % 3);
        out[i] = ((float)in[i] / 255.0f - mean[c]) / stdv[c];
    }
}

// the same arithmetic, written as the planar [N,3,H,W] batch the stem kernel reads (vd_stem.hip): one thread = one pixel
__global__ void k_preprocess_u8_nchw(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t npix, int64_t hw) {
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    GRID_STRIDE(i, npix) {
        const int64_t n = i / hw, r = i - n * hw;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(n * 3 + c) * hw + r] = ((float)in[i * 3 + c] / 255.0f - mean[c]) / stdv[c];
    }
}

__global__ void k_tpool(const float* __restrict__ x, float* __restrict__ y, int32_t* __restrict__ arg, int B,
                        int K, int64_t inner, int type) {
    const int64_t total = (int64_t)B * inner;
    GRID_STRIDE(i, total) {
        const int64_t b = i / inner, r = i % inner;
        const float* src = x + b * K * inner + r;
        float v = src[0];
        int am = 0;
        if (type == 0) {
            for (int k = 1; k < K; ++k) {
                const float t = src[(int64_t)k * inner];
                if (t > v) { v = t; am = k; }
            }
        } else {
            for (int k = 1; k < K; ++k) v += src[(int64_t)k * inner];
            v /= (float)K;
        }
        y[i] = v;
        if (arg) arg[i] = am;
    }
}

// bf16 tensors (bf16 inference of the k > 1 networks): 8 elements per thread, the max / mean in fp32
__global__ void k_tpool_bf16(const __bf16* __restrict__ x, __bf16* __restrict__ y, int B, int K, int64_t inner8, int type) {
    const int64_t total = (int64_t)B * inner8;
    GRID_STRIDE(i, total) {
        const int64_t b = i / inner8, r = i % inner8;
        f32x8 v = vd_ld8(x, b * K * inner8 + r);
        for (int k = 1; k < K; ++k) {
            const f32x8 t = vd_ld8(x, (b * K + k) * inner8 + r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = type == 0 ? fmaxf(v[e], t[e]) : v[e] + t[e];
        }
        if (type != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] /= (float)K;
        }
        vd_st8(y, i, v);
    }
}

__global__ void k_tpool_bwd(const float* __restrict__ dy, const int32_t* __restrict__ arg, float* __restrict__ dx,
                            int B, int K, int64_t inner, int type) {
    const int64_t total = (int64_t)B * K * inner;
    GRID_STRIDE(i, total) {
        const int64_t r = i % inner;
        const int64_t t = i / inner;
        const int k = (int)(t % K);
        const int64_t b = t / K;
        const float g = dy[b * inner + r];
        dx[i] = (type == 0) ? (arg[b * inner + r] == k ? g : 0.f) : g / (float)K;
    }
}

// 'cat' join of the K frames (yolo3.py:1108,1136 F.reshape(x,(0,-3,-2)): (B,K,C,h,w) -> (B,K*C,h,w)):
// NHWC folded [B*K, hw, C] -> [B, hw, K*C] with channel index k*C + c; 16-byte units; fwd = gather, bwd = scatter
// frames [k0, k0 + kc) of every K-frame window: forward y[b][j] = x[b][k0 + j]; backward (the gradient of that) fills the
// whole K-frame tensor: dx[b][k] = dy[b][k - k0] inside the range, 0 outside
__global__ void k_frame_slice(const float* __restrict__ x, float* __restrict__ y, int K, int k0, int kc, int64_t inner4, int bwd,
                              int64_t total) {
    GRID_STRIDE(i, total) {
        const int64_t r = i % inner4;
        int64_t t = i / inner4;
        if (!bwd) {                              // i runs over y [B][kc][inner4]
            const int j = (int)(t % kc);
            const int64_t b = t / kc;
            reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(x)[((b * K) + k0 + j) * inner4 + r];
        } else {                                 // i runs over y = dx [B][K][inner4]; x = dy [B][kc][inner4]
            const int k = (int)(t % K);
            const int64_t b = t / K;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k >= k0 && k < k0 + kc) v = reinterpret_cast<const f32x4*>(x)[((b * kc) + k - k0) * inner4 + r];
            reinterpret_cast<f32x4*>(y)[i] = v;
        }
    }
}

__global__ void k_tcat(const float* __restrict__ x, float* __restrict__ y, int B, int K, int64_t hw, int C4, int bwd) {
    const int64_t total = (int64_t)B * K * hw * C4;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int64_t px = t % hw;
        t /= hw;
        const int k = (int)(t % K);
        const int64_t b = t / K;
        const int64_t j = ((b * hw + px) * K + k) * C4 + c;       // position in the stacked tensor
        if (bwd) reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(x)[j];
        else reinterpret_cast<f32x4*>(y)[j] = reinterpret_cast<const f32x4*>(x)[i];
    }
}

__global__ void k_sgd(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, int64_t n4,
                      int64_t n, float lr, float mom, float wd, float rescale) {
    GRID_STRIDE(i, n4) {
        f32x4 wv = reinterpret_cast<f32x4*>(w)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // mom = momentum*mom - lr*(rescale*grad + wd*w) ; w += mom   (mxnet sgd_mom_update)
            mv[e] = mom * mv[e] - lr * (rescale * gv[e] + wd * wv[e]);
            wv[e] += mv[e];
        }
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(w)[i] = wv;
    }
    const int64_t t = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float mv = mom * m[t] - lr * (rescale * g[t] + wd * w[t]);
        m[t] = mv;
        w[t] += mv;
    }
}

}  // namespace

extern "C" {

int vd_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    VD_REQUIRE(a && b && out && n > 0, "vd_add: bad args");
    hipLaunchKernelGGL(k_add, dim3(sblocks(n / 4 + 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4, n);
    VD_CHECK_LAUNCH("vd_add");
    return VD_OK;
}

int vd_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    VD_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "vd_add_bf16: bad args (n must be a multiple of 8)");
    hipLaunchKernelGGL(k_add_bf16, dim3(sblocks(n / 8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, (const __bf16*)b,
                       (__bf16*)out, n / 8);
    VD_CHECK_LAUNCH("vd_add_bf16");
    return VD_OK;
}

int vd_fill(float* out, float v, int64_t n, void* stream) {
    VD_REQUIRE(out && n > 0, "vd_fill: bad args");
    hipLaunchKernelGGL(k_fill, dim3(sblocks(n)), dim3(256), 0, (hipStream_t)stream, out, v, n);
    VD_CHECK_LAUNCH("vd_fill");
    return VD_OK;
}

int vd_amax(const float* x, int64_t n, float* amax, void* stream) {
    VD_REQUIRE(x && amax && n > 0, "vd_amax: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(amax, 0, sizeof(float) * VD_AMAX_FLOATS, s) != hipSuccess) {
        vd_set_error("vd_amax: memset failed");
        return VD_ELAUNCH;
    }
    int nb = sblocks(n / 4 + 1);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_amax, dim3(nb), dim3(256), 0, s, x, n, amax);
    VD_CHECK_LAUNCH("vd_amax");
    return VD_OK;
}

int vd_amax_segments(const float* base, const int64_t* seg, int nseg, float* amax, void* stream) {
    VD_REQUIRE(base && seg && amax && nseg > 0 && nseg < 65536, "vd_amax_segments: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(amax, 0, sizeof(float) * VD_AMAX_FLOATS * (size_t)nseg, s) != hipSuccess) {
        vd_set_error("vd_amax_segments: memset failed");
        return VD_ELAUNCH;
    }
    hipLaunchKernelGGL(k_amax_segments, dim3(32, nseg), dim3(256), 0, s, base, seg, amax);
    VD_CHECK_LAUNCH("vd_amax_segments");
    return VD_OK;
}

int vd_amax_merge(const float* a, const float* b, float* out, void* stream) {
    VD_REQUIRE(a && out, "vd_amax_merge: bad args");
    hipLaunchKernelGGL(k_amax_merge, dim3(1), dim3(VD_AMAX_SLOTS), 0, (hipStream_t)stream, a, b, out);
    VD_CHECK_LAUNCH("vd_amax_merge");
    return VD_OK;
}

int vd_upsample2x_concat(const float* up, const float* route, float* out, int N, int Ho, int Wo, int Cu, int Cr,
                         void* stream) {
    VD_REQUIRE(up && route && out && N > 0 && Ho % 2 == 0 && Wo % 2 == 0 && Cu % 4 == 0 && Cr % 4 == 0,
               "vd_upsample2x_concat: bad args (Ho=%d Wo=%d Cu=%d Cr=%d)", Ho, Wo, Cu, Cr);
    const int64_t total = (int64_t)N * Ho * Wo * ((Cu + Cr) / 4);
    hipLaunchKernelGGL(k_upcat, dim3(sblocks(total)), dim3(256), 0, (hipStream_t)stream, up, route, out, N, Ho, Wo,
                       Cu / 4, Cr / 4);
    VD_CHECK_LAUNCH("vd_upsample2x_concat");
    return VD_OK;
}

int vd_upsample2x_concat_bwd(const float* dout, float* dup, float* droute, int N, int Ho, int Wo, int Cu, int Cr,
                             void* stream) {
    // dup / droute may be NULL: that half is not needed (no trainable parameter upstream of it, grad_req 'null')
    VD_REQUIRE(dout && (dup || droute) && N > 0 && Ho % 2 == 0 && Wo % 2 == 0 && Cu % 4 == 0 && Cr % 4 == 0,
               "vd_upsample2x_concat_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (dup) {
        const int64_t t1 = (int64_t)N * (Ho / 2) * (Wo / 2) * (Cu / 4);
        hipLaunchKernelGGL(k_upcat_bwd_up, dim3(sblocks(t1)), dim3(256), 0, s, dout, dup, N, Ho, Wo, Cu / 4, (Cu + Cr) / 4);
        VD_CHECK_LAUNCH("vd_upsample2x_concat_bwd/up");
    }
    if (droute) {
        const int64_t npix = (int64_t)N * Ho * Wo;
        hipLaunchKernelGGL(k_upcat_bwd_route, dim3(sblocks(npix * (Cr / 4))), dim3(256), 0, s, dout, droute, npix, Cu / 4,
                           Cr / 4);
        VD_CHECK_LAUNCH("vd_upsample2x_concat_bwd/route");
    }
    return VD_OK;
}

/* bf16 tensors.  The forward concat and the route half of the backward are copies: run them through vd_upsample2x_concat /
 * this entry's route launch with the channel counts halved (two bf16 = one 4-byte word); the up half sums four gradients. */
int vd_upsample2x_concat_bwd_bf16(const void* dout, void* dup, void* droute, int N, int Ho, int Wo, int Cu, int Cr, void* stream) {
    VD_REQUIRE(dout && (dup || droute) && N > 0 && Ho % 2 == 0 && Wo % 2 == 0 && Cu % 8 == 0 && Cr % 8 == 0,
               "vd_upsample2x_concat_bwd_bf16: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (dup) {
        const int64_t t1 = (int64_t)N * (Ho / 2) * (Wo / 2) * (Cu / 8);
        hipLaunchKernelGGL(k_upcat_bwd_up_bf16, dim3(sblocks(t1)), dim3(256), 0, s, (const __bf16*)dout, (__bf16*)dup, N, Ho, Wo, Cu / 8,
                           (Cu + Cr) / 8);
        VD_CHECK_LAUNCH("vd_upsample2x_concat_bwd_bf16/up");
    }
    if (droute) {
        const int64_t npix = (int64_t)N * Ho * Wo;
        hipLaunchKernelGGL(k_upcat_bwd_route, dim3(sblocks(npix * (Cr / 8))), dim3(256), 0, s, (const float*)dout, (float*)droute, npix,
                           Cu / 8, Cr / 8);
        VD_CHECK_LAUNCH("vd_upsample2x_concat_bwd_bf16/route");
    }
    return VD_OK;
}

int vd_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, void* stream) {
    VD_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0, "vd_nchw_to_nhwc: bad args");
    const int64_t total = (int64_t)N * C * H * W;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(sblocks(total)), dim3(256), 0, (hipStream_t)stream, in, out, N, C, H * W);
    VD_CHECK_LAUNCH("vd_nchw_to_nhwc");
    return VD_OK;
}

int vd_preprocess_u8_nhwc(const uint8_t* in, float* out, int64_t npix, void* stream) {
    VD_REQUIRE(in && out && npix > 0, "vd_preprocess_u8_nhwc: bad args");
    hipLaunchKernelGGL(k_preprocess_u8, dim3(sblocks(npix * 3)), dim3(256), 0, (hipStream_t)stream, in, out, npix * 3);
    VD_CHECK_LAUNCH("vd_preprocess_u8_nhwc");
    return VD_OK;
}

int vd_preprocess_u8_nchw(const uint8_t* in, float* out, int N, int H, int W, void* stream) {
    VD_REQUIRE(in && out && N > 0 && H > 0 && W > 0, "vd_preprocess_u8_nchw: bad args");
    const int64_t hw = (int64_t)H * W;
    hipLaunchKernelGGL(k_preprocess_u8_nchw, dim3(sblocks(N * hw)), dim3(256), 0, (hipStream_t)stream, in, out, N * hw, hw);
    VD_CHECK_LAUNCH("vd_preprocess_u8_nchw");
    return VD_OK;
}

int vd_temporal_pool(const float* x, float* y, int32_t* argmax, int B, int K, int64_t inner, int type, void* stream) {
    VD_REQUIRE(x && y && B > 0 && K > 0 && inner > 0 && (type == 0 || type == 1), "vd_temporal_pool: bad args");
    hipLaunchKernelGGL(k_tpool, dim3(sblocks((int64_t)B * inner)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, B, K,
                       inner, type);
    VD_CHECK_LAUNCH("vd_temporal_pool");
    return VD_OK;
}

int vd_temporal_pool_bf16(const void* x, void* y, int B, int K, int64_t inner, int type, void* stream) {
    VD_REQUIRE(x && y && B > 0 && K > 0 && inner > 0 && inner % 8 == 0 && (type == 0 || type == 1), "vd_temporal_pool_bf16: bad args");
    hipLaunchKernelGGL(k_tpool_bf16, dim3(sblocks((int64_t)B * (inner / 8))), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x,
                       (__bf16*)y, B, K, inner / 8, type);
    VD_CHECK_LAUNCH("vd_temporal_pool_bf16");
    return VD_OK;
}

int vd_temporal_pool_bwd(const float* dy, const int32_t* argmax, float* dx, int B, int K, int64_t inner, int type,
                         void* stream) {
    VD_REQUIRE(dy && dx && B > 0 && K > 0 && inner > 0 && (type == 1 || argmax), "vd_temporal_pool_bwd: bad args");
    hipLaunchKernelGGL(k_tpool_bwd, dim3(sblocks((int64_t)B * K * inner)), dim3(256), 0, (hipStream_t)stream, dy, argmax,
                       dx, B, K, inner, type);
    VD_CHECK_LAUNCH("vd_temporal_pool_bwd");
    return VD_OK;
}

int vd_frame_slice(const float* x, float* y, int B, int K, int k0, int kc, int64_t inner, int backward, void* stream) {
    VD_REQUIRE(x && y && B > 0 && K > 0 && k0 >= 0 && kc > 0 && k0 + kc <= K && inner > 0 && inner % 4 == 0,
               "vd_frame_slice: bad args (K=%d k0=%d kc=%d)", K, k0, kc);
    const int64_t total = (int64_t)B * (backward ? K : kc) * (inner / 4);
    hipLaunchKernelGGL(k_frame_slice, dim3(sblocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, K, k0, kc, inner / 4,
                       backward, total);
    VD_CHECK_LAUNCH("vd_frame_slice");
    return VD_OK;
}

int vd_temporal_cat(const float* x, float* y, int B, int K, int64_t hw, int C, int backward, void* stream) {
    VD_REQUIRE(x && y && B > 0 && K > 0 && hw > 0 && C > 0 && C % 4 == 0, "vd_temporal_cat: bad args");
    hipLaunchKernelGGL(k_tcat, dim3(sblocks((int64_t)B * K * hw * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, B, K, hw,
                       C / 4, backward);
    VD_CHECK_LAUNCH("vd_temporal_cat");
    return VD_OK;
}

int vd_sgd_momentum(float* w, const float* grad, float* mom, int64_t n, float lr, float momentum, float wd,
                    float rescale, void* stream) {
    VD_REQUIRE(w && grad && mom && n > 0, "vd_sgd_momentum: bad args");
    hipLaunchKernelGGL(k_sgd, dim3(sblocks(n / 4 + 4)), dim3(256), 0, (hipStream_t)stream, w, grad, mom, n / 4, n, lr,
                       momentum, wd, rescale);
    VD_CHECK_LAUNCH("vd_sgd_momentum");
    return VD_OK;
}

}  // extern "C"
