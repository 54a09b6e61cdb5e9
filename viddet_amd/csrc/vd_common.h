// vd_common.h — shared helpers for the gfx950 kernel library (not part of the public C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/viddet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void vd_set_error(const char* fmt, ...);

// std::conditional without <type_traits> in device code
template <bool B, typename T, typename F> struct vd_select { typedef T type; };
template <typename T, typename F> struct vd_select<false, T, F> { typedef F type; };

// ---- storage-type generic 4- / 8-element accesses of the streaming kernels (fp32 or bf16 tensors, fp32 arithmetic) ----
typedef __bf16 vd_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 vd_bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 vd_ld4(const float* p, int64_t i4) { return reinterpret_cast<const f32x4*>(p)[i4]; }
__device__ __forceinline__ f32x4 vd_ld4(const __bf16* p, int64_t i4) {
    const vd_bf16x4 v = reinterpret_cast<const vd_bf16x4*>(p)[i4];
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void vd_st4(float* p, int64_t i4, const f32x4 v) { reinterpret_cast<f32x4*>(p)[i4] = v; }
__device__ __forceinline__ void vd_st4(__bf16* p, int64_t i4, const f32x4 v) {
    vd_bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];           // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    reinterpret_cast<vd_bf16x4*>(p)[i4] = o;
}
__device__ __forceinline__ f32x8 vd_ld8(const __bf16* p, int64_t i8) {
    const vd_bf16x8 v = reinterpret_cast<const vd_bf16x8*>(p)[i8];
    f32x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
    return o;
}
__device__ __forceinline__ void vd_st8(__bf16* p, int64_t i8, const f32x8 v) {
    vd_bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    reinterpret_cast<vd_bf16x8*>(p)[i8] = o;
}

#define VD_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            vd_set_error(__VA_ARGS__);                          \
            return VD_EINVAL;                                   \
        }                                                       \
    } while (0)

#define VD_CHECK_LAUNCH(name)                                                       \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            vd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return VD_ELAUNCH;                                                      \
        }                                                                           \
    } while (0)

__host__ __device__ static inline int64_t vd_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD a contiguous run of logical tile ids (L2 reuse of the shared
// operand panel).  Placement only affects speed, never correctness.
__device__ __forceinline__ int vd_xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// accurate expf (not __expf): decoded boxes are compared with the fp64 oracle at 1e-3 px
__device__ __forceinline__ float vd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- per-tensor max-abs ("amax") exchange between producer and consumer kernels (VD_MATH_F16X2, viddet_hip.h) ----
// A tensor's amax lives in VD_AMAX_SLOTS sub-slots, VD_AMAX_STRIDE floats (256 B) apart, so that the thousands of
// workgroups of a streaming producer do not queue their atomics on one address; the tensor's amax is the maximum of
// the sub-slots.  Values are non-negative floats, so unsigned-integer max on the bit patterns is float max; the
// caller zeroes the slots before the producer runs.
__device__ __forceinline__ float vd_wave_max(float m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    return m;
}

// every lane of the wave calls with its own running max (>= 0); one atomic per wave
__device__ __forceinline__ void vd_amax_publish(float* amax, float m) {
    m = vd_wave_max(m);
    if ((threadIdx.x & 63) == 0)
        atomicMax(reinterpret_cast<unsigned int*>(amax + (size_t)(blockIdx.x & (VD_AMAX_SLOTS - 1)) * VD_AMAX_STRIDE),
                  __float_as_uint(m));
}

// wave-uniform amax of a tensor (all 64 lanes call)
__device__ __forceinline__ float vd_amax_read(const float* amax) {
    const int lane = threadIdx.x & 63;
    float v = lane < VD_AMAX_SLOTS ? amax[(size_t)lane * VD_AMAX_STRIDE] : 0.f;
    v = vd_wave_max(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// exponent e with amax * 2^e in [2^14, 2^15): the top of the fp16 range with a factor 2 of headroom below 65504.
// amax == 0 / denormal / inf / nan -> 0 (no scaling).
__device__ __forceinline__ int vd_f16_scale_exp(float amax) {
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    if (E == 0 || E == 255) return 0;
    const int e = 141 - E;                  // 14 - (E - 127)
    return e > 126 ? 126 : e;
}
