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
