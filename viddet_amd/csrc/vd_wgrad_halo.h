// vd_wgrad_halo.h - internal interface between vd_conv_wgrad's dispatch (vd_conv.hip) and the halo-ring weight-gradient
// kernel (vd_wgrad_halo.hip).  Not part of the public C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/viddet_hip.h"

// VD_WGRAD_HALO set and the launch is one the halo kernel serves (3x3 / stride 1 / 'same', Co >= 128, Ci % 32 == 0,
// VD_MATH_F16X2 or VD_STORE_BF16 operands, the ring fits LDS)
bool vd_wgrad_halo_ok(const vd_wgrad_desc& d);
// pixel ranges (= workspace slabs of Co * 9 * Ci floats) the launch will use; d.splits > 0 overrides
int vd_wgrad_halo_splits(const vd_wgrad_desc& d);
// dst: the slab array [splits][Co][9 * Ci] (or dwp itself when splits == 1)
void vd_wgrad_halo_launch(const vd_wgrad_desc& d, float* dst, int splits, hipStream_t s);
