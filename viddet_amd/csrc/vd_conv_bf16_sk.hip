// vd_conv_bf16_sk.hip - the persistent stream-K instantiations of k_conv_igemm_bf16 (vd_conv_igemm_bf16.h, template
// parameter SK): bf16 inference (BASELINE configs[1]) and the forward / data-gradient convs of bf16-storage training
// (configs[4]).  Same reference call sites as vd_conv_bf16.hip (nn.Conv2D, models/definitions/layers.py:66-67); this file
// changes how a launch is cut into workgroups, not what it computes (stream-K: bit-identical outputs; the split-K form of
// launches with too few tiles, VD_CONV_SPLITK: the same sums in another, fixed, association).
#include "vd_conv_igemm_bf16.h"

namespace {

template <int WM, int WN, int TM, int TN>
int launch_sk_b(const vd_conv_desc& d, hipStream_t s, bool q) {
    if constexpr (TM * TN <= 4) {
        if (d.bs_part) {
            if constexpr (WM * WN == 8) {
                if (halo_ok_b(d, WM * TM * 32, WN * TN * 32)) return launch_b2<WM, WN, TM, TN, false, false, true, true, true>(d, s, q);
            }
            return launch_b2<WM, WN, TM, TN, false, false, false, true, true>(d, s, q);
        }
        if constexpr (WM * WN == 8) {
            if (halo_ok_b(d, WM * TM * 32, WN * TN * 32)) return launch_b2<WM, WN, TM, TN, false, false, true, false, true>(d, s, q);
        }
    }
    if (d.bs_part) return 1;
    return launch_b2<WM, WN, TM, TN, false, false, false, false, true>(d, s, q);
}

}  // namespace

// tile numbering of dispatch_b (vd_conv_bf16.hip), bf16 outputs, Ci a multiple of 64.  0 = launched (or, with query_only,
// would launch) as a stream-K grid; 1 = the form does not apply.
int vd_igemm_bf16_sk_dispatch(const vd_conv_desc& d, int tile, hipStream_t s, bool query_only) {
    if (d.Ci % 64 || d.in_scale) return 1;
    switch (tile) {
        case 1: return launch_sk_b<2, 2, 2, 2>(d, s, query_only);
        case 2: return launch_sk_b<4, 2, 1, 2>(d, s, query_only);
        case 3: return launch_sk_b<2, 4, 2, 1>(d, s, query_only);
        case 4: return launch_sk_b<2, 2, 1, 2>(d, s, query_only);      // 64 x 128: the split-K form's small partials (batch-1 detection)
        case 5: return launch_sk_b<2, 4, 1, 1>(d, s, query_only);
        case 6: return launch_sk_b<4, 2, 2, 2>(d, s, query_only);
        case 7: return launch_sk_b<2, 4, 2, 2>(d, s, query_only);
        case 8: return launch_sk_b<2, 4, 4, 2>(d, s, query_only);
        case 9: return launch_sk_b<4, 2, 2, 4>(d, s, query_only);
        case 10: return launch_sk_b<4, 2, 2, 1>(d, s, query_only);
        default: return 1;
    }
}
