// vd_conv_sk.hip - the persistent stream-K instantiations of k_conv_igemm (vd_conv_igemm.h, template parameter SK) for the
// fp16-split tiles, and their dispatch.  Reference arithmetic replaced: the same nn.Conv2D / autograd call sites as
// vd_conv.hip (models/definitions/layers.py:66-67, train_yolov3.py:631); this file only changes HOW a launch is cut into
// workgroups, never what it computes (bit-identical outputs).
#include "vd_conv_igemm.h"

namespace {

template <int WM, int WN, int TM, int TN, bool M16>
int launch_sk(const vd_conv_desc& d, hipStream_t s) {
    constexpr bool HT_ = WM * WN == 8 && WN * TN * 32 >= 64;      // tiles the halo loop is instantiated for
    if (HT_ && halo_ok(d, WM * TM * 32, WN * TN * 32)) {
        if (d.bs_part) return launch_igemm_bs<WM, WN, TM, TN, false, true, M16, true, 2, HT_, true>(d, s);
        return launch_igemm_bs<WM, WN, TM, TN, false, true, M16, false, 2, HT_, true>(d, s);
    }
    if (d.bs_part) return launch_igemm_bs<WM, WN, TM, TN, false, true, M16, true, 2, false, true>(d, s);
    return launch_igemm_bs<WM, WN, TM, TN, false, true, M16, false, 2, false, true>(d, s);
}

}  // namespace

// tile numbering of dispatch_igemm_split (vd_conv.hip); `tile` is already resolved (1..16).  Returns 0 when the launch was
// issued as a stream-K grid, 1 when the form does not apply to this launch (the caller runs the one-tile-per-workgroup form).
int vd_igemm_sk_dispatch(const vd_conv_desc& d, int tile, hipStream_t s) {
    if (!(d.flags & VD_MATH_F16X2) || d.in_scale) return 1;
    switch (tile) {
        case 1: return launch_sk<4, 2, 2, 2, false>(d, s);
        case 2: return launch_sk<4, 2, 1, 2, false>(d, s);
        case 3: return launch_sk<4, 2, 2, 1, false>(d, s);
        case 4: return launch_sk<4, 2, 1, 1, false>(d, s);
        case 5: return launch_sk<4, 2, 2, 2, true>(d, s);
        case 6: return launch_sk<4, 2, 1, 2, true>(d, s);
        case 7: return launch_sk<4, 2, 2, 1, true>(d, s);
        case 8: return launch_sk<4, 2, 1, 1, true>(d, s);
        case 9: return launch_sk<8, 1, 1, 1, false>(d, s);
        case 10: return launch_sk<8, 1, 1, 1, true>(d, s);
        case 11: return launch_sk<2, 2, 2, 2, false>(d, s);
        case 12: return launch_sk<2, 2, 2, 2, true>(d, s);
        case 13: return launch_sk<4, 1, 2, 1, false>(d, s);
        case 14: return launch_sk<4, 1, 2, 1, true>(d, s);
        case 15: return launch_sk<4, 1, 2, 2, false>(d, s);
        case 16: return launch_sk<4, 1, 2, 2, true>(d, s);
        default: return 1;
    }
}

// would vd_igemm_sk_dispatch(d, tile) launch?  (host autotuners, tests)
int vd_igemm_sk_applies(const vd_conv_desc& d, int tile) {
    if (!(d.flags & VD_MATH_F16X2) || d.in_scale || tile < 1 || tile > 16) return 0;
    static const int WMs[17] = {0, 4, 4, 4, 4, 4, 4, 4, 4, 8, 8, 2, 2, 4, 4, 4, 4}, WNs[17] = {0, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 2, 2, 1, 1, 1, 1},
                     TMs[17] = {0, 2, 1, 2, 1, 2, 1, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2}, TNs[17] = {0, 2, 2, 1, 1, 2, 2, 1, 1, 1, 1, 2, 2, 1, 1, 2, 2};
    const int BM = WMs[tile] * TMs[tile] * 32, BN = WNs[tile] * TNs[tile] * 32, waves = WMs[tile] * WNs[tile];
    if (streamk_grid(d, BM, BN, waves == 4 ? 2 : 1) == 0) return 0;
    if (waves == 8 && BN >= 64 && halo_ok(d, BM, BN)) return ((halo_lds_bytes(BM, BN, d.Wi) + 15) & ~15ll) + 16 <= 160 * 1024;
    return 1;
}
