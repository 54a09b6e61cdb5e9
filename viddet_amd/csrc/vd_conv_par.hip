// vd_conv_par.hip - the parity-fused instantiations of k_conv_igemm (vd_conv_igemm.h, template parameter PAR): the data
// gradient of a 3x3 / stride-2 / pad-1 convolution as ONE launch (VD_CONV_PARITY4, include/viddet_hip.h), and the packer of
// its weight image.  Reference arithmetic replaced: autograd's backward of the five stride-2 nn.Conv2D of Darknet-53
// (models/definitions/darknet/three_darknet.py:182-183; train_yolov3.py:631).
#include "vd_conv_igemm.h"

namespace {

template <int WM, int WN, int TM, int TN, bool M16>
int launch_par(const vd_conv_desc& d, hipStream_t s) {
    if (d.bs_part) return launch_igemm_bs<WM, WN, TM, TN, false, true, M16, true, 2, false, false, true>(d, s);
    return launch_igemm_bs<WM, WN, TM, TN, false, true, M16, false, 2, false, false, true>(d, s);
}

// (class, offset) -> kernel tap along one axis: parity 0 reaches k = 1 at offset 0; parity 1 reaches k = 2 at offset 0 and
// k = 0 at offset 1 (pad 1, stride 2: input index = (2 q + parity + 1 - k) / 2 = q + offset)
__device__ __host__ inline int par_tap(int parity, int off) { return parity == 0 ? (off == 0 ? 1 : -1) : (off == 0 ? 2 : 0); }

__global__ void k_pack_dgrad_s2(const float* __restrict__ w, float* __restrict__ wp4, int Co, int Co_pad, int Ci) {
    // wp4[(c * Ci + ci)][o * Co_pad + co] = w[co][(ky * 3 + kx) * Ci + ci]   (w fwd-packed [Co][9 * Ci])
    const int64_t total = (int64_t)4 * Ci * 4 * Co_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co_pad);
        int64_t r = i / Co_pad;
        const int o = (int)(r % 4);
        r /= 4;
        const int ci = (int)(r % Ci), c = (int)(r / Ci);
        const int ky = par_tap(c >> 1, o >> 1), kx = par_tap(c & 1, o & 1);
        float v = 0.f;
        if (co < Co && ky >= 0 && kx >= 0) v = w[(int64_t)co * 9 * Ci + (ky * 3 + kx) * Ci + ci];
        wp4[i] = v;
    }
}

}  // namespace

// tile numbering of dispatch_igemm_split (vd_conv.hip); the instantiated ones: 256 x 128 and 128 x 128 on both MFMA shapes
int vd_igemm_par_dispatch(const vd_conv_desc& d, int tile, hipStream_t s) {
    switch (tile) {
        case 1: return launch_par<4, 2, 2, 2, false>(d, s);
        case 5: return launch_par<4, 2, 2, 2, true>(d, s);
        case 2: return launch_par<4, 2, 1, 2, false>(d, s);
        case 6: return launch_par<4, 2, 1, 2, true>(d, s);
        case 11: return launch_par<2, 2, 2, 2, false>(d, s);
        default: return launch_par<2, 2, 2, 2, true>(d, s);       // 12
    }
}
int vd_igemm_par_tile(int tile) { return (tile == 1 || tile == 5 || tile == 2 || tile == 6 || tile == 11 || tile == 12) ? tile : 5; }
int vd_igemm_par_bm(int tile) { return (tile == 1 || tile == 5) ? 256 : 128; }

extern "C" int vd_pack_weight_dgrad_s2(const float* wp_fwd, float* wp4, int Co, int Co_pad, int Ci, int32_t* par_mask, void* stream) {
    VD_REQUIRE(wp_fwd && wp4 && Co > 0 && Co_pad >= Co && Ci > 0, "vd_pack_weight_dgrad_s2: bad args");
    const int64_t total = (int64_t)16 * Ci * Co_pad;
    const int nb = (int)(vd_cdiv(total, 256) < 4096 ? vd_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(k_pack_dgrad_s2, dim3(nb), dim3(256), 0, (hipStream_t)stream, wp_fwd, wp4, Co, Co_pad, Ci);
    VD_CHECK_LAUNCH("vd_pack_weight_dgrad_s2");
    if (par_mask) {
        int m = 0;
        for (int o = 0; o < 4; ++o)
            for (int c = 0; c < 4; ++c)
                if (par_tap(c >> 1, o >> 1) >= 0 && par_tap(c & 1, o & 1) >= 0) m |= 1 << (4 * o + c);
        *par_mask = m;
    }
    return VD_OK;
}
