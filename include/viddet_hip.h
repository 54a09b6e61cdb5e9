/*
 * viddet_hip.h — C-ABI of libviddet_hip.so, the MI355X (gfx950) kernel library behind the
 * yolo3_darknet53 train / detect hot path.
 *
 * The reference (HaydenFaulkner/VidDet) has no FFI: its hot path is a Gluon HybridBlock that
 * composes MXNet/GluonCV operators.  Each entry point below therefore names the reference
 * *operator call site* (file:line under /root/reference) whose arithmetic it replaces.  The
 * Python host (viddet_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every function returns int: 0 = ok, <0 = VD_E* ; vd_last_error() gives a thread-local text.
 *  - the library never allocates or frees device memory: every buffer (incl. workspaces) is
 *    owned by the caller; pointers are raw device addresses.
 *  - `stream` is a hipStream_t passed as void*; ordering is by stream only; calls are
 *    asynchronous and re-entrant.
 *  - ONE device per process (the design is one process per GPU): the library caches the device
 *    address of its resident zero pages and its kernels' dynamic-LDS attributes in process-wide statics the first time
 *    an entry point runs; a process that drove a second device through it would read the first device's addresses.
 *  - activations are NHWC fp32 (bf16 where stated); conv weights arrive in the reference's OIHW
 *    layout and are re-packed by vd_pack_* into the K-contiguous layouts the kernels read.
 */
#ifndef VIDDET_HIP_H
#define VIDDET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VD_OK            0
#define VD_EINVAL       -1   /* bad argument / unsupported shape */
#define VD_ELAUNCH      -2   /* hip launch error */
#define VD_EWORKSPACE   -3   /* workspace too small */

#define VD_MAX_TAPS 27

/* epilogue flags of vd_conv_igemm */
#define VD_EPI_AFFINE    1   /* v = v*scale[c] + shift[c]   (BN-eval fold, or bias with scale==NULL) */
#define VD_EPI_LEAKY     2   /* v = v>0 ? v : slope*v */
#define VD_EPI_RESIDUAL  4   /* v += residual[m][c]  (after the activation) */
/* Arithmetic of the fp32 products (vd_conv_desc.flags / vd_wgrad_desc.flags).  Default: v_mfma_f32_32x32x2_f32, an
 * exact fp32 fma chain.  VD_MATH_SPLIT: each fp32 operand is split exactly into three bf16 pieces (24 significand
 * bits) and six of the nine partial products are accumulated in fp32 on the bf16 matrix pipe; the dropped terms are
 * below 2^-23 of the product (one fp32 rounding).  Storage, accumulation and the epilogue stay fp32. */
#define VD_MATH_SPLIT    16
/* VD_MATH_BF16: products on bf16-ROUNDED operands (only the leading piece of the split: one bf16 MFMA per product
 * block), fp32 tensors / accumulation / epilogues unchanged - the mixed-precision training arithmetic of BASELINE
 * configs[4]; results are bf16-accurate (2^-8 per operand), NOT fp32-accurate.  Same kernels and tiles as VD_MATH_SPLIT. */
#define VD_MATH_BF16     32

/* VD_MATH_F16X2: each fp32 operand x is scaled by a per-tensor power of two s (exact) and split into two fp16 pieces,
 * h = fp16(x*s), l = fp16(x*s - h) (round-to-nearest): 22-23 significand bits, |x*s - h - l| <= 2^-23 |x*s|.  A product is
 * accumulated in fp32 from THREE partial products al*bh, ah*bl, ah*bh on the f16 matrix pipe (the dropped al*bl is below
 * 2^-22 of the product) - half the matrix-pipe work of VD_MATH_SPLIT - and the epilogue multiplies by the inverse powers
 * of two (exact).  The scale puts the tensor's max-abs in [2^14, 2^15): every element within 2^19 of it keeps full
 * precision, smaller ones degrade gracefully towards an absolute floor of 2^-40 of the max.  The max-abs values are
 * read from device memory (vd_conv_desc.amax_in / amax_w, vd_wgrad_desc.amax_in / amax_dout): VD_AMAX_SLOTS sub-slots,
 * VD_AMAX_STRIDE floats apart, written by the producing kernels (atomic max; the caller zeroes them before the producer
 * runs) or by vd_amax / vd_amax_segments.  Measured against fp64 the error is not above the fp32 MFMA's. */
#define VD_MATH_F16X2    64
/* with VD_MATH_F16X2: keep the generic K loop where the library would stage the activation operand as a pixel halo tile
 * (3x3 stride-1 convs and their data gradients; vd_conv.hip, HALO) - for A/B timing by the host autotuner */
#define VD_MATH_NOHALO   128
/* VD_STORE_BF16 (vd_wgrad_desc.flags): `in` and `dout` are bf16 tensors (bf16-storage training, BASELINE configs[4]): the
 * products are the one-term bf16 MFMA of VD_MATH_BF16 on operands that are ALREADY bf16, the weight gradient stays fp32.
 * The forward / data-gradient convs of that mode are vd_conv_igemm_bf16. */
#define VD_STORE_BF16    256
/* VD_WGRAD_HALO (vd_wgrad_desc.flags, with VD_MATH_F16X2 or VD_STORE_BF16): where the launch is a 3x3 / stride-1 / 'same'
 * weight gradient with Co >= 128, a workgroup owns BM output channels x (9 taps x 32 input channels) and stages the
 * activation rows ONCE per pixel in a sliding LDS ring that the nine taps read at shifted rows (vd_wgrad_halo.hip), instead
 * of gathering the activation tile once per tap.  Same arithmetic, slabs and reduction; other geometries ignore the flag. */
#define VD_WGRAD_HALO    512
/* VD_CONV_STREAMK (vd_conv_desc.flags, with VD_MATH_F16X2): run the launch as a PERSISTENT stream-K grid - one workgroup
 * per CU slot, each owning an equal contiguous run of (tile, K-unit) units instead of whole tiles, so the launch ends
 * together on every CU whatever the tile count (vd_conv_sk.hip).  A tile cut by a run boundary is finished by the next
 * workgroup FROM the first one's accumulators (handed over through sk_ws), i.e. the same MFMA chain in the same order:
 * results are bit-identical to the launch without the flag.  Needs vd_conv_desc.sk_ws / sk_ws_bytes; where the form does
 * not apply (too few tiles, workspace too small, another arithmetic) the flag is ignored.  vd_conv_igemm_streamk(d) tells. */
#define VD_CONV_STREAMK  1024
/* VD_CONV_PARITY4 (vd_conv_desc.flags, with VD_MATH_F16X2): the data gradient of a 3x3 / stride-2 / pad-1 convolution as
 * ONE launch instead of four parity launches.  `in` = the incoming gradient dz [N, Hi, Wi, Ci = Cout], Hg = Hi, Wg = Wi,
 * in_stride 1, T = 4 taps with (dy, dx) = (0,0), (0,1), (1,0), (1,1); Co = 4 * par_cin GEMM columns = (parity class c,
 * channel), `wp` packed by vd_pack_weight_dgrad_s2; out = dx [N, Ho = 2 Hi, Wo = 2 Wi, ldo >= par_cin], out_stride 2: row q,
 * class c goes to pixel (2 qy + (c >> 1), 2 qx + (c & 1)).  Epilogue: residual (accumulate) and the fused BatchNorm-backward
 * reductions (bs_*: one table row [2 par_cin] per (M tile, column tile): vd_conv_igemm_mtiles() rows, the caller zeroes
 * the table first); no affine / LeakyReLU / statistics.  par_mask: bit (4 * tap + class) set where that weight block is
 * nonzero (vd_pack_weight_dgrad_s2 returns it).  Replaces autograd's backward of nn.Conv2D(strides=2),
 * three_darknet.py:182-183. */
#define VD_CONV_PARITY4  2048
/* vd_conv_igemm_bf16 only: SPLIT-K for launches whose tiles cannot fill the chip (batch-1 detection: 24 tiles of 128 x 128
 * at 19 x 19).  Each tile is cut along K into 2..8 parts run by as many workgroups; every part publishes its raw
 * accumulators in `sk_ws` (the stream-K workspace and hand-off protocol) and the part that arrives LAST sums them in part
 * order and runs the epilogue - nobody waits.  Deterministic, but another association than the one-tile launch (stream-K
 * proper is bit-identical to it), so it is its own flag: the host sets it for inference plans only.  Ignored where the
 * launch has tiles enough, where fewer than 4 K-steps per part would be left, or with the fused backward reductions. */
#define VD_CONV_SPLITK   4096
#define VD_SK_MAX_WG        2048     /* seam counters per workspace (the last one counts the polls that gave up: diagnostics) */
#define VD_SK_HEADER_BYTES  32768    /* [VD_SK_MAX_WG] u32 hand-off counters + [VD_SK_MAX_WG] u32 consumed counts + [VD_SK_SPLIT_MAX_TILES] u32 arrivals */
#define VD_SK_SPLIT_CNT_OFF   4096   /* (u32 words) the split-K form's per-tile arrival counters: zero between launches */
#define VD_SK_SPLIT_MAX_TILES 4096
#define VD_SK_TIMEOUT_TICKS 4000     /* bound of the hand-off poll in 10 ns ticks, after which the consumer recomputes */
#define VD_AMAX_SLOTS    32
#define VD_AMAX_STRIDE   64   /* floats between sub-slots (256 B) */
#define VD_AMAX_FLOATS   (VD_AMAX_SLOTS * VD_AMAX_STRIDE)   /* floats per tensor */

const char* vd_last_error(void);
int vd_version(void);
/* The ABI's revision: bumped whenever an entry point's argument list or a descriptor's layout changes (round 2 added the
 * amax_out / dhead_amax pointers in front of ws / stream and grew both descriptors: a caller built against the older
 * header would have passed its stream handle as amax_out).  A binding checks vd_abi_version() == VD_ABI_VERSION and
 * vd_sizeof_desc(i) == sizeof(its mirror of the descriptor) at load, before the first compute call: viddet_amd/lib.py
 * does, INTEGRATION.md shows it.  i: 0 = vd_conv_desc, 1 = vd_wgrad_desc, 2 = vd_head_desc; unknown i -> -1. */
#define VD_ABI_VERSION 6
int vd_abi_version(void);
int64_t vd_sizeof_desc(int which);

/* ---------------------------------------------------------------------------------------
 * Generic tap-list implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *
 *   out[n, gy*os+oy, gx*os+ox, co] = epi( sum_{t<T} sum_{ci<Ci}
 *          in[n, gy*is+dy[t], gx*is+dx[t], ci] * wp[co][t*Ci + ci] )        (zero outside the image)
 *
 * One kernel serves: forward 1x1/3x3 s1/s2 (layers.py:66-67 nn.Conv2D inside _conv2d;
 * yolo3.py:62 prediction conv), the 3-tap / 27-tap temporal convs (layers.py:73-89), and the
 * data gradient of all of them (autograd.backward, train_yolov3.py:631) through flipped /
 * parity-split tap lists built by the host.
 * Requirements: Ci % 32 == 0; in/out 16-byte aligned; ldo >= Co (output pixel pitch in floats).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* in;        /* [N, Hi, Wi, Ci] */
    const float* wp;        /* [Co][T*Ci] packed, k contiguous */
    float*       out;       /* [N, Ho, Wo, ldo] */
    const float* scale;     /* [Co] or NULL */
    const float* shift;     /* [Co] or NULL */
    const float* residual;  /* same geometry as out, pitch ldr, or NULL */
    int32_t N, Hi, Wi, Ci;
    int32_t Hg, Wg;         /* GEMM row grid per image */
    int32_t in_stride;      /* is */
    int32_t T;
    int32_t dy[VD_MAX_TAPS], dx[VD_MAX_TAPS];
    int32_t dz[VD_MAX_TAPS];/* temporal tap offset (frames); 0 for 2-D convs */
    int32_t Kfr;            /* frames per window for temporal convs (N = windows*Kfr); 1 otherwise */
    int32_t Ho, Wo, Co;
    int32_t out_stride, out_oy, out_ox;
    int32_t ldo, ldr;
    int32_t flags;
    float   slope;
    /* in-load transform: the A operand is leaky(in*in_scale[ci]+in_shift[ci]) when in_scale!=NULL
     * (fuses the producer's BatchNorm+LeakyReLU into this conv's gather; padding stays 0) */
    const float* in_scale;
    const float* in_shift;
    float   in_slope;
    int32_t tile;           /* 0 = library heuristic; 1..8 (fp32 MFMA) / 1..16 (split arithmetics) = explicit tile variant (host autotuner) */
    /* fused BatchNorm statistics (training forward, flags must be 0): per-M-tile partial sums
     * stats_part[tile_m][0..Co) = sum, [Co..2Co) = sum of squares; vd_conv_igemm_mtiles() rows;
     * finish with vd_bn_sum_partials().  NULL = off. */
    float*  stats_part;
    /* fused BatchNorm BACKWARD reductions (training; this launch is the data-gradient conv that writes the final
     * gradient dy of a BatchNorm+LeakyReLU output, direct geometry: out_stride 1, Ho==Hg, Wo==Wg).  With
     * g = dy * leaky'(z*bs_scale+bs_shift) and xhat = (z - bs_mean)*bs_invstd, the epilogue adds the per-M-tile
     * partial sums  bs_part[tile_m][0..Co) = sum g,  [Co..2Co) = sum g*xhat  (rows as vd_conv_igemm_mtiles());
     * bs_z is that layer's pre-BN conv output, laid out like `out` (pitch ldo).  Finish with
     * vd_bn_sum_partials(); replaces vd_bn_bwd_reduce.  bs_part NULL = off. */
    const float* bs_z;
    const float* bs_scale;
    const float* bs_shift;
    const float* bs_mean;
    const float* bs_invstd;
    float*  bs_part;
    float   bs_slope;
    /* VD_MATH_F16X2: max-abs slots (VD_AMAX_FLOATS floats each) of `in` and of `wp`; required with that flag */
    const float* amax_in;
    const float* amax_w;
    /* optional, any arithmetic: the epilogue publishes the max-abs of what it stores (zeroed by the caller) */
    float*  amax_out;
    /* VD_CONV_STREAMK: hand-off workspace, vd_conv_igemm_streamk_ws_bytes() bytes.  The caller zeroes its first
     * VD_SK_HEADER_BYTES ONCE, when it allocates it (the counters are monotonic across launches), and never shares one
     * workspace between launches that may run at the same time (one per stream).  VD_CONV_SPLITK uses the same workspace
     * (its per-tile arrival counters are zero between launches). */
    void*   sk_ws;
    int64_t sk_ws_bytes;
    /* VD_CONV_PARITY4: channels per parity class, nonzero (tap, class) weight blocks */
    int32_t par_cin;
    int32_t par_mask;
} vd_conv_desc;

int vd_conv_igemm(const vd_conv_desc* d, void* stream);
/* 1 when vd_conv_igemm(d) would run as a persistent stream-K grid (flag set, arithmetic / tile / tile count it serves and
 * a large enough workspace), else 0: for host autotuners and tests. */
int vd_conv_igemm_streamk(const vd_conv_desc* d);
/* bytes of a stream-K workspace that serves every launch shape (header + one fp32 accumulator tile per CU slot) */
int64_t vd_conv_igemm_streamk_ws_bytes(void);
/* number of M tiles (rows of stats_part) the launch described by d will use */
int vd_conv_igemm_mtiles(const vd_conv_desc* d);

/* bf16 variant of vd_conv_igemm for inference (BASELINE configs[1] asks for bf16; the reference is fp32-only):
 * in / wp / residual are bf16 (NHWC with Ci % 64 == 0, or Ci == 32 unpadded: a K-step then holds two taps;
 * weights [Co][T*Ci]), accumulation and epilogue fp32,
 * out is bf16 (out_f32 = 0) or fp32 (out_f32 = 1: prediction heads, shared decode / NMS kernels).
 * Any output geometry of vd_conv_igemm (the strided / offset outputs of the stride-2 data gradients included) and, for
 * bf16-storage training, d->stats_part: the fused BatchNorm statistics of the raw outputs, taken from the fp32
 * accumulators before they are rounded to bf16 - [vd_conv_igemm_bf16_mtiles(d)][2 * Co] floats, as vd_conv_igemm writes
 * them.  d->tile: 0 = default, 1..13 = tile variant (256x256 .. 128x64; the host autotunes it). */
int vd_conv_igemm_bf16(const vd_conv_desc* d, int out_f32, void* stream);
/* VD_CONV_STREAMK in d->flags (with sk_ws / sk_ws_bytes) runs the launch as a persistent stream-K grid where that form
 * applies (bf16 outputs, Ci % 64 == 0, enough tiles: as vd_conv_igemm), VD_CONV_SPLITK as a split-K grid where THAT
 * applies (too few tiles for the chip).  Returns 1 when vd_conv_igemm_bf16(d, out_f32) would run the stream-K form, 2 when
 * it would run the split-K form, 0 when it would launch one workgroup per tile. */
int vd_conv_igemm_bf16_streamk(const vd_conv_desc* d, int out_f32);
int vd_conv_igemm_bf16_mtiles(const vd_conv_desc* d);
/* fp32 fwd-packed [>=Co][T*Ci] -> bf16 [Co_pad][T*Ci_pad] (zero padded rows / channels) */
int vd_pack_weight_bf16(const float* wp_f32, void* wp_bf16, int Co, int Co_pad, int Ci, int Ci_pad, int T,
                        void* stream);
/* stem im2col into 64 bf16 columns per pixel (27 real), from [N,H,W,3] (nchw=0) or [N,3,H,W] (nchw=1) fp32.
 * Off the product path since vd_stem_conv (direct stem kernel): kept as the explicit lowering the tests compare with. */
int vd_stem_im2col_bf16(const float* in, void* col, int N, int H, int W, int nchw, void* stream);

/* Weight gradient (autograd.backward wrt nn.Conv2D weight, train_yolov3.py:631):
 *   dwp[co][t*Ci+ci] = sum_{n,gy,gx} dout[n,gy,gx,co] * in[n, gy*is+dy[t], gx*is+dx[t], ci]
 * split over `splits` pixel ranges into workspace slabs, then reduced deterministically.
 * ws must hold vd_conv_wgrad_ws_bytes() bytes. */
typedef struct {
    const float* in;        /* [N, Hi, Wi, Ci] */
    const float* dout;      /* [N, Hg, Wg, ldd] */
    float*       dwp;       /* [Co][T*Ci] */
    int32_t N, Hi, Wi, Ci;
    int32_t Hg, Wg, Co, ldd;
    int32_t in_stride;
    int32_t T;
    int32_t dy[VD_MAX_TAPS], dx[VD_MAX_TAPS];
    int32_t dz[VD_MAX_TAPS];
    int32_t Kfr;
    int32_t splits;         /* 0 = choose */
    const float* in_scale;  /* optional in-load transform as in vd_conv_desc */
    const float* in_shift;
    float   in_slope;
    int32_t flags;          /* VD_MATH_SPLIT / VD_MATH_F16X2: split-operand products (Co >= 64; narrower layers stay on the fp32 MFMA) */
    const float* amax_in;   /* VD_MATH_F16X2: max-abs slots of `in` and of `dout` */
    const float* amax_dout;
} vd_wgrad_desc;

int64_t vd_conv_wgrad_ws_bytes(const vd_wgrad_desc* d);
int vd_conv_wgrad(const vd_wgrad_desc* d, void* ws, int64_t ws_bytes, void* stream);
/* 1 when vd_conv_wgrad(d) would run the halo-ring kernel (VD_WGRAD_HALO set and the geometry / arithmetic is one it
 * serves), else 0: for host autotuners (no point timing the flag where it is ignored) and tests. */
int vd_conv_wgrad_uses_halo(const vd_wgrad_desc* d);

/* Direct stem convolution (3x3, 3 -> 32 channels, stride 1, pad 1; three_darknet.py:163-164) straight from the NCHW
 * fp32 frame batch (transforms.py:239-245): no im2col round trip.  wp is the fwd-packed stem weight [32][32]
 * (k = (ky*3+kx)*3 + c, k >= 27 zero).  out: NHWC [N,H,W,ldo >= 32] (32 channels written per pixel; a wider pitch
 * leaves the pad channels alone) fp32, or bf16 when out_bf16; flags = VD_EPI_AFFINE /
 * VD_EPI_LEAKY (BN-eval fold + LeakyReLU); stats_part (flags 0, fp32 out): per-block BatchNorm partial sums
 * [vd_stem_conv_blocks()][2*32], finish with vd_bn_sum_partials(). */
/* bf16 inference: the stem and the stride-2 conv behind it (three_darknet.py:163-164 + :182-183) in ONE launch of the
 * first-stage patch kernel (vd_conv_c32_bf16.hip): `d` describes the 3x3 / stride-2 / 32 -> 64 channel conv as for
 * vd_conv_igemm_bf16 (N, Hi = H, Wi = W of the frames; folded BatchNorm + LeakyReLU epilogue, bf16 output; d->in is not
 * read), the stem's 32-channel map is computed inside the kernel from the fp32 NCHW frames and never stored.  Outputs are
 * bit-identical to vd_stem_conv(..., VD_EPI_AFFINE | VD_EPI_LEAKY, out_bf16 = 1) followed by vd_conv_igemm_bf16 with tile
 * 16.  VD_EINVAL where the descriptor is not that conv. */
int vd_stem_conv_c32_bf16(const float* x_nchw, const float* stem_wp, const float* stem_scale, const float* stem_shift,
                          float stem_slope, const vd_conv_desc* d, void* stream);
int vd_stem_conv_blocks(int N, int H, int W);
int vd_stem_conv(const float* x_nchw, const float* wp, void* out, int ldo, int N, int H, int W, const float* scale,
                 const float* shift, float slope, int flags, int out_bf16, float* stats_part, void* stream);
/* Weight gradient of the stem from the same NCHW batch and the gradient dz [N*H*W][ldd >= 32] of its output:
 * dwp [32][32] in the fwd-packed layout (columns >= 27 come out zero). */
int64_t vd_stem_wgrad_ws_bytes(int N, int H, int W);
int vd_stem_wgrad(const float* x_nchw, const float* dz, int ldd, float* dwp, int N, int H, int W, void* ws,
                  int64_t ws_bytes, void* stream);
/* Stem conv 3->32 3x3 s1 p1 (three_darknet.py:163-164): Ci=3 is too thin for the GEMM path, so
 * the stem is lowered to an explicit 32-wide im2col ( col[n,y,x,(ky*3+kx)*3+c], entries 27..31
 * zero ) followed by vd_conv_igemm / vd_conv_wgrad with T=1, Ci=32.  `in` is [N,H,W,3] (nchw=0)
 * or the reference's [N,3,H,W] batch layout (nchw=1, transforms.py:239-245). */
int vd_stem_im2col(const float* in, float* col, int N, int H, int W, int nchw, void* stream);

/* OIHW -> packed [Co_pad][T*Ci] (forward) ; rows >= Co are zero.  kd = temporal kernel depth (1 for 2-D). */
int vd_pack_weight_fwd(const float* w_oihw, float* wp, int Co, int Co_pad, int Ci,
                       int kd, int kh, int kw, void* stream);
/* OIHW -> dgrad pack [Ci][T'*Co_pad] for the tap subset `taps` (indices into kd*kh*kw, already
 * in the order the dgrad tap list uses; HOST pointer).  src_packed=1: w is already in the
 * forward-packed layout [Co][T*Ci] (the layout the parameter arena keeps) instead of OIHW. */
int vd_pack_weight_dgrad(const float* w_oihw, float* wp, int Co, int Co_pad, int Ci,
                         int kd, int kh, int kw, const int32_t* taps, int ntaps, int src_packed,
                         void* stream);
/* VD_CONV_PARITY4 weight image from the fwd-packed weights [Co][9 * Ci] of a 3x3 conv: wp4 [4 * Ci][4 * Co_pad],
 * wp4[(c * Ci + ci)][o * Co_pad + co] = w[co][ci][ky][kx] for the kernel tap (ky, kx) that output parity class c = 2 py + px
 * reaches at gradient offset o = 2 dy + dx (py = 0: ky = 1 at dy = 0; py = 1: ky = 2 at dy = 0, ky = 0 at dy = 1; likewise
 * px / kx / dx), zero where there is none.  Returns the 16-bit block mask through *par_mask (host pointer; a constant). */
int vd_pack_weight_dgrad_s2(const float* wp_fwd, float* wp4, int Co, int Co_pad, int Ci, int32_t* par_mask, void* stream);
/* packed gradient [Co_pad][T*Ci] -> OIHW [Co][Ci][kd][kh][kw] (accumulate=0 overwrites) */
int vd_unpack_wgrad(const float* dwp, float* dw_oihw, int Co, int Ci, int kd, int kh, int kw,
                    void* stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm (layers.py:68 norm_layer(epsilon=1e-5, momentum=0.9)) + LeakyReLU(0.1) (layers.py:69)
 * on NHWC [M, C] (M = N*H*W).
 * ------------------------------------------------------------------------------------- */
/* partial sums: sums[0..C) = sum x, sums[C..2C) = sum x^2 (fp64 accumulators, deterministic) */
int64_t vd_bn_stats_ws_bytes(int64_t M, int C);
int vd_bn_stats(const float* x, int64_t M, int C, double* sums, void* ws, int64_t ws_bytes,
                void* stream);
/* fp64 reduction of a partial table part[nblk][2C] (vd_conv_igemm stats_part) into sums[2C] */
int64_t vd_bn_sum_partials_ws_bytes(int nblk, int C);
int vd_bn_sum_partials(const float* part, int nblk, int C, double* sums, void* ws, int64_t ws_bytes, void* stream);
/* The same reduction fused with what follows it (one launch for tables of up to 1024 rows, else the calls above):
 * forward  = vd_bn_sum_partials + vd_bn_finalize;  backward = vd_bn_sum_partials + vd_bn_param_grads.  The fp64 sums
 * are written as well.  (With SyncBN the unfused calls are used: the all-reduce sits between the two halves.) */
int vd_bn_sum_finalize(const float* part, int nblk, int C, double* sums, double count, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                       float* scale, float* shift, float* save_mean, float* save_invstd, void* ws, int64_t ws_bytes,
                       void* stream);
int vd_bn_sum_param_grads(const float* part, int nblk, int C, double* sums2, float* dgamma, float* dbeta, void* ws,
                          int64_t ws_bytes, void* stream);
/* from (possibly all-reduced) sums and total count: mean, biased var -> scale/shift for the
 * apply, saved mean/invstd for backward, running-stat update run = mom*run + (1-mom)*batch */
int vd_bn_finalize(const double* sums, double count, int C, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var,
                   float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
/* eval: scale/shift from running stats */
int vd_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                    const float* running_var, float eps, int C, float* scale, float* shift,
                    void* stream);
/* y = leaky(x*scale+shift) (+ residual); amax_out (optional): max-abs slots of y (VD_AMAX_FLOATS floats, zeroed by the
 * caller) for the fp16 operand scale of the convs that read y (VD_MATH_F16X2) */
int vd_bn_apply_leaky(const float* x, const float* scale, const float* shift, const float* residual,
                      float* y, int64_t M, int C, float slope, float* amax_out, void* stream);
/* backward, pass 1: partial sums of g=dy*leaky'(.) and g*xhat -> sums2[0..C)=sum g, [C..2C)=sum g*xhat */
int vd_bn_bwd_reduce(const float* x, const float* dy, const float* scale, const float* shift,
                     const float* save_mean, const float* save_invstd, int64_t M, int C, float slope,
                     double* sums2, void* ws, int64_t ws_bytes, void* stream);
/* dgamma[c] = sum g*xhat, dbeta[c] = sum g from the LOCAL sums2 (before any SyncBN all-reduce) */
int vd_bn_param_grads(const double* sums2, int C, float* dgamma, float* dbeta, void* stream);
/* backward, pass 2: dx = scale*(g - mean_g - xhat*mean_gx), means from (all-reduced) sums2 / count */
int vd_bn_bwd_apply(const float* x, const float* dy, const float* scale, const float* shift,
                    const float* save_mean, const float* save_invstd,
                    const double* sums2, double count, int64_t M, int C, float slope,
                    float* dx, float* amax_out /* optional, as in vd_bn_apply_leaky */, void* stream);

/* Operand-range guard of VD_MATH_F16X2.  That arithmetic scales a whole tensor by ONE power of two; elements more than
 * ~2^18 below the tensor's max-abs are staged with fewer than 22 significant bits.  Dense tensors (activations, gradients
 * behind a BatchNorm backward) only get there through their per-channel scales, which are known from the layer's vectors:
 * per BatchNorm layer i, ratios[2i] = min_c / max_c of (|gamma_c| + |beta_c|) (the cell's output y), ratios[2i+1] the same
 * of |scale_c| (= gamma_c * invstd_c: the gradient dz the layer's backward writes; scale may be NULL: ratio 1), and
 * flags[2i+k] = ratios[2i+k] < thresh.  `items` is a DEVICE array.  The host reads the flags when it synchronises anyway
 * and rebuilds the plans of flagged tensors' consumers in VD_MATH_SPLIT, whose bf16 pieces have fp32's exponent range
 * (viddet_amd/model.py check_operand_ranges).  Sparse saturated tensors - the loss gradient - never run in VD_MATH_F16X2. */
typedef struct {
    const float* gamma;
    const float* beta;
    const float* scale;     /* gamma * invstd of the last training forward, or NULL */
    int32_t C;
    int32_t pad_;
} vd_guard_item;
int vd_range_guard(const vd_guard_item* items, int n, float thresh, float* ratios, int32_t* flags, void* stream);

/* out[slot] = max(a[slot], b[slot]) over the sub-slots (b may be NULL: copy): the max-abs of a tensor assembled from, or
 * bounded by, other tensors (upsample+concat, temporal pooling / stacking) without another pass over it */
int vd_amax_merge(const float* a, const float* b, float* out, void* stream);
/* max-abs of a tensor into its slots (zeroes them first): amax[VD_AMAX_FLOATS] */
int vd_amax(const float* x, int64_t n, float* amax, void* stream);
/* the same for `nseg` ranges of one buffer in ONE launch (the conv weights of the parameter arena, once per optimiser
 * step): seg is a DEVICE array [nseg][2] = (element offset, element count); amax [nseg][VD_AMAX_FLOATS] */
int vd_amax_segments(const float* base, const int64_t* seg, int nseg, float* amax, void* stream);

/* ---------------------------------------------------------------------------------------
 * Pointwise helpers
 * ------------------------------------------------------------------------------------- */
/* out = a + b (residual backward fan-in etc.) */
int vd_add(const float* a, const float* b, float* out, int64_t n, void* stream);
int vd_fill(float* out, float v, int64_t n, void* stream);
/* layers.py:11-20 _upsample (nearest x2) + yolo3.py:1170-1177 slice_like + concat(dim=1):
 * out[n, y, x, 0:Cu) = up[n, y/2, x/2, :], out[..., Cu:Cu+Cr) = route[n, y, x, :] */
int vd_upsample2x_concat(const float* up, const float* route, float* out,
                         int N, int Ho, int Wo, int Cu, int Cr, void* stream);
/* backward: dup[n,y2,x2,:] = sum of the 4 children of dout[..., 0:Cu); droute = dout[..., Cu:).  Either output may be
 * NULL (that gradient is not needed: every parameter upstream of it has grad_req 'null', wrappers.py:55-57). */
int vd_upsample2x_concat_bwd(const float* dout, float* dup, float* droute,
                             int N, int Ho, int Wo, int Cu, int Cr, void* stream);
/* NCHW fp32 -> NHWC fp32 (the reference feeds NCHW, transforms.py:239-245) */
int vd_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, void* stream);
/* transforms.py:229-245: uint8 HWC -> /255 -> (x-mean)/std, NHWC fp32 */
int vd_preprocess_u8_nhwc(const uint8_t* in, float* out, int64_t npix, void* stream);
/* the same arithmetic from uint8 NHWC frames [N,H,W,3] into the planar fp32 batch [N,3,H,W] the network takes
 * (transforms.py:239-245 to_tensor + normalize): loaders ship uint8, a quarter of the bytes */
int vd_preprocess_u8_nchw(const uint8_t* in, float* out, int N, int H, int W, void* stream);
/* layers.py:161-205 TemporalPooling over K frames: x [B,K,HW*C] -> y [B,HW*C]; type 0=max 1=mean */
int vd_temporal_pool(const float* x, float* y, int32_t* argmax, int B, int K, int64_t inner, int type,
                     void* stream);
int vd_temporal_pool_bwd(const float* dy, const int32_t* argmax, float* dx, int B, int K,
                         int64_t inner, int type, void* stream);
/* x.slice_axis(axis=1, begin=k0, end=k0+kc) on folded frames (yolo3_temporal.py:437-446): x [B*K, inner] -> y [B*kc, inner];
 * backward=1 is its gradient (x = dy [B*kc, inner], y = dx [B*K, inner], zero outside the range).  Also how the un-padded
 * temporal convs of _conv21d(padding=[1,0]) (:329-332) are taken out of the 'same'-padded ones: frames [1, K-1). */
int vd_frame_slice(const float* x, float* y, int B, int K, int k0, int kc, int64_t inner, int backward, void* stream);
/* 'cat' join (yolo3.py:1108,1136 reshape (B,K,C,h,w)->(B,K*C,h,w)): NHWC [B*K,hw,C] -> [B,hw,K*C];
 * backward=1 runs the inverse (x = stacked gradient, y = per-frame gradient) */
int vd_temporal_cat(const float* x, float* y, int B, int K, int64_t hw, int C, int backward, void* stream);

/* ---------------------------------------------------------------------------------------
 * YOLO head: decode / filter / NMS / targets / loss
 * Head tensors are the raw prediction-conv outputs, NHWC [B, g, g, ldh], channel = a*(5+C)+j,
 * j: 0,1 = raw xy, 2,3 = raw wh, 4 = objectness, 5.. = classes (yolo3.py:158-165).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* head[3];   /* scale order = reference output order: stride 32, 16, 8 (yolo3.py:1013) */
    int32_t g[3];           /* grid side per scale */
    int32_t ldh;            /* channel pitch of head tensors */
    float   stride[3];
    float   anchors[3][6];  /* (w,h) x3 per scale, in reference output order */
    int32_t B, C;
} vd_head_desc;

/* Inference (yolo3.py:167-199 + 1195-1206): fused decode + valid_thresh filter.
 * Emits per image a candidate list of (score, row) where row is the row index into the
 * (B, C*P, 6) tensor the reference would materialise.  counts[b] = number emitted (may exceed
 * cap: then only the first cap are stored and vd_nms_topk reports overflow). */
int vd_yolo_decode_filter(const vd_head_desc* h, float valid_thresh,
                          float* cand_score, int32_t* cand_row, int32_t cap, int32_t* counts,
                          void* stream);
/* F.contrib.box_nms(overlap_thresh, topk, id_index=0, score_index=1, coord_start=2,
 * force_suppress=False) + slice_axis(0:post_nms)  (yolo3.py:1197-1202).
 * out_ids/out_scores [B,post_nms], out_boxes [B,post_nms,4], out_rows [B,post_nms] (original
 * row index of each kept detection, -1 padded).  ws >= vd_nms_ws_bytes. */
int64_t vd_nms_ws_bytes(int B, int cap, int topk);
int vd_nms_topk(const vd_head_desc* h, const float* cand_score, const int32_t* cand_row,
                int32_t cap, const int32_t* counts, float nms_thresh, int topk, int post_nms,
                float* out_ids, float* out_scores, float* out_boxes, int32_t* out_rows,
                void* ws, int64_t ws_bytes, void* stream);

/* Training (yolo3.py:1140-1187 + yolo_target.py:173-281 + gluoncv YOLOV3Loss):
 * fused decode -> dynamic ignore mask (IoU>thr vs gt) -> target merge -> 4 losses and their
 * gradient wrt the raw head outputs.  targets are the prefetched ones, (B,P,.) in reference row
 * order; gt [B,M,4] corner boxes padded with -1.  losses [B,4] = obj, center, scale, cls.
 * dhead[s] has the geometry of head[s] (pad channels get 0).  box_out (optional) [B,P,4]. */
int vd_yolo_loss_fwd_bwd(const vd_head_desc* h, const float* gt, int M,
                         const float* obj_t, const float* center_t, const float* scale_t,
                         const float* weight_t, const float* class_t,
                         float ignore_thresh, int label_smooth,
                         float* losses, float* const dhead[3], float* box_out,
                         float* const dhead_amax[3] /* optional: max-abs slots of the three gradients, zeroed by the caller */,
                         void* ws, int64_t ws_bytes, void* stream);
int64_t vd_yolo_loss_ws_bytes(const vd_head_desc* h);

/* ---------------------------------------------------------------------------------------
 * Optimiser (gluon.Trainer('sgd'), train_yolov3.py:527-530,634):
 *   g = rescale*grad ; mom = momentum*mom - lr*(g + wd*w) ; w += mom
 * over a flat parameter arena.
 * ------------------------------------------------------------------------------------- */
int vd_sgd_momentum(float* w, const float* grad, float* mom, int64_t n,
                    float lr, float momentum, float wd, float rescale, void* stream);

/* ---------------------------------------------------------------------------------------
 * bf16-STORAGE training (BASELINE configs[4]: "combined-dataset training ... bf16"; the reference is fp32-only,
 * train_yolov3.py:623-636, so this mode is judged against the fp32 oracle at a stated bf16 tolerance).
 * Activations (conv outputs z, cell outputs y) and their gradients (dy, dz) are bf16 NHWC tensors; weights (fp32 master +
 * bf16 images), weight gradients, BatchNorm vectors / statistics, the head logits, the losses and the optimiser stay fp32.
 *   convs fwd / dgrad      vd_conv_igemm_bf16 (stats_part; strided outputs)
 *   weight gradients       vd_conv_wgrad with VD_STORE_BF16 in vd_wgrad_desc.flags; vd_stem_wgrad_bf16
 *   BatchNorm passes       the entry points below: the fp32 kernels instantiated on bf16 tensors (fp32 arithmetic)
 *   concat                 vd_upsample2x_concat on half the channel counts (a copy); vd_upsample2x_concat_bwd_bf16
 *   loss                   vd_yolo_loss_fwd_bwd_bf16: fp32 logits in, bf16 gradient rows out
 * ------------------------------------------------------------------------------------- */
int vd_bn_stats_bf16(const void* x, int64_t M, int C, double* sums, void* ws, int64_t ws_bytes, void* stream);
int vd_bn_apply_leaky_bf16(const void* x, const float* scale, const float* shift, const void* residual, void* y,
                           int64_t M, int C, float slope, void* stream);
int vd_bn_bwd_reduce_bf16(const void* x, const void* dy, const float* scale, const float* shift, const float* save_mean,
                          const float* save_invstd, int64_t M, int C, float slope, double* sums2, void* ws, int64_t ws_bytes,
                          void* stream);
int vd_bn_bwd_apply_bf16(const void* x, const void* dy, const float* scale, const float* shift, const float* save_mean,
                         const float* save_invstd, const double* sums2, double count, int64_t M, int C, float slope,
                         void* dx, void* stream);
int vd_pack_weight_dgrad_bf16(const float* w, void* wp_bf16, int Co, int Co_pad, int Ci, int kd, int kh, int kw,
                              const int32_t* taps, int ntaps, int src_packed, void* stream);
int vd_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* TemporalPooling (layers.py:161-205; type 0 = max, 1 = mean) over the K frames of a window on bf16 tensors: the bf16
 * inference of the k > 1 networks.  vd_frame_slice / vd_temporal_cat are copies: they take bf16 tensors with the inner /
 * channel counts halved (two bf16 = one 4-byte word). */
int vd_temporal_pool_bf16(const void* x, void* y, int B, int K, int64_t inner, int type, void* stream);
int vd_upsample2x_concat_bwd_bf16(const void* dout, void* dup, void* droute, int N, int Ho, int Wo, int Cu, int Cr, void* stream);
int vd_stem_wgrad_bf16(const float* x_nchw, const void* dz, int ldd, float* dwp, int N, int H, int W, void* ws, int64_t ws_bytes,
                       void* stream);
int vd_yolo_loss_fwd_bwd_bf16(const vd_head_desc* h, const float* gt, int M, const float* obj_t, const float* center_t,
                              const float* scale_t, const float* weight_t, const float* class_t, float ignore_thresh,
                              int label_smooth, float* losses, void* const dhead[3], float* box_out, void* ws, int64_t ws_bytes,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDDET_HIP_H */
