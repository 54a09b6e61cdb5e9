"""Thin operator layer over the C-ABI: builds descriptors (tap lists, geometry) for torch tensors
used as device storage and launches the HIP kernels on torch's current stream.

No arithmetic happens in Python/torch here; every function ends in a `lib.vd_*` call.
Reference call sites are cited in include/viddet_hip.h next to each entry point.
"""
import ctypes as C

import torch

from . import lib as L
from .lib import (ConvDesc, WgradDesc, HeadDesc, EPI_AFFINE, EPI_LEAKY, EPI_RESIDUAL, MATH_SPLIT, MATH_BF16, MATH_F16X2,
                  MATH_NOHALO, AMAX_FLOATS, check, ptr)


def round_up(v, m):
    return (v + m - 1) // m * m


def _lib():
    return L.load()


def _s():
    return L.stream_ptr()


# --------------------------------------------------------------------------------------------
# tap lists
# --------------------------------------------------------------------------------------------
def fwd_taps(k, pad, kd=1, pad_d=0):
    """Taps of a (kd,k,k) kernel in OI[D]HW order t = (kz*k + ky)*k + kx."""
    taps = []
    for kz in range(kd):
        for ky in range(k):
            for kx in range(k):
                taps.append((ky - pad, kx - pad, kz - pad_d))
    return taps


def dgrad_plans(k, pad, stride, Hi, Wi, kd=1, pad_d=0):
    """Data-gradient launches of a conv with kernel (kd,k,k), spatial `stride`, temporal stride 1.

    Returns a list of plans; each plan = dict(py, px, Hg, Wg, taps=[(dy,dx,dz)], tap_ids=[t]) meaning:
    dX[n, q_y*stride+py, q_x*stride+px, :] = sum_j dZ[n, q_y+dy_j, q_x+dx_j, frame+dz_j, :] . W[:, :, tap_ids[j]]
    """
    plans = []
    for py in range(stride):
        for px in range(stride):
            Hg = (Hi - py + stride - 1) // stride
            Wg = (Wi - px + stride - 1) // stride
            if Hg <= 0 or Wg <= 0:
                continue
            taps, ids = [], []
            for kz in range(kd):
                for ky in range(k):
                    if (py + pad - ky) % stride:
                        continue
                    for kx in range(k):
                        if (px + pad - kx) % stride:
                            continue
                        taps.append(((py + pad - ky) // stride, (px + pad - kx) // stride, pad_d - kz))
                        ids.append((kz * k + ky) * k + kx)
            plans.append(dict(py=py, px=px, Hg=Hg, Wg=Wg, taps=taps, tap_ids=ids))
    return plans


def _set_taps(d, taps):
    d.T = len(taps)
    for i, (dy, dx, dz) in enumerate(taps):
        d.dy[i], d.dx[i], d.dz[i] = dy, dx, dz


# --------------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------------
def conv_igemm(x, wp, out, *, N, Hi, Wi, Ci, Hg, Wg, in_stride, taps, Ho, Wo, Co, ldo,
               out_stride=1, out_oy=0, out_ox=0, scale=None, shift=None, residual=None, ldr=0,
               leaky=False, slope=0.1, kfr=1, in_scale=None, in_shift=None, in_slope=0.1, tile=0, split=False,
               amax_in=None, amax_w=None, amax_out=None, streamk_ws=None, desc_out=None):
    """streamk_ws: a workspace from streamk_workspace() - the launch runs as a persistent stream-K grid where that form
    applies (VD_CONV_STREAMK; bit-identical results).
    split: False = fp32 MFMA; True / 'split' = exact 3-plane bf16 split; 'bf16' = one plane (bf16-rounded operands);
    'f16x2' = two fp16 planes with per-tensor scales - the max-abs slots of x and wp (amax_in / amax_w, see amax())
    are computed here when not given."""
    d = ConvDesc()
    d.tile = tile
    d.in_, d.wp, d.out = ptr(x), ptr(wp), ptr(out)
    d.scale, d.shift, d.residual = ptr(scale), ptr(shift), ptr(residual)
    d.N, d.Hi, d.Wi, d.Ci = N, Hi, Wi, Ci
    d.Hg, d.Wg, d.in_stride = Hg, Wg, in_stride
    _set_taps(d, taps)
    d.Kfr = kfr
    d.Ho, d.Wo, d.Co = Ho, Wo, Co
    d.out_stride, d.out_oy, d.out_ox = out_stride, out_oy, out_ox
    d.ldo, d.ldr = ldo, ldr
    flags = 0
    if scale is not None or shift is not None:
        flags |= EPI_AFFINE
    if leaky:
        flags |= EPI_LEAKY
    if residual is not None:
        flags |= EPI_RESIDUAL
    if split in ('f16x2', 'f16x2nh'):           # 'f16x2nh': without the halo-staged loop of the 3x3 stride-1 convs
        flags |= MATH_F16X2 | (MATH_NOHALO if split == 'f16x2nh' else 0)
        amax_in = amax(x) if amax_in is None else amax_in
        amax_w = amax(wp) if amax_w is None else amax_w
        d.amax_in, d.amax_w = ptr(amax_in), ptr(amax_w)
    elif split:
        flags |= MATH_BF16 if split == 'bf16' else MATH_SPLIT
    d.amax_out = ptr(amax_out)
    if streamk_ws is not None:
        flags |= L.CONV_STREAMK
        d.sk_ws, d.sk_ws_bytes = ptr(streamk_ws), streamk_ws.numel() * streamk_ws.element_size()
    d.flags, d.slope = flags, slope
    d.in_scale, d.in_shift, d.in_slope = ptr(in_scale), ptr(in_shift), in_slope
    if desc_out is not None:
        desc_out.append(d)
    check(_lib().vd_conv_igemm(C.byref(d), _s()), "vd_conv_igemm")


def streamk_workspace(device=None):
    """A stream-K hand-off workspace (vd_conv_desc.sk_ws): its counter header zeroed once, here; one per stream that may
    run such launches at the same time."""
    n = int(_lib().vd_conv_igemm_streamk_ws_bytes())
    ws = torch.empty(n, dtype=torch.uint8, device=device or torch.device("cuda", torch.cuda.current_device()))
    ws[:L.SK_HEADER_BYTES].zero_()
    return ws


def conv_fwd(x, wp, out, *, k, stride, pad, Co, ldo=None, scale=None, shift=None, residual=None,
             leaky=False, slope=0.1, kd=1, pad_d=0, kfr=1, tile=0, split=False, amax_in=None, amax_w=None,
             amax_out=None, in_scale=None, in_shift=None, in_slope=0.1, streamk_ws=None, desc_out=None):
    """Forward conv on NHWC x [N,Hi,Wi,Ci] with fwd-packed weights wp [>=Co][T*Ci] -> out [N,Ho,Wo,ldo].
    in_scale / in_shift: per-input-channel transform leaky(x * s + b) applied in the operand gather (the padding stays zero)."""
    N, Hi, Wi, Ci = x.shape
    Ho = (Hi + 2 * pad - k) // stride + 1
    Wo = (Wi + 2 * pad - k) // stride + 1
    ldo = Co if ldo is None else ldo
    conv_igemm(x, wp, out, N=N, Hi=Hi, Wi=Wi, Ci=Ci, Hg=Ho, Wg=Wo, in_stride=stride,
               taps=fwd_taps(k, pad, kd, pad_d), Ho=Ho, Wo=Wo, Co=Co, ldo=ldo, scale=scale, shift=shift,
               residual=residual, ldr=ldo, leaky=leaky, slope=slope, kfr=kfr, tile=tile, split=split,
               amax_in=amax_in, amax_w=amax_w, amax_out=amax_out, in_scale=in_scale, in_shift=in_shift, in_slope=in_slope,
               streamk_ws=streamk_ws, desc_out=desc_out)
    return Ho, Wo


def conv_igemm_bf16(x, wb, out, *, N, Hi, Wi, Ci, Hg, Wg, in_stride, taps, Ho, Wo, Co, ldo, out_stride=1, out_oy=0, out_ox=0,
                    scale=None, shift=None, residual=None, ldr=0, leaky=False, slope=0.1, tile=0, stats_part=None, nohalo=False,
                    streamk_ws=None, splitk=False):
    """vd_conv_igemm_bf16: x / wb / residual bf16, out bf16 or fp32 (by its dtype); any output geometry; stats_part =
    fused BatchNorm statistics table [mtiles][2 * Co] (conv_bf16_mtiles).  streamk_ws: VD_CONV_STREAMK (bit-identical);
    splitk=True with it: VD_CONV_SPLITK too (launches with too few tiles for the chip are cut along K)."""
    d = ConvDesc()
    d.tile = tile
    d.in_, d.wp, d.out = ptr(x), ptr(wb), ptr(out)
    d.scale, d.shift, d.residual = ptr(scale), ptr(shift), ptr(residual)
    d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = N, Hi, Wi, Ci, Hg, Wg, in_stride
    _set_taps(d, taps)
    d.Kfr, d.Ho, d.Wo, d.Co = 1, Ho, Wo, Co
    d.out_stride, d.out_oy, d.out_ox, d.ldo, d.ldr = out_stride, out_oy, out_ox, ldo, ldr
    d.flags = (EPI_AFFINE if (scale is not None or shift is not None) else 0) | (EPI_LEAKY if leaky else 0) | \
              (EPI_RESIDUAL if residual is not None else 0) | (MATH_NOHALO if nohalo else 0)
    d.slope, d.stats_part = slope, ptr(stats_part)
    if streamk_ws is not None:
        d.flags |= L.CONV_STREAMK | (L.CONV_SPLITK if splitk else 0)
        d.sk_ws, d.sk_ws_bytes = ptr(streamk_ws), streamk_ws.numel() * streamk_ws.element_size()
    check(_lib().vd_conv_igemm_bf16(C.byref(d), 1 if out.dtype == torch.float32 else 0, _s()), "vd_conv_igemm_bf16")
    return d


def conv_bf16_mtiles(N, Hg, Wg, Ci, Co, tile=0):
    d = ConvDesc()
    d.N, d.Hg, d.Wg, d.Ci, d.Co, d.tile = N, Hg, Wg, Ci, Co, tile
    return _lib().vd_conv_igemm_bf16_mtiles(C.byref(d))


def pack_weight_bf16(wp32, wb, *, Co, Co_pad, Ci, Ci_pad, T):
    """fp32 packed [>=Co][T*Ci] -> bf16 [Co_pad][T*Ci_pad] (zero-padded rows / channels)."""
    check(_lib().vd_pack_weight_bf16(ptr(wp32), ptr(wb), Co, Co_pad, Ci, Ci_pad, T, _s()), "vd_pack_weight_bf16")


def conv_wgrad(x, dout, dwp, ws, *, k, stride, pad, Co, kd=1, pad_d=0, kfr=1, splits=0, split=False, amax_in=None,
               amax_dout=None, in_scale=None, in_shift=None, in_slope=0.1):
    """dwp [Co][T*Ci] (fwd-packed layout) = wgrad(x [N,Hi,Wi,Ci], dout [N,Ho,Wo,Co])."""
    N, Hi, Wi, Ci = x.shape
    _, Ho, Wo, ldd = dout.shape
    d = WgradDesc()
    d.in_, d.dout, d.dwp = ptr(x), ptr(dout), ptr(dwp)
    d.N, d.Hi, d.Wi, d.Ci = N, Hi, Wi, Ci
    d.Hg, d.Wg, d.Co, d.ldd = Ho, Wo, Co, ldd
    d.in_stride = stride
    _set_taps(d, fwd_taps(k, pad, kd, pad_d))
    d.Kfr, d.splits = kfr, splits
    d.in_scale, d.in_shift, d.in_slope = ptr(in_scale), ptr(in_shift), in_slope
    if split in ('f16x2', 'f16x2nh', 'f16x2h'):
        d.flags = MATH_F16X2 | (L.WGRAD_HALO if split == 'f16x2h' else 0)     # 'f16x2h': the halo-ring kernel where it applies
        amax_in = amax(x) if amax_in is None else amax_in
        amax_dout = amax(dout) if amax_dout is None else amax_dout
        d.amax_in, d.amax_dout = ptr(amax_in), ptr(amax_dout)
    else:
        d.flags = (MATH_BF16 if split == 'bf16' else MATH_SPLIT) if split else 0
    if x.dtype == torch.bfloat16:               # bf16-storage training: both operands bf16, the gradient fp32
        assert dout.dtype == torch.bfloat16 and dwp.dtype == torch.float32
        assert Ci % 8 == 0 and Co % 8 == 0 and ldd % 8 == 0 and x.data_ptr() % 16 == 0 and dout.data_ptr() % 16 == 0, \
            "bf16-stored operands: Ci, Co and the gradient pitch multiples of 8, 16-byte aligned tensors"
        d.flags = L.STORE_BF16 | MATH_BF16 | (L.WGRAD_HALO if split == 'halo' else 0)
    lib = _lib()
    need = lib.vd_conv_wgrad_ws_bytes(C.byref(d))
    if need > ws.numel() * ws.element_size():
        raise L.VidDetHipError("conv_wgrad: workspace %d < %d bytes" % (ws.numel() * ws.element_size(), need))
    check(lib.vd_conv_wgrad(C.byref(d), ptr(ws), ws.numel() * ws.element_size(), _s()), "vd_conv_wgrad")


def wgrad_ws_bytes(N, Hi, Wi, Ci, Ho, Wo, Co, k, stride, pad, kd=1, pad_d=0):
    d = WgradDesc()
    d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.Co, d.ldd = N, Hi, Wi, Ci, Ho, Wo, Co, Co
    d.in_stride = stride
    _set_taps(d, fwd_taps(k, pad, kd, pad_d))
    d.Kfr, d.splits = 1, 0
    need = 0
    # the product arithmetics pick different split counts (every split form: the same); the halo-ring kernel its own.  The
    # library only looks at shapes and flags here, so a non-null placeholder stands for the max-abs pointers of MATH_F16X2
    for fl in (0, MATH_SPLIT, MATH_F16X2 | L.WGRAD_HALO, L.STORE_BF16 | L.WGRAD_HALO):
        d.flags = fl
        d.amax_in = d.amax_dout = 1 if fl & MATH_F16X2 else None
        need = max(need, _lib().vd_conv_wgrad_ws_bytes(C.byref(d)))
    return need


def pack_weight_fwd(w_oihw, wp, Co_pad):
    Co, Ci = w_oihw.shape[0], w_oihw.shape[1]
    if w_oihw.dim() == 4:
        kd, kh, kw = 1, w_oihw.shape[2], w_oihw.shape[3]
    else:
        kd, kh, kw = w_oihw.shape[2:]
    check(_lib().vd_pack_weight_fwd(ptr(w_oihw), ptr(wp), Co, Co_pad, Ci, kd, kh, kw, _s()), "vd_pack_weight_fwd")


def pack_weight_dgrad(w, wp, *, Co, Co_pad, Ci, kd, kh, kw, tap_ids, src_packed):
    arr = (C.c_int32 * len(tap_ids))(*tap_ids)
    check(_lib().vd_pack_weight_dgrad(ptr(w), ptr(wp), Co, Co_pad, Ci, kd, kh, kw, arr, len(tap_ids),
                                      1 if src_packed else 0, _s()), "vd_pack_weight_dgrad")


PARITY4_TAPS = [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0)]


def _parity4_mask():
    """bit (4 * offset + class) set where a 3x3 / stride-2 / pad-1 kernel has a tap for output parity class c = 2 py + px at
    gradient offset o = 2 dy + dx (vd_conv_par.hip par_tap: parity 0 -> k = 1 at offset 0; parity 1 -> k = 2 at 0, k = 0 at 1)"""
    tap = lambda parity, off: (1 if off == 0 else -1) if parity == 0 else (2 if off == 0 else 0)
    m = 0
    for o in range(4):
        for c in range(4):
            if tap(c >> 1, o >> 1) >= 0 and tap(c & 1, o & 1) >= 0:
                m |= 1 << (4 * o + c)
    return m


PARITY4_MASK = _parity4_mask()          # 0x8caf: nine of the sixteen blocks


def pack_weight_dgrad_s2(wp_fwd, wp4, *, Co, Co_pad, Ci):
    """weight image of the parity-fused stride-2 data gradient (VD_CONV_PARITY4): fwd-packed [Co][9 * Ci] ->
    [4 * Ci][4 * Co_pad]; returns the 16-bit mask of its nonzero (offset, class) blocks"""
    m = C.c_int32(0)
    check(_lib().vd_pack_weight_dgrad_s2(ptr(wp_fwd), ptr(wp4), Co, Co_pad, Ci, C.byref(m), _s()), "vd_pack_weight_dgrad_s2")
    return int(m.value)


def conv_dgrad_s2_fused(dz, wp4, dx, *, Cin, par_mask, residual=None, tile=0, amax_in=None, amax_w=None, desc_out=None):
    """dx [N, 2 Ho, 2 Wo, Cin] = data gradient of a 3x3 / stride-2 / pad-1 conv from dz [N, Ho, Wo, Cout] in ONE launch"""
    N, Ho, Wo, Cout = dz.shape
    d = ConvDesc()
    d.tile = tile
    d.in_, d.wp, d.out, d.residual = ptr(dz), ptr(wp4), ptr(dx), ptr(residual)
    d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = N, Ho, Wo, Cout, Ho, Wo, 1
    _set_taps(d, PARITY4_TAPS)
    d.Kfr, d.Ho, d.Wo, d.Co = 1, 2 * Ho, 2 * Wo, 4 * Cin
    d.out_stride, d.out_oy, d.out_ox, d.ldo, d.ldr = 2, 0, 0, dx.shape[-1], dx.shape[-1]
    d.flags = MATH_F16X2 | L.CONV_PARITY4 | (EPI_RESIDUAL if residual is not None else 0)
    d.par_cin, d.par_mask, d.slope = Cin, par_mask, 0.1
    amax_in = amax(dz) if amax_in is None else amax_in
    amax_w = amax(wp4) if amax_w is None else amax_w
    d.amax_in, d.amax_w = ptr(amax_in), ptr(amax_w)
    if desc_out is not None:
        desc_out.append((d, amax_in, amax_w))
    check(_lib().vd_conv_igemm(C.byref(d), _s()), "vd_conv_igemm/parity4")


def unpack_weight(wp, w_oihw):
    """fwd-packed [>=Co][T*Ci] -> OIHW (also used for gradients)."""
    Co, Ci = w_oihw.shape[0], w_oihw.shape[1]
    if w_oihw.dim() == 4:
        kd, kh, kw = 1, w_oihw.shape[2], w_oihw.shape[3]
    else:
        kd, kh, kw = w_oihw.shape[2:]
    check(_lib().vd_unpack_wgrad(ptr(wp), ptr(w_oihw), Co, Ci, kd, kh, kw, _s()), "vd_unpack_wgrad")


def stem_im2col(x, col, nchw):
    if nchw:
        N, _, H, W = x.shape
    else:
        N, H, W, _ = x.shape
    check(_lib().vd_stem_im2col(ptr(x), ptr(col), N, H, W, 1 if nchw else 0, _s()), "vd_stem_im2col")


# --------------------------------------------------------------------------------------------
# batch norm
# --------------------------------------------------------------------------------------------
def stem_conv(x_nchw, wp, out, *, scale=None, shift=None, leaky=False, slope=0.1, stats_part=None):
    """Direct stem conv (3x3, 3->32) from the NCHW fp32 batch; out NHWC fp32 or bf16 [N,H,W,32]."""
    N, _, H, W = x_nchw.shape
    flags = (EPI_AFFINE if scale is not None else 0) | (EPI_LEAKY if leaky else 0)
    check(_lib().vd_stem_conv(ptr(x_nchw), ptr(wp), ptr(out), out.shape[-1], N, H, W, ptr(scale), ptr(shift), slope, flags,
                              1 if out.dtype == torch.bfloat16 else 0, ptr(stats_part), _s()), "vd_stem_conv")


def stem_wgrad(x_nchw, dz, dwp, ws):
    """Weight gradient of the stem: dwp [32][32] fwd-packed from x (N,3,H,W) and dz [N,H,W,ldd]."""
    N, _, H, W = x_nchw.shape
    fn = _lib().vd_stem_wgrad_bf16 if dz.dtype == torch.bfloat16 else _lib().vd_stem_wgrad
    check(fn(ptr(x_nchw), ptr(dz), dz.shape[-1], ptr(dwp), N, H, W, ptr(ws), ws.numel() * ws.element_size(), _s()), "vd_stem_wgrad")


def bn_stats(x2d_rows, C_, x, sums, ws):
    fn = _lib().vd_bn_stats_bf16 if x.dtype == torch.bfloat16 else _lib().vd_bn_stats
    check(fn(ptr(x), x2d_rows, C_, ptr(sums), ptr(ws), ws.numel() * ws.element_size(), _s()), "vd_bn_stats")


def bn_stats_ws_bytes(M, C_):
    return _lib().vd_bn_stats_ws_bytes(M, C_)


def bn_finalize(sums, count, C_, gamma, beta, eps, momentum, rmean, rvar, scale, shift, smean, sinv):
    check(_lib().vd_bn_finalize(ptr(sums), float(count), C_, ptr(gamma), ptr(beta), eps, momentum, ptr(rmean),
                                ptr(rvar), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), _s()), "vd_bn_finalize")


def bn_fold_eval(gamma, beta, rmean, rvar, eps, scale, shift):
    check(_lib().vd_bn_fold_eval(ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), eps, gamma.numel(), ptr(scale),
                                 ptr(shift), _s()), "vd_bn_fold_eval")


def bn_apply_leaky(x, scale, shift, residual, y, M, C_, slope=0.1, amax_out=None):
    if x.dtype == torch.bfloat16:
        check(_lib().vd_bn_apply_leaky_bf16(ptr(x), ptr(scale), ptr(shift), ptr(residual), ptr(y), M, C_, slope, _s()),
              "vd_bn_apply_leaky_bf16")
        return
    check(_lib().vd_bn_apply_leaky(ptr(x), ptr(scale), ptr(shift), ptr(residual), ptr(y), M, C_, slope, ptr(amax_out),
                                   _s()), "vd_bn_apply_leaky")


def bn_bwd_reduce(x, dy, scale, shift, smean, sinv, M, C_, sums2, ws, slope=0.1):
    if x.dtype == torch.bfloat16:
        check(_lib().vd_bn_bwd_reduce_bf16(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), M, C_, slope,
                                           ptr(sums2), ptr(ws), ws.numel() * ws.element_size(), _s()), "vd_bn_bwd_reduce_bf16")
        return
    check(_lib().vd_bn_bwd_reduce(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), M, C_, slope,
                                  ptr(sums2), ptr(ws), ws.numel() * ws.element_size(), _s()), "vd_bn_bwd_reduce")


def bn_param_grads(sums2, C_, dgamma, dbeta):
    check(_lib().vd_bn_param_grads(ptr(sums2), C_, ptr(dgamma), ptr(dbeta), _s()), "vd_bn_param_grads")


def bn_bwd_apply(x, dy, scale, shift, smean, sinv, sums2, count, M, C_, dx, slope=0.1, amax_out=None):
    if x.dtype == torch.bfloat16:
        check(_lib().vd_bn_bwd_apply_bf16(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), ptr(sums2),
                                          float(count), M, C_, slope, ptr(dx), _s()), "vd_bn_bwd_apply_bf16")
        return
    check(_lib().vd_bn_bwd_apply(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), ptr(sums2),
                                 float(count), M, C_, slope, ptr(dx), ptr(amax_out), _s()), "vd_bn_bwd_apply")


# --------------------------------------------------------------------------------------------
# pointwise
# --------------------------------------------------------------------------------------------
def amax(x, out=None):
    """Max-abs slots (AMAX_FLOATS floats) of a tensor: the operand-scale input of the VD_MATH_F16X2 arithmetic."""
    out = torch.empty(AMAX_FLOATS, device=x.device) if out is None else out
    check(_lib().vd_amax(ptr(x), x.numel(), ptr(out), _s()), "vd_amax")
    return out


def amax_value(slots):
    """The tensor's max-abs as a Python float (tests / diagnostics; synchronises)."""
    return float(slots.view(-1)[::L.AMAX_STRIDE][:L.AMAX_SLOTS].max())


def add(a, b, out):
    if out.dtype == torch.bfloat16:
        check(_lib().vd_add_bf16(ptr(a), ptr(b), ptr(out), out.numel(), _s()), "vd_add_bf16")
        return
    check(_lib().vd_add(ptr(a), ptr(b), ptr(out), out.numel(), _s()), "vd_add")


def fill(t, v):
    check(_lib().vd_fill(ptr(t), float(v), t.numel(), _s()), "vd_fill")


def upsample2x_concat(up, route, out):
    N, Ho, Wo, Cr = route.shape
    Cu = up.shape[3]
    if out.dtype == torch.bfloat16:             # a copy: two bf16 = one 4-byte word
        Cu, Cr = Cu // 2, Cr // 2
    check(_lib().vd_upsample2x_concat(ptr(up), ptr(route), ptr(out), N, Ho, Wo, Cu, Cr, _s()), "vd_upsample2x_concat")


def upsample2x_concat_bwd(dout, dup, droute):
    N, Ho, Wo, Cr = droute.shape
    Cu = dup.shape[3]
    if dout.dtype == torch.bfloat16:
        check(_lib().vd_upsample2x_concat_bwd_bf16(ptr(dout), ptr(dup), ptr(droute), N, Ho, Wo, Cu, Cr, _s()),
              "vd_upsample2x_concat_bwd_bf16")
        return
    check(_lib().vd_upsample2x_concat_bwd(ptr(dout), ptr(dup), ptr(droute), N, Ho, Wo, Cu, Cr, _s()),
          "vd_upsample2x_concat_bwd")


def nchw_to_nhwc(x, out):
    N, Cc, H, W = x.shape
    check(_lib().vd_nchw_to_nhwc(ptr(x), ptr(out), N, Cc, H, W, _s()), "vd_nchw_to_nhwc")


def preprocess_u8(x_u8, out):
    check(_lib().vd_preprocess_u8_nhwc(ptr(x_u8), ptr(out), x_u8.numel() // 3, _s()), "vd_preprocess_u8_nhwc")


def temporal_pool(x, y, argmax, B, K, inner, type_):
    check(_lib().vd_temporal_pool(ptr(x), ptr(y), ptr(argmax), B, K, inner, type_, _s()), "vd_temporal_pool")


def temporal_pool_bwd(dy, argmax, dx, B, K, inner, type_):
    check(_lib().vd_temporal_pool_bwd(ptr(dy), ptr(argmax), ptr(dx), B, K, inner, type_, _s()), "vd_temporal_pool_bwd")


def sgd_momentum(w, g, m, lr, momentum, wd, rescale):
    check(_lib().vd_sgd_momentum(ptr(w), ptr(g), ptr(m), w.numel(), lr, momentum, wd, rescale, _s()), "vd_sgd_momentum")


# --------------------------------------------------------------------------------------------
# yolo head
# --------------------------------------------------------------------------------------------
def make_head_desc(heads, grids, ldh, strides, anchors, B, num_class):
    h = HeadDesc()
    for s in range(3):
        h.head[s] = heads[s].data_ptr()
        h.g[s] = grids[s]
        h.stride[s] = float(strides[s])
        for j in range(6):
            h.anchors[s][j] = float(anchors[s][j])
    h.ldh, h.B, h.C = ldh, B, num_class
    return h


def yolo_decode_filter(h, valid_thresh, cand_score, cand_row, cap, counts):
    check(_lib().vd_yolo_decode_filter(C.byref(h), valid_thresh, ptr(cand_score), ptr(cand_row), cap, ptr(counts), _s()),
          "vd_yolo_decode_filter")


def nms_topk(h, cand_score, cand_row, cap, counts, nms_thresh, topk, post_nms, ids, scores, boxes, rows, ws):
    check(_lib().vd_nms_topk(C.byref(h), ptr(cand_score), ptr(cand_row), cap, ptr(counts), nms_thresh, topk, post_nms,
                             ptr(ids), ptr(scores), ptr(boxes), ptr(rows), ptr(ws), ws.numel() * ws.element_size(),
                             _s()), "vd_nms_topk")


def yolo_loss_fwd_bwd(h, gt, M, obj_t, center_t, scale_t, weight_t, class_t, ignore_thresh, label_smooth, losses,
                      dheads, box_out, ws, dhead_amax=None):
    arr = (C.c_void_p * 3)(*[t.data_ptr() for t in dheads])
    if dheads[0].dtype == torch.bfloat16:       # bf16-storage training: bf16 gradient rows (fp32 logits)
        check(_lib().vd_yolo_loss_fwd_bwd_bf16(C.byref(h), ptr(gt), M, ptr(obj_t), ptr(center_t), ptr(scale_t), ptr(weight_t),
                                               ptr(class_t), ignore_thresh, 1 if label_smooth else 0, ptr(losses),
                                               C.byref(arr), ptr(box_out), ptr(ws), ws.numel() * ws.element_size(), _s()),
              "vd_yolo_loss_fwd_bwd_bf16")
        return
    am = None if dhead_amax is None else C.byref((C.c_void_p * 3)(*[t.data_ptr() for t in dhead_amax]))
    check(_lib().vd_yolo_loss_fwd_bwd(C.byref(h), ptr(gt), M, ptr(obj_t), ptr(center_t), ptr(scale_t), ptr(weight_t),
                                      ptr(class_t), ignore_thresh, 1 if label_smooth else 0, ptr(losses),
                                      C.byref(arr), ptr(box_out), am, ptr(ws), ws.numel() * ws.element_size(), _s()),
          "vd_yolo_loss_fwd_bwd")


def yolo_loss_ws_bytes(h):
    return _lib().vd_yolo_loss_ws_bytes(C.byref(h))
