"""GPU parity: tap-list implicit-GEMM conv (forward, data gradient, weight gradient, stem) vs the
fp64 oracle (oracle/ops.py).  Tolerance 2e-4 abs on O(1) outputs (fp32 MFMA = exact fp32 fma chain)."""
import numpy as np
import pytest
import torch

from oracle import ops as R
from tests.util import dev, nchw_to_dev_nhwc, dev_nhwc_to_nchw, maxdiff

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _mk(n, ci, h, w, co, k, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, ci, h, w))
    wt = rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)
    return rng, x, wt


def _packed(wt, co_pad):
    from viddet_amd import ops
    co, ci, kh, kw = wt.shape
    wp = torch.empty(co_pad, kh * kw * ci, device="cuda")
    ops.pack_weight_fwd(dev(wt), wp, co_pad)
    return wp


FWD_CASES = [
    # n, ci, h, w, co, k, stride, pad
    (2, 32, 16, 16, 64, 3, 1, 1),
    (2, 64, 13, 13, 32, 1, 1, 0),      # 128x32 tile
    (1, 32, 20, 20, 64, 3, 2, 1),      # stride 2, 128x64 tile
    (3, 128, 13, 13, 256, 3, 1, 1),    # 64x128 tile (few blocks)
    (8, 64, 52, 52, 128, 3, 1, 1),     # 128x128 tile (many blocks)
    (2, 96, 7, 9, 160, 1, 1, 0),       # ragged: Co not a tile multiple, odd spatial
    (1, 64, 19, 19, 255, 1, 1, 0),     # head-like Co
    (2, 32, 15, 17, 64, 3, 2, 1),      # stride 2 on odd sizes
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_conv_fwd_plain(case):
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = case
    rng, x, wt = _mk(n, ci, h, w, co, k, 1)
    ref = R.conv2d(x, wt, s, p)
    co_pad = ops.round_up(co, 32)
    out = torch.full((n, ref.shape[2], ref.shape[3], co_pad), 7.0, device="cuda")
    ops.conv_fwd(nchw_to_dev_nhwc(x), _packed(wt, co_pad), out, k=k, stride=s, pad=p, Co=co_pad, ldo=co_pad)
    torch.cuda.synchronize()
    got = dev_nhwc_to_nchw(out, co)
    assert maxdiff(got, ref) < TOL
    if co_pad > co:   # pad channels come out as exact zeros (zero weight rows)
        assert float(out[..., co:].abs().max()) == 0.0


def test_conv_fwd_epilogue_affine_leaky_residual():
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = 2, 64, 26, 26, 128, 3, 1, 1
    rng, x, wt = _mk(n, ci, h, w, co, k, 2)
    scale = rng.uniform(0.5, 1.5, co)
    shift = rng.standard_normal(co)
    res = rng.standard_normal((n, co, h, w))
    z = R.conv2d(x, wt, s, p) * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
    ref = R.leaky(z) + res
    out = torch.empty(n, h, w, co, device="cuda")
    ops.conv_fwd(nchw_to_dev_nhwc(x), _packed(wt, co), out, k=k, stride=s, pad=p, Co=co, scale=dev(scale),
                 shift=dev(shift), residual=nchw_to_dev_nhwc(res), leaky=True)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out), ref) < TOL


def test_conv_fwd_bias_only():
    from viddet_amd import ops
    n, ci, h, w, co, k = 2, 128, 13, 13, 75, 1
    rng, x, wt = _mk(n, ci, h, w, co, k, 3)
    bias = rng.standard_normal(co)
    ref = R.conv2d(x, wt, 1, 0, bias)
    co_pad = 96
    bpad = np.zeros(co_pad); bpad[:co] = bias
    out = torch.empty(n, h, w, co_pad, device="cuda")
    ops.conv_fwd(nchw_to_dev_nhwc(x), _packed(wt, co_pad), out, k=k, stride=1, pad=0, Co=co_pad, shift=dev(bpad))
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out, co), ref) < TOL


DGRAD_CASES = [
    (2, 32, 16, 16, 64, 3, 1, 1),
    (2, 64, 13, 13, 32, 1, 1, 0),
    (2, 32, 20, 20, 64, 3, 2, 1),
    (1, 64, 15, 17, 128, 3, 2, 1),
    (2, 96, 9, 9, 75, 1, 1, 0),        # head-like: Co padded to 96 on the dZ side
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv_dgrad(case):
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = case
    rng, x, wt = _mk(n, ci, h, w, co, k, 4)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    dy = rng.standard_normal((n, co, ho, wo))
    dx_ref, _ = R.conv2d_backward(x, wt, dy, s, p)
    co_pad = ops.round_up(co, 32)
    dz = nchw_to_dev_nhwc(dy, co_pad)
    dx = torch.full((n, h, w, ci), 3.0, device="cuda")
    wdev = dev(wt)
    for plan in ops.dgrad_plans(k, p, s, h, w):
        if not plan["taps"]:
            # no tap reaches this parity class: gradient is zero there
            dx[:, plan["py"]::s, plan["px"]::s, :] = 0
            continue
        wp = torch.empty(ci, len(plan["taps"]) * co_pad, device="cuda")
        ops.pack_weight_dgrad(wdev, wp, Co=co, Co_pad=co_pad, Ci=ci, kd=1, kh=k, kw=k, tap_ids=plan["tap_ids"],
                              src_packed=False)
        ops.conv_igemm(dz, wp, dx, N=n, Hi=ho, Wi=wo, Ci=co_pad, Hg=plan["Hg"], Wg=plan["Wg"], in_stride=1,
                       taps=plan["taps"], Ho=h, Wo=w, Co=ci, ldo=ci, out_stride=s, out_oy=plan["py"],
                       out_ox=plan["px"])
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(dx), dx_ref) < TOL


def test_conv_dgrad_accumulate_and_packed_source():
    """dgrad from the fwd-packed weight layout, accumulating into an existing gradient."""
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = 2, 64, 12, 12, 128, 3, 1, 1
    rng, x, wt = _mk(n, ci, h, w, co, k, 5)
    dy = rng.standard_normal((n, co, h, w))
    prev = rng.standard_normal((n, ci, h, w))
    dx_ref, _ = R.conv2d_backward(x, wt, dy, s, p)
    dx = nchw_to_dev_nhwc(prev)
    wpf = _packed(wt, co)
    plan = ops.dgrad_plans(k, p, s, h, w)[0]
    wp = torch.empty(ci, len(plan["taps"]) * co, device="cuda")
    ops.pack_weight_dgrad(wpf, wp, Co=co, Co_pad=co, Ci=ci, kd=1, kh=k, kw=k, tap_ids=plan["tap_ids"], src_packed=True)
    ops.conv_igemm(nchw_to_dev_nhwc(dy), wp, dx, N=n, Hi=h, Wi=w, Ci=co, Hg=h, Wg=w, in_stride=1, taps=plan["taps"],
                   Ho=h, Wo=w, Co=ci, ldo=ci, residual=dx, ldr=ci)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(dx), dx_ref + prev) < TOL


WGRAD_CASES = [
    (2, 32, 16, 16, 64, 3, 1, 1, 0),
    (2, 64, 13, 13, 32, 1, 1, 0, 0),
    (2, 32, 20, 20, 64, 3, 2, 1, 0),
    (4, 128, 26, 26, 256, 3, 1, 1, 0),
    (2, 160, 9, 11, 96, 1, 1, 0, 1),   # Ci, Co not multiples of 128; single split
    (2, 64, 15, 17, 128, 3, 2, 1, 7),  # forced odd split count
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(case):
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p, splits = case
    rng, x, wt = _mk(n, ci, h, w, co, k, 6)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    dy = rng.standard_normal((n, co, ho, wo))
    _, dw_ref = R.conv2d_backward(x, wt, dy, s, p)
    dwp = torch.empty(co, k * k * ci, device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ops.conv_wgrad(nchw_to_dev_nhwc(x), nchw_to_dev_nhwc(dy), dwp, ws, k=k, stride=s, pad=p, Co=co, splits=splits)
    dw = torch.empty(co, ci, k, k, device="cuda")
    ops.unpack_weight(dwp, dw)
    torch.cuda.synchronize()
    scale = np.sqrt(n * ho * wo)
    assert maxdiff(dw.cpu().numpy(), dw_ref) < TOL * scale


XF_CASES = [(2, 64, 13, 11, 128, 3, 1, 1), (3, 32, 9, 12, 64, 1, 1, 0), (2, 64, 14, 15, 96, 3, 2, 1), (2, 128, 8, 8, 256, 3, 1, 1)]


@pytest.mark.parametrize("mode", [False, True], ids=["fp32-mfma", "bf16x3"])
@pytest.mark.parametrize("case", XF_CASES)
def test_in_load_transform_forward_and_wgrad(case, mode):
    """The header's in-load transform (vd_conv_desc / vd_wgrad_desc in_scale, in_shift, in_slope): the activation operand is
    leaky(x * s[c] + b[c]) applied in the gather, zero padding staying zero - i.e. the conv / weight gradient of the
    transformed tensor (what a consumer would read if the producer's BatchNorm+LeakyReLU pass were skipped).  Off the
    product path, but part of the C-ABI; the fp16 split has no such variant and must say so."""
    from viddet_amd import ops, lib as L
    n, ci, h, w, co, k, s, p = case
    rng, x, wt = _mk(n, ci, h, w, co, k, 31)
    sc, sh, slope = rng.uniform(0.5, 1.5, ci), rng.standard_normal(ci), 0.1
    xt = R.leaky(x * sc.reshape(1, -1, 1, 1) + sh.reshape(1, -1, 1, 1), slope)
    ref = R.conv2d(xt, wt, s, p)
    ho, wo = ref.shape[2:]
    out = torch.empty(n, ho, wo, co, device="cuda")
    xd = nchw_to_dev_nhwc(x)
    for tile in ((0, 1, 4) if not mode else (0, 2, 4, 11, 16 if co <= 64 else 12)):
        out.zero_()
        ops.conv_fwd(xd, _packed(wt, co), out, k=k, stride=s, pad=p, Co=co, tile=tile, split=mode,
                     in_scale=dev(sc), in_shift=dev(sh), in_slope=slope)
        torch.cuda.synchronize()
        assert maxdiff(dev_nhwc_to_nchw(out), ref) < TOL, tile
    dy = rng.standard_normal((n, co, ho, wo))
    _, dw_ref = R.conv2d_backward(xt, wt, dy, s, p)
    dwp = torch.empty(co, k * k * ci, device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ops.conv_wgrad(xd, nchw_to_dev_nhwc(dy), dwp, ws, k=k, stride=s, pad=p, Co=co, split=mode,
                   in_scale=dev(sc), in_shift=dev(sh), in_slope=slope)
    dw = torch.empty(co, ci, k, k, device="cuda")
    ops.unpack_weight(dwp, dw)
    torch.cuda.synchronize()
    assert maxdiff(dw.cpu().numpy(), dw_ref) < TOL * np.sqrt(n * ho * wo)
    with pytest.raises(L.VidDetHipError):
        ops.conv_fwd(xd, _packed(wt, co), out, k=k, stride=s, pad=p, Co=co, split='f16x2', in_scale=dev(sc), in_shift=dev(sh))


def test_stem_im2col_conv_and_wgrad():
    from viddet_amd import ops
    n, h, w, co = 2, 24, 20, 32
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, 3, h, w))
    wt = rng.standard_normal((co, 3, 3, 3)) / np.sqrt(27)
    ref = R.conv2d(x, wt, 1, 1)
    col = torch.empty(n, h, w, 32, device="cuda")
    ops.stem_im2col(dev(x), col, nchw=True)
    col2 = torch.empty(n, h, w, 32, device="cuda")
    ops.stem_im2col(nchw_to_dev_nhwc(x), col2, nchw=False)
    # stem weight as a 1x1 conv over the 32-wide im2col rows: wp[co][(ky*3+kx)*3+c]
    wp_np = np.zeros((co, 32))
    wp_np[:, :27] = wt.transpose(0, 2, 3, 1).reshape(co, 27)
    wp = dev(wp_np)
    out = torch.empty(n, h, w, co, device="cuda")
    ops.conv_fwd(col, wp, out, k=1, stride=1, pad=0, Co=co)
    torch.cuda.synchronize()
    assert float((col - col2).abs().max()) == 0.0
    assert maxdiff(dev_nhwc_to_nchw(out), ref) < TOL
    dy = rng.standard_normal((n, co, h, w))
    _, dw_ref = R.conv2d_backward(x, wt, dy, 1, 1)
    dwp = torch.empty(co, 32, device="cuda")
    ws = torch.empty(16 << 20, dtype=torch.uint8, device="cuda")
    ops.conv_wgrad(col, nchw_to_dev_nhwc(dy), dwp, ws, k=1, stride=1, pad=0, Co=co)
    torch.cuda.synchronize()
    got = dwp.cpu().numpy()[:, :27].reshape(co, 3, 3, 3).transpose(0, 3, 1, 2)
    assert maxdiff(got, dw_ref) < TOL * np.sqrt(n * h * w)
    assert float(dwp[:, 27:].abs().max()) == 0.0


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("shape", [(2, 64, 8, 8, 128, 3, 1, 1), (6, 128, 8, 8, 256, 3, 1, 1), (3, 32, 13, 11, 96, 1, 1, 0),
                                   (2, 64, 17, 15, 160, 3, 2, 1)])
def test_conv_fwd_every_tile_variant(tile, shape):
    """Every k_conv_igemm tile variant the autotuner may pick (vd_conv_desc.tile = 1..8) on ragged shapes."""
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = shape
    rng, x, wt = _mk(n, ci, h, w, co, k, 40 + tile)
    res = rng.standard_normal((n, co, (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1))
    ref = R.conv2d(x, wt, s, p) + res
    co_pad = ops.round_up(co, 32)
    out = torch.full((n, ref.shape[2], ref.shape[3], co_pad), 7.0, device="cuda")
    ops.conv_fwd(nchw_to_dev_nhwc(x), _packed(wt, co_pad), out, k=k, stride=s, pad=p, Co=co_pad, ldo=co_pad,
                 residual=nchw_to_dev_nhwc(res, co_pad), tile=tile)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out, co), ref) < TOL


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8])
def test_conv_fused_bn_statistics(tile):
    """Per-channel sum / sum-of-squares produced by the conv epilogue == statistics of the conv output."""
    import ctypes as C
    from viddet_amd import ops, lib as L
    n, ci, h, w, co, k, s, p = 3, 64, 13, 11, 96, 3, 1, 1
    rng, x, wt = _mk(n, ci, h, w, co, k, 60 + tile)
    ref = R.conv2d(x, wt, s, p)
    out = torch.empty(n, h, w, co, device="cuda")
    xd, wp = nchw_to_dev_nhwc(x), _packed(wt, co)
    d = L.ConvDesc()
    d.in_, d.wp, d.out = xd.data_ptr(), wp.data_ptr(), out.data_ptr()
    d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = n, h, w, ci, h, w, s
    ops._set_taps(d, ops.fwd_taps(k, p))
    d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.ldo, d.ldr, d.tile = 1, h, w, co, 1, co, co, tile
    mt = L.load().vd_conv_igemm_mtiles(C.byref(d))
    part = torch.full((mt, 2 * co), 9.0, device="cuda")
    d.stats_part = part.data_ptr()
    L.check(L.load().vd_conv_igemm(C.byref(d), L.stream_ptr()), "vd_conv_igemm")
    sums = torch.empty(2 * co, dtype=torch.float64, device="cuda")
    L.check(L.load().vd_bn_sum_partials(part.data_ptr(), mt, co, sums.data_ptr(), None, 0, L.stream_ptr()), "vd_bn_sum_partials")
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out), ref) < TOL
    sv = sums.cpu().numpy()
    assert maxdiff(sv[:co], ref.sum(axis=(0, 2, 3))) < 1e-3
    assert maxdiff(sv[co:], (ref ** 2).sum(axis=(0, 2, 3))) < 1e-3 * max(1.0, float((ref ** 2).sum(axis=(0, 2, 3)).max()) / 100)


def test_bn_sum_partials_tall_table():
    """Two-level path (more than 256 partial rows) of the fused-statistics reduction."""
    from viddet_amd import lib as L
    rng = np.random.default_rng(70)
    nblk, c = 5000, 48
    part = rng.standard_normal((nblk, 2 * c)).astype(np.float32)
    pd = dev(part)
    sums = torch.empty(2 * c, dtype=torch.float64, device="cuda")
    need = L.load().vd_bn_sum_partials_ws_bytes(nblk, c)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    L.check(L.load().vd_bn_sum_partials(pd.data_ptr(), nblk, c, sums.data_ptr(), ws.data_ptr(), need, L.stream_ptr()), "sum")
    torch.cuda.synchronize()
    assert maxdiff(sums.cpu().numpy(), part.astype(np.float64).sum(axis=0)) < 1e-9


@pytest.mark.parametrize("shape", [(2, 16, 16), (3, 13, 21), (1, 64, 64), (5, 7, 9)])
def test_stem_direct_conv_forward_wgrad_stats(shape):
    """vd_stem_conv / vd_stem_wgrad (3x3, 3 -> 32 from the NCHW batch, no im2col) vs the oracle conv; the fused
    per-block BatchNorm partial sums; bf16 output at its rounding."""
    import ctypes as C
    from viddet_amd import ops, lib as L
    n, h, w = shape
    rng, x, wt = _mk(n, 3, h, w, 32, 3, 80 + n)
    ref = R.conv2d(x, wt, 1, 1)
    xd = dev(x)                                                   # (n,3,h,w) NCHW
    # k index of the packed stem weight: (ky*3+kx)*3 + c
    wk = np.zeros((32, 32), np.float32)
    for ky in range(3):
        for kx in range(3):
            for c in range(3):
                wk[:, (ky * 3 + kx) * 3 + c] = wt[:, c, ky, kx]
    wp = dev(wk)
    out = torch.full((n, h, w, 32), 7.0, device="cuda")
    nb = L.load().vd_stem_conv_blocks(n, h, w)
    part = torch.full((nb, 64), 9.0, device="cuda")
    ops.stem_conv(xd, wp, out, stats_part=part)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out), ref) < TOL
    assert maxdiff(part[:, :32].double().sum(0).cpu().numpy(), ref.sum((0, 2, 3))) < 1e-3
    assert maxdiff(part[:, 32:].double().sum(0).cpu().numpy(), (ref ** 2).sum((0, 2, 3))) < 1e-3 * max(1.0, float((ref ** 2).sum((0, 2, 3)).max()) / 100)
    # BN-eval fold + LeakyReLU epilogue, fp32 and bf16 outputs
    sc, sh = rng.uniform(0.5, 1.5, 32), rng.standard_normal(32)
    u = ref * sc[None, :, None, None] + sh[None, :, None, None]
    refa = np.where(u > 0, u, 0.1 * u)
    ops.stem_conv(xd, wp, out, scale=dev(sc), shift=dev(sh), leaky=True)
    ob = torch.zeros(n, h, w, 64, device="cuda", dtype=torch.bfloat16)       # pitch 64: pad channels stay untouched
    ops.stem_conv(xd, wp, ob, scale=dev(sc), shift=dev(sh), leaky=True)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out), refa) < TOL
    # bf16 output = the matrix-pipe form (two bf16 MFMAs per 32 pixels on the frame values and weights rounded to bf16, fp32
    # accumulation): exact against the oracle on the ROUNDED operands up to fp32 summation and one rounding of the output,
    # and within bf16 operand accuracy of the fp32 result
    rb = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)
    ub = R.conv2d(rb(x), rb(wt), 1, 1) * sc[None, :, None, None] + sh[None, :, None, None]
    refb = np.where(ub > 0, ub, 0.1 * ub)
    gotb = dev_nhwc_to_nchw(ob.float(), 32)
    assert maxdiff(gotb, refb) < 2e-4 + 2.0 ** -8 * float(np.abs(refb).max())
    assert maxdiff(gotb, refa) < 3e-2 * max(1.0, float(np.abs(refa).max()))
    assert float(ob[..., 32:].float().abs().max()) == 0.0
    # weight gradient
    dy = rng.standard_normal((n, 32, h, w))
    _, dw_ref = R.conv2d_backward(x, wt, dy, 1, 1)
    dwp = torch.full((32, 32), 5.0, device="cuda")
    ws = torch.empty(max(16, L.load().vd_stem_wgrad_ws_bytes(n, h, w)), dtype=torch.uint8, device="cuda")
    ops.stem_wgrad(xd, nchw_to_dev_nhwc(dy), dwp, ws)
    torch.cuda.synchronize()
    got = dwp.cpu().numpy()
    dwk = np.zeros((32, 32))
    for ky in range(3):
        for kx in range(3):
            for c in range(3):
                dwk[:, (ky * 3 + kx) * 3 + c] = dw_ref[:, c, ky, kx]
    assert maxdiff(got, dwk) < TOL * np.sqrt(n * h * w)
    assert float(np.abs(got[:, 27:]).max()) == 0.0


def test_stem_im2col_bf16_and_nms_workspace_size():
    """Two exports no other test reaches directly: the bf16 form of the explicit stem lowering (64 columns per pixel, 27
    real) equals the fp32 one rounded to bf16, and the NMS workspace is one overflow flag per image."""
    import ctypes as C
    from viddet_amd import ops, lib as L
    n, h, w = 2, 11, 9
    rng = np.random.default_rng(8)
    x = rng.standard_normal((n, 3, h, w))
    col32 = torch.empty(n, h, w, 32, device="cuda")
    ops.stem_im2col(dev(x), col32, nchw=True)
    for nchw, src in ((1, dev(x)), (0, nchw_to_dev_nhwc(x))):
        col = torch.full((n, h, w, 64), 7.0, dtype=torch.bfloat16, device="cuda")
        L.check(L.load().vd_stem_im2col_bf16(src.data_ptr(), col.data_ptr(), n, h, w, nchw, L.stream_ptr()), "vd_stem_im2col_bf16")
        torch.cuda.synchronize()
        assert torch.equal(col[..., :27], col32[..., :27].to(torch.bfloat16)) and float(col[..., 27:].abs().max()) == 0.0
    assert L.load().vd_nms_ws_bytes(5, 1000, 400) == 5 * 4


@pytest.mark.parametrize("tile", [5, 1, 6, 2, 12, 11])
@pytest.mark.parametrize("case", [(3, 32, 16, 64), (2, 64, 12, 128), (2, 128, 8, 256), (5, 32, 26, 64)])
def test_conv_dgrad_stride2_fused_one_launch(case, tile):
    """VD_CONV_PARITY4 (vd_conv_par.hip): the data gradient of a 3x3 / stride-2 / pad-1 conv as ONE launch - GEMM columns =
    (parity class, channel), taps = the four offsets of a 2x2 window on dz's grid, zero weight blocks skipped - against the
    fp64 oracle, accumulating into an existing gradient (residual epilogue), with the fused BatchNorm-backward reductions
    checked against the same sums taken on the host from the result."""
    import ctypes as C
    from viddet_amd import ops, lib as L
    n, ci, ho, co = case                 # the conv: ci -> co, input 2 ho x 2 ho, output ho x ho
    h = 2 * ho
    rng, x, wt = _mk(n, ci, h, h, co, 3, 40 + tile)
    dy = rng.standard_normal((n, co, ho, ho))
    prev = rng.standard_normal((n, ci, h, h))
    dx_ref, _ = R.conv2d_backward(x, wt, dy, 2, 1)
    wpf = _packed(wt, co)
    wp4 = torch.empty(4 * ci, 4 * co, device="cuda")
    mask = ops.pack_weight_dgrad_s2(wpf, wp4, Co=co, Co_pad=co, Ci=ci)
    assert mask == ops.PARITY4_MASK and bin(mask).count("1") == 9
    dz = nchw_to_dev_nhwc(dy)
    dx = nchw_to_dev_nhwc(prev)
    ds = []
    ops.conv_dgrad_s2_fused(dz, wp4, dx, Cin=ci, par_mask=mask, residual=dx, tile=tile, desc_out=ds)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(dx), dx_ref + prev) < TOL
    # the same launch carrying the fused BatchNorm-backward reductions of the producer of `x`
    z = rng.standard_normal((n, ci, h, h))
    bsc, bsh, bmu, bis = rng.uniform(0.5, 1.5, ci), rng.standard_normal(ci), rng.standard_normal(ci), rng.uniform(0.5, 2.0, ci)
    d, ax, aw = ds[0]
    dx2 = nchw_to_dev_nhwc(prev)
    zd = nchw_to_dev_nhwc(z)
    vec = [dev(v) for v in (bsc, bsh, bmu, bis)]
    d.out, d.residual, d.bs_z = dx2.data_ptr(), dx2.data_ptr(), zd.data_ptr()
    d.bs_scale, d.bs_shift, d.bs_mean, d.bs_invstd = [v.data_ptr() for v in vec]
    rows = L.load().vd_conv_igemm_mtiles(C.byref(d))
    part = torch.zeros(rows, 2 * ci, device="cuda")
    d.bs_part, d.bs_slope = part.data_ptr(), 0.1
    L.check(L.load().vd_conv_igemm(C.byref(d), L.stream_ptr()), "vd_conv_igemm")
    torch.cuda.synchronize()
    g_all = dx_ref + prev
    assert maxdiff(dev_nhwc_to_nchw(dx2), g_all) < TOL
    u = z * bsc.reshape(1, -1, 1, 1) + bsh.reshape(1, -1, 1, 1)
    g = np.where(u > 0, g_all, 0.1 * g_all)
    s1 = g.sum(axis=(0, 2, 3))
    s2 = (g * (z - bmu.reshape(1, -1, 1, 1)) * bis.reshape(1, -1, 1, 1)).sum(axis=(0, 2, 3))
    got = part.double().sum(0).cpu().numpy()
    tol = 2e-4 * np.sqrt(n * h * h) * 4
    assert np.abs(got[:ci] - s1).max() < tol and np.abs(got[ci:] - s2).max() < tol * 4
