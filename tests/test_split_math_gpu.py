"""GPU parity of the split-operand fp32 product arithmetics (include/viddet_hip.h).  VD_MATH_SPLIT: every fp32
operand is split exactly into three bf16 pieces and six partial products are accumulated in fp32 on the bf16 matrix
pipe.  VD_MATH_F16X2 ('f16x2' below): every operand is scaled by a per-tensor power of two and split into two fp16
pieces, three partial products.  Both are checked against the fp64 oracle (oracle/ops.py) with the SAME tolerance as
the fp32-MFMA kernels (2e-4 abs on O(1) outputs), plus the claim the design rests on: their error against fp64 is not
larger than the fp32 MFMA's (the fp32 fma chain) on the same data - for the fp16 form also on tensors far outside the
fp16 range and on tensors with outliers (the scale comes from the tensor's max-abs)."""
import numpy as np
import pytest
import torch

from oracle import ops as R
from tests.util import dev, nchw_to_dev_nhwc, dev_nhwc_to_nchw, maxdiff
from tests.test_conv_gpu import _mk, _packed

pytestmark = pytest.mark.gpu
TOL = 2e-4

SHAPES = [  # n, ci, h, w, co, k, stride, pad
    (2, 64, 8, 8, 128, 3, 1, 1), (6, 128, 8, 8, 256, 3, 1, 1), (3, 32, 13, 11, 96, 1, 1, 0),
    (2, 64, 17, 15, 160, 3, 2, 1), (5, 96, 21, 19, 320, 3, 1, 1), (1, 64, 19, 19, 255, 1, 1, 0),
    (7, 64, 13, 13, 128, 3, 1, 1),        # several 13x13 images per 256-pixel tile: halo rows cross image borders
    (1, 32, 104, 104, 64, 3, 1, 1),       # the widest map the halo loop takes (halo = 466 rows, one 32-channel chunk)
    (2, 160, 52, 52, 96, 3, 1, 1),        # five chunks: the stream wraps the halo buffers
]


@pytest.mark.parametrize("mode", [True, "f16x2", "f16x2nh"])      # f16x2: halo-staged loop on the 3x3 stride-1 shapes
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 15, 16])      # 5..8: the same tiles on the 16x16x32 MFMA shape; 11, 12: 128x128 as four waves, both shapes
@pytest.mark.parametrize("shape", SHAPES)
def test_split_fwd_every_tile(tile, shape, mode):
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = shape
    rng, x, wt = _mk(n, ci, h, w, co, k, 70 + tile)
    res = rng.standard_normal((n, co, (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1))
    sc, sh = rng.uniform(0.5, 1.5, co), rng.standard_normal(co)
    u = R.conv2d(x, wt, s, p) * sc[None, :, None, None] + sh[None, :, None, None]
    ref = np.where(u > 0, u, 0.1 * u) + res
    co_pad = ops.round_up(co, 32)
    out = torch.full((n, ref.shape[2], ref.shape[3], co_pad), 7.0, device="cuda")
    scd, shd = torch.zeros(co_pad, device="cuda"), torch.zeros(co_pad, device="cuda")
    scd[:co], shd[:co] = dev(sc), dev(sh)
    ops.conv_fwd(nchw_to_dev_nhwc(x), _packed(wt, co_pad), out, k=k, stride=s, pad=p, Co=co_pad, ldo=co_pad,
                 scale=scd, shift=shd, leaky=True, residual=nchw_to_dev_nhwc(res, co_pad), tile=tile, split=mode)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out, co), ref) < TOL


@pytest.mark.parametrize("mode", [True, "f16x2"])
@pytest.mark.parametrize("tile", [9, 10, 13, 14])
@pytest.mark.parametrize("shape", [(2, 64, 23, 19, 32, 1, 1, 0), (3, 64, 17, 15, 32, 3, 1, 1), (2, 96, 20, 22, 24, 3, 2, 1),
                                   (5, 32, 9, 31, 32, 3, 1, 1)])
def test_split_fwd_32_column_tiles(tile, shape, mode):
    """Tiles 9 / 10: 256 x 32 (8 waves of 32x32; half of the loader lanes carry no weight row) for outputs of up to 32
    channels - the data gradients of the first-stage 3x3 convs and the 1x1 64->32 forward."""
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = shape
    rng, x, wt = _mk(n, ci, h, w, co, k, 270 + tile)
    res = rng.standard_normal((n, co, (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1))
    sc, sh = rng.uniform(0.5, 1.5, co), rng.standard_normal(co)
    u = R.conv2d(x, wt, s, p) * sc[None, :, None, None] + sh[None, :, None, None]
    ref = np.where(u > 0, u, 0.1 * u) + res
    co_pad = ops.round_up(co, 32)
    out = torch.full((n, ref.shape[2], ref.shape[3], co_pad), 7.0, device="cuda")
    scd, shd = torch.zeros(co_pad, device="cuda"), torch.zeros(co_pad, device="cuda")
    scd[:co], shd[:co] = dev(sc), dev(sh)
    ops.conv_fwd(nchw_to_dev_nhwc(x), _packed(wt, co_pad), out, k=k, stride=s, pad=p, Co=co_pad, ldo=co_pad,
                 scale=scd, shift=shd, leaky=True, residual=nchw_to_dev_nhwc(res, co_pad), tile=tile, split=mode)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out, co), ref) < TOL


@pytest.mark.parametrize("case", [(2, 64, 12, 12, 128, 3, 1, 1), (2, 32, 20, 20, 64, 3, 2, 1), (1, 64, 15, 17, 128, 3, 2, 1),
                                  (3, 256, 13, 13, 128, 1, 1, 0), (2, 96, 9, 9, 75, 1, 1, 0), (3, 64, 26, 26, 128, 3, 1, 1),
                                  (1, 128, 104, 104, 64, 3, 1, 1)])
@pytest.mark.parametrize("mode", [True, "f16x2", "f16x2nh"])
def test_split_dgrad(case, mode):
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
            dx[:, plan["py"]::s, plan["px"]::s, :] = 0
            continue
        wp = torch.empty(ci, len(plan["taps"]) * co_pad, device="cuda")
        ops.pack_weight_dgrad(wdev, wp, Co=co, Co_pad=co_pad, Ci=ci, kd=1, kh=k, kw=k, tap_ids=plan["tap_ids"],
                              src_packed=False)
        ops.conv_igemm(dz, wp, dx, N=n, Hi=ho, Wi=wo, Ci=co_pad, Hg=plan["Hg"], Wg=plan["Wg"], in_stride=1,
                       taps=plan["taps"], Ho=h, Wo=w, Co=ci, ldo=ci, out_stride=s, out_oy=plan["py"],
                       out_ox=plan["px"], split=mode)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(dx), dx_ref) < TOL


@pytest.mark.parametrize("case", [
    (4, 128, 26, 26, 256, 3, 1, 1, 0),   # 256-row tile
    (2, 64, 15, 17, 128, 3, 2, 1, 7),    # 128-row tile, forced odd split count, ragged pixel ranges
    (2, 160, 9, 11, 288, 1, 1, 0, 1),    # Ci, Co not multiples of the tiles; single split
    (3, 32, 14, 14, 128, 3, 1, 1, 0),    # four taps share one 128-wide column tile
    (2, 64, 13, 13, 64, 3, 1, 1, 0),     # 64-row tile (128-B plane rows, two-chunk swizzle)
    (3, 32, 21, 19, 64, 3, 2, 1, 5),     # same tile, stride 2, ragged
    (2, 64, 13, 13, 32, 1, 1, 0, 0),     # Co < 64: stays on the fp32 MFMA even when the flag is set
])
@pytest.mark.parametrize("mode", [True, "f16x2"])
def test_split_wgrad(case, mode):
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p, splits = case
    rng, x, wt = _mk(n, ci, h, w, co, k, 6)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    dy = rng.standard_normal((n, co, ho, wo))
    _, dw_ref = R.conv2d_backward(x, wt, dy, s, p)
    dwp = torch.empty(co, k * k * ci, device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ops.conv_wgrad(nchw_to_dev_nhwc(x), nchw_to_dev_nhwc(dy), dwp, ws, k=k, stride=s, pad=p, Co=co, splits=splits,
                   split=mode)
    dw = torch.empty(co, ci, k, k, device="cuda")
    ops.unpack_weight(dwp, dw)
    torch.cuda.synchronize()
    assert maxdiff(dw.cpu().numpy(), dw_ref) < TOL * np.sqrt(n * ho * wo)


@pytest.mark.parametrize("mode", [True, "f16x2"])
@pytest.mark.parametrize("xs,ws_", [(1.0, 1.0), (2.0 ** -60, 2.0 ** -30), (3.0e7, 1.0e-9), (1.0, 4.0e4)])
def test_split_error_not_above_fp32_mfma(mode, xs, ws_):
    """The accuracy claim: on a deep reduction (K = 9 * 512) the rms error of the split-operand products against
    fp64 is no larger than that of the fp32 MFMA (an exact fp32 fma chain) - forward and weight gradient.  The
    operand magnitudes (xs, ws_) range far outside fp16's 6e-8 .. 65504: the fp16 form scales by the tensors' max-abs."""
    from viddet_amd import ops
    n, ci, h, w, co, k = 4, 512, 13, 13, 256, 3
    rng, x, wt = _mk(n, ci, h, w, co, k, 99)
    x, wt = x * xs, wt * ws_
    xd, wp = nchw_to_dev_nhwc(x), _packed(wt, co)
    x32 = xd.permute(0, 3, 1, 2).double().cpu().numpy()           # the fp32-rounded operands the kernels see
    w32 = dev(wt).double().cpu().numpy()
    ref = R.conv2d(x32, w32, 1, 1)
    err = {}
    for split in (False, mode):
        out = torch.empty(n, h, w, co, device="cuda")
        ops.conv_fwd(xd, wp, out, k=k, stride=1, pad=1, Co=co, split=split)
        torch.cuda.synchronize()
        e = dev_nhwc_to_nchw(out) - ref
        err[split] = float(np.sqrt((e ** 2).mean()))
    rms = float(np.sqrt((ref ** 2).mean()))
    print("fwd rms error / rms: fp32 MFMA %.3e, %s %.3e" % (err[False] / rms, mode, err[mode] / rms))
    assert err[mode] <= 1.25 * err[False] + 1e-9 * rms, err
    assert err[mode] < 2e-5 * rms
    # weight gradient: reduction over n*h*w = 676 pixels
    dy = rng.standard_normal((n, co, h, w)) * ws_
    dyd = nchw_to_dev_nhwc(dy)
    dy32 = dyd.permute(0, 3, 1, 2).double().cpu().numpy()
    _, dw_ref = R.conv2d_backward(x32, w32, dy32, 1, 1)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for split in (False, mode):
        dwp = torch.empty(co, k * k * ci, device="cuda")
        ops.conv_wgrad(xd, dyd, dwp, ws, k=k, stride=1, pad=1, Co=co, split=split)
        dw = torch.empty(co, ci, k, k, device="cuda")
        ops.unpack_weight(dwp, dw)
        torch.cuda.synchronize()
        e = dw.double().cpu().numpy() - dw_ref
        err[split] = float(np.sqrt((e ** 2).mean()))
    rms = float(np.sqrt((dw_ref ** 2).mean()))
    print("wgrad rms error / rms: fp32 MFMA %.3e, %s %.3e" % (err[False] / rms, mode, err[mode] / rms))
    assert err[mode] <= 1.25 * err[False] + 1e-9 * rms, err


def test_f16x2_outliers_and_degenerate_tensors():
    """A tensor's scale comes from its LARGEST element: with one element in a thousand 10^4 times the rest, the small
    elements sit 13 binades below the top of the fp16 window and still keep their 22 bits (full precision down to
    2^-19 of the max); an all-zero operand and a single huge element are handled (scale 1 / exact)."""
    from viddet_amd import ops
    n, ci, h, w, co, k = 2, 256, 13, 13, 128, 3
    rng, x, wt = _mk(n, ci, h, w, co, k, 77)
    x = np.where(rng.random(x.shape) < 1e-3, x * 1e4, x)
    xd, wp = nchw_to_dev_nhwc(x), _packed(wt, co)
    x32 = xd.permute(0, 3, 1, 2).double().cpu().numpy()
    w32 = dev(wt).double().cpu().numpy()
    ref = R.conv2d(x32, w32, 1, 1)
    err = {}
    for split in (False, "f16x2"):
        out = torch.empty(n, h, w, co, device="cuda")
        ops.conv_fwd(xd, wp, out, k=k, stride=1, pad=1, Co=co, split=split)
        torch.cuda.synchronize()
        # the outputs the outliers do not reach show what the small elements kept
        e = (dev_nhwc_to_nchw(out) - ref)
        err[split] = float(np.sqrt((e ** 2).mean()))
        small = np.abs(ref) < 50
        err[(split, 's')] = float(np.sqrt((e[small] ** 2).mean()))
    print(err)
    assert err["f16x2"] <= 1.25 * err[False] and err[("f16x2", 's')] <= 1.5 * err[(False, 's')] + 1e-7
    zero = torch.zeros_like(xd)
    out = torch.full((n, h, w, co), 3.0, device="cuda")
    ops.conv_fwd(zero, wp, out, k=k, stride=1, pad=1, Co=co, split="f16x2")
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0
    one = torch.zeros_like(xd)
    one[0, 5, 5, 7] = 3.0e38
    ops.conv_fwd(one, wp, out, k=k, stride=1, pad=1, Co=co, split="f16x2")
    torch.cuda.synchronize()
    got = out[0, 5, 5, :].double().cpu().numpy()
    want = 3.0e38 * w32[:, 7, 1, 1]
    ok = np.isfinite(want.astype(np.float32))
    assert np.allclose(got[ok], want[ok], rtol=5e-7, atol=0)       # one product: 2^-23 per operand


def test_amax_slots_from_every_producer():
    """The max-abs exchange of the fp16 arithmetic: vd_amax, vd_amax_segments, vd_amax_merge, the conv epilogue, the two
    BatchNorm streaming kernels and the loss kernel all publish exactly max|tensor| into the tensor's slots."""
    from viddet_amd import ops, lib as L
    import ctypes as C
    rng = np.random.default_rng(5)
    lib = L.load()
    x = dev(rng.standard_normal((3, 17, 19, 64)) * 3.0)
    assert ops.amax_value(ops.amax(x)) == float(x.abs().max())
    arena = dev(rng.standard_normal(5000))
    seg = torch.tensor([[0, 1000], [1000, 37], [1037, 3963]], dtype=torch.int64, device="cuda")
    am = torch.empty(3 * L.AMAX_FLOATS, device="cuda")
    L.check(lib.vd_amax_segments(arena.data_ptr(), seg.data_ptr(), 3, am.data_ptr(), L.stream_ptr()), "vd_amax_segments")
    for i, (o, c) in enumerate(seg.cpu().numpy()):
        assert ops.amax_value(am[i * L.AMAX_FLOATS:(i + 1) * L.AMAX_FLOATS]) == float(arena[o:o + c].abs().max())
    mg = torch.empty(L.AMAX_FLOATS, device="cuda")
    L.check(lib.vd_amax_merge(am.data_ptr(), am[L.AMAX_FLOATS:].data_ptr(), mg.data_ptr(), L.stream_ptr()), "vd_amax_merge")
    assert ops.amax_value(mg) == float(arena[:1037].abs().max())
    # conv epilogue
    n, ci, h, w, co, k = 2, 64, 11, 9, 96, 3
    _, xx, wt = _mk(n, ci, h, w, co, k, 3)
    out = torch.empty(n, h, w, co, device="cuda")
    for mode in (False, True, "f16x2"):
        slot = torch.zeros(L.AMAX_FLOATS, device="cuda")
        ops.conv_fwd(nchw_to_dev_nhwc(xx), _packed(wt, co), out, k=k, stride=1, pad=1, Co=co, split=mode, amax_out=slot)
        torch.cuda.synchronize()
        assert ops.amax_value(slot) == float(out.abs().max()), mode
    # BatchNorm apply / backward apply
    M, c = 3 * 17 * 19, 64
    sc, sh = dev(rng.uniform(0.5, 1.5, c)), dev(rng.standard_normal(c))
    y = torch.empty_like(x)
    slot = torch.zeros(L.AMAX_FLOATS, device="cuda")
    ops.bn_apply_leaky(x, sc, sh, None, y, M, c, amax_out=slot)
    torch.cuda.synchronize()
    assert ops.amax_value(slot) == float(y.abs().max())
    dy, mu, iv = dev(rng.standard_normal(x.shape)), dev(rng.standard_normal(c)), dev(rng.uniform(0.5, 2, c))
    sums2 = torch.from_numpy(rng.standard_normal(2 * c)).cuda()
    dx = torch.empty_like(x)
    slot.zero_()
    ops.bn_bwd_apply(x, dy, sc, sh, mu, iv, sums2, float(M), M, c, dx, amax_out=slot)
    torch.cuda.synchronize()
    assert ops.amax_value(slot) == float(dx.abs().max())


@pytest.mark.parametrize("math", ["split", "f16x2", "f16x2nh"])
def test_split_fused_bn_statistics(math):
    """Fused per-M-tile BatchNorm partial sums out of the split-math epilogue (256- and 128-row tiles; generic and
    halo-staged K loops)."""
    from viddet_amd import ops
    from viddet_amd import lib as L
    import ctypes as C
    n, ci, h, w, co, k = 3, 64, 19, 17, 160, 3
    rng, x, wt = _mk(n, ci, h, w, co, k, 123)
    ref = R.conv2d(x, wt, 1, 1)
    M = n * h * w
    fl = {"split": L.MATH_SPLIT, "f16x2": L.MATH_F16X2, "f16x2nh": L.MATH_F16X2 | L.MATH_NOHALO}[math]
    for tile in (1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 15, 16):
        d = L.ConvDesc()
        xd, wp = nchw_to_dev_nhwc(x), _packed(wt, co)
        ax, aw = ops.amax(xd), ops.amax(wp)
        d.amax_in, d.amax_w = ax.data_ptr(), aw.data_ptr()
        out = torch.empty(n, h, w, co, device="cuda")
        part = torch.zeros(64, 2 * co, device="cuda")
        d.in_, d.wp, d.out = xd.data_ptr(), wp.data_ptr(), out.data_ptr()
        d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = n, h, w, ci, h, w, 1
        ops._set_taps(d, ops.fwd_taps(k, 1))
        d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.ldo = 1, h, w, co, 1, co
        d.flags, d.tile, d.stats_part = fl, tile, part.data_ptr()
        lib = L.load()
        L.check(lib.vd_conv_igemm(C.byref(d), L.stream_ptr()), "vd_conv_igemm")
        mt = lib.vd_conv_igemm_mtiles(C.byref(d))
        torch.cuda.synchronize()
        assert mt == -(-M // (256 if tile in (1, 3, 5, 7, 15, 16) else 128))
        s1 = part[:mt, :co].double().sum(0).cpu().numpy()
        s2 = part[:mt, co:].double().sum(0).cpu().numpy()
        assert np.abs(s1 - ref.sum((0, 2, 3))).max() < 2e-3
        assert np.abs(s2 - (ref ** 2).sum((0, 2, 3))).max() < 2e-3 * max(1.0, float((ref ** 2).sum((0, 2, 3)).max()) / 100)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8])
def test_bf16_products_forward_and_gradients(tile):
    """VD_MATH_BF16: the same kernels with only the leading bf16 piece of each operand (one MFMA term): results carry
    the bf16 rounding of the operands (2^-8 each) and nothing worse - rms error within 1e-2 of the output rms and well
    above the fp32 error (the mode really rounds)."""
    from viddet_amd import ops
    n, ci, h, w, co, k = 3, 128, 13, 11, 160, 3
    rng, x, wt = _mk(n, ci, h, w, co, k, 200 + tile)
    xd, wp = nchw_to_dev_nhwc(x), _packed(wt, co)
    ref = R.conv2d(x, wt, 1, 1)
    out = torch.empty(n, h, w, co, device="cuda")
    ops.conv_fwd(xd, wp, out, k=k, stride=1, pad=1, Co=co, tile=tile, split='bf16')
    torch.cuda.synchronize()
    e = dev_nhwc_to_nchw(out) - ref
    rel = float(np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean()))
    assert 5e-4 < rel < 1e-2, rel
    if tile == 1:
        dy = rng.standard_normal((n, co, h, w))
        dx_ref, dw_ref = R.conv2d_backward(x, wt, dy, 1, 1)
        dyd = nchw_to_dev_nhwc(dy)
        dwp = torch.empty(co, k * k * ci, device="cuda")
        ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        ops.conv_wgrad(xd, dyd, dwp, ws, k=k, stride=1, pad=1, Co=co, split='bf16')
        dw = torch.empty(co, ci, k, k, device="cuda")
        ops.unpack_weight(dwp, dw)
        plan = ops.dgrad_plans(k, 1, 1, h, w)[0]
        wpk = torch.empty(ci, len(plan["taps"]) * co, device="cuda")
        ops.pack_weight_dgrad(wp, wpk, Co=co, Co_pad=co, Ci=ci, kd=1, kh=k, kw=k, tap_ids=plan["tap_ids"], src_packed=True)
        dx = torch.empty(n, h, w, ci, device="cuda")
        ops.conv_igemm(dyd, wpk, dx, N=n, Hi=h, Wi=w, Ci=co, Hg=h, Wg=w, in_stride=1, taps=plan["taps"], Ho=h, Wo=w,
                       Co=ci, ldo=ci, split='bf16')
        torch.cuda.synchronize()
        for got, r_ in ((dw.cpu().numpy(), dw_ref), (dev_nhwc_to_nchw(dx), dx_ref)):
            rel = float(np.sqrt(((got - r_) ** 2).mean()) / np.sqrt((r_ ** 2).mean()))
            assert 5e-4 < rel < 1e-2, rel


def _errs(got, ref, S):
    e = got - ref
    return float(np.sqrt((e ** 2).mean())), float(np.abs(e).max()), float((np.abs(e) / np.maximum(S, 1e-300)).max())


U = 2.0 ** -24          # fp32 unit round-off


@pytest.mark.parametrize("mode", [True, "f16x2"])
@pytest.mark.parametrize("ci,k", [(32, 1), (64, 1), (32, 3), (128, 1)])          # K = 32, 64, 288, 128
def test_split_error_shallow_k(mode, ci, k):
    """Shallow reductions (K = 32 .. 288: the 1x1 cells and first-stage 3x3s), where the fp32 fma chain makes ~1 ulp of
    error and the split's per-product error (operand representation 2^-23 each + the dropped low x low term 2^-22) is
    NOT averaged away by a long sum.  Measured here: rms and max error against fp64 and the largest error relative to
    sum |a||b| (the quantity every fp32 dot-product error bound is stated in).  The claim: still fp32-grade - within the
    a-priori bound of the arithmetic, 2^-21 sum |a||b| (+ the accumulation), which is below the gamma_K = K u bound ANY
    fp32 summation order carries for K >= 8 - not 'never above the fma chain': at K = 32 the two are the same size."""
    from viddet_amd import ops
    n, h, w, co = 3, 26, 26, 128
    rng, x, wt = _mk(n, ci, h, w, co, k, 500 + ci + k)
    xd, wp = nchw_to_dev_nhwc(x), _packed(wt, co)
    x32 = xd.permute(0, 3, 1, 2).double().cpu().numpy()
    w32 = dev(wt).double().cpu().numpy()
    ref = R.conv2d(x32, w32, 1, k // 2)
    S = R.conv2d(np.abs(x32), np.abs(w32), 1, k // 2)
    res = {}
    for split in (False, mode):
        out = torch.empty(n, h, w, co, device="cuda")
        ops.conv_fwd(xd, wp, out, k=k, stride=1, pad=k // 2, Co=co, split=split)
        torch.cuda.synchronize()
        res[split] = _errs(dev_nhwc_to_nchw(out), ref, S)
    K = ci * k * k
    print("K = %4d  fp32 MFMA: rms %.2e max %.2e max/sum|ab| %.2f u   %s: rms %.2e max %.2e max/sum|ab| %.2f u" % (
        (K,) + res[False][:2] + (res[False][2] / U, str(mode)) + res[mode][:2] + (res[mode][2] / U,)))
    assert res[mode][2] <= 8 * U + K * U / 16, res            # a-priori: 2^-21 sum|a||b| = 8 u, + the accumulation's share
    assert res[mode][2] <= max(1.25 * res[False][2], 8 * U)   # and never worse than the fma chain beyond that bound
    assert res[mode][0] <= max(1.25 * res[False][0], 4 * U * float(np.sqrt((S ** 2).mean())))


def test_split_error_on_operands_of_a_real_backward_pass():
    """Operands captured from a real training step instead of Gaussians: the head gradients (dL/dlogit: mostly exact
    zeros and sigmoid tails, a dynamic range far beyond 2^19) and the incoming gradient of a first-stage 3x3 conv, each
    as the `dout` operand of its weight gradient (reduction over pixels) and as the activation operand of a 1x1 product
    with the layer's channel count as K (the data gradient's shape: K = 96 / 64, shallow)."""
    from viddet_amd import ops
    from viddet_amd.model import ConvNode
    from tests.test_model_gpu import _mk_net, _targets
    c, size, B = 20, 128, 4
    net, P = _mk_net(c, 61, obj_bias=-1.0)
    rng = np.random.default_rng(61)
    x = rng.standard_normal((B, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, B, c, size, 3)
    net(dev(x), dev(gt), *[dev(t) for t in tg])
    net.backward()
    torch.cuda.synchronize()
    bufs = net._last_train['bufs']
    nodes = {n.name: n for n in net.nodes if isinstance(n, ConvNode)}
    picks = [nodes["yolo_outputs.0.prediction"], nodes["yolo_outputs.2.prediction"], nodes["stages.0.2.body.1"]]
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for n in picks:
        xin, dout = bufs[n.src].contiguous(), bufs['d:' + n.dst].contiguous()
        co = dout.shape[-1]
        amax, nz = float(dout.abs().max()), dout[dout != 0]
        span = np.log2(amax / float(nz.abs().min())) if nz.numel() else 0.0
        print("%s: dout %s  zeros %.1f %%  dynamic range 2^%.0f" % (n.name, tuple(dout.shape), 100.0 * (1 - nz.numel() / dout.numel()), span))
        x32 = xin.permute(0, 3, 1, 2).double().cpu().numpy()
        d32 = dout.permute(0, 3, 1, 2).double().cpu().numpy()
        wz = np.zeros((co, n.cin, n.k, n.k))
        # (1) weight gradient with the real (activation, gradient) pair
        _, dw_ref = R.conv2d_backward(x32, wz, d32, 1, n.pad)
        _, S = R.conv2d_backward(np.abs(x32), wz, np.abs(d32), 1, n.pad)
        res = {}
        for split in (False, True, "f16x2"):
            dwp = torch.empty(co, n.k * n.k * n.cin, device="cuda")
            ops.conv_wgrad(xin, dout, dwp, ws, k=n.k, stride=1, pad=n.pad, Co=co, split=split)
            dw = torch.empty(co, n.cin, n.k, n.k, device="cuda")
            ops.unpack_weight(dwp, dw)
            torch.cuda.synchronize()
            res[split] = _errs(dw.double().cpu().numpy(), dw_ref, S)
        print("  wgrad     fp32 MFMA rms %.2e max %.2e | 3-way bf16 rms %.2e max %.2e | 2-way fp16 rms %.2e max %.2e" % (
            res[False][:2] + res[True][:2] + res["f16x2"][:2]))
        for m in (True, "f16x2"):
            assert res[m][0] <= 1.25 * res[False][0] + 1e-9 * float(np.sqrt((dw_ref ** 2).mean())) + 1e-30, (n.name, m, res)
            assert res[m][1] <= 2.0 * res[False][1] + 8 * U * float(S.max()), (n.name, m, res)
        # (2) the gradient tensor as the activation operand of a shallow-K product (K = its channel count)
        wt = rng.standard_normal((64, co, 1, 1)) * np.sqrt(1.0 / co)
        w32 = dev(wt).double().cpu().numpy()
        ref = R.conv2d(d32, w32, 1, 0)
        S2 = R.conv2d(np.abs(d32), np.abs(w32), 1, 0)
        wp = _packed(wt, 64)
        for split in (False, True, "f16x2"):
            out = torch.empty(dout.shape[0], dout.shape[1], dout.shape[2], 64, device="cuda")
            ops.conv_fwd(dout, wp, out, k=1, stride=1, pad=0, Co=64, split=split)
            torch.cuda.synchronize()
            res[split] = _errs(dev_nhwc_to_nchw(out), ref, S2)
        print("  K = %3d   fp32 MFMA rms %.2e max/sum|ab| %.2f u | 3-way bf16 rms %.2e %.2f u | 2-way fp16 rms %.2e %.2f u" % (
            co, res[False][0], res[False][2] / U, res[True][0], res[True][2] / U, res["f16x2"][0], res["f16x2"][2] / U))
        for m in (True, "f16x2"):
            assert res[m][1] <= 2.0 * res[False][1] + 8 * U * float(S2.max()), (n.name, m, res)
            assert res[m][0] <= 1.25 * res[False][0] + 4 * U * float(np.sqrt((S2 ** 2).mean())), (n.name, m, res)
