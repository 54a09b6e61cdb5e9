"""GPU: the kernels of bf16-STORAGE training (BASELINE configs[4]; include/viddet_hip.h, last section), one by one.
The reference is fp32-only (train_yolov3.py:623-636), so every bound here is defined against the fp64 oracle on operands
that are exactly bf16-representable: a product of two bf16 values is exact in fp32 and the kernels accumulate in fp32, so a
convolution differs from the oracle by fp32 summation error plus ONE rounding of its output to bf16 (2^-9 relative); the
streaming kernels do fp32 arithmetic on the widened values and round once."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ops as R
from oracle import yolo as Y
from tests.util import dev, maxdiff

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
EPS = 2.0 ** -8          # bf16 round-to-nearest: relative error <= 2^-9; bounds below use 2^-8


def _r(a):
    """round to bf16, return float64"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(BF).float().numpy().astype(np.float64)


def _nhwc_b(a):
    return torch.from_numpy(np.moveaxis(a, 1, -1).copy()).to(BF).cuda()


def _nchw(t):
    return np.moveaxis(t.float().cpu().numpy().astype(np.float64), -1, 1)


@pytest.mark.parametrize("shape", [(2, 64, 18, 20, 128, 3, 2, 1), (2, 32, 18, 20, 64, 3, 2, 1), (3, 64, 13, 13, 128, 3, 1, 1),
                                   (2, 128, 9, 11, 64, 1, 1, 0), (2, 64, 16, 16, 32, 3, 1, 1)])     # n, ci, h, w, co, k, stride, pad
def test_data_gradient_through_the_bf16_conv_kernel(shape):
    """dX of a conv from bf16 dZ: the tap plans of ops.dgrad_plans through vd_conv_igemm_bf16 - for stride 2 four parity
    launches with strided / offset outputs - with in-place accumulation onto an existing bf16 gradient (residual)."""
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = shape
    rng = np.random.default_rng(sum(shape))
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    x = rng.standard_normal((n, ci, h, w))
    wt = _r(rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k))
    dz = _r(rng.standard_normal((n, co, ho, wo)))
    skip = _r(rng.standard_normal((n, ci, h, w)))
    dx_ref, _ = R.conv2d_backward(x, wt, dz, s, p)
    ref = dx_ref + skip
    dzd, skd = _nhwc_b(dz), _nhwc_b(skip)
    wp32 = torch.empty(co, k * k * ci, device="cuda")
    ops.pack_weight_fwd(dev(wt), wp32, co)
    out = torch.zeros(n, h, w, ci, dtype=BF, device="cuda")
    for pl in ops.dgrad_plans(k, p, s, h, w):
        T = len(pl["taps"])
        wpk = torch.empty(ci, T * co, device="cuda")
        ops.pack_weight_dgrad(wp32, wpk, Co=co, Co_pad=co, Ci=ci, kd=1, kh=k, kw=k, tap_ids=pl["tap_ids"], src_packed=True)
        wb = torch.empty(ci, T * co, dtype=BF, device="cuda")
        ops.pack_weight_bf16(wpk, wb, Co=ci, Co_pad=ci, Ci=co, Ci_pad=co, T=T)
        ops.conv_igemm_bf16(dzd, wb, out, N=n, Hi=ho, Wi=wo, Ci=co, Hg=pl["Hg"], Wg=pl["Wg"], in_stride=1, taps=pl["taps"],
                            Ho=h, Wo=w, Co=ci, ldo=ci, out_stride=s, out_oy=pl["py"], out_ox=pl["px"], residual=skd, ldr=ci)
    torch.cuda.synchronize()
    got = _nchw(out)
    tol = 2e-4 + np.abs(ref).max() * EPS
    assert maxdiff(got, ref) < tol, (maxdiff(got, ref), tol)


@pytest.mark.parametrize("shape", [(3, 64, 13, 13, 128, 3, 1, 1), (2, 64, 18, 20, 128, 3, 2, 1), (2, 128, 9, 11, 64, 1, 1, 0)])
def test_data_gradient_with_fused_batchnorm_backward_reductions(shape):
    """vd_conv_desc.bs_*: sum g and sum g * xhat of the layer whose dy the data gradient completes (g = dy * leaky'(z * scale
    + shift), xhat = (z - mean) * invstd), from the fp32 values of dy before they are rounded to bf16; a stride-2 data
    gradient adds its four parity launches' table rows."""
    from viddet_amd import ops, lib as L
    n, ci, h, w, co, k, s, p = shape
    rng = np.random.default_rng(sum(shape) + 1)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    wt = _r(rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k))
    dz = _r(rng.standard_normal((n, co, ho, wo)))
    skip = _r(rng.standard_normal((n, ci, h, w)))
    z = _r(rng.standard_normal((n, ci, h, w)) * 1.5)                       # the producer's pre-BatchNorm output
    scale, shift = rng.uniform(0.5, 1.5, ci), rng.standard_normal(ci) * 0.3
    mean, invstd = rng.standard_normal(ci) * 0.1, rng.uniform(0.5, 2.0, ci)
    dy = R.conv2d_backward(np.zeros((n, ci, h, w)), wt, dz, s, p)[0] + skip
    sh4 = (1, -1, 1, 1)
    u = z * scale.reshape(sh4) + shift.reshape(sh4)
    g = np.where(u > 0, dy, 0.1 * dy)
    ref = np.concatenate([g.sum(axis=(0, 2, 3)), (g * (z - mean.reshape(sh4)) * invstd.reshape(sh4)).sum(axis=(0, 2, 3))])
    dzd, skd, zd = _nhwc_b(dz), _nhwc_b(skip), _nhwc_b(z)
    wp32 = torch.empty(co, k * k * ci, device="cuda")
    ops.pack_weight_fwd(dev(wt), wp32, co)
    out = torch.zeros(n, h, w, ci, dtype=BF, device="cuda")
    sc, sh, mu, iv = dev(scale), dev(shift), dev(mean), dev(invstd)
    part = torch.zeros(4096, 2 * ci, device="cuda")
    rows = 0
    for pl in ops.dgrad_plans(k, p, s, h, w):
        T = len(pl["taps"])
        wpk = torch.empty(ci, T * co, device="cuda")
        ops.pack_weight_dgrad(wp32, wpk, Co=co, Co_pad=co, Ci=ci, kd=1, kh=k, kw=k, tap_ids=pl["tap_ids"], src_packed=True)
        wb = torch.empty(ci, T * co, dtype=BF, device="cuda")
        ops.pack_weight_bf16(wpk, wb, Co=ci, Co_pad=ci, Ci=co, Ci_pad=co, T=T)
        d = L.ConvDesc()
        d.in_, d.wp, d.out, d.residual = dzd.data_ptr(), wb.data_ptr(), out.data_ptr(), skd.data_ptr()
        d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = n, ho, wo, co, pl["Hg"], pl["Wg"], 1
        ops._set_taps(d, pl["taps"])
        d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.out_oy, d.out_ox = 1, h, w, ci, s, pl["py"], pl["px"]
        d.ldo = d.ldr = ci
        d.flags, d.slope = L.EPI_RESIDUAL, 0.1
        d.bs_z, d.bs_scale, d.bs_shift, d.bs_mean, d.bs_invstd = zd.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), iv.data_ptr()
        d.bs_part, d.bs_slope = part.data_ptr() + rows * 2 * ci * 4, 0.1
        L.check(L.load().vd_conv_igemm_bf16(C.byref(d), 0, L.stream_ptr()), "vd_conv_igemm_bf16")
        rows += L.load().vd_conv_igemm_bf16_mtiles(C.byref(d))
    torch.cuda.synchronize()
    assert maxdiff(_nchw(out), dy) < 2e-4 + np.abs(dy).max() * EPS
    got = part[:rows].double().sum(dim=0).cpu().numpy()
    assert np.allclose(got, ref, rtol=1e-4, atol=2e-3 * np.sqrt(n * h * w)), np.abs(got - ref).max()
    # tile 8 (256 x 256) has no fused form: refused, never silently unreduced
    d.tile = 8
    assert L.load().vd_conv_igemm_bf16(C.byref(d), 0, L.stream_ptr()) != 0 and b"tile other than 8" in L.load().vd_last_error()


@pytest.mark.parametrize("tile", [0, 1, 2, 6, 10, 13])
def test_forward_conv_with_fused_statistics(tile):
    """Training forward: raw bf16 output + the BatchNorm partial sums of the fp32 accumulators (one table row per M tile),
    finished by vd_bn_sum_partials: sum and sum of squares of the UNROUNDED conv output."""
    from viddet_amd import ops
    n, ci, h, w, co, k = 3, 64, 15, 13, 64 if tile in (10, 13) else 128, 3
    rng = np.random.default_rng(400 + tile)
    x = _r(rng.standard_normal((n, ci, h, w)))
    wt = _r(rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k))
    z = R.conv2d(x, wt, 1, 1)
    wp32 = torch.empty(co, k * k * ci, device="cuda")
    ops.pack_weight_fwd(dev(wt), wp32, co)
    wb = torch.empty(co, k * k * ci, dtype=BF, device="cuda")
    ops.pack_weight_bf16(wp32, wb, Co=co, Co_pad=co, Ci=ci, Ci_pad=ci, T=k * k)
    mt = ops.conv_bf16_mtiles(n, h, w, ci, co, tile)
    part = torch.full((mt, 2 * co), 7.0, device="cuda")
    out = torch.empty(n, h, w, co, dtype=BF, device="cuda")
    ops.conv_igemm_bf16(_nhwc_b(x), wb, out, N=n, Hi=h, Wi=w, Ci=ci, Hg=h, Wg=w, in_stride=1, taps=ops.fwd_taps(k, 1), Ho=h, Wo=w,
                        Co=co, ldo=co, tile=tile, stats_part=part)
    torch.cuda.synchronize()
    assert maxdiff(_nchw(out), z) < 2e-4 + np.abs(z).max() * EPS
    sums = part.double().sum(dim=0).cpu().numpy()
    assert np.allclose(sums[:co], z.sum(axis=(0, 2, 3)), atol=2e-3) and np.allclose(sums[co:], (z ** 2).sum(axis=(0, 2, 3)), rtol=1e-5)


@pytest.mark.parametrize("shape", [(3, 21, 37, 1), (2, 41, 67, 2), (1, 8, 304, 1), (2, 17, 70, 2)])
def test_first_stage_patch_kernel_with_fused_statistics(shape):
    """tile 16 (vd_conv_c32_bf16.hip) in its training form: raw bf16 outputs of the 32 -> 64 channel 3x3 conv (stride 1 / 2) and
    one row of partial sums per 2-D PATCH (pixels of a border patch that lie outside the image are not counted)."""
    from viddet_amd import ops, lib as L
    import ctypes as C
    n, h, w, s_ = shape
    ci, co, k = 32, 64, 3
    rng = np.random.default_rng(500 + h + w)
    x = _r(rng.standard_normal((n, ci, h, w)))
    wt = _r(rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k))
    z = R.conv2d(x, wt, s_, 1)
    ho, wo = z.shape[2], z.shape[3]
    wp32 = torch.empty(co, k * k * ci, device="cuda")
    ops.pack_weight_fwd(dev(wt), wp32, co)
    wb = torch.empty(co, k * k * ci, dtype=BF, device="cuda")
    ops.pack_weight_bf16(wp32, wb, Co=co, Co_pad=co, Ci=ci, Ci_pad=ci, T=k * k)
    out = torch.empty(n, ho, wo, co, dtype=BF, device="cuda")
    part = torch.full((n * ((wo + 31) // 32) * ((ho + 3) // 4) + 1, 2 * co), 7.0, device="cuda")
    d = ops.conv_igemm_bf16(_nhwc_b(x), wb, out, N=n, Hi=h, Wi=w, Ci=ci, Hg=ho, Wg=wo, in_stride=s_, taps=ops.fwd_taps(k, 1), Ho=ho, Wo=wo,
                            Co=co, ldo=co, tile=16, stats_part=part)
    rows = int(L.load().vd_conv_igemm_bf16_mtiles(C.byref(d)))
    torch.cuda.synchronize()
    assert rows == n * ((wo + 31) // 32) * ((ho + (8 if s_ == 1 else 4) - 1) // (8 if s_ == 1 else 4))      # the patch kernel ran
    assert bool((part[rows:] == 7.0).all())
    assert maxdiff(_nchw(out), z) < 2e-4 + np.abs(z).max() * EPS
    sums = part[:rows].double().sum(dim=0).cpu().numpy()
    assert np.allclose(sums[:co], z.sum(axis=(0, 2, 3)), atol=2e-3) and np.allclose(sums[co:], (z ** 2).sum(axis=(0, 2, 3)), rtol=1e-5)


@pytest.mark.parametrize("shape", [(4, 64, 13, 13, 256, 3, 1, 1), (2, 128, 9, 11, 128, 1, 1, 0), (2, 64, 18, 20, 64, 3, 2, 1),
                                   (2, 64, 16, 16, 32, 1, 1, 0), (3, 32, 12, 12, 64, 3, 1, 1)])
def test_weight_gradient_from_bf16_operands(shape):
    from viddet_amd import ops
    n, ci, h, w, co, k, s, p = shape
    rng = np.random.default_rng(sum(shape) + 7)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    x = _r(rng.standard_normal((n, ci, h, w)))
    dz = _r(rng.standard_normal((n, co, ho, wo)))
    _, dw_ref = R.conv2d_backward(x, np.zeros((co, ci, k, k)), dz, s, p)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    dwp = torch.full((co, k * k * ci), 3.0, device="cuda")
    ops.conv_wgrad(_nhwc_b(x), _nhwc_b(dz), dwp, ws, k=k, stride=s, pad=p, Co=co)
    dw = torch.empty(co, ci, k, k, device="cuda")
    ops.unpack_weight(dwp, dw)
    torch.cuda.synchronize()
    assert maxdiff(dw.cpu().numpy(), dw_ref) < 2e-4 * np.sqrt(n * ho * wo)       # exact products, fp32 sums: no bf16 term


def test_batchnorm_passes_on_bf16_tensors():
    """apply (+ residual), backward reductions, backward apply: fp32 arithmetic on the widened bf16 values, one rounding
    of the result - against the fp64 formulas on the same bf16 inputs."""
    from viddet_amd import ops
    M, Cc = 3 * 17 * 19, 96
    rng = np.random.default_rng(12)
    z, res, dy = [_r(rng.standard_normal((M, Cc)) * s_) for s_ in (2.0, 1.0, 0.7)]
    scale, shift = rng.uniform(0.5, 1.5, Cc), rng.standard_normal(Cc)
    mean, invstd = rng.standard_normal(Cc) * 0.1, rng.uniform(0.5, 2.0, Cc)
    b = lambda a: torch.from_numpy(a.astype(np.float32)).to(BF).cuda()
    zd, rd, dyd = b(z), b(res), b(dy)
    sc, sh, mu, iv = dev(scale), dev(shift), dev(mean), dev(invstd)
    y = torch.empty(M, Cc, dtype=BF, device="cuda")
    ops.bn_apply_leaky(zd, sc, sh, rd, y, M, Cc)
    u = z * scale + shift
    ref = np.where(u > 0, u, 0.1 * u) + res
    torch.cuda.synchronize()
    assert maxdiff(y.float().cpu().numpy(), ref) < 1e-5 + np.abs(ref).max() * EPS
    # statistics of a bf16 tensor (the head-bias gradient: column sums of the bf16 dhead)
    ws = torch.empty(max(ops.bn_stats_ws_bytes(M, Cc), 16), dtype=torch.uint8, device="cuda")
    sums = torch.empty(2 * Cc, dtype=torch.float64, device="cuda")
    ops.bn_stats(M, Cc, dyd, sums, ws)
    torch.cuda.synchronize()
    assert np.allclose(sums.cpu().numpy(), np.concatenate([dy.sum(0), (dy ** 2).sum(0)]), rtol=1e-5, atol=1e-3)
    # backward reductions
    sums2 = torch.empty(2 * Cc, dtype=torch.float64, device="cuda")
    ops.bn_bwd_reduce(zd, dyd, sc, sh, mu, iv, M, Cc, sums2, ws)
    g = np.where(u > 0, dy, 0.1 * dy)
    xh = (z - mean) * invstd
    torch.cuda.synchronize()
    assert np.allclose(sums2.cpu().numpy(), np.concatenate([g.sum(0), (g * xh).sum(0)]), rtol=1e-4, atol=2e-3)
    dx = torch.empty(M, Cc, dtype=BF, device="cuda")
    s2 = torch.from_numpy(np.concatenate([g.sum(0), (g * xh).sum(0)])).cuda()
    ops.bn_bwd_apply(zd, dyd, sc, sh, mu, iv, s2, float(M), M, Cc, dx)
    refd = scale * (g - g.sum(0) / M - xh * (g * xh).sum(0) / M)
    torch.cuda.synchronize()
    assert maxdiff(dx.float().cpu().numpy(), refd) < 1e-5 + np.abs(refd).max() * EPS


def test_concat_and_add_on_bf16_tensors():
    from viddet_amd import ops
    n, hu, wu, cu, cr = 2, 5, 7, 64, 128
    rng = np.random.default_rng(3)
    up, route = _r(rng.standard_normal((n, hu, wu, cu))), _r(rng.standard_normal((n, 2 * hu, 2 * wu, cr)))
    b = lambda a: torch.from_numpy(a.astype(np.float32)).to(BF).cuda()
    out = torch.empty(n, 2 * hu, 2 * wu, cu + cr, dtype=BF, device="cuda")
    ops.upsample2x_concat(b(up), b(route), out)
    ref = np.concatenate([up.repeat(2, axis=1).repeat(2, axis=2), route], axis=-1)
    torch.cuda.synchronize()
    assert maxdiff(out.float().cpu().numpy(), ref) == 0.0                      # a copy
    dout = _r(rng.standard_normal(ref.shape))
    dup, drt = torch.empty(n, hu, wu, cu, dtype=BF, device="cuda"), torch.empty(n, 2 * hu, 2 * wu, cr, dtype=BF, device="cuda")
    ops.upsample2x_concat_bwd(b(dout), dup, drt)
    r_up = dout[..., :cu].reshape(n, hu, 2, wu, 2, cu).sum(axis=(2, 4))
    torch.cuda.synchronize()
    assert maxdiff(drt.float().cpu().numpy(), dout[..., cu:]) == 0.0
    assert maxdiff(dup.float().cpu().numpy(), r_up) < np.abs(r_up).max() * EPS
    a_, b_ = _r(rng.standard_normal(4096)), _r(rng.standard_normal(4096))
    o = torch.empty(4096, dtype=BF, device="cuda")
    ops.add(b(a_), b(b_), o)
    torch.cuda.synchronize()
    assert maxdiff(o.float().cpu().numpy(), a_ + b_) <= np.abs(a_ + b_).max() * EPS


def test_loss_gradient_rows_in_bf16_and_stem_weight_gradient():
    """The loss kernel with bf16 gradient rows = its fp32 rows rounded once; the stem's weight gradient from a bf16 dz =
    the fp32 kernel on the widened dz."""
    from viddet_amd import ops
    from tests.test_yolo_gpu import _heads, _gt, _desc
    b, c, size, m = 2, 20, 128, 5
    grids = [size // 32, size // 16, size // 8]
    rng = np.random.default_rng(77)
    heads = _heads(rng, b, c, grids, -1.0)
    gt, ids = _gt(rng, b, m, size, c, [4, 2])
    tg = Y.prefetch_targets(size, size, grids, gt, ids, c)
    h, hd, _, ldh = _desc(ops, heads, c, b)
    P = 3 * sum(g * g for g in grids)
    ws = torch.empty(max(16, ops.yolo_loss_ws_bytes(h)), dtype=torch.uint8, device="cuda")
    res = {}
    for dt in (torch.float32, BF):
        dh = [torch.full(t.shape, 5.0, dtype=dt, device="cuda") for t in hd]
        losses = torch.empty(b, 4, device="cuda")
        ops.yolo_loss_fwd_bwd(h, dev(gt), m, *[dev(t) for t in tg], 0.7, False, losses, dh, None, ws)
        torch.cuda.synchronize()
        res[dt] = (losses.clone(), dh)
    assert torch.equal(res[torch.float32][0], res[BF][0])
    for a_, b_ in zip(res[torch.float32][1], res[BF][1]):
        assert torch.equal(a_.to(BF), b_)
    # stem weight gradient
    n, hh, ww = 2, 24, 20
    x = torch.randn(n, 3, hh, ww, device="cuda")
    dz = torch.randn(n, hh, ww, 32, device="cuda").to(BF)
    wsz = torch.empty(int(ops._lib().vd_stem_wgrad_ws_bytes(n, hh, ww)), dtype=torch.uint8, device="cuda")
    d1, d2 = torch.empty(32, 32, device="cuda"), torch.empty(32, 32, device="cuda")
    ops.stem_wgrad(x, dz, d1, wsz)
    ops.stem_wgrad(x, dz.float(), d2, wsz)
    torch.cuda.synchronize()
    assert torch.equal(d1, d2)


# ---------------------------------------------------------------------------------------------------------------------
# the network in bf16 storage (net.set_storage('bf16'))
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_compare(net, P, c, x, gt, tg, out, loss_tol, whole_min, mean_min):
    from oracle import net as ON
    onet = ON.Net(P, c)
    losses_r, G, _ = onet.train_step(x.astype(np.float64), gt, *tg)
    for i in range(4):
        got = out[i].cpu().numpy()
        assert np.all(np.abs(got - losses_r[i]) <= loss_tol * np.maximum(1.0, np.abs(losses_r[i]))), (i, got, losses_r[i])
    cos, dot, ng, nr = [], 0.0, 0.0, 0.0
    for k in G.keys():
        g, r_ = net.collect_params()[k].grad().cpu().numpy().ravel().astype(np.float64), G[k].ravel()
        assert np.all(np.isfinite(g)), k
        cos.append(float(g @ r_ / (np.linalg.norm(g) * np.linalg.norm(r_) + 1e-30)))
        dot, ng, nr = dot + float(g @ r_), ng + float(g @ g), nr + float(r_ @ r_)
    whole = dot / np.sqrt(ng * nr)
    print("bf16-storage gradients vs the fp64 oracle: whole cosine %.4f, per-tensor mean %.3f min %.3f, norm ratio %.3f" % (
        whole, np.mean(cos), min(cos), np.sqrt(ng / nr)))
    assert whole > whole_min and np.mean(cos) > mean_min, (whole, np.mean(cos), min(cos))
    assert 0.8 < np.sqrt(ng / nr) < 1.25


@pytest.mark.parametrize("cfg", [(2, 4, 64, 3), (2, 20, 128, 4), (2, 4, 64, 3, "fused")])
def test_training_step_in_bf16_storage_against_the_oracle(cfg, monkeypatch):
    """One training step with bf16 activations and gradients against the fp64 oracle of the fp32 reference arithmetic.
    Tolerances are those of the bf16-PRODUCT arithmetic (tests/test_model_gpu.py::test_training_step_in_bf16_products:
    losses 5 %, whole-gradient cosine > 0.9): storing the tensors in bf16 adds one rounding per tensor to the rounding every
    conv operand already got there.  Every launch of the step must be a bf16-tensor kernel."""
    from tests.test_model_gpu import _mk_net, _targets
    b, c, size, m = cfg[:4]
    # "fused": the BatchNorm-backward reductions in the data-gradient epilogues (VD_FUSE_BWD_BF16=1; off by default)
    monkeypatch.setenv("VD_FUSE_BWD_BF16", "1" if len(cfg) > 4 else "0")
    net, P = _mk_net(c, 6, obj_bias=-1.0)
    net.set_storage('bf16')
    rng = np.random.default_rng(6)
    x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, b, c, size, m)
    out = net(dev(x), dev(gt), *[dev(t) for t in tg])
    net.backward()
    torch.cuda.synchronize()
    tp = net._last_train
    names = [fn_ for seg in tp['fwd'] + tp['bwd'] if hasattr(seg, 'recs') for (fn_, _, a) in seg.recs if fn_]
    assert names.count('vd_conv_igemm_bf16') > 140 and 'vd_conv_igemm' not in names and 'vd_bn_apply_leaky' not in names
    assert (names.count('vd_bn_sum_param_grads') > 40) == (len(cfg) > 4)
    assert all(t.dtype == BF for k, t in tp['bufs'].items() if torch.is_tensor(t) and t.dim() == 4 and k not in net.head_names and k != 'in')
    _oracle_compare(net, P, c, x, gt, tg, out, 5e-2, 0.9, 0.7)
    # the optimiser step and a second forward on the moved weights (bf16 weight images are re-packed per step)
    w0, first = net.weights.clone(), float(sum(o.sum() for o in out))      # (the loss tensors are views of the plan's buffer)
    net.sgd_step(lr=1e-3, momentum=0.9, wd=5e-4, batch_size=b)
    out2 = net(dev(x), dev(gt), *[dev(t) for t in tg])
    net.backward()
    torch.cuda.synchronize()
    assert not torch.equal(w0, net.weights) and all(bool(torch.isfinite(o).all()) for o in out2)
    assert float(sum(o.sum() for o in out2)) < first                         # one SGD step on the same batch lowers the loss
    # back to fp32 storage: the fp32 plan of the same shape is independent of the bf16 one
    net.set_storage('fp32')
    out3 = net(dev(x), dev(gt), *[dev(t) for t in tg])
    torch.cuda.synchronize()
    assert ('train', b, size, size) in net._programs and ('train_bf16', b, size, size) in net._programs
    assert all(bool(torch.isfinite(o).all()) for o in out3)


def test_bf16_storage_loop_learns_its_batch():
    """60 steps on one batch in bf16 storage: the loss falls as in fp32 storage (tests/test_training_loop_gpu.py) and the
    trained weights find their boxes through the (fp32) inference path."""
    from tests.test_model_gpu import _mk_net, _targets
    c, size, B = 4, 128, 4
    net, P = _mk_net(c, 21, obj_bias=-1.0)
    net.set_storage('bf16')
    rng = np.random.default_rng(21)
    x = rng.standard_normal((B, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, B, c, size, 2)
    xs, gts, tgs = dev(x), dev(gt), [dev(t) for t in tg]
    hist = []
    for it in range(60):
        out = net(xs, gts, *tgs)
        net.backward()
        net.sgd_step(lr=1e-3, momentum=0.9, wd=5e-4, batch_size=B)
        hist.append(float(sum(o.sum() for o in out)))
    torch.cuda.synchronize()
    print("bf16-storage summed loss: step 0 %.1f, 10 %.1f, 59 %.1f" % (hist[0], hist[10], hist[-1]))
    assert np.all(np.isfinite(hist)) and bool(torch.isfinite(net.weights).all())
    assert hist[10] < 0.8 * hist[0] and hist[-1] < 0.5 * hist[0], (hist[0], hist[10], hist[-1])
