"""GPU parity: BatchNorm (stats / apply / backward), pointwise kernels and SGD vs the fp64 oracle."""
import numpy as np
import pytest
import torch

from oracle import ops as R
from tests.util import dev, nchw_to_dev_nhwc, dev_nhwc_to_nchw, maxdiff

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 32, 20, 20), (3, 64, 13, 13), (2, 256, 7, 5), (1, 1024, 13, 13), (64, 32, 52, 52),
                                   (2, 96, 9, 9), (1, 1056, 5, 5)])       # channel counts that do not tile the 256-lane sweep
def test_bn_train_forward_backward(shape):
    from viddet_amd import ops
    n, c, h, w = shape
    rng = np.random.default_rng(11)
    x = rng.standard_normal(shape) * 1.7 + rng.standard_normal((1, c, 1, 1))
    gamma = rng.uniform(0.5, 1.5, c)
    beta = rng.standard_normal(c) * 0.3
    res = rng.standard_normal(shape)
    dy = rng.standard_normal(shape)
    rmean0, rvar0 = rng.standard_normal(c), rng.uniform(0.5, 2.0, c)
    # oracle
    u, mean, var = R.bn_train(x, gamma, beta)
    y_ref = R.leaky(u) + res
    g = R.leaky_backward(u, dy)
    dx_ref, dgamma_ref, dbeta_ref = R.bn_train_backward(x, gamma, mean, var, g)
    # device
    M = n * h * w
    xd = nchw_to_dev_nhwc(x)
    sums = torch.empty(2 * c, dtype=torch.float64, device="cuda")
    ws = torch.empty(ops.bn_stats_ws_bytes(M, c), dtype=torch.uint8, device="cuda")
    ops.bn_stats(M, c, xd, sums, ws)
    scale, shift, smean, sinv = [torch.empty(c, device="cuda") for _ in range(4)]
    rmean, rvar = dev(rmean0), dev(rvar0)
    ops.bn_finalize(sums, M, c, dev(gamma), dev(beta), 1e-5, 0.9, rmean, rvar, scale, shift, smean, sinv)
    y = torch.empty_like(xd)
    ops.bn_apply_leaky(xd, scale, shift, nchw_to_dev_nhwc(res), y, M, c)
    sums2 = torch.empty(2 * c, dtype=torch.float64, device="cuda")
    dyd = nchw_to_dev_nhwc(dy)
    ops.bn_bwd_reduce(xd, dyd, scale, shift, smean, sinv, M, c, sums2, ws)
    dgamma, dbeta = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    ops.bn_param_grads(sums2, c, dgamma, dbeta)
    dx = torch.empty_like(xd)
    ops.bn_bwd_apply(xd, dyd, scale, shift, smean, sinv, sums2, M, M, c, dx)
    torch.cuda.synchronize()
    assert maxdiff(smean.cpu().numpy(), mean) < 1e-5
    assert maxdiff(1.0 / sinv.cpu().numpy() ** 2 - 1e-5, var) < 1e-4 * max(1.0, var.max())
    assert maxdiff(dev_nhwc_to_nchw(y), y_ref) < 1e-4
    assert maxdiff(rmean.cpu().numpy(), R.bn_running_update(rmean0, mean)) < 1e-5
    assert maxdiff(rvar.cpu().numpy(), R.bn_running_update(rvar0, var)) < 1e-4
    tolg = 1e-5 * M + 1e-3
    assert maxdiff(dgamma.cpu().numpy(), dgamma_ref) < tolg
    assert maxdiff(dbeta.cpu().numpy(), dbeta_ref) < tolg
    assert maxdiff(dev_nhwc_to_nchw(dx), dx_ref) < 2e-4


def test_bn_fold_eval():
    from viddet_amd import ops
    rng = np.random.default_rng(12)
    c = 96
    x = rng.standard_normal((2, c, 5, 7))
    gamma, beta = rng.uniform(0.5, 1.5, c), rng.standard_normal(c)
    rmean, rvar = rng.standard_normal(c), rng.uniform(0.2, 2.0, c)
    ref = R.leaky(R.bn_eval(x, gamma, beta, rmean, rvar))
    scale, shift = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    ops.bn_fold_eval(dev(gamma), dev(beta), dev(rmean), dev(rvar), 1e-5, scale, shift)
    xd = nchw_to_dev_nhwc(x)
    y = torch.empty_like(xd)
    ops.bn_apply_leaky(xd, scale, shift, None, y, 2 * 5 * 7, c)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(y), ref) < 1e-5


def test_upsample_concat_fwd_bwd():
    from viddet_amd import ops
    rng = np.random.default_rng(13)
    n, cu, cr, h, w = 2, 32, 64, 6, 10
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    up = f32(rng.standard_normal((n, cu, h // 2, w // 2)))
    route = f32(rng.standard_normal((n, cr, h, w)))
    ref = np.concatenate([R.upsample2x(up), route], axis=1)
    out = torch.empty(n, h, w, cu + cr, device="cuda")
    ops.upsample2x_concat(nchw_to_dev_nhwc(up), nchw_to_dev_nhwc(route), out)
    dout = f32(rng.standard_normal(ref.shape))
    dup, droute = torch.empty(n, h // 2, w // 2, cu, device="cuda"), torch.empty(n, h, w, cr, device="cuda")
    ops.upsample2x_concat_bwd(nchw_to_dev_nhwc(dout), dup, droute)
    torch.cuda.synchronize()
    assert maxdiff(dev_nhwc_to_nchw(out), ref) == 0.0
    assert maxdiff(dev_nhwc_to_nchw(dup), R.upsample2x_backward(dout[:, :cu])) < 1e-6
    assert maxdiff(dev_nhwc_to_nchw(droute), dout[:, cu:]) == 0.0


def test_layout_preprocess_add_fill():
    from viddet_amd import ops
    rng = np.random.default_rng(14)
    x = rng.standard_normal((2, 3, 9, 11)).astype(np.float32)
    out = torch.empty(2, 9, 11, 3, device="cuda")
    ops.nchw_to_nhwc(dev(x), out)
    img = rng.integers(0, 256, (2, 9, 11, 3), dtype=np.uint8)
    pre = torch.empty(2, 9, 11, 3, device="cuda")
    ops.preprocess_u8(torch.from_numpy(img).cuda(), pre)
    a, b = rng.standard_normal(1003).astype(np.float32), rng.standard_normal(1003).astype(np.float32)
    o = torch.empty(1003, device="cuda")
    ops.add(dev(a), dev(b), o)
    f = torch.empty(777, device="cuda")
    ops.fill(f, 2.5)
    torch.cuda.synchronize()
    assert maxdiff(out.cpu().numpy(), np.moveaxis(x, 1, -1)) == 0.0
    ref = np.stack([np.moveaxis(R.preprocess_u8(img[i]), 0, -1) for i in range(2)])
    assert maxdiff(pre.cpu().numpy(), ref) < 1e-6
    assert maxdiff(o.cpu().numpy(), a + b) == 0.0
    assert float(f.min()) == 2.5 and float(f.max()) == 2.5


@pytest.mark.parametrize("type_", [0, 1])
def test_temporal_pool(type_):
    from viddet_amd import ops
    rng = np.random.default_rng(15)
    b, k, inner = 3, 3, 5 * 5 * 32
    x = rng.standard_normal((b, k, inner))
    ref = R.temporal_pool(x, 'max' if type_ == 0 else 'mean')
    y = torch.empty(b, inner, device="cuda")
    am = torch.empty(b, inner, dtype=torch.int32, device="cuda")
    ops.temporal_pool(dev(x), y, am, b, k, inner, type_)
    dy = rng.standard_normal((b, inner))
    dx = torch.empty(b, k, inner, device="cuda")
    ops.temporal_pool_bwd(dev(dy), am, dx, b, k, inner, type_)
    torch.cuda.synchronize()
    assert maxdiff(y.cpu().numpy(), ref) < 1e-6
    if type_ == 0:
        dref = (x == x.max(axis=1, keepdims=True)) * dy[:, None, :]
    else:
        dref = np.broadcast_to(dy[:, None, :] / k, x.shape)
    assert maxdiff(dx.cpu().numpy(), dref) < 1e-6


def test_sgd_momentum():
    from viddet_amd import ops
    rng = np.random.default_rng(16)
    n = 4099
    w, g, m = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    wr, mr = R.sgd_momentum(w, g, m, 0.01, 0.9, 5e-4, 1.0 / 64)
    wd_, md_ = dev(w), dev(m)
    ops.sgd_momentum(wd_, dev(g), md_, 0.01, 0.9, 5e-4, 1.0 / 64)
    torch.cuda.synchronize()
    assert maxdiff(wd_.cpu().numpy(), wr) < 1e-6
    assert maxdiff(md_.cpu().numpy(), mr) < 1e-6
