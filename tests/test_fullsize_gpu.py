"""BASELINE.json's configurations at their FULL sizes - too large for the fp64 oracle - through a property the domain
offers: a batch made of B copies of one frame (window) has the batch statistics of that frame, so

  * every activation and every data gradient of the batch-B run is the same for all B frames, bit for bit (the rows of a
    conv / BatchNorm / pointwise launch are computed independently of where they sit in a tile): this is the check on the
    256-row tiles, the halo loop across frame boundaries, the split-K slabs and the XCD tile order, none of which a
    batch-1 run of the same geometry exercises;
  * head rows, losses and detections of every copy equal the batch-1 run's - and the batch-1 geometry is what
    tests/test_model_gpu.py compares with the oracle (one 416x416 / 608x608 frame);
  * the gradients of the three prediction layers are B times the batch-1 ones.  The other gradients are B times the batch-1
    ones only up to what LeakyReLU branch flips at |pre-activation| < 5e-5 do: the two runs round differently, a handful of
    the 10^7 pre-activations of a frame change branch, and the gradient of this random-initialised network moves by percents
    of a tensor's maximum per flip.  The fp64 oracle shows the very same deviations when it is given the two mask sets
    (measured at 128x128, batch 4: device vs oracle 2.5e-5 on both runs, oracle(masks B) vs B x oracle(masks 1) 5.4e-2,
    device B vs B x device 1 5.4e-2), so that comparison is a loose sanity bound here.
"""
import numpy as np
import pytest
import torch

from tests.test_model_gpu import _mk_net, _targets
from tests.util import dev

pytestmark = pytest.mark.gpu


def _frames_identical(net, bufs, B, c, min_tensors):
    n = 0
    for k, v in bufs.items():
        if not (torch.is_tensor(v) and v.dtype in (torch.float32, torch.bfloat16) and v.dim() == 4 and isinstance(k, str)):
            continue
        if v.shape[0] % B or v.shape[0] < B:
            continue
        if k.split(':')[-1] in net.head_names:
            v = v[..., :3 * (5 + c)]             # the padding columns of a head row are never written
        w = v.reshape(B, -1)
        assert bool(torch.isfinite(w[0].float()).all()), k
        assert bool((w == w[0:1]).all()), "frames of %s differ" % k
        n += 1
    assert n >= min_tensors, n


def _train_replication(net, x1, gt1, tg1, B, c, min_tensors, head_tol=2e-4, loss_tol=1e-4):
    rep = lambda a: np.ascontiguousarray(np.repeat(a, B, axis=0))
    out1 = [t.clone() for t in net(dev(x1), dev(gt1), *[dev(t) for t in tg1])]
    net.backward()
    torch.cuda.synchronize()
    heads1 = [net._last_train['bufs'][h].clone() for h in net.head_names]
    g1 = {k: p.grad().clone() for k, p in net.collect_params().items() if p.span is not None}
    outB = [t.clone() for t in net(dev(rep(x1)), dev(rep(gt1)), *[dev(rep(t)) for t in tg1])]
    net.backward()
    torch.cuda.synchronize()
    bufs = net._last_train['bufs']
    _frames_identical(net, bufs, B, c, min_tensors)
    for s, h in enumerate(net.head_names):
        tol = head_tol * float(heads1[s].abs().max())
        assert float((bufs[h] - heads1[s]).abs().max()) < tol, ("head", s, float((bufs[h] - heads1[s]).abs().max()), tol)
    for i in range(4):
        a, b_ = outB[i].cpu().numpy().reshape(-1), out1[i].cpu().numpy().reshape(-1)
        assert np.all(np.abs(a - b_[0]) <= loss_tol * max(1.0, abs(float(b_[0])))), (i, a[:4], b_)
    dev_, dot, na, nb = [], 0.0, 0.0, 0.0
    for k, g in g1.items():
        gb = net.collect_params()[k].grad()
        assert bool(torch.isfinite(gb).all()), k
        dev_.append((float((gb - B * g).abs().max()) / (max(1e-3, float(g.abs().max())) * B), k))
        dot, na, nb = dot + float((gb.double() * g.double()).sum()), na + float((gb.double() ** 2).sum()), nb + float((g.double() ** 2).sum())
    dev_.sort(reverse=True)
    cos = dot / np.sqrt(na * nb)
    print("largest gradient deviations from %d x batch-1:" % B, dev_[:4], "median", dev_[len(dev_) // 2], "whole-gradient cosine", cos)
    return outB, dev_, cos


@pytest.mark.parametrize("math", [None, "native", "split", "split2"])
def test_configs2_headline_batch_64_frames_416_80_classes(math):
    """BASELINE configs[2], the bench.py workload: yolo3_darknet53_coco, batch 64, 416x416, fp32 - under the autotuner's mix
    (None) and with every convolution forced onto one product arithmetic (fp32 MFMA / 3-way bf16 split / 2-way fp16 split),
    so that each family's full-size tiles run the whole network."""
    from viddet_amd import model as M
    M.set_conv_math(math)
    try:
        _headline(math)
    finally:
        M.set_conv_math(None)


def _headline(math):
    c, size, B = 80, 416, 64
    net, P = _mk_net(c, 8, obj_bias=-1.0)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((1, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, 1, c, size, 6)
    outB, dev_, cos = _train_replication(net, x, gt, tg, B, c, min_tensors=150)
    assert outB[0].shape == (B,)
    for e, k in dev_:
        if k.startswith('yolo_outputs.'):        # no LeakyReLU between these and the loss
            assert e < 1e-4, (k, e)
    assert dev_[0][0] < 0.3 and dev_[len(dev_) // 2][0] < 0.05 and cos > 0.999, (dev_[:8], cos)
    # and the optimiser step of the full-size arena
    w0 = net.weights.clone()
    net.sgd_step(lr=1e-3, momentum=0.9, wd=5e-4, batch_size=B)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(net.weights).all()) and not torch.equal(w0, net.weights)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_configs1_detect_batch_32_frames_608(precision):
    """BASELINE configs[1]: detect path, 608x608, batch 32 (bench.py --mode detect), fp32 tensors and bf16 storage."""
    c, size, B = 80, 608, 32
    net, P = _mk_net(c, 9, obj_bias=0.0)
    net.set_precision(precision)
    rng = np.random.default_rng(9)
    x = rng.standard_normal((1, 3, size, size)).astype(np.float32)
    ids1, sc1, bx1 = [t.clone() for t in net(dev(x))]
    rows1 = net.last_rows.clone()
    idsB, scB, bxB = [t.clone() for t in net(dev(np.ascontiguousarray(np.repeat(x, B, axis=0))))]
    rowsB = net.last_rows.clone()
    torch.cuda.synchronize()
    assert int(net.last_overflow.max()) == 0 and int((ids1 >= 0).sum()) >= 20, "fixture produced too few detections"
    bufs = net._programs[('infer_bf16', B, size, size)][1] if precision == 'bf16' else net._programs[('buf', B, size, size, False)]
    _frames_identical(net, bufs, B, c, min_tensors=70)
    for t in (idsB, scB, bxB, rowsB):
        assert bool((t == t[0:1]).all())
    if precision == 'bf16':
        # against the batch-1 run.  Its 19 x 19 / 38 x 38 layers run as split-K grids (VD_CONV_SPLITK: 24 tiles cannot fill the
        # chip), i.e. the same fp32 sums in another association, and a bf16 rounding that lands one unit away in a layer is the
        # noise floor of bf16 storage - so the comparison is the one bf16 inference is held to against the fp32 oracle
        # (tests/test_bf16_gpu.py): raw heads within 3e-2 of the head's largest magnitude (measured 0.9e-2 / 1.4e-2: two runs that
        # each sit ~1.2e-2 from the oracle), the
        # batch-1 run's ten best detections found again with the same class and IoU > 0.9
        b1 = net._programs[('infer_bf16', 1, size, size)][1]
        for hname in net.head_names:
            a, b_ = b1[hname][..., :3 * (5 + c)].float(), bufs[hname][:1, ..., :3 * (5 + c)].float()
            rel = float((a - b_).abs().max()) / float(a.abs().max())
            print(hname, "batch 1 vs batch 32 (bf16): max difference / max|head| = %.2e" % rel)
            assert rel < 3e-2, (hname, rel)
        i1, s1, q1 = ids1.cpu().numpy()[0, :, 0], sc1.cpu().numpy()[0, :, 0], bx1.cpu().numpy()[0]
        iB, qB = idsB.cpu().numpy()[0, :, 0], bxB.cpu().numpy()[0]
        for j in [j for j in range(len(i1)) if i1[j] >= 0][:10]:
            best = 0.0
            for q in np.nonzero(iB == i1[j])[0]:
                a, b_ = q1[j], qB[q]
                iw, ih = max(0.0, min(a[2], b_[2]) - max(a[0], b_[0])), max(0.0, min(a[3], b_[3]) - max(a[1], b_[1]))
                u = (a[2] - a[0]) * (a[3] - a[1]) + (b_[2] - b_[0]) * (b_[3] - b_[1]) - iw * ih
                best = max(best, iw * ih / u if u > 0 else 0.0)
            assert best > 0.9, (j, s1[j], best)
        return
    # fp32, against the batch-1 run: the same rows in the same order, except where two scores tie to the 6th digit
    from tests.util import assert_rows_match, take_ranks
    perm = assert_rows_match(rowsB[:1].cpu().numpy(), rows1.cpu().numpy(), sc1.cpu().numpy(), tie=1e-6)
    assert np.array_equal(take_ranks(idsB[:1], perm), ids1.cpu().numpy())
    assert float(np.abs(take_ranks(scB[:1], perm) - sc1.cpu().numpy()).max()) < 1e-5
    assert float(np.abs(take_ranks(bxB[:1], perm) - bx1.cpu().numpy()).max()) < 3e-3


def test_configs3_temporal_windows_batch_16_416():
    """BASELINE configs[3]: k = 3 frame stack, 416x416, 30 classes (ImageNet-VID), per-GPU batch 16
    (bench.py --window 3 --batch 16 --classes 30): windows replicated."""
    from tests.test_temporal_gpu import _mk
    c, size, B, K = 30, 416, 16, 3
    net, P = _mk(dict(jt="max", jp="late", bct="2"), c, 43)
    rng = np.random.default_rng(43)
    x = rng.standard_normal((1, K, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, 1, c, size, 5)
    outB, dev_, cos = _train_replication(net, x, gt, tg, B, c, min_tensors=150)
    for e, k in dev_:
        if k.startswith('yolo_outputs.'):
            assert e < 1e-4, (k, e)
    assert dev_[0][0] < 0.3 and dev_[len(dev_) // 2][0] < 0.05 and cos > 0.999, (dev_[:8], cos)


def test_configs4_combined_classes_608_batch_32_bf16_products():
    """BASELINE configs[4] per-GPU shape: 285 classes, 608x608, batch 32, the bf16-product arithmetic built for it
    (set_conv_math('bf16')); the frame property does not depend on the arithmetic."""
    from viddet_amd import model as M
    M.set_conv_math("bf16")
    M._TUNE_CACHE.clear()
    try:
        c, size, B = 285, 608, 32
        net, P = _mk_net(c, 17, obj_bias=-1.0)
        rng = np.random.default_rng(17)
        x = rng.standard_normal((1, 3, size, size)).astype(np.float32)
        gt, tg = _targets(rng, 1, c, size, 8)
        # bf16-rounded operands: an fp32 activation that differs in its last bit between the two runs can round to the
        # other bf16 neighbour (2^-9 relative), so heads / losses agree to bf16 accuracy, not to fp32 round-off
        outB, dev_, cos = _train_replication(net, x, gt, tg, B, c, min_tensors=150, head_tol=3e-2, loss_tol=5e-2)
        for e, k in dev_:
            if k.startswith('yolo_outputs.'):
                assert e < 5e-2, (k, e)
        # deeper gradients: direction only (test_training_step_in_bf16_products has the same bound against the oracle)
        assert cos > 0.9, cos
    finally:
        M.set_conv_math(None)
        M._TUNE_CACHE.clear()


def test_configs4_combined_classes_608_batch_32_bf16_storage():
    """BASELINE configs[4] per-GPU shape in bf16 STORAGE (net.set_storage('bf16')): 285 classes, 608x608, batch 32 - bf16
    activations and gradients through vd_conv_igemm_bf16 / VD_STORE_BF16 weight gradients.  The frame property holds for
    bf16 tensors exactly as for fp32 ones (rows are computed independently of their place in a tile); against the batch-1
    run heads and losses agree to bf16 accuracy (another tile shape rounds a layer's output differently)."""
    c, size, B = 285, 608, 32
    net, P = _mk_net(c, 17, obj_bias=-1.0)
    net.set_storage('bf16')
    rng = np.random.default_rng(17)
    x = rng.standard_normal((1, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, 1, c, size, 8)
    outB, dev_, cos = _train_replication(net, x, gt, tg, B, c, min_tensors=150, head_tol=3e-2, loss_tol=5e-2)
    assert net._last_train.get('storage') == 'bf16'
    for e, k in dev_:
        if k.startswith('yolo_outputs.'):
            assert e < 5e-2, (k, e)
    assert cos > 0.9, cos
    net.sgd_step(lr=1e-3, momentum=0.9, wd=5e-4, batch_size=B)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(net.weights).all())
