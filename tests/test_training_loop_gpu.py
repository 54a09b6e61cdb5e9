"""The training loop as a loop: many forward / backward / SGD-momentum steps on one fixed batch.  No oracle can follow 60
steps (fp64 NumPy, minutes per step), so this checks what a loop must do whatever the arithmetic: the summed loss falls
steadily (every repack after a step - data-gradient weight layout, weight max-abs slots, BN fold - feeds the next step),
nothing leaves the finite range, the two fp32-grade product arithmetics stay on the same trajectory for the first steps, and
the trained weights then find the boxes they were trained on through the inference path (running statistics, fold, decode, NMS)."""
import numpy as np
import pytest
import torch

from tests.test_model_gpu import _mk_net, _targets
from tests.util import dev

pytestmark = pytest.mark.gpu


def _loop(math, steps, c=4, size=128, B=4, lr=1e-3, ulp=False):
    from viddet_amd import model as M
    M.set_conv_math(math)
    try:
        net, P = _mk_net(c, 21, obj_bias=-1.0)
        if ulp:
            # every conv weight moved by -1 / 0 / +1 unit in the last place (a perturbation of fp32 rounding size)
            g = np.random.default_rng(5)
            for k, p in net.collect_params().items():
                if k.endswith("weight"):
                    w = P[k].astype(np.float32)
                    step = np.where(w == 0, 0, g.integers(-1, 2, w.shape)).astype(np.int32)
                    p.set_data(torch.from_numpy((w.view(np.int32) + step).view(np.float32)))
        rng = np.random.default_rng(21)
        x = rng.standard_normal((B, 3, size, size)).astype(np.float32)
        gt, tg = _targets(rng, B, c, size, 2)
        xs, gts, tgs = dev(x), dev(gt), [dev(t) for t in tg]
        hist = []
        for it in range(steps):
            out = net(xs, gts, *tgs)
            net.backward()
            net.sgd_step(lr=lr, momentum=0.9, wd=5e-4, batch_size=B)
            hist.append([float(o.sum()) for o in out])
        torch.cuda.synchronize()
        return net, np.asarray(hist), (x, gt, tg)
    finally:
        M.set_conv_math(None)


def test_fixed_batch_loss_falls_and_the_trained_net_finds_its_boxes():
    net, hist, (x, gt, tg) = _loop(None, 150)
    tot = hist.sum(axis=1)
    print("summed loss: step 0 %.1f, 10 %.1f, 50 %.1f, 149 %.1f" % (tot[0], tot[10], tot[50], tot[-1]))
    assert np.all(np.isfinite(hist)) and bool(torch.isfinite(net.weights).all()) and bool(torch.isfinite(net.running).all())
    assert tot[10] < 0.8 * tot[0] and tot[50] < 0.5 * tot[0] and tot[-1] < 0.25 * tot[0], tot[[0, 10, 50, -1]]
    # smoothed over windows of 25 steps the loss falls steeply, then reaches its plateau under this learning rate.  WHERE the
    # plateau starts and how it wobbles depends on the kernels in use (tile choices and summation orders move the tail by a
    # few per cent: one build had windows 19.18, 19.26, 19.33, 18.32 where another fell monotonically), so the property is:
    # the first two windows fall, and from then on no window is more than 5 % above the lowest one before it
    w = tot[:150].reshape(6, 25).mean(axis=1)
    assert w[1] < 0.5 * w[0] and w[2] < w[1], w
    assert all(w[i] <= 1.05 * w[:i].min() for i in range(3, 6)) and w[-1] < 1.15 * w.min(), w
    # inference path on the training frames: every ground-truth box is found by a detection with IoU > 0.5
    ids, sc, bx = [t.cpu().numpy() for t in net(dev(x))]
    from viddet_amd.bbox import bbox_iou
    found = total = 0
    for b in range(x.shape[0]):
        keep = ids[b, :, 0] >= 0
        for j in range(gt.shape[1]):
            if gt[b, j, 0] < 0:
                continue
            total += 1
            if keep.any():
                iou = bbox_iou(bx[b][keep], gt[b, j:j + 1])[:, 0]
                found += int((iou > 0.5).any())
    print("ground-truth boxes found by the trained net: %d of %d" % (found, total))
    assert total >= 4 and found >= total - 1, (found, total)


def _pinned(monkeypatch, alt=False):
    """Deterministic kernels: no timing-based autotuner, the kernels' heuristic tiles (or the second fixed tile set)."""
    monkeypatch.setenv("VD_AUTOTUNE", "0")
    monkeypatch.setenv("VD_TILE_ALT", "1" if alt else "0")


def test_product_arithmetics_share_the_trajectory(monkeypatch):
    """train_yolov3.py:623-640 as a loop under each fp32-grade product arithmetic, every run with PINNED kernels
    (VD_AUTOTUNE=0: which tile and arithmetic a launch uses no longer depends on the box's timings).  A loop that drops the
    loss 1331 -> 70 in four steps amplifies any rounding-level difference ~10x per step, so no fixed bound on
    |split - native| after step 1 means anything by itself; what IS boundable is the rate: the split arithmetics must drift
    from the fp32 MFMA's trajectory no faster than the fp32 MFMA drifts from ITSELF under perturbations of rounding size -
    controls: (a) the same arithmetic under a second tile set (other groupings of the BatchNorm partial sums and split-K
    slabs), (b) every weight moved by at most one unit in the last place."""
    steps = 8
    _pinned(monkeypatch)
    _, h_native, _ = _loop("native", steps)
    _, h_again, _ = _loop("native", steps)
    _, h_split, _ = _loop("split", steps)
    _, h_f16, _ = _loop("split2", steps)
    _, h_ulp, _ = _loop("native", steps, ulp=True)
    _pinned(monkeypatch, alt=True)
    _, h_alt, _ = _loop("native", steps)
    tn, ts, tf, tu, tt = [h.sum(1) for h in (h_native, h_split, h_f16, h_ulp, h_alt)]
    rel = lambda t: np.abs(t - tn) / tn
    for name, t in (("3-way bf16 split", ts), ("2-way fp16 split", tf), ("control: 1-ulp weights", tu), ("control: tile set 2", tt)):
        print("%-24s |loss - native| / native per step: %s" % (name, " ".join("%.1e" % v for v in rel(t))))
    # pinned kernels are deterministic: the same run twice is the same bits (nothing timing-dependent is left)
    assert np.array_equal(h_native, h_again)
    # identical weights at step 0: the first losses agree to fp32 round-off for every arithmetic, and after one update still
    assert np.all(rel(ts)[:2] < 2e-4) and np.all(rel(tf)[:2] < 2e-4), (rel(ts)[:2], rel(tf)[:2])
    # the rate: at every step the split runs are no further from native than 32x the larger control has been so far
    # (32x = one and a half steps of the ~10x-per-step amplification; a bf16-grade product would start 1e4 x the
    # controls and stay there), plus the round-off floor of the loss sum itself
    ctrl = np.maximum.accumulate(np.maximum(rel(tu), rel(tt)))
    assert ctrl[0] < 2e-4
    for name, t in (("split", ts), ("split2", tf)):
        bad = rel(t) > 32.0 * ctrl + 1e-6
        assert not bad.any(), (name, rel(t), ctrl)
    # and none of them leaves the neighbourhood of the trajectory
    for t in (ts, tf, tu, tt):
        assert np.all(np.abs(t - tn) < 0.6 * tn), (t, tn)


def test_tuning_table_persists_and_a_second_start_is_bit_identical(tmp_path, monkeypatch):
    """VD_TUNE_CACHE: the autotuner's choices are written when a plan is built and read by the next start, which then times
    nothing and repeats the first run bit for bit (with choices made by timing, two runs of one seed used to differ)."""
    from viddet_amd import model as M
    path = tmp_path / "tune.json"
    monkeypatch.setenv("VD_TUNE_CACHE", str(path))
    monkeypatch.setenv("VD_AUTOTUNE", "1")
    M._TUNE_CACHE.clear()
    t0 = M._TUNE_CACHE.tuned
    net1, h1, _ = _loop(None, 3)
    assert M._TUNE_CACHE.tuned - t0 > 20 and path.exists()
    import json
    doc = json.load(open(path))
    assert doc["library"] == M.TuneCache.library_id() and len(doc["entries"]) >= M._TUNE_CACHE.tuned - t0
    M._TUNE_CACHE.clear()                                   # a second process start: nothing in memory
    t1 = M._TUNE_CACHE.tuned
    net2, h2, _ = _loop(None, 3)
    assert M._TUNE_CACHE.tuned == t1, "the second start timed kernels again"
    assert np.array_equal(h1, h2)
    assert torch.equal(net1.weights, net2.weights) and torch.equal(net1.momentum_buf, net2.momentum_buf)
    M._TUNE_CACHE.clear()


def test_parity_streams_do_not_change_a_bit(monkeypatch):
    """The four parity launches of a stride-2 data gradient write disjoint outputs and disjoint rows of the BatchNorm
    partial table: on four streams (VD_PARITY_STREAMS=1, the default) or one after the other, same bits - a race on the
    shared table or on d:src would show here.  The fused stride-2 reductions (VD_FUSE_BWD_S2) sum the same terms in
    another kernel: round-off apart."""
    _pinned(monkeypatch)
    res = {}
    for ps, fs in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("VD_PARITY_STREAMS", ps)
        monkeypatch.setenv("VD_FUSE_BWD_S2", fs)
        net, hist, _ = _loop("split2", 1)             # one step: same forward, so no LeakyReLU branch can flip
        res[(ps, fs)] = (hist, net.grads.clone(), net.weights.clone())
    a, b, c = res[("1", "1")], res[("0", "1")], res[("1", "0")]
    assert np.array_equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    gmax = float(a[1].abs().max())
    assert float((a[1] - c[1]).abs().max()) <= 1e-4 * gmax and abs(a[0][0].sum() - c[0][0].sum()) <= 1e-5 * a[0][0].sum()
