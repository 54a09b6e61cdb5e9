"""GPU parity of YOLOV3Temporal with per-frame outputs (--temp --mult_out, t = 5) against the fp64
oracle (oracle/net_temporal.py TemporalOutNet): per-frame detections (B,5,100,.), per-frame losses reduced to the four
means the reference returns, every parameter gradient, and the optimiser step that carries the 1/(B*t) of the mean."""
import numpy as np
import pytest
import torch

from oracle import net_temporal as OT
from oracle import ops as R
from oracle import yolo as Y
from tests.util import dev, maxdiff, boxes_close

pytestmark = pytest.mark.gpu
T_ = 5


def _mk(bct, c, seed):
    from viddet_amd.model import yolo3_darknet53
    net = yolo3_darknet53(["c%d" % i for i in range(c)], k=T_, block_conv_type=bct, temporal=True, t_out=True)
    P = OT.init_params(c, T_, 'tout', bct, seed=seed, obj_bias=-1.0)
    assert set(P) == set(net.collect_params().keys()), sorted(set(P) ^ set(net.collect_params().keys()))[:6]
    for k, p in net.collect_params().items():
        assert tuple(P[k].shape) == p.shape, (k, P[k].shape, p.shape)
        p.set_data(torch.from_numpy(P[k].astype(np.float32)))
    return net, P


def test_factory_guards():
    from viddet_amd.model import yolo3_darknet53
    with pytest.raises(NotImplementedError):
        yolo3_darknet53(["a"], k=5, temporal=True, block_conv_type='3')    # t_out=False squeezes the frame axis: 2-D blocks only
    with pytest.raises(NotImplementedError):
        yolo3_darknet53(["a"], k=5, temporal=True, t_out=True, corr_d=4)
    with pytest.raises(AssertionError):
        yolo3_darknet53(["a"], k=3, temporal=True, t_out=True)     # "Currently only support t=5"


@pytest.mark.parametrize("bct", ["2", "3", "21"])
def test_temporal_out_inference_and_training(bct):
    c, b, size = 3, 1, 64
    net, P = _mk(bct, c, 51)
    rng = np.random.default_rng(51)
    x = rng.standard_normal((b, T_, 3, size, size)).astype(np.float32)
    onet = OT.TemporalOutNet(P, c, T_, bct)
    ids_r, sc_r, bx_r, rows_r, heads_r = onet.detect(x.astype(np.float64))
    ids, sc, bx = net(dev(x))
    torch.cuda.synchronize()
    assert tuple(ids.shape) == (b, T_, 100, 1) and tuple(bx.shape) == (b, T_, 100, 4)
    bufs = net._programs[('buf', b, size, size, False)]
    for s, hname in enumerate(net.head_names):
        got = bufs[hname].cpu().numpy()[..., :3 * (5 + c)]
        assert got.shape[0] == b * T_
        assert maxdiff(got, np.moveaxis(heads_r[s], 1, -1)) < 1e-3, "head %d" % s
    from tests.util import assert_rows_match, take_ranks
    perm = assert_rows_match(net.last_rows.cpu().numpy(), rows_r, sc_r)
    assert np.array_equal(take_ranks(ids.reshape(b * T_, 100, 1), perm), ids_r) and int((ids_r >= 0).sum()) > 0
    assert maxdiff(take_ranks(sc.reshape(b * T_, 100, 1), perm), sc_r) < 1e-3
    boxes_close(take_ranks(bx.reshape(b * T_, 100, 4), perm), bx_r)
    # ---- training: every frame of the window has its own ground truth and prefetch targets
    grids = [size // 32, size // 16, size // 8]
    gt = np.full((b, T_, 2, 4), -1.0)
    gid = np.full((b, T_, 2, 1), -1.0)
    for t in range(T_):
        gt[0, t, 0] = [4.0 + 2 * t, 6.0 + t, 38.0 + 2 * t, 48.0 + t]
        gid[0, t, 0] = t % c
        if t % 2:
            gt[0, t, 1] = [20.0, 10.0 + 3 * t, 60.0, 40.0 + 3 * t]
            gid[0, t, 1] = (t + 1) % c
    gt_f, gid_f = gt.reshape(b * T_, 2, 4), gid.reshape(b * T_, 2, 1)
    tg_f = Y.prefetch_targets(size, size, grids, gt_f, gid_f, c)
    tg = [t.reshape((b, T_) + t.shape[1:]) for t in tg_f]
    out = net(dev(x), dev(gt), *[dev(t) for t in tg])
    net.backward()
    torch.cuda.synchronize()
    tb = net._programs[('buf', b, size, size, True)]
    from tests.util import device_leaky_masks, check_masks_differ_only_at_ties
    onet.mask_override = device_leaky_masks(net, tb)
    losses_r, G, _ = onet.train_step(x.astype(np.float64), gt_f, *tg_f)
    check_masks_differ_only_at_ties(onet.pre, onet.mask_override)
    for i in range(4):                                   # yolo3_temporal.py:535: mean over the B*t per-frame losses
        ref = float(np.mean(losses_r[i]))
        assert out[i].dim() == 0 and abs(float(out[i]) - ref) <= 2e-3 * max(1.0, abs(ref)), (i, float(out[i]), ref)
    for k, v in onet.new_running.items():
        assert maxdiff(net.collect_params()[k].data().cpu().numpy(), v) < 1e-4, k
    bad = []
    for k, gref in G.items():                            # the arena holds the gradient of the SUM of per-frame losses
        got = net.collect_params()[k].grad().cpu().numpy()
        scale = max(1e-3, float(np.abs(gref).max()))
        if maxdiff(got, gref) / scale >= 5e-4:
            bad.append((k, maxdiff(got, gref) / scale))
    assert not bad, bad[:6]
    # the optimiser step applies the 1/(B*t) of the mean through its rescale (trainer.step(batch_size) on mean losses)
    keys = list(G.keys())[::9]
    w0 = {k: net.collect_params()[k].data().cpu().numpy().astype(np.float64) for k in keys}
    net.sgd_step(lr=0.01, momentum=0.9, wd=5e-4, batch_size=b)
    torch.cuda.synchronize()
    for k in keys:
        gdev = net.collect_params()[k].grad().cpu().numpy().astype(np.float64)
        wr, _ = R.sgd_momentum(w0[k], gdev, np.zeros_like(gdev), 0.01, 0.9, 5e-4, 1.0 / (b * T_) / b)
        assert maxdiff(net.collect_params()[k].data().cpu().numpy(), wr) < 1e-6, k


@pytest.mark.parametrize("flags", [
    ["--window", "3,1", "--k_join_type", "max", "--k_join_pos", "late"],             # YOLOV3T k=3 (BASELINE configs[3])
    ["--window", "5,1", "--temp", "--mult_out"],                                      # YOLOV3Temporal, per-frame outputs
    ["--window", "5,1", "--temp"],                                                    # YOLOV3Temporal, one output (side branches)
])
def test_train_script_windows_end_to_end(tmp_path, monkeypatch, flags):
    """train_yolov3.py on synthetic VID windows: loader (window batches, per-frame targets), network, loss logging,
    validation with VOCMApMetric / VOCMApMetricTemporal, checkpoint."""
    import glob
    import os
    import train_yolov3 as T
    monkeypatch.chdir(tmp_path)
    T.main(["--dataset", "vid", "--batch_size", "2", "--data_shape", "64", "--epochs", "1", "--synthetic_samples", "8",
            "--save_prefix", "w", "--val_interval", "1", "--log_interval", "1", "--no_random_shape"] + flags)
    (log,) = glob.glob(os.path.join("models", "experiments", "w", "*_train.log"))
    logs = open(log).read()
    assert "Training cost" in logs and "Validation" in logs and "mAP" in logs, logs[-600:]
    if "--mult_out" in flags:
        assert "mAP t=4/5" in logs
    assert glob.glob(os.path.join("models", "experiments", "w", "*.params"))


def test_temporal_side_branches_inference_and_training():
    """YOLOV3Temporal(t_out=False) (--temp without --mult_out; yolo3_temporal.py:326-333,436-447): stages on 5 / 3 / 1 frames,
    the strided 2+1-D side branches convs1 / convs2 (the second cell of each a (3,1,1) conv WITHOUT temporal padding) added to
    the stage outputs, centre-frame routes, single-frame neck and heads - against oracle/net_temporal.py TemporalSideNet:
    heads, identical post-NMS rows, the four per-sample losses, all 234 gradients, running statistics."""
    from viddet_amd.model import yolo3_darknet53
    from tests.util import assert_rows_match, take_ranks, device_leaky_masks, check_masks_differ_only_at_ties
    c, b, size = 3, 2, 64
    net = yolo3_darknet53(["c%d" % i for i in range(c)], k=T_, temporal=True)
    P = OT.side_init_params(c, seed=61, obj_bias=-1.0)
    assert set(P) == set(net.collect_params().keys()), sorted(set(P) ^ set(net.collect_params().keys()))[:6]
    for k, p in net.collect_params().items():
        assert tuple(P[k].shape) == p.shape, (k, P[k].shape, p.shape)
        p.set_data(torch.from_numpy(P[k].astype(np.float32)))
    rng = np.random.default_rng(61)
    x = rng.standard_normal((b, T_, 3, size, size)).astype(np.float32)
    onet = OT.TemporalSideNet(P, c)
    ids_r, sc_r, bx_r, rows_r, heads_r = onet.detect(x.astype(np.float64))
    ids, sc, bx = net(dev(x))
    torch.cuda.synchronize()
    assert tuple(ids.shape) == (b, 100, 1)
    bufs = net._programs[('buf', b, size, size, False)]
    for s_, hname in enumerate(net.head_names):
        got = bufs[hname].cpu().numpy()[..., :3 * (5 + c)]
        assert got.shape[0] == b and maxdiff(got, np.moveaxis(heads_r[s_], 1, -1)) < 1e-3, "head %d" % s_
    perm = assert_rows_match(net.last_rows.cpu().numpy(), rows_r, sc_r)
    assert np.array_equal(take_ranks(ids, perm), ids_r) and int((ids_r >= 0).sum()) > 0
    assert maxdiff(take_ranks(sc, perm), sc_r) < 1e-3 and boxes_close(take_ranks(bx, perm), bx_r)
    gt = np.array([[[5., 8., 40., 50.], [-1, -1, -1, -1]], [[10., 12., 30., 28.], [20., 5., 60., 62.]]])
    gid = np.array([[[1.], [-1.]], [[0.], [2.]]])
    tg = Y.prefetch_targets(size, size, [size // 32, size // 16, size // 8], gt, gid, c)
    out = net(dev(x), dev(gt), *[dev(t) for t in tg])
    net.backward()
    torch.cuda.synchronize()
    onet.mask_override = device_leaky_masks(net, net._programs[('buf', b, size, size, True)])
    losses_r, G, _ = onet.train_step(x.astype(np.float64), gt, *tg)
    check_masks_differ_only_at_ties(onet.pre, onet.mask_override)
    for i in range(4):
        assert np.all(np.abs(out[i].cpu().numpy() - losses_r[i]) <= 2e-3 * np.maximum(1.0, np.abs(losses_r[i]))), i
    for k, v in onet.new_running.items():
        assert maxdiff(net.collect_params()[k].data().cpu().numpy(), v) < 1e-4, k
    bad = []
    for k, gref in G.items():
        got = net.collect_params()[k].grad().cpu().numpy()
        scale = max(1e-3, float(np.abs(gref).max()))
        if maxdiff(got, gref) / scale >= 5e-4:
            bad.append((k, maxdiff(got, gref) / scale))
    assert len(G) == 234 and not bad, bad[:6]
