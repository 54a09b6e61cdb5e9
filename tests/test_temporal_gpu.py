"""GPU parity of the temporal-window variants of YOLOV3T (k=3; SURVEY 8 row a15, BASELINE config 4) against the
fp64 oracle (oracle/net_temporal.py): early / late joins, max / mean pooling, 2-D per-frame neck, 3-D (3x3x3) and
2+1-D ((1,3,3)+(3,1,1)) neck convs across the K frames."""
import numpy as np
import pytest
import torch

from viddet_amd.model import ConvNode as ConvNode_

from oracle import net_temporal as OT
from oracle import yolo as Y
from tests.util import dev, maxdiff, boxes_close

pytestmark = pytest.mark.gpu

CFGS = [
    dict(jt="max", jp="early", bct="2"),
    dict(jt="mean", jp="late", bct="2"),
    dict(jt="max", jp="late", bct="3"),
    dict(jt="mean", jp="late", bct="21"),
    dict(jt="max", jp="late", bct="2"),
    dict(jt="mean", jp="early", bct="2"),
    dict(jt="cat", jp="early", bct="2"),
    dict(jt="cat", jp="late", bct="3"),
]


def _mk(cfg, c, seed):
    from viddet_amd.model import yolo3_darknet53
    net = yolo3_darknet53(["c%d" % i for i in range(c)], k=3, k_join_type=cfg["jt"], k_join_pos=cfg["jp"],
                          block_conv_type=cfg["bct"])
    P = OT.init_params(c, 3, cfg["jp"], cfg["bct"], seed=seed, obj_bias=-1.0, k_join_type=cfg["jt"])
    assert set(P) == set(net.collect_params().keys()), sorted(set(P) ^ set(net.collect_params().keys()))[:6]
    for k, p in net.collect_params().items():
        assert tuple(P[k].shape) == p.shape, (k, P[k].shape, p.shape)
        p.set_data(torch.from_numpy(P[k].astype(np.float32)))
    return net, P


@pytest.mark.parametrize("cfg", CFGS)
def test_temporal_inference_and_training(cfg):
    c, b, size, K = 3, 2, int(__import__('os').environ.get('VD_TSIZE', '64')), 3
    net, P = _mk(cfg, c, 41)
    rng = np.random.default_rng(41)
    x = rng.standard_normal((b, K, 3, size, size)).astype(np.float32)
    onet = OT.TemporalNet(P, c, K, cfg["jt"], cfg["jp"], cfg["bct"])
    ids_r, sc_r, bx_r, rows_r, heads_r = onet.detect(x.astype(np.float64))
    ids, sc, bx = net(dev(x))
    torch.cuda.synchronize()
    bufs = net._programs[('buf', b, size, size, False)]
    for s, hname in enumerate(net.head_names):
        got = bufs[hname].cpu().numpy()[..., :3 * (5 + c)]
        assert maxdiff(got, np.moveaxis(heads_r[s], 1, -1)) < 1e-3, "head %d" % s
    from tests.util import assert_rows_match, take_ranks
    perm = assert_rows_match(net.last_rows.cpu().numpy(), rows_r, sc_r)
    assert maxdiff(take_ranks(sc, perm), sc_r) < 1e-3 and boxes_close(take_ranks(bx, perm), bx_r)
    # training step (labels belong to the window's centre frame: one gt set per window)
    gt = np.array([[[5., 8., 40., 50.], [-1, -1, -1, -1]], [[10., 12., 30., 28.], [20., 5., 60., 62.]]])
    gid = np.array([[[1.], [-1.]], [[0.], [2.]]])
    tg = Y.prefetch_targets(size, size, [size // 32, size // 16, size // 8], gt, gid, c)
    out = net(dev(x), dev(gt), *[dev(t) for t in tg])
    net.backward()
    torch.cuda.synchronize()
    tb = net._programs[('buf', b, size, size, True)]
    from tests.util import device_leaky_masks, check_masks_differ_only_at_ties
    onet.mask_override = device_leaky_masks(net, tb)        # see oracle/ops.py leaky: tie-proof comparison
    from viddet_amd.model import PoolNode
    for n in net.nodes:                                      # same for the max-pool winner among the K frames
        if isinstance(n, PoolNode) and n.type == 0:
            onet.argmax_override[n.name] = np.moveaxis(tb['am:' + n.dst].cpu().numpy().astype(np.int64), -1, 1)
    losses_r, G, heads_t = onet.train_step(x.astype(np.float64), gt, *tg)
    print("leaky tie flips:", check_masks_differ_only_at_ties(onet.pre, onet.mask_override))
    for name, (am, v5) in onet.argmax_natural.items():
        if name in onet.argmax_override:
            d = am != onet.argmax_override[name]
            if d.any():      # a different winner is only acceptable between (numerically) equal candidates
                a = np.take_along_axis(v5, am[:, None], axis=1)[:, 0]
                bwin = np.take_along_axis(v5, onet.argmax_override[name][:, None], axis=1)[:, 0]
                assert np.abs(a - bwin)[d].max() < 2e-4, name
    for s, hname in enumerate(net.head_names):
        got = tb[hname].cpu().numpy()[..., :3 * (5 + c)]
        assert maxdiff(got, np.moveaxis(heads_t[s], 1, -1)) < 1e-3, "training-mode head %d" % s
    for i in range(4):
        assert np.all(np.abs(out[i].cpu().numpy() - losses_r[i]) <= 2e-3 * np.maximum(1.0, np.abs(losses_r[i])))
    for k, v in onet.new_running.items():
        assert maxdiff(net.collect_params()[k].data().cpu().numpy(), v) < 1e-4, k
    for n in net.nodes:
        if isinstance(n, ConvNode_) and n.name in onet.vars:
            ref = onet.vars[n.name].v
            got = np.moveaxis(tb[n.dst].cpu().numpy(), -1, 1)
            sc = max(1e-6, float(np.abs(ref).max()))
            print("Y  %-40s rel %.3e scale %.3e" % (n.name, maxdiff(got, ref) / sc, sc))
    # localise: gradient wrt every conv cell's output, in backward order
    from viddet_amd.model import ConvNode
    for n in reversed(net.nodes):
        if isinstance(n, ConvNode) and n.name in onet.vars and onet.vars[n.name].g is not None:
            gref = onet.vars[n.name].g
            got = np.moveaxis(tb['d:' + n.dst].cpu().numpy(), -1, 1)
            sc = max(1e-6, float(np.abs(gref).max()))
            print("dY %-40s rel %.3e scale %.3e" % (n.name, maxdiff(got, gref) / sc, sc))
            if maxdiff(got, gref) / sc > 1e-3 and not globals().get("_dumped"):
                globals()["_dumped"] = True
                e = np.abs(got - gref)
                print("   per-frame max err", e.max(axis=(1, 2, 3)))
                print("   per-row(y) max err", e.max(axis=(0, 1, 3)))
                print("   per-col(x) max err", e.max(axis=(0, 1, 2)))
                ch = e.max(axis=(0, 2, 3))
                print("   worst channels", np.argsort(-ch)[:8], ch[np.argsort(-ch)[:8]], "median ch err", np.median(ch))
                print("   ratio got/ref at worst", got.ravel()[e.argmax()], gref.ravel()[e.argmax()])
    bad, table = [], []
    for k, gref in G.items():
        got = net.collect_params()[k].grad().cpu().numpy()
        scale = max(1e-3, float(np.abs(gref).max()))
        err = maxdiff(got, gref) / scale
        table.append("%-48s %.3e %.3e" % (k, err, scale))
        if err >= 5e-4:
            bad.append((k, err))
    print("\n".join(table))
    assert not bad, bad[:6]


def test_temporal_flag_guards():
    from viddet_amd.model import yolo3_darknet53
    with pytest.raises(AssertionError):
        yolo3_darknet53(["a"], k=1, block_conv_type="3")                      # yolo3.py:979
    with pytest.raises(AssertionError):
        yolo3_darknet53(["a"], k=3, k_join_type="max", k_join_pos="early", block_conv_type="3")   # :980
    with pytest.raises(NotImplementedError):
        yolo3_darknet53(["a"], k=3)                                           # a join type / position is required
