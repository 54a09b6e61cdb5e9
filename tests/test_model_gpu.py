"""GPU parity of the whole yolo3_darknet53 path (k=1) against the fp64 oracle network (oracle/net.py):
inference (boxes/scores within 1e-3, identical post-NMS row indices) and one training step
(losses, every parameter gradient, BN running statistics, SGD-momentum update)."""
import numpy as np
import pytest
import torch

from oracle import net as ON
from oracle import ops as R
from oracle import yolo as Y
from tests.util import dev, maxdiff

pytestmark = pytest.mark.gpu


def _mk_net(num_class, seed, obj_bias):
    from viddet_amd.model import yolo3_darknet53
    classes = ["c%d" % i for i in range(num_class)]
    net = yolo3_darknet53(classes)
    P = ON.init_params(num_class, seed=seed, obj_bias=obj_bias)
    for k, p in net.collect_params().items():
        p.set_data(torch.from_numpy(P[k].astype(np.float32)))
    return net, P


def test_param_roundtrip_and_count():
    net, P = _mk_net(80, 1, 0.0)
    tot = sum(int(np.prod(p.shape)) for p in net.collect_params().values())
    assert tot == 62001757, tot      # darknet53 + yolo heads incl. BN running stats (C=80), SURVEY 6: 62.00 M
    for k in ["stages.0.0.0.weight", "stages.0.2.body.1.0.weight", "yolo_outputs.1.prediction.weight",
              "yolo_outputs.2.prediction.bias", "transitions.0.1.gamma", "stages.2.4.body.0.1.running_var"]:
        assert maxdiff(net.collect_params()[k].data().cpu().numpy(), P[k]) == 0.0, k


@pytest.mark.parametrize("cfg", [dict(b=2, c=4, size=64, seed=3), dict(b=1, c=20, size=96, seed=4),
                                 dict(b=1, c=20, size=608, seed=5, obj_bias=1.0)])      # one full 608x608 frame, most of its
                                                                                      # C*P = 454,860 rows above valid_thresh
def test_inference_matches_oracle(cfg):
    b, c, size = cfg["b"], cfg["c"], cfg["size"]
    net, P = _mk_net(c, cfg["seed"], obj_bias=cfg.get("obj_bias", -1.0))
    rng = np.random.default_rng(cfg["seed"])
    x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
    onet = ON.Net(P, c)
    ids_r, sc_r, bx_r, rows_r, heads_r = onet.detect(x.astype(np.float64))
    ids, sc, bx = net(dev(x))
    torch.cuda.synchronize()
    # raw head tensors first (localises a failure to the conv stack)
    bufs = net._programs[('buf', b, size, size, False)]
    for s, hname in enumerate(net.head_names):
        got = bufs[hname].cpu().numpy()[..., :3 * (5 + c)]
        ref = np.moveaxis(heads_r[s], 1, -1)
        assert maxdiff(got, ref) < 1e-3, "head %d" % s
    assert int(net.last_overflow.max()) == 0
    if size == 608:      # more candidates than the 2^18 cap of round 1, which dropped rows in arrival order here
        assert int(net._programs[('infer', b, size, size)][2]['counts'].max()) > (1 << 18)
    from tests.util import assert_rows_match, take_ranks
    perm = assert_rows_match(net.last_rows.cpu().numpy(), rows_r, sc_r)
    ids, sc, bx = [torch.from_numpy(take_ranks(t, perm)) for t in (ids, sc, bx)]
    assert np.array_equal(ids.cpu().numpy(), ids_r)
    assert maxdiff(sc.cpu().numpy(), sc_r) < 1e-3
    # boxes: 1e-3 px + 1e-5 of the coordinate (tests/util.py boxes_close: twice the measured fp32 round-off of the 75-layer
    # stack at every box scale); in detect()'s normalised units (detect_yolo3.py:257, boxes / W) north_star's 1e-3 is met
    # with a 10x margin
    from tests.util import boxes_close
    boxes_close(bx.cpu().numpy(), bx_r)
    assert maxdiff(bx.cpu().numpy() / size, bx_r / size) < 1e-4
    assert int((ids_r >= 0).sum()) > 0, "fixture produced no detections"


def test_inference_hip_graph_replay_identical():
    net, P = _mk_net(4, 5, obj_bias=-1.0)
    rng = np.random.default_rng(5)
    x = dev(rng.standard_normal((2, 3, 64, 64)).astype(np.float32))
    a = [t.clone() for t in net(x)]
    net.use_graphs = True
    b1 = [t.clone() for t in net(x)]
    b2 = [t.clone() for t in net(x)]
    torch.cuda.synchronize()
    for u, v, w in zip(a, b1, b2):
        assert torch.equal(u, v) and torch.equal(u, w)


def _targets(rng, b, c, size, m):
    gt = np.full((b, m, 4), -1.0)
    ids = np.full((b, m, 1), -1.0)
    for bi in range(b):
        for j in range(rng.integers(1, m + 1)):
            cx, cy = rng.uniform(0.15, 0.85, 2) * size
            w, h = rng.uniform(6, 0.6 * size, 2)
            gt[bi, j] = [max(cx - w / 2, 0), max(cy - h / 2, 0), min(cx + w / 2, size - 1), min(cy + h / 2, size - 1)]
            ids[bi, j, 0] = rng.integers(0, c)
    grids = [size // 32, size // 16, size // 8]
    return gt, Y.prefetch_targets(size, size, grids, gt, ids, c)


@pytest.mark.parametrize("cfg", [(2, 4, 64, 3), (1, 20, 416, 6), (3, 4, 64, 3, "chunks")])     # the second: one 416x416 voc frame, full-size grids
def test_training_step_matches_oracle(cfg, monkeypatch):
    # the third: the stride-2 data gradients cut into frame chunks (VD_S2_CHUNK_MB, model.py: one HBM read of dz instead of
    # four at BASELINE's sizes) - forced here by a tiny chunk size, three frames in chunks of one and two
    if len(cfg) == 5:
        monkeypatch.setenv("VD_S2_CHUNK_MB", "0.3")
        cfg = cfg[:4]
    b, c, size, m = cfg
    net, P = _mk_net(c, 6, obj_bias=-1.0)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, b, c, size, m)
    out = net(dev(x), dev(gt), *[dev(t) for t in tg])
    net.backward()
    torch.cuda.synchronize()
    # the oracle takes the device's LeakyReLU branch decisions (they can differ only at |pre-activation| ~ 1e-7
    # ties, about one element per step; see oracle/ops.py leaky) and must agree with them everywhere else
    from tests.util import device_leaky_masks, check_masks_differ_only_at_ties
    onet = ON.Net(P, c)
    onet.mask_override = device_leaky_masks(net, net._last_train['bufs'])
    losses_r, G, _ = onet.train_step(x.astype(np.float64), gt, *tg)
    nflip = check_masks_differ_only_at_ties(onet.pre, onet.mask_override)
    print("leaky tie flips:", nflip)
    for i in range(4):
        lr = losses_r[i]
        assert np.all(np.abs(out[i].cpu().numpy() - lr) <= 2e-3 * np.maximum(1.0, np.abs(lr))), (i, out[i], lr)
    # running stats
    for k, v in onet.new_running.items():
        assert maxdiff(net.collect_params()[k].data().cpu().numpy(), v) < 1e-4, k
    # every parameter gradient, relative to that tensor's own scale
    errs = []
    for k, gref in G.items():
        got = net.collect_params()[k].grad().cpu().numpy()
        scale = max(1e-3, float(np.abs(gref).max()))
        errs.append((maxdiff(got, gref) / scale, k, scale))
    # (measured <= 5e-5 of a tensor's max on both fixtures: the bound is 10x that)
    bad = [e for e in errs if e[0] >= 5e-4]
    print("\n".join("%-45s rel_err %.3e  scale %.3e" % (k, e, sc) for e, k, sc in errs))
    assert not bad, "gradient mismatch: %s" % bad[:8]
    # SGD-momentum step on top (wd on everything, rescale 1/batch)
    w0 = {k: p.data().cpu().numpy().astype(np.float64) for k, p in net.collect_params().items() if k in G}
    net.sgd_step(lr=0.01, momentum=0.9, wd=5e-4, batch_size=b)
    torch.cuda.synchronize()
    for k in list(G.keys())[::7]:
        gdev = net.collect_params()[k].grad().cpu().numpy().astype(np.float64)
        wr, _ = R.sgd_momentum(w0[k], gdev, np.zeros_like(gdev), 0.01, 0.9, 5e-4, 1.0 / b)
        assert maxdiff(net.collect_params()[k].data().cpu().numpy(), wr) < 1e-6, k


def test_plans_of_two_shapes_each_follow_the_weights():
    """Every (batch, height, width) gets its own plan with its own repacked weight images (data-gradient layout for
    training, bf16 for inference).  Alternating shapes with no optimiser step in between, and changing the weights between
    calls, must give what a fresh network gives (round 2: a net-wide 'packed' flag left the second shape's images unpacked)."""
    c = 4
    rng = np.random.default_rng(12)
    xa = rng.standard_normal((1, 3, 64, 64)).astype(np.float32)
    xb = rng.standard_normal((2, 3, 96, 96)).astype(np.float32)
    ta, tb = _targets(rng, 1, c, 64, 3), _targets(rng, 2, c, 96, 3)

    def step(net, x, t):
        net(dev(x), dev(t[0]), *[dev(v) for v in t[1]])
        net.backward()
        torch.cuda.synchronize()
        return {k: p.grad().clone() for k, p in net.collect_params().items() if p.span is not None}

    net, P = _mk_net(c, 12, obj_bias=-1.0)
    step(net, xa, ta)
    gb = step(net, xb, tb)                      # second shape, no step in between
    fresh, _ = _mk_net(c, 12, obj_bias=-1.0)
    gref = step(fresh, xb, tb)
    for k in gref:
        assert bool(torch.isfinite(gb[k]).all()) and torch.equal(gb[k], gref[k]), k
    # bf16 inference: two shapes, then new weights, then the shapes in the other order
    net.set_precision('bf16'); fresh.set_precision('bf16')
    net(dev(xa)); net(dev(xb))
    P2 = ON.init_params(c, seed=13, obj_bias=-1.0)
    for n_ in (net, fresh):
        for k, p in n_.collect_params().items():
            p.set_data(torch.from_numpy(P2[k].astype(np.float32)))
    ob = [t.clone() for t in net(dev(xb))]
    oa = [t.clone() for t in net(dev(xa))]
    ra = [t.clone() for t in fresh(dev(xa))]
    rb = [t.clone() for t in fresh(dev(xb))]
    torch.cuda.synchronize()
    for u, v in zip(oa + ob, ra + rb):
        assert torch.equal(u, v)


def test_freeze_base_leaves_the_backbone_untouched():
    """wrappers.py:55-57: freeze_base sets grad_req = 'null' on every Darknet parameter - no gradient, no update, no
    weight decay for stages.*; the neck / heads train exactly as in the unfrozen network (their gradients do not
    depend on anything below the three route tensors); BatchNorm still normalises with batch statistics and moves
    its running statistics (training mode).  The backward schedule drops the backbone's launches altogether."""
    from viddet_amd.model import yolo3_darknet53
    b, c, size, m = 2, 4, 64, 3
    rng = np.random.default_rng(8)
    x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, b, c, size, m)
    P = ON.init_params(c, seed=8, obj_bias=-1.0)
    nets = []
    for freeze in (False, True):
        net = yolo3_darknet53(["c%d" % i for i in range(c)], freeze_base=freeze)
        for k, p in net.collect_params().items():
            p.set_data(torch.from_numpy(P[k].astype(np.float32)))
        nets.append(net)
    free, frozen = nets
    assert all(p.grad_req == 'null' for k, p in frozen.collect_params('stages.*').items())
    assert all(p.grad_req == 'write' for k, p in frozen.collect_params('(yolo|trans).*(weight|gamma|beta|bias)').items())
    outs = []
    before = {k: p.data().clone() for k, p in frozen.collect_params().items()}
    for net in nets:
        out = net(dev(x), dev(gt), *[dev(t) for t in tg])
        net.backward()
        outs.append([o.clone() for o in out])
    torch.cuda.synchronize()
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)                  # the forward pass is the same program
    nw = lambda net: sum(1 for seg in net._last_train['bwd'] if hasattr(seg, 'recs') for r in seg.recs
                         if r[0] in ('vd_conv_wgrad', 'vd_stem_wgrad'))
    assert nw(free) == 75 and nw(frozen) == 75 - 52, (nw(free), nw(frozen))
    for net in nets:
        net.sgd_step(lr=0.01, momentum=0.9, wd=5e-4, batch_size=b)
    torch.cuda.synchronize()
    for k, p in frozen.collect_params().items():
        now = p.data()
        if k.startswith("stages") and "running" not in k:
            assert torch.equal(now, before[k]), "frozen tensor %s changed" % k
            assert float(p.grad().abs().max()) == 0.0, k
        elif k.startswith("stages"):
            assert not torch.equal(now, before[k]), "running statistic %s did not move" % k
        else:
            # same gradient as in the unfrozen network (same launches on the same inputs; tile choices of the two plans
            # may differ, hence a tolerance) and the same update
            if "running" not in k:
                g0, g1 = free.collect_params()[k].grad(), p.grad()
                assert float((g0 - g1).abs().max()) <= 1e-5 * max(1e-3, float(g0.abs().max())), k
                assert float((free.collect_params()[k].data() - now).abs().max()) <= 1e-7 + 1e-5 * float(now.abs().max()), k
                assert not torch.equal(now, before[k]), k
    assert float(frozen.momentum_buf[:frozen.conv_nodes[52].w_off].abs().max()) == 0.0
    # thawing rebuilds the schedule
    for k, p in frozen.collect_params('stages.*(weight|gamma|beta)').items():
        p.grad_req = 'write'
    frozen(dev(x), dev(gt), *[dev(t) for t in tg])
    frozen.backward()
    torch.cuda.synchronize()
    assert nw(frozen) == 75


def test_wd_mult_and_lr_mult_are_applied():
    """Per-parameter multipliers of the optimiser (train_yolov3.py:495-497 sets wd_mult = 0 on gamma / beta / bias)."""
    net, P = _mk_net(4, 9, obj_bias=-1.0)
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 3, 64, 64)).astype(np.float32)
    gt, tg = _targets(rng, 2, 4, 64, 3)
    net(dev(x), dev(gt), *[dev(t) for t in tg])
    net.backward()
    pg, pw = net.collect_params()["yolo_blocks.1.body.2.1.gamma"], net.collect_params()["yolo_blocks.0.tip.0.weight"]
    pg.wd_mult = 0.0
    pw.lr_mult, pw.wd_mult = 0.5, 2.0
    others = ["transitions.0.1.beta", "stages.1.3.body.1.0.weight"]
    w0 = {k: net.collect_params()[k].data().cpu().numpy().astype(np.float64) for k in [pg.name, pw.name] + others}
    g = {k: net.collect_params()[k].grad().cpu().numpy().astype(np.float64) for k in w0}
    net.sgd_step(lr=0.01, momentum=0.9, wd=5e-4, batch_size=2)
    torch.cuda.synchronize()
    for k, (lm, wm) in {pg.name: (1.0, 0.0), pw.name: (0.5, 2.0), others[0]: (1.0, 1.0), others[1]: (1.0, 1.0)}.items():
        wr, _ = R.sgd_momentum(w0[k], g[k], np.zeros_like(g[k]), 0.01 * lm, 0.9, 5e-4 * wm, 0.5)
        assert maxdiff(net.collect_params()[k].data().cpu().numpy(), wr) < 1e-6, k
    assert len(net._optimizer_ranges()) >= 4


def test_uint8_frames_are_normalised_on_the_device():
    """Loaders ship uint8 (B,H,W,3) frames; vd_preprocess_u8_nchw does transforms.py:239-245 (to_tensor: /255, normalize:
    (x - mean) / std, HWC -> CHW) on the GPU.  The planar batch equals the oracle's preprocessing, and the detections are
    BIT-IDENTICAL to feeding the host-normalised fp32 batch."""
    from viddet_amd.data import _to_tensor_normalize
    net, P = _mk_net(4, 12, obj_bias=-1.0)
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    a = [t.clone() for t in net(torch.from_numpy(img).cuda())]
    inbuf = net._programs[('buf', 2, 64, 64, False)]['in'].cpu().numpy()
    ref = np.stack([R.preprocess_u8(img[i]) for i in range(2)])
    assert maxdiff(inbuf, ref) < 1e-6
    host = np.stack([_to_tensor_normalize(img[i]) for i in range(2)])
    assert np.array_equal(inbuf, host)                      # the same fp32 operations in the same order
    b_ = net(torch.from_numpy(host).cuda())
    torch.cuda.synchronize()
    for u, v in zip(a, b_):
        assert torch.equal(u, v)
    with pytest.raises(ValueError):
        net(torch.from_numpy(img[:, :, :, :2].copy()).cuda())


def test_reset_class_reuses_rows():
    net, P = _mk_net(4, 7, obj_bias=0.0)
    old = net.collect_params()["yolo_outputs.0.prediction.weight"].data().cpu().numpy()
    net.reset_class(["c2", "new"], reuse_weights={"c2": "c2"})
    new = net.collect_params()["yolo_outputs.0.prediction.weight"].data().cpu().numpy()
    assert new.shape[0] == 3 * 7
    for a in range(3):
        assert np.array_equal(new[a * 7:a * 7 + 5], old[a * 9:a * 9 + 5])
        assert np.array_equal(new[a * 7 + 5], old[a * 9 + 5 + 2])
    x = dev(np.random.default_rng(0).standard_normal((1, 3, 64, 64)).astype(np.float32))
    ids, sc, bx = net(x)
    torch.cuda.synchronize()
    assert ids.shape == (1, 100, 1)
    # the fine-tuning order of train_yolov3.py (--freeze_base --trained_on ...): freeze, then reset the classes - the
    # backbone stays frozen, the rebuilt prediction convs train
    net2, _ = _mk_net(4, 7, obj_bias=0.0)
    for k, p in net2.collect_params().items():
        if k.startswith("stages.") and p.span is not None:
            p.grad_req = 'null'
    net2.reset_class(["a", "b", "c"])
    P2 = net2.collect_params()
    assert P2["stages.0.0.0.weight"].grad_req == 'null' and P2["yolo_outputs.0.prediction.weight"].grad_req == 'write'
    assert P2["yolo_blocks.0.tip.0.weight"].grad_req == 'write'
    rng = np.random.default_rng(3)
    x2 = rng.standard_normal((2, 3, 64, 64)).astype(np.float32)
    gt, tg = _targets(rng, 2, 3, 64, 3)
    w0 = {k: p.data().clone() for k, p in P2.items()}
    net2(dev(x2), dev(gt), *[dev(t) for t in tg])
    net2.backward()
    net2.sgd_step(lr=0.01, momentum=0.9, wd=5e-4, batch_size=2)
    torch.cuda.synchronize()
    assert torch.equal(P2["stages.2.4.body.1.0.weight"].data(), w0["stages.2.4.body.1.0.weight"])
    assert not torch.equal(P2["yolo_outputs.1.prediction.weight"].data(), w0["yolo_outputs.1.prediction.weight"])


def test_data_parallel_two_ranks_equal_one_process():
    """tools/dp_equivalence.py: two ranks (gloo on the one GPU; RCCL needs one device per rank) with SyncBN('all'),
    bucketed gradient all-reduce and the global-batch rescale reproduce the single-process step on the whole batch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if not k.startswith("VD_DP_")}
    # duplicate shards: bit-identical step; real shards: the update agrees as a whole; then the BASELINE configs[3] family -
    # k = 3 windows with the SyncBN scope the reference's --syncbn reaches (stem + stride-2 convs), duplicate shards; and
    # three consecutive steps (momentum, the repacked weight layouts and max-abs slots after each update), bit-identical
    # ... and bf16-storage training (BASELINE configs[4]'s mode: 8 GPUs, bf16) under the same SyncBN('all') + bucketed
    # all-reduce: duplicate shards bit-identical over two steps, real shards within the bf16 tolerance
    # (one set of processes runs the seven cases one after the other: a process start costs more than a case)
    import json
    cases = [dict(DUP=1), dict(), dict(DUP=1, K=3, SCOPE="reference"), dict(K=3), dict(DUP=1, STEPS=3),
             dict(STORAGE="bf16", DUP=1, STEPS=2), dict(STORAGE="bf16")]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dp_equivalence.py"), "2"], capture_output=True,
                       text=True, timeout=900, env=dict(base, VD_DP_CASES=json.dumps(cases)))
    assert r.returncode == 0 and r.stdout.count("dp_equivalence ok") == len(cases), (r.stdout[-1500:], r.stderr[-1500:])


def test_single_rank_rccl_path_and_exchange_diagnostics():
    """bench.py with VD_FORCE_DIST=1: the RCCL code path (communicator, bucketed asynchronous all-reduce behind the
    weight-gradient stream, tail all-reduce) on ONE rank, and the diagnostics the first multi-GPU run will print
    (buckets per step, exposed all-reduce time, bus GB/s of one all-reduce of the whole gradient arena)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VD_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", VD_BUCKET_MB="8")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--size", "64", "--classes", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900,
                       env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    ph = out["phases"]
    assert out["n_gpus"] == 1 and ph["allreduce"]["buckets_per_step"] >= 4 and ph["allreduce"]["arena_mb"] > 200
    assert ph["allreduce_exposed_ms"] >= 0 and ph["bwd_local_ms"] > 0 and ph["allreduce"]["algo_gb_s"] > 0


def test_bench_two_ranks_launched_the_drivers_way():
    """bench.py under the launcher line the driver uses for N > 1 (python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...), rehearsed with two ranks sharing the
    one GPU over gloo (VD_REHEARSE_SHARED_GPU=1; RCCL wants one device per rank): ONE JSON line from rank 0, whole-job frames/s
    (both ranks' frames over the max-over-ranks time), the exchange diagnostics, weak scaling."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VD_REHEARSE_SHARED_GPU="1", VD_BUCKET_MB="8")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "2", "--size", "64", "--classes", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert abs(out["value"] - 4 / (out["ms_per_step"] * 1e-3)) < 0.02 * out["value"]
    assert out["cpu_baseline"] is None and out["roofline"]["frac"] > 0
    assert out["phases"]["allreduce"]["buckets_per_step"] >= 4 and out["phases"]["allreduce_exposed_ms"] >= 0


@pytest.mark.parametrize("cfg", [(2, 4, 64, 3), (1, 20, 416, 6)])
def test_fp16_split_arithmetic_matches_oracle(cfg):
    """set_conv_math('split2') (VD_MATH_F16X2: two fp16 pieces per operand, per-tensor power-of-two scales from the
    producers' max-abs, three MFMAs per product block): the SAME tolerances as the fp32 arithmetics - heads 1e-3, identical
    post-NMS rows, losses 2e-3, every one of the 222 gradients 5e-4 of its tensor's max - on the small fixture and on one
    full 416x416 frame; and the launches really ran in it."""
    from viddet_amd import model as M
    from viddet_amd import lib as L
    from tests.util import device_leaky_masks, check_masks_differ_only_at_ties, assert_rows_match, take_ranks
    M.set_conv_math("split2")
    M._TUNE_CACHE.clear()
    try:
        b, c, size, m = cfg
        net, P = _mk_net(c, 6, obj_bias=-1.0)
        rng = np.random.default_rng(6)
        x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
        onet = ON.Net(P, c)
        ids_r, sc_r, bx_r, rows_r, heads_r = onet.detect(x.astype(np.float64))
        ids, sc, bx = net(dev(x))
        torch.cuda.synchronize()
        bufs = net._programs[('buf', b, size, size, False)]
        for s_, hname in enumerate(net.head_names):
            assert maxdiff(bufs[hname].cpu().numpy()[..., :3 * (5 + c)], np.moveaxis(heads_r[s_], 1, -1)) < 1e-3, s_
        perm = assert_rows_match(net.last_rows.cpu().numpy(), rows_r, sc_r)
        assert maxdiff(take_ranks(sc, perm), sc_r) < 1e-3
        # boxes: 1e-3 px, or 2e-5 of the coordinate for the exp()-blown boxes of this random-init fixture (+-1700 px)
        assert np.all(np.abs(take_ranks(bx, perm) - bx_r) <= 1e-3 + 2e-5 * np.abs(bx_r))
        used = [bool(a[0]._obj.flags & L.MATH_F16X2) for (fn_, _, a) in net._programs[('infer', b, size, size)][0].recs
                if fn_ == 'vd_conv_igemm']
        assert sum(used) >= 70, sum(used)
        gt, tg = _targets(rng, b, c, size, m)
        out = net(dev(x), dev(gt), *[dev(t) for t in tg])
        net.backward()
        torch.cuda.synchronize()
        used = [bool(a[0]._obj.flags & L.MATH_F16X2) for seg in net._last_train['fwd'] + net._last_train['bwd']
                if hasattr(seg, 'recs') for (fn_, _, a) in seg.recs if fn_ in ('vd_conv_igemm', 'vd_conv_wgrad')]
        assert sum(used) > 200, "the fp16-split arithmetic was not selected (%d launches)" % sum(used)
        onet = ON.Net(P, c)
        onet.mask_override = device_leaky_masks(net, net._last_train['bufs'])
        losses_r, G, _ = onet.train_step(x.astype(np.float64), gt, *tg)
        check_masks_differ_only_at_ties(onet.pre, onet.mask_override)
        for i in range(4):
            assert np.all(np.abs(out[i].cpu().numpy() - losses_r[i]) <= 2e-3 * np.maximum(1.0, np.abs(losses_r[i]))), i
        worst = 0.0
        for k, gref in G.items():
            got = net.collect_params()[k].grad().cpu().numpy()
            e = maxdiff(got, gref) / max(1e-3, float(np.abs(gref).max()))
            worst = max(worst, e)
            assert e < 5e-4, (k, e)
        print("fp16-split arithmetic: worst gradient error / tensor max = %.2e" % worst)
        for k, v in onet.new_running.items():
            assert maxdiff(net.collect_params()[k].data().cpu().numpy(), v) < 1e-4, k
    finally:
        M.set_conv_math(None)
        M._TUNE_CACHE.clear()


def test_training_step_in_bf16_products():
    """set_conv_math('bf16'): convolution products on bf16-rounded operands, everything else fp32 (the mixed-precision
    training arithmetic of BASELINE configs[4]).  Losses within 5 % of the fp64 oracle's, gradients correlated to it."""
    from viddet_amd import model as M
    M.set_conv_math("bf16")
    try:
        b, c, size, m = 2, 4, 64, 3
        net, P = _mk_net(c, 6, obj_bias=-1.0)
        rng = np.random.default_rng(6)
        x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
        gt, tg = _targets(rng, b, c, size, m)
        out = net(dev(x), dev(gt), *[dev(t) for t in tg])
        net.backward()
        torch.cuda.synchronize()
        used = [bool(a[0]._obj.flags & 32) for seg in net._last_train['fwd'] + net._last_train['bwd'] if hasattr(seg, 'recs')
                for (fn_, _, a) in seg.recs if fn_ in ('vd_conv_igemm', 'vd_conv_wgrad')]
        assert sum(used) > 100, "the bf16-product arithmetic was not selected"
        onet = ON.Net(P, c)
        losses_r, G, _ = onet.train_step(x.astype(np.float64), gt, *tg)
        for i in range(4):
            # 5 %: bf16 operands (2^-8 each) through 75 layers; which tiles the autotuner picks moves a loss by ~1 %
            assert np.all(np.abs(out[i].cpu().numpy() - losses_r[i]) <= 5e-2 * np.maximum(1.0, np.abs(losses_r[i]))), i
        # bf16 rounding of every operand, through 75 layers whose deepest BatchNorms see 8 samples on this fixture,
        # leaves the gradient close in direction to the oracle's, not equal: whole-gradient cosine, and per tensor
        cos, dot, ng, nr = [], 0.0, 0.0, 0.0
        for k in G.keys():
            g, r_ = net.collect_params()[k].grad().cpu().numpy().ravel().astype(np.float64), G[k].ravel()
            cos.append(float(g @ r_ / (np.linalg.norm(g) * np.linalg.norm(r_) + 1e-30)))
            dot, ng, nr = dot + float(g @ r_), ng + float(g @ g), nr + float(r_ @ r_)
        whole = dot / np.sqrt(ng * nr)
        print("bf16-product gradients: whole cosine %.4f, per-tensor mean %.3f min %.3f" % (whole, np.mean(cos), min(cos)))
        assert whole > 0.9 and np.mean(cos) > 0.7 and min(cos) > 0.3, (whole, np.mean(cos), min(cos))
    finally:
        M.set_conv_math(None)
        M._TUNE_CACHE.clear()


def test_configs4_shape_combined_classes_bf16_products():
    """BASELINE configs[4] (combined dataset: 285 classes, 608x608, bf16): the mixed-precision arithmetic built for it
    (set_conv_math('bf16'): bf16-rounded conv operands, fp32 accumulation / tensors / optimiser) with the 285-class heads
    (A = 870 channels) - a training step against the fp64 oracle at the bf16 tolerance of
    test_training_step_in_bf16_products, and one full 608x608 frame through forward, loss, backward and the update."""
    from viddet_amd import model as M
    M.set_conv_math("bf16")
    M._TUNE_CACHE.clear()
    try:
        b, c, size, m = 2, 285, 96, 4
        net, P = _mk_net(c, 16, obj_bias=-1.0)
        rng = np.random.default_rng(16)
        x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
        gt, tg = _targets(rng, b, c, size, m)
        out = net(dev(x), dev(gt), *[dev(t) for t in tg])
        net.backward()
        torch.cuda.synchronize()
        losses_r, G, _ = ON.Net(P, c).train_step(x.astype(np.float64), gt, *tg)
        for i in range(4):
            assert np.all(np.abs(out[i].cpu().numpy() - losses_r[i]) <= 5e-2 * np.maximum(1.0, np.abs(losses_r[i]))), i
        dot = ng = nr = 0.0
        for k in G.keys():
            g, r_ = net.collect_params()[k].grad().cpu().numpy().ravel().astype(np.float64), G[k].ravel()
            dot, ng, nr = dot + float(g @ r_), ng + float(g @ g), nr + float(r_ @ r_)
        assert dot / np.sqrt(ng * nr) > 0.9
        size = 608
        x = rng.standard_normal((1, 3, size, size)).astype(np.float32)
        gt, tg = _targets(rng, 1, c, size, 8)
        w0 = net.collect_params()["yolo_outputs.2.prediction.weight"].data().clone()
        out = net(dev(x), dev(gt), *[dev(t) for t in tg])
        net.backward()
        net.sgd_step(lr=1e-3, momentum=0.9, wd=5e-4, batch_size=1)
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(o).all()) for o in out) and bool(torch.isfinite(net.weights).all())
        assert not torch.equal(w0, net.collect_params()["yolo_outputs.2.prediction.weight"].data())
    finally:
        M.set_conv_math(None)
        M._TUNE_CACHE.clear()


def test_plan_eviction_when_the_next_shape_does_not_fit_beside_the_others(monkeypatch):
    """Random-shape training keeps one plan (buffers + programs) per input shape; when the next shape's plan does not fit
    beside the cached ones they are dropped and the build is retried (`_build_or_evict`).  The retry must run OUTSIDE the
    except block: inside it the failed build's half-allocated buffers are still referenced by the live traceback and
    cannot be returned to the driver (ADVICE, round 2).  Forced here with a per-process memory cap between 'both plans' and
    'the larger plan alone'; the step after eviction must equal a fresh network's, bit for bit (pinned kernels)."""
    monkeypatch.setenv("VD_AUTOTUNE", "0")
    c = 4
    rng = np.random.default_rng(77)
    xa = rng.standard_normal((8, 3, 256, 256)).astype(np.float32)
    xb = rng.standard_normal((8, 3, 320, 320)).astype(np.float32)
    ta, tb = _targets(rng, 8, c, 256, 3), _targets(rng, 8, c, 320, 3)

    def step(net, x, t):
        out = net(dev(x), dev(t[0]), *[dev(v) for v in t[1]])
        net.backward()
        torch.cuda.synchronize()
        return [o.clone() for o in out], net.grads.clone()

    def reserved():
        torch.cuda.synchronize()
        return torch.cuda.memory_reserved()

    total = torch.cuda.get_device_properties(0).total_memory
    try:
        import gc
        fresh, _ = _mk_net(c, 15, obj_bias=-1.0)
        ref = step(fresh, xb, tb)
        del fresh
        gc.collect(); torch.cuda.empty_cache()
        net, P = _mk_net(c, 15, obj_bias=-1.0)
        base = reserved()
        step(net, xa, ta)
        m_a = reserved() - base
        # room for plan A plus HALF of what plan B needs beside it (B's buffers are (320/256)^2 = 1.56 x A's)
        cap = reserved() + int(0.8 * m_a)
        torch.cuda.set_per_process_memory_fraction(min(1.0, cap / total))
        got = step(net, xb, tb)                       # does not fit beside A: A is evicted, B is built again and runs
        assert ('train', 8, 320, 320) in net._programs and ('train', 8, 256, 256) not in net._programs
        assert all(torch.equal(a, b) for a, b in zip(got[0], ref[0])) and torch.equal(got[1], ref[1])
        step(net, xa, ta)                             # and back: B is evicted in turn
        assert ('train', 8, 256, 256) in net._programs
    finally:
        torch.cuda.set_per_process_memory_fraction(1.0)
