"""GPU parity of the cached-feature path: extract_features() of the full network, and the
no-backbone network YOLOV3_noback (net(x1, x2, x3[, targets])) in inference and in one training step, against the
fp64 oracle (oracle/net.py backbone / NoBackNet).  Same tolerances as tests/test_model_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import net as ON
from tests.util import dev, maxdiff, boxes_close
from tests.test_model_gpu import _mk_net, _targets

pytestmark = pytest.mark.gpu


def _mk_noback(num_class, P):
    from viddet_amd.model import yolo3_no_backbone
    net = yolo3_no_backbone(["c%d" % i for i in range(num_class)])
    names = list(net.collect_params().keys())
    assert names and not any(k.startswith("stages.") for k in names)
    for k, p in net.collect_params().items():            # the neck/head subset of the full network's parameters
        p.set_data(torch.from_numpy(P[k].astype(np.float32)))
    return net


def test_extract_features_and_noback_inference():
    b, c, size = 2, 4, 64
    full, P = _mk_net(c, 3, obj_bias=-1.0)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
    f1, f2, f3 = full.extract_features(dev(x))
    torch.cuda.synchronize()
    assert tuple(f1.shape) == (b, 256, size // 8, size // 8) and tuple(f3.shape) == (b, 1024, size // 32, size // 32)
    onet = ON.Net(P, c)
    routes = onet.backbone(x.astype(np.float64), train=False)
    for got, ref in zip((f1, f2, f3), routes):
        assert maxdiff(got.cpu().numpy(), ref.v) < 1e-3
    # the no-backbone network on those features reproduces the full network's detections bit for bit
    nb = _mk_noback(c, P)
    ids_f, sc_f, bx_f = [t.clone() for t in full(dev(x))]
    ids, sc, bx = nb(f1, f2, f3)
    torch.cuda.synchronize()
    assert torch.equal(ids, ids_f) and int((ids >= 0).sum()) > 0
    assert float((sc - sc_f).abs().max()) < 1e-6 and float((bx - bx_f).abs().max()) < 1e-4
    # ... and the oracle's, from the oracle's own fp64 features
    ids_r, sc_r, bx_r, rows_r, _ = ON.NoBackNet(P, c).detect([r.v for r in routes])
    from tests.util import assert_rows_match, take_ranks
    perm = assert_rows_match(nb.last_rows.cpu().numpy(), rows_r, sc_r)
    assert np.array_equal(take_ranks(ids, perm), ids_r)
    assert maxdiff(take_ranks(sc, perm), sc_r) < 1e-3 and boxes_close(take_ranks(bx, perm), bx_r)


def test_noback_call_protocol_errors():
    nb = _mk_noback(4, ON.init_params(4, seed=1))
    with pytest.raises(TypeError):
        nb(torch.zeros(1, 256, 8, 8, device="cuda"))
    with pytest.raises(ValueError):
        nb(torch.zeros(1, 255, 8, 8, device="cuda"), torch.zeros(1, 512, 4, 4, device="cuda"), torch.zeros(1, 1024, 2, 2, device="cuda"))
    with pytest.raises(NotImplementedError):
        nb.extract_features(torch.zeros(1, 3, 64, 64, device="cuda"))


def test_noback_training_step_matches_oracle():
    b, c, size, m = 2, 20, 64, 3          # 20 classes: 75 head channels padded to 96 (a pitch that is no power of two)
    P = ON.init_params(c, seed=8, obj_bias=-1.0)
    nb = _mk_noback(c, P)
    rng = np.random.default_rng(8)
    feats = [rng.standard_normal((b, ch, size // d, size // d)).astype(np.float32) * 0.7
             for ch, d in ((256, 8), (512, 16), (1024, 32))]
    gt, tg = _targets(rng, b, c, size, m)
    out = nb(*[dev(f) for f in feats], dev(gt), *[dev(t) for t in tg])
    nb.backward()
    torch.cuda.synchronize()
    from tests.util import device_leaky_masks, check_masks_differ_only_at_ties
    onet = ON.NoBackNet(P, c)
    onet.mask_override = device_leaky_masks(nb, nb._last_train['bufs'])
    losses_r, G, _ = onet.train_step([f.astype(np.float64) for f in feats], gt, *tg)
    check_masks_differ_only_at_ties(onet.pre, onet.mask_override)
    for i in range(4):
        lr = losses_r[i]
        assert np.all(np.abs(out[i].cpu().numpy() - lr) <= 2e-3 * np.maximum(1.0, np.abs(lr))), (i, out[i], lr)
    assert set(G.keys()) == {k for k in nb.collect_params().keys() if not k.endswith(("running_mean", "running_var"))}
    for k, gref in G.items():
        got = nb.collect_params()[k].grad().cpu().numpy()
        scale = max(1e-3, float(np.abs(gref).max()))
        assert maxdiff(got, gref) / scale < 5e-4, k
    for k, v in onet.new_running.items():
        assert maxdiff(nb.collect_params()[k].data().cpu().numpy(), v) < 1e-4, k


def test_features_dir_scripts_end_to_end(tmp_path, monkeypatch):
    """extract_base_features.py writes <id>_F{1,2,3}.npy; train_yolov3.py --features_dir trains the no-backbone
    network on them for one epoch (host plumbing of SURVEY §8(f) N3: FeatureDataset, YOLO3NB* transforms, loader)."""
    import os
    import extract_base_features as E
    import train_yolov3 as T
    monkeypatch.chdir(tmp_path)
    dirs = set()
    for seed in (233, 234):            # the train and the val stand-in datasets of train_yolov3.py (seed, seed + 1)
        dirs.add(E.main(["--dataset", "voc", "--random_init", "--synthetic_samples", "8", "--batch_size", "4",
                         "--data_shape", "64", "--dataset_seed", str(seed)]))
    (fdir,) = dirs
    files = sorted(os.listdir(fdir))
    assert len(files) == 2 * 8 * 3 and files[0].endswith("_F1.npy")
    f1 = np.load(os.path.join(fdir, files[0]))
    assert f1.shape == (256, 8, 8) and f1.dtype == np.float32
    T.main(["--dataset", "voc", "--features_dir", fdir, "--batch_size", "4", "--data_shape", "64", "--epochs", "1",
            "--synthetic_samples", "8", "--save_prefix", "nb", "--val_interval", "1", "--log_interval", "1"])
    import glob
    (log,) = glob.glob(os.path.join("models", "experiments", "nb", "*_train.log"))
    logs = open(log).read()
    assert "Training cost" in logs and "Validation" in logs, logs[-400:]
