"""CPU, world_size 2, gloo: the N>1 exchange steps of the path (viddet_amd/dist.py) — frame sharding, the
gradient-arena all-reduce (+ bucketed form), and the SyncBN exchange units — checked against the oracle's
single-process full-batch results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ops as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from viddet_amd import dist as vd
    vd.init_from_env(backend="gloo")
    try:
        rng = np.random.default_rng(0)                  # same data on both ranks; each takes its shard
        n, c, h, w = 6, 8, 5, 7
        x = rng.standard_normal((n, c, h, w)) * 1.5 + 0.3
        gamma, beta = rng.uniform(0.5, 1.5, c), rng.standard_normal(c)
        dy = rng.standard_normal((n, c, h, w))
        lo, hi = vd.shard_range(n, rank, world)
        xs, dys = x[lo:hi], dy[lo:hi]
        # --- SyncBN forward exchange: fp64 [sum x, sum x^2]
        sums = torch.from_numpy(np.concatenate([xs.sum(axis=(0, 2, 3)), (xs ** 2).sum(axis=(0, 2, 3))]))
        vd.allreduce_sum_(sums)
        cnt = n * h * w
        mean = sums[:c].numpy() / cnt
        var = sums[c:].numpy() / cnt - mean ** 2
        u_ref, mean_ref, var_ref = R.bn_train(x, gamma, beta)
        assert np.allclose(mean, mean_ref, atol=1e-12) and np.allclose(var, var_ref, atol=1e-10)
        # --- SyncBN backward exchange: fp64 [sum g, sum g*xhat]; dgamma/dbeta stay LOCAL (summed later with the grads)
        invstd = 1 / np.sqrt(var + 1e-5)
        shp = (1, -1, 1, 1)
        xh = (xs - mean.reshape(shp)) * invstd.reshape(shp)
        us = xh * gamma.reshape(shp) + beta.reshape(shp)
        g = R.leaky_backward(us, dys)
        s2_local = np.concatenate([g.sum(axis=(0, 2, 3)), (g * xh).sum(axis=(0, 2, 3))])
        s2 = torch.from_numpy(s2_local.copy())
        vd.allreduce_sum_(s2)
        s2 = s2.numpy()
        dx = (gamma * invstd).reshape(shp) * (g - s2[:c].reshape(shp) / cnt - xh * s2[c:].reshape(shp) / cnt)
        dx_ref, dgamma_ref, dbeta_ref = R.bn_train_backward(x, gamma, mean_ref, var_ref, R.leaky_backward(u_ref, dy))
        assert np.allclose(dx, dx_ref[lo:hi], atol=1e-10)
        # --- gradient arena all-reduce (fp32), plain and bucketed, then the identical SGD step on every rank
        grads = torch.from_numpy(np.concatenate([s2_local[c:], s2_local[:c]]).astype(np.float32))   # local dgamma, dbeta
        g2 = grads.clone()
        vd.allreduce_sum_(grads)
        vd.bucketed_allreduce_sum_(g2, bucket_elems=5)
        assert torch.equal(grads, g2)
        assert np.allclose(grads.numpy(), np.concatenate([dgamma_ref, dbeta_ref]), atol=1e-4)
        wts = np.concatenate([gamma, beta])
        w_new, _ = R.sgd_momentum(wts, grads.numpy().astype(np.float64), np.zeros(2 * c), 0.01, 0.9, 5e-4, 1.0 / n)
        gathered = [torch.zeros(2 * c, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(w_new))
        assert torch.equal(gathered[0], gathered[1]), "replicas diverged"
        q.put((rank, "ok", (lo, hi)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "fail: %r" % (e,), None))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_exchanges():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    ranges = sorted(r[2] for r in res)
    assert ranges == [(0, 3), (3, 6)]


def test_shard_range_covers_without_overlap():
    from viddet_amd.dist import shard_range
    for n in (1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


def _val_records(idx, c=3):
    """Deterministic fake detections / ground truth of validation sample `idx`."""
    rng = np.random.default_rng(1000 + idx)
    n, m = 5, 3
    lo = rng.uniform(0, 60, (n, 2)); pb = np.concatenate([lo, lo + rng.uniform(5, 40, (n, 2))], axis=1)
    gl = rng.integers(0, c, (m, 1)).astype(np.float64)
    gb = pb[:m] + rng.uniform(-3, 3, (m, 4))            # the first m detections overlap a ground-truth box
    pl = np.concatenate([gl[:, 0], rng.integers(0, c, n - m)]).reshape(n, 1).astype(np.float64)
    ps = rng.uniform(0.05, 0.95, (n, 1))
    return (idx, pb, pl, ps, gb, gl, None)


def _metric_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from viddet_amd import dist as vd
    from viddet_amd.metrics import VOCMApMetric, update_metric_sharded
    vd.init_from_env(backend="gloo")
    try:
        nsamp = 9                                         # not divisible by the world size
        names = ["a", "b", "c"]
        m = VOCMApMetric(iou_thresh=0.5, class_names=names)
        mine = [_val_records(i) for i in range(rank, nsamp, world)]     # Loader's frame sharding: idx[rank::world]
        n = update_metric_sharded(m, mine)
        full = VOCMApMetric(iou_thresh=0.5, class_names=names)
        for i in range(nsamp):
            _, pb, pl, ps, gb, gl, gd = _val_records(i)
            full.update([pb], [pl], [ps], [gb], [gl], None)
        assert n == nsamp
        a, b = m.get()[1], full.get()[1]
        assert np.allclose(a, b, rtol=0, atol=0, equal_nan=True), (a, b)      # the single-process metric, bit for bit
        # a decision of rank 0 reaches every rank (train_yolov3.py start-up: refuse an existing save_dir together)
        assert vd.broadcast_object(rank == 0 and "refuse", src=0) == "refuse"
        vd.barrier()
        q.put((rank, "ok", float(a[-1])))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "fail: %r" % (e,), None))
    finally:
        dist.destroy_process_group()


def test_validation_metric_covers_the_whole_set_on_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metric_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    assert res[0][2] == res[1][2] and res[0][2] > 0


def test_loader_shards_cover_the_validation_set():
    from viddet_amd.data import SyntheticDetection, YOLO3VideoInferenceTransform, Loader
    ds = SyntheticDetection("voc", num_samples=7, size=(64, 48))
    seen = []
    for r in range(2):
        ld = Loader(ds, YOLO3VideoInferenceTransform(32, 32), 2, train=False, last_batch="keep", rank=r, world=2)
        for batch in ld:
            seen += [int(i) for i in batch[-1]]
    assert sorted(seen) == list(range(7))
