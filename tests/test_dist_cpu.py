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
