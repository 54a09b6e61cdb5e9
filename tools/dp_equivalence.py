#!/usr/bin/env python
"""Data-parallel equivalence rehearsal on ONE GPU: world_size ranks (gloo, all on cuda:0) each run one training step
on their shard of a global batch with SyncBN('all') + the bucketed gradient all-reduce + the SGD step with the GLOBAL
batch rescale; rank 0 then compares its weights / running statistics with a single-process step on the whole batch.
(RCCL refuses two ranks on one device, so the collective library here is gloo; the schedule - where collectives sit
in the launch programs, what they reduce, the rescale - is the code the N-GPU run executes.)
usage: python tools/dp_equivalence.py [world_size=2]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

# Fixed kernel variants on both sides: with the timing-based autotuner the two runs may pick different tiles /
# product arithmetics per layer, whose 1e-7 differences flip a few LeakyReLU decisions at |pre-activation| ~ 0 and
# move small gradients by ~1e-2 - noise that would hide what this check is about.  With fixed variants the
# duplicate-shard form (VD_DP_DUP=1) is bit-identical and the sharded form differs by summation order only (4e-4).
os.environ.setdefault("VD_AUTOTUNE", "0")
C, SIZE, PER_RANK = 4, 64, 2


def cases():
    """VD_DP_CASES = a JSON list of cases run one after the other by ONE set of processes (the GPU suite: a process start
    costs more than a case), else one case from the VD_DP_* variables:
      K        frames per window: 3 = the BASELINE configs[3] family (YOLOV3T, late max join)
      SCOPE    SyncBN scope: 'all', or 'reference' = the six layers --syncbn reaches
      STORAGE  'bf16' = bf16-storage training (net.set_storage('bf16'), BASELINE configs[4]'s mode)
      STEPS    optimiser steps on the same batch (> 1: the repacks after a step, under DP)
      DUP      every rank gets shard 0 (compared with a single-process step on shard 0: bit-identical)
      NOSYNCBN plain BatchNorm on the ranks (diagnostic)"""
    js = os.environ.get("VD_DP_CASES")
    raw = json.loads(js) if js else [dict(K=os.environ.get("VD_DP_K", "1"), SCOPE=os.environ.get("VD_DP_SCOPE", "all"),
                                          STORAGE=os.environ.get("VD_DP_STORAGE", "fp32"), STEPS=os.environ.get("VD_DP_STEPS", "1"),
                                          DUP=os.environ.get("VD_DP_DUP", ""), NOSYNCBN=os.environ.get("VD_DP_NOSYNCBN", ""))]
    out = []
    for c in raw:
        out.append(dict(K=int(c.get("K", 1)), SCOPE=c.get("SCOPE", "all"), STORAGE=c.get("STORAGE", "fp32"),
                        STEPS=int(c.get("STEPS", 1)), DUP=bool(c.get("DUP")), NOSYNCBN=bool(c.get("NOSYNCBN"))))
        # with the 'reference' scope most layers normalise with their OWN shard's statistics, so one process on the joint
        # batch is not the same computation: only the duplicate-shard form has a single-process equal
        assert out[-1]["SCOPE"] == "all" or out[-1]["DUP"], "SCOPE=reference needs DUP=1"
    return out


def make(cfg, world):
    from viddet_amd.targets import synthetic_batch, prefetch_targets
    K = cfg["K"]
    x, gt, ids = synthetic_batch(PER_RANK * world * K, SIZE, C, 5)
    if K > 1:                                           # windows of K frames; the labels are the centre frame's
        x = x.reshape(PER_RANK * world, K, 3, SIZE, SIZE)
        gt, ids = gt.reshape(PER_RANK * world, K, -1, 4)[:, K // 2], ids.reshape(PER_RANK * world, K, -1, 1)[:, K // 2]
    tg = prefetch_targets(SIZE, SIZE, gt, ids, C)
    return x, gt, tg


def one_step(cfg, net, x, gt, tg, global_batch):
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    for _ in range(cfg["STEPS"]):
        out = net(dv(x), dv(gt), *[dv(t) for t in tg])
        net.backward()
        net.allreduce_grads()
        net.sgd_step(0.001 if cfg["STEPS"] > 1 else 0.01, 0.9, 5e-4, batch_size=global_batch)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out]


def build(cfg, syncbn):
    from viddet_amd.model import yolo3_darknet53
    K = cfg["K"]
    net = yolo3_darknet53(["c%d" % i for i in range(C)], norm_layer="syncbn" if syncbn else None,
                          norm_kwargs={"scope": cfg["SCOPE"]} if syncbn else None,
                          **(dict(k=K, k_join_type="max", k_join_pos="late") if K > 1 else {}))
    net.initialize(init="he", seed=3, obj_bias=-1.0)
    if cfg["STORAGE"] == "bf16":
        net.set_storage("bf16")
    return net


def release(net):
    import gc
    del net
    gc.collect()
    torch.cuda.empty_cache()


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for ci, cfg in enumerate(cases()):
        x, gt, tg = make(cfg, world)
        lo, hi = rank * PER_RANK, (rank + 1) * PER_RANK
        if cfg["DUP"]:
            lo, hi = 0, PER_RANK
        net = build(cfg, syncbn=not cfg["NOSYNCBN"])
        losses = one_step(cfg, net, x[lo:hi], gt[lo:hi], [t[lo:hi] for t in tg], PER_RANK * world)
        state = {k: p.data().cpu().numpy() for k, p in net.collect_params().items()}
        gathered = [None] * world
        dist.all_gather_object(gathered, (losses, state if rank == 0 else None))
        if rank == 0:
            ret["dp%d" % ci] = gathered
        release(net)
    dist.destroy_process_group()


def check(cfg, dp, world):
    DUP, STORAGE = cfg["DUP"], cfg["STORAGE"]
    # single process, whole batch (plain BatchNorm over the global batch == SyncBN over the shards)
    x, gt, tg = make(cfg, world)
    net = build(cfg, syncbn=False)
    w0 = {k: p.data().cpu().numpy().copy() for k, p in net.collect_params().items()}
    if DUP:
        x, gt, tg = x[:PER_RANK], gt[:PER_RANK], [t[:PER_RANK] for t in tg]
        losses = one_step(cfg, net, x, gt, tg, PER_RANK)
        dp = [dp[0]]
        world = 1
    else:
        losses = one_step(cfg, net, x, gt, tg, PER_RANK * world)
    ref = {k: p.data().cpu().numpy() for k, p in net.collect_params().items()}
    release(net)
    dp_losses = [np.concatenate([dp[r][0][i] for r in range(world)]) for i in range(4)]
    worst, table, num, den = 0.0, [], 0.0, 0.0
    # bf16 storage, real shards: the shards' statistics arrive in another summation order, which moves bf16 roundings of
    # activations (2^-9 relative each) instead of fp32 ones - the same comparison at the bf16 tolerance
    ltol = 2e-4 if (STORAGE == "fp32" or DUP) else 2e-2
    for i in range(4):
        assert np.allclose(dp_losses[i], losses[i], rtol=ltol, atol=ltol), (cfg, i, dp_losses[i], losses[i])
    for k, v in ref.items():
        # compare the UPDATE each side applied (weights and running statistics start identical): both sides compute it
        # in fp32 through 75 layers with different summation orders, so the bound is the gradient-parity tolerance of
        # tests/test_model_gpu.py (5e-4 of the tensor's largest update)
        upd_ref, upd_dp = v - w0[k], dp[0][1][k] - w0[k]
        d = float(np.abs(upd_dp - upd_ref).max())
        s = max(1e-7, float(np.abs(upd_ref).max()))
        worst = max(worst, d / s)
        table.append((d / s, k))
        num += float(((upd_dp - upd_ref).astype(np.float64) ** 2).sum())
        den += float((upd_ref.astype(np.float64) ** 2).sum())
    if os.environ.get("VD_DP_VERBOSE"):
        for r, k in sorted(table)[::max(1, len(table) // 40)]:
            print("%-50s %.3e" % (k, r))
    l2 = (num / max(den, 1e-30)) ** 0.5
    if DUP:
        # identical shards on every rank: the collectives only scale sums by the world size, so with fixed kernel
        # variants the step is reproduced to the last bit
        bad = [(r, k) for r, k in table if r > 1e-6]
        assert not bad, (cfg, sorted(bad)[-5:])
    else:
        # real shards: per-shard partial sums change the summation order (1e-7), which can flip a LeakyReLU decision at
        # |pre-activation| ~ 0 and move one small gradient tensor by a few %; the update as a whole must agree
        if STORAGE == "bf16":
            # measured 4.1e-2 on this 64x64 / 4-class fixture (its deepest BatchNorms see 8 samples per rank): the size of
            # bf16 storage's own distance from the fp32 oracle (whole-gradient cosine 0.906 there), not of the
            # exchange - the duplicate-shard form, which has the same collectives, is bit-identical
            assert l2 < 1e-1, (cfg, l2)
        else:
            assert l2 < 2e-3, (cfg, l2)
            bad = [(r, k) for r, k in table if r >= 0.2]
            assert not bad, (cfg, sorted(bad)[-5:])
    print("update L2 difference %.2e" % l2)
    print("dp_equivalence ok: storage=%s k=%d scope=%s steps=%d dup=%d world=%d, %d tensors, worst relative difference %.2e" % (
        STORAGE, cfg["K"], cfg["SCOPE"], cfg["STEPS"], int(DUP), world, len(ref), worst), flush=True)


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cs = cases()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, int(os.environ.get("VD_DP_PORT", "29533")), ret), nprocs=world, join=True)
    for ci, cfg in enumerate(cs):
        check(cfg, ret["dp%d" % ci], world)


if __name__ == "__main__":
    main()
