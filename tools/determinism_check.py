#!/usr/bin/env python
"""Run the same training step several times and report which gradient tensors differ bitwise between runs
(no kernel on the training path uses atomics, so every run must be identical)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from viddet_amd.model import yolo3_darknet53, ConvNode
from viddet_amd.targets import synthetic_batch, prefetch_targets

k = int(os.environ.get("K", "1"))
B, S, C = int(os.environ.get("B", "2")), int(os.environ.get("S", "64")), 3
kw = dict(k=3, k_join_type="mean", k_join_pos="late") if k == 3 else {}
net = yolo3_darknet53(["a", "b", "c"], **kw)
net.initialize(init="he", seed=1)
x, gt, ids = synthetic_batch(B * (3 if k == 3 else 1), S, C, 5, max_gt=3)
if k == 3:
    x = x.reshape(B, 3, 3, S, S)
    gt, ids = gt[:B], ids[:B]
tg = prefetch_targets(S, S, gt, ids, C)
xd, gtd = torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda()
tgd = [torch.from_numpy(t).cuda() for t in tg]
ref = None
for it in range(6):
    net(xd, gtd, *tgd)
    net.backward()
    torch.cuda.synchronize()
    g = net.grads.clone()
    tb = net._last_train['bufs']
    acts = {n.name: tb['d:' + n.dst].clone() for n in net.nodes if isinstance(n, ConvNode)}
    if ref is None:
        ref, ract = g, acts
        continue
    nd = int((g != ref).sum())
    bad = [k_ for k_ in acts if not torch.equal(acts[k_], ract[k_])]
    print("run %d: grad elements differing from run 0: %d ; first differing dY tensors (backward order): %s"
          % (it, nd, bad[::-1][:4]))
