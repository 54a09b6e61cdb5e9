"""CPU twin of the k = 1 network on torch-CPU operators (oneDNN convolutions, autograd backward).  TEST INFRASTRUCTURE
ONLY, like the rest of oracle/ (parity-status statement: oracle/ops.py): an INDEPENDENT construction of the architecture
that tests/test_oracle_cpu.py checks the NumPy oracle against (fp64), and - in fp32 on all host threads - the
`cpu_baseline` leg of bench.py.  It stands where north_star names "the reference MXNet CPU path": MXNet / GluonCV cannot
be installed here or shipped to the GPU box, MXNet's CPU backend is MKL-DNN (= oneDNN), and this is the same
network on PyTorch's oneDNN build - kind "port", NOT MXNet.

Follows, under /root/reference: layers.py:63-70 (_conv2d cell), three_darknet.py:100-123,152-194 (Darknet-53),
yolo3.py:218-263 (detection block), :1047-1054,1167-1177 (transition, upsample, concat), :43-74 (prediction conv).
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import net as ON
from . import yolo as Y


def torch_net(tp, x, train):
    """Heads (B, A, g, g) x 3 of the network with parameters `tp` (dict name -> tensor) on input x (B,3,H,W)."""

    def cell(name, x, k, s, res=None):
        z = F.conv2d(x, tp[name + ".0.weight"], stride=s, padding=k // 2)
        u = F.batch_norm(z, tp[name + ".1.running_mean"].clone(), tp[name + ".1.running_var"].clone(),
                         tp[name + ".1.gamma"], tp[name + ".1.beta"], training=train, momentum=0.1, eps=1e-5)
        y = F.leaky_relu(u, 0.1)
        return y if res is None else y + res

    nm = ON.stage_names()
    h = cell(nm(0), x, 3, 1)
    f, routes = 1, []
    for nl, ch in zip([1, 2, 8, 8, 4], [64, 128, 256, 512, 1024]):
        h = cell(nm(f), h, 3, 2)
        f += 1
        for _ in range(nl):
            h = cell(nm(f) + ".body.1", cell(nm(f) + ".body.0", h, 1, 1), 3, 1, res=h)
            f += 1
        if f in (15, 24, 29):
            routes.append(h)
    heads, h = [], routes[2]
    for i in range(3):
        for j in range(5):
            h = cell("yolo_blocks.%d.body.%d" % (i, j), h, 1 if j % 2 == 0 else 3, 1)
        tip = cell("yolo_blocks.%d.tip" % i, h, 3, 1)
        heads.append(F.conv2d(tip, tp["yolo_outputs.%d.prediction.weight" % i], tp["yolo_outputs.%d.prediction.bias" % i]))
        if i < 2:
            t = cell("transitions.%d" % i, h, 1, 1)
            h = torch.cat([F.interpolate(t, scale_factor=2, mode="nearest"), routes[1 - i]], dim=1)
    return heads


def params(P, dtype=torch.float64):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype).clone().requires_grad_(
        not k.endswith(("running_mean", "running_var"))) for k, v in P.items()}


def loss_sum(heads, merged, c):
    """Sum of the four YOLOV3Loss terms over the batch, on the oracle's merged targets."""
    b = heads[0].shape[0]
    dt = heads[0].dtype
    objness_t, center_t, scale_t, weight_t, class_t, class_mask = [torch.from_numpy(np.ascontiguousarray(m)).to(dt) for m in merged]
    p = torch.cat([h.reshape(b, 3, 5 + c, -1).permute(0, 3, 1, 2).reshape(b, -1, 5 + c) for h in heads], dim=1)
    bce = lambda lg, z: F.binary_cross_entropy_with_logits(lg, z, reduction="none")
    w = weight_t * objness_t
    hard = torch.where(objness_t > 0, torch.ones_like(objness_t), objness_t)
    om = torch.where(objness_t > 0, objness_t, (objness_t >= 0).to(dt))
    return (bce(p[..., 4:5], hard) * om).sum() + (bce(p[..., 0:2], center_t) * w).sum() + \
           ((p[..., 2:4] - scale_t).abs() * w).sum() + (bce(p[..., 5:], class_t) * class_mask * objness_t).sum()


def baseline(mode, size, classes, batch, seed=233, budget_s=20.0):
    """Timed fp32 steps on all host threads: train = forward + 4 losses + backward + SGD-momentum update; detect = forward +
    decode + NMS.  Runs whole steps until `budget_s` seconds are spent (at least one after a warm-up step).
    Returns (frames per second, steps timed, threads)."""
    from viddet_amd.targets import synthetic_batch          # the bench's own synthetic inputs: data, not arithmetic
    P = ON.init_params(classes, seed=seed, obj_bias=-4.0)
    x, gt, ids = synthetic_batch(batch, size, classes, seed)
    xt = torch.from_numpy(x)
    tp = params(P, torch.float32)
    grids = [size // 32, size // 16, size // 8]
    if mode == "train":
        tg = Y.prefetch_targets(size, size, grids, gt.astype(np.float64), ids, classes)
        mom = {k: torch.zeros_like(v) for k, v in tp.items() if v.requires_grad}

        def step():
            heads = torch_net(tp, xt, True)
            with torch.no_grad():
                outs = [Y.yolo_output(h.detach().double().numpy(), classes, Y.OUT_ANCHORS[s], Y.OUT_STRIDES[s], True)
                        for s, h in enumerate(heads)]
                merged = Y.merge_targets(np.concatenate([o[0] for o in outs], axis=1), gt.astype(np.float64), *tg, classes,
                                         0.7, False)
            loss_sum(heads, merged, classes).backward()
            with torch.no_grad():
                for k, m in mom.items():                     # gluon 'sgd': mom = 0.9 mom - lr (g / B + wd w); w += mom
                    m.mul_(0.9).sub_(1e-3 * (tp[k].grad / batch + 5e-4 * tp[k]))
                    tp[k].add_(m)
                    tp[k].grad = None
    else:
        def step():
            with torch.no_grad():
                heads = torch_net(tp, xt, False)
                dets = [Y.yolo_output(h.double().numpy(), classes, Y.OUT_ANCHORS[s], Y.OUT_STRIDES[s], False)
                        for s, h in enumerate(heads)]
                Y.detect_postprocess(dets, 0.45, 400, 100)
    step()                                                   # warm-up (oneDNN primitive creation)
    n, t0 = 0, time.time()
    while n == 0 or time.time() - t0 < budget_s:
        step()
        n += 1
    return batch * n / (time.time() - t0), n, torch.get_num_threads()
