#!/usr/bin/env python
"""Experiment: the training step (forward + losses + backward + SGD, batch 64 / 416 x 416, bench.py's workload) replayed as ONE
captured HIP graph against the plain launch loop - how much of the step is launch gaps (tools/step_gaps.py: 1.9 ms).  The
learning rate is a kernel argument, so this graph is valid for one learning rate only: a measurement, not a product path.
usage: python tools/graph_train_probe.py [steps=20]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd.model import yolo3_darknet53
from viddet_amd.targets import synthetic_batch, prefetch_targets


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B, S, C = 64, 416, 80
    net = yolo3_darknet53(["c%d" % i for i in range(C)])
    net.initialize(init="he", seed=233, obj_bias=-4.0)
    x_np, gt_np, ids_np = synthetic_batch(B, S, C, 233)
    tg = prefetch_targets(S, S, gt_np, ids_np, C)
    x, gt = torch.from_numpy(x_np).cuda(), torch.from_numpy(gt_np).cuda()
    tgd = [torch.from_numpy(t).cuda() for t in tg]

    def step():
        net(x, gt, *tgd)
        net.backward()
        net.sgd_step(1e-3, 0.9, 5e-4, batch_size=B)

    def timeit(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    for _ in range(4):
        step()
    t_plain = timeit(step, steps)
    print("plain launch loop: %.3f ms per step (%.1f frames/s)" % (t_plain, B / t_plain * 1e3), flush=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        step()                                   # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            step()
    torch.cuda.synchronize()
    t_graph = timeit(g.replay, steps)
    print("one captured graph: %.3f ms per step (%.1f frames/s)" % (t_graph, B / t_graph * 1e3), flush=True)
    t_plain2 = timeit(step, steps)
    print("plain again: %.3f ms" % t_plain2)


if __name__ == "__main__":
    main()
