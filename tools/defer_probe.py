#!/usr/bin/env python
"""What would overlapping the deep layers' weight-gradient GEMMs with the NEXT forward pass buy?  Times the forward
segments alone, a prefix of the step's vd_conv_wgrad launches (backward order: heads, neck, stage 5 ...) alone on a side
stream, and both together.  Developer probe (results of the concurrent run are not used)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from viddet_amd import lib as L
from viddet_amd.model import yolo3_darknet53, Program, Slot
from viddet_amd.targets import synthetic_batch, prefetch_targets


def main():
    B, S, Cn = 64, 416, 80
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.45
    net = yolo3_darknet53(["c%d" % i for i in range(Cn)])
    net.initialize(init="he", seed=233, obj_bias=-4.0)
    x_np, gt_np, ids_np = synthetic_batch(B, S, Cn, 233)
    tg = prefetch_targets(S, S, gt_np, ids_np, Cn)
    x = torch.from_numpy(x_np).cuda()
    gt = torch.from_numpy(gt_np).cuda()
    tgd = [torch.from_numpy(t).cuda() for t in tg]
    for _ in range(3):
        net(x, gt, *tgd)
        net.backward()
        net.sgd_step(1e-3, 0.9, 5e-4, batch_size=B)
    torch.cuda.synchronize()
    tp = net._last_train
    wg = []
    for seg in tp['bwd']:
        if isinstance(seg, Program):
            for (fname, fn, args), st in zip(seg.recs, seg.streams):
                if fname == 'vd_conv_wgrad':
                    wg.append((fn, args))
    total_fl = sum(2.0 for _ in wg)
    ndef = int(len(wg) * frac)
    side = torch.cuda.Stream()

    def run_wgrads(n):
        sp = C.c_void_p(side.cuda_stream)
        for fn, args in wg[:n]:
            a = [v.value if isinstance(v, Slot) else v for v in args]
            fn(*a, sp)

    def fwd():
        net._run_segments(tp['fwd'])

    def timed(f, reps=5):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_f = timed(fwd)

    def only_w():
        side.wait_stream(torch.cuda.current_stream())
        run_wgrads(ndef)
        torch.cuda.current_stream().wait_stream(side)
    t_w = timed(only_w)

    def both():
        side.wait_stream(torch.cuda.current_stream())
        run_wgrads(ndef)
        fwd()
        torch.cuda.current_stream().wait_stream(side)
    t_b = timed(both)
    print("wgrad launches %d of %d | forward alone %.2f ms | deferred wgrads alone %.2f ms | together %.2f ms | saved %.2f ms"
          % (ndef, len(wg), t_f, t_w, t_b, t_f + t_w - t_b))


if __name__ == "__main__":
    main()
