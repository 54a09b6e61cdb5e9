#!/usr/bin/env python
"""Where the persistent stream-K form spends time: arbitrary (cin, cout, k, spatial, batch, tile) with the number of polls
that gave up.  usage: python tools/sk_probe.py cin cout k hin batch tile [tile ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from viddet_amd import ops
from sk_bench import timeit


def main():
    cin, cout, k, hin, B = [int(v) for v in sys.argv[1:6]]
    tiles = [int(v) for v in sys.argv[6:]]
    pad = k // 2
    x = torch.randn(B, hin, hin, cin, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wp = torch.empty(cout, k * k * cin, device="cuda")
    ops.pack_weight_fwd(w, wp, cout)
    y0, y1 = torch.empty(B, hin, hin, cout, device="cuda"), torch.empty(B, hin, hin, cout, device="cuda")
    ax, aw = ops.amax(x), ops.amax(wp)
    skws = ops.streamk_workspace()
    flops = 2.0 * cin * cout * k * k * hin * hin * B
    for tile in tiles:
        kw = dict(k=k, stride=1, pad=pad, Co=cout, tile=tile, split="f16x2", amax_in=ax, amax_w=aw)
        t0 = timeit(lambda: ops.conv_fwd(x, wp, y0, **kw), 10)
        c0 = int(skws.view(torch.int32)[2047])
        t1 = timeit(lambda: ops.conv_fwd(x, wp, y1, streamk_ws=skws, **kw), 10)
        c1 = int(skws.view(torch.int32)[2047])
        print("%d->%d k%d @%d B%d tile %d: classic %.4f ms (%.1f TF)  stream-K %.4f ms (%.1f TF)  polls given up in 12 launches: %d  equal %s"
              % (cin, cout, k, hin, B, tile, t0, flops / t0 / 1e9, t1, flops / t1 / 1e9, c1 - c0, bool(torch.equal(y0, y1))), flush=True)


if __name__ == "__main__":
    main()
