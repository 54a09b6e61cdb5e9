"""The persistent stream-K form of k_conv_igemm (VD_CONV_STREAMK, viddet_amd/csrc/vd_conv_sk.hip) against the
one-tile-per-workgroup form of the same launch.

The form cuts a launch into equal runs of (tile, K-unit) units per workgroup; a tile cut by a run boundary is finished by
the next workgroup FROM the first one's accumulators, i.e. the same MFMA chain in the same order.  So the claim under
test is the strongest one there is: EVERY output - the tensor, the fused BatchNorm statistics table, the fused
BatchNorm-backward reduction table, the published max-abs - is bit-identical to the classic launch, on every tile variant,
for forward convs (1x1, 3x3 stride 1 / halo loop, stride 2, temporal taps), data gradients (flipped taps, strided parity
outputs, accumulating epilogue) and when every consumer gives up its hand-off poll and recomputes the prefix
(VD_SK_TIMEOUT_TICKS=0, a subprocess).  The classic launch itself is checked against the fp64 oracle in test_conv_gpu.py.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ops as R
from tests.util import dev, nchw_to_dev_nhwc, dev_nhwc_to_nchw, maxdiff

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _streamk_used(desc):
    from viddet_amd import lib as L
    return bool(L.load().vd_conv_igemm_streamk(C.byref(desc)))


def _pair(run):
    """run(streamk_ws or None, out tensors...) twice; returns ([classic outputs], [stream-K outputs], used)"""
    from viddet_amd import ops
    ws = ops.streamk_workspace()
    a = run(None)
    b = run(ws)
    torch.cuda.synchronize()
    return a, b, ws


# (n, ci, h, w, co, k, stride): batch sizes chosen so that the tile counts exceed the 256 / 512 workgroup slots
FWD = [
    (110, 64, 26, 26, 128, 3, 1),     # halo loop, 2 chunks per tile
    (64, 128, 26, 26, 256, 3, 1),     # halo loop, two column tiles: BASELINE's batch on the 26 x 26 layers
    (64, 256, 13, 13, 512, 3, 1),     # 13 x 13: the 1.3-round case the form exists for (128-row tiles)
    (28, 64, 52, 52, 128, 3, 1),
    (112, 128, 27, 25, 128, 3, 1),    # odd map, rows not a tile multiple
    (112, 256, 26, 26, 128, 1, 1),    # 1x1: generic loop, 8 K-steps per tile
    (110, 64, 52, 52, 128, 3, 2),     # stride 2: generic loop, taps innermost
    (110, 64, 26, 26, 96, 3, 1),      # Co not a multiple of the tile width (head-like padding)
]


@pytest.mark.parametrize("tile", [1, 5, 2, 6, 3, 7, 4, 8, 11, 12, 15, 16])
@pytest.mark.parametrize("case", FWD)
def test_forward_bit_identical(case, tile):
    from viddet_amd import ops
    n, ci, h, w, co, k, s = case
    g = torch.Generator(device="cuda").manual_seed(1234 + tile)
    x = torch.randn(n, h, w, ci, device="cuda", generator=g)
    wt = torch.randn(co, ci, k, k, device="cuda", generator=g) / np.sqrt(ci * k * k)
    wp = torch.empty(co, k * k * ci, device="cuda")
    ops.pack_weight_fwd(wt, wp, co)
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    res = torch.randn(n, ho, wo, co, device="cuda", generator=g)
    scale, shift = torch.rand(co, device="cuda", generator=g) + 0.5, torch.randn(co, device="cuda", generator=g)
    ax, aw = ops.amax(x), ops.amax(wp)
    used = []

    def run(ws):
        out = torch.full((n, ho, wo, co), 3.0, device="cuda")
        am = torch.zeros(ops.AMAX_FLOATS, device="cuda")
        ds = []
        ops.conv_fwd(x, wp, out, k=k, stride=s, pad=pad, Co=co, tile=tile, split="f16x2", amax_in=ax, amax_w=aw, scale=scale,
                     shift=shift, residual=res, leaky=True, amax_out=am, streamk_ws=ws, desc_out=ds)
        if ws is not None:
            used.append(_streamk_used(ds[0]))
        return out, am

    (o0, a0), (o1, a1), ws = _pair(run)
    if not used[0]:
        pytest.skip("the form does not apply to this tile count")
    assert torch.equal(o0, o1), float((o0 - o1).abs().max())
    assert ops.amax_value(a0) == ops.amax_value(a1)
    assert int(ws.view(torch.int32)[2047]) == 0, "a hand-off poll gave up on an idle GPU"
    # a second launch on the same workspace (monotonic counters) and a different geometry after it
    (o0b, _), (o1b, _), _ = (run(None), run(ws), None)
    torch.cuda.synchronize()
    assert torch.equal(o0b, o1b)


def test_forward_matches_oracle_and_training_statistics_table():
    """one case against the fp64 oracle (the classic launch is what test_conv_gpu.py checks; this pins the new form to the
    same yardstick directly), with the fused BatchNorm statistics rows of a training forward"""
    from viddet_amd import ops, lib as L
    n, ci, h, w, co, k = 110, 64, 26, 26, 128, 3
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, ci, h, w))
    wt = rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)
    ref = R.conv2d(x, wt, 1, 1)
    xd, wp = nchw_to_dev_nhwc(x), torch.empty(co, k * k * ci, device="cuda")
    ops.pack_weight_fwd(dev(wt), wp, co)
    ax, aw = ops.amax(xd), ops.amax(wp)
    sk = ops.streamk_workspace()
    outs = []
    for ws in (None, sk):
        d = L.ConvDesc()
        out = torch.empty(n, h, w, co, device="cuda")
        d.in_, d.wp, d.out = xd.data_ptr(), wp.data_ptr(), out.data_ptr()
        d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = n, h, w, ci, h, w, 1
        ops._set_taps(d, ops.fwd_taps(k, 1))
        d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.ldo, d.ldr = 1, h, w, co, 1, co, co
        d.flags, d.tile = L.MATH_F16X2 | (L.CONV_STREAMK if ws is not None else 0), 5
        d.amax_in, d.amax_w = ax.data_ptr(), aw.data_ptr()
        if ws is not None:
            d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
        mt = L.load().vd_conv_igemm_mtiles(C.byref(d))
        part = torch.zeros(mt, 2 * co, device="cuda")
        d.stats_part = part.data_ptr()
        L.check(L.load().vd_conv_igemm(C.byref(d), L.stream_ptr()), "vd_conv_igemm")
        if ws is not None:
            assert _streamk_used(d)
        outs.append((out, part))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert maxdiff(dev_nhwc_to_nchw(outs[1][0]), ref) < 2e-4
    s1 = outs[1][1][:, :co].double().sum(0).cpu().numpy()
    assert np.allclose(s1, ref.sum(axis=(0, 2, 3)), rtol=0, atol=2e-3 * np.sqrt(n * h * w))


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("tile", [5, 1, 12])
def test_data_gradient_with_fused_bn_backward_reductions(stride, tile):
    """the data gradient launches of a 3x3 conv (one launch at stride 1, four parity launches with strided outputs at
    stride 2), accumulating into an existing gradient (residual epilogue) and carrying the fused BatchNorm-backward
    reductions: output rows and the partial table bit-identical"""
    from viddet_amd import ops, lib as L
    n, cin, cout, hin = 110, 128, 128, 26 if stride == 1 else 52
    k, pad = 3, 1
    ho = (hin + 2 * pad - k) // stride + 1
    g = torch.Generator(device="cuda").manual_seed(77 + tile)
    wt = torch.randn(cout, cin, k, k, device="cuda", generator=g) / np.sqrt(cin * 9)
    wp = torch.empty(cout, 9 * cin, device="cuda")
    ops.pack_weight_fwd(wt, wp, cout)
    dz = torch.randn(n, ho, ho, cout, device="cuda", generator=g)
    skip = torch.randn(n, hin, hin, cin, device="cuda", generator=g)
    z = torch.randn(n, hin, hin, cin, device="cuda", generator=g)
    vec = lambda: torch.randn(cin, device="cuda", generator=g)
    bsc, bsh, bmu, bis = vec(), vec(), vec(), vec().abs() + 0.5
    adz = ops.amax(dz)
    plans = ops.dgrad_plans(k, pad, stride, hin, hin)
    sk = ops.streamk_workspace()
    res = []
    for ws in (None, sk):
        dx = torch.zeros(n, hin, hin, cin, device="cuda")
        tables, used = [], []
        for pl in plans:
            wpk = torch.empty(cin, len(pl["taps"]) * cout, device="cuda")
            ops.pack_weight_dgrad(wp, wpk, Co=cout, Co_pad=cout, Ci=cin, kd=1, kh=k, kw=k, tap_ids=pl["tap_ids"], src_packed=True)
            d = L.ConvDesc()
            d.in_, d.wp, d.out, d.residual = dz.data_ptr(), wpk.data_ptr(), dx.data_ptr(), skip.data_ptr()
            d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = n, ho, ho, cout, pl["Hg"], pl["Wg"], 1
            ops._set_taps(d, pl["taps"])
            d.Kfr, d.Ho, d.Wo, d.Co = 1, hin, hin, cin
            d.out_stride, d.out_oy, d.out_ox, d.ldo, d.ldr = stride, pl["py"], pl["px"], cin, cin
            d.flags, d.tile = L.MATH_F16X2 | L.EPI_RESIDUAL | (L.CONV_STREAMK if ws is not None else 0), tile
            aw = ops.amax(wpk)
            d.amax_in, d.amax_w = adz.data_ptr(), aw.data_ptr()
            if ws is not None:
                d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
            mt = L.load().vd_conv_igemm_mtiles(C.byref(d))
            part = torch.zeros(mt, 2 * cin, device="cuda")
            d.bs_z, d.bs_scale, d.bs_shift, d.bs_mean, d.bs_invstd = z.data_ptr(), bsc.data_ptr(), bsh.data_ptr(), bmu.data_ptr(), bis.data_ptr()
            d.bs_part, d.bs_slope = part.data_ptr(), 0.1
            L.check(L.load().vd_conv_igemm(C.byref(d), L.stream_ptr()), "vd_conv_igemm")
            used.append(ws is not None and _streamk_used(d))
            tables.append(part)
            torch.cuda.synchronize()
        res.append((dx, tables, used))
    assert any(res[1][2]), "no launch of this case took the stream-K form"
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


def test_temporal_taps_bit_identical():
    """27-tap (3x3x3) conv over K = 3 frame windows (layers.py:73-78 through the same kernel: dz taps, Kfr)"""
    from viddet_amd import ops
    B, K, ci, co, hw = 37, 3, 64, 128, 26
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(B * K, hw, hw, ci, device="cuda", generator=g)
    wt = torch.randn(co, ci, 3, 3, 3, device="cuda", generator=g) / np.sqrt(27 * ci)
    wp = torch.empty(co, 27 * ci, device="cuda")
    ops.pack_weight_fwd(wt, wp, co)
    ax, aw = ops.amax(x), ops.amax(wp)
    used = []

    def run(ws):
        out = torch.empty(B * K, hw, hw, co, device="cuda")
        ds = []
        ops.conv_fwd(x, wp, out, k=3, stride=1, pad=1, Co=co, kd=3, pad_d=1, kfr=K, tile=5, split="f16x2", amax_in=ax, amax_w=aw,
                     streamk_ws=ws, desc_out=ds)
        if ws is not None:
            used.append(_streamk_used(ds[0]))
        return (out,)

    (o0,), (o1,), _ = _pair(run)
    assert used[0] and torch.equal(o0, o1)


_GIVE_UP = r"""
import sys, torch
sys.path.insert(0, %r)
from viddet_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
n, ci, co, hw = 110, 64, 128, 26
x = torch.randn(n, hw, hw, ci, device="cuda", generator=g)
w = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * 0.05
wp = torch.empty(co, 9 * ci, device="cuda"); ops.pack_weight_fwd(w, wp, co)
ax, aw = ops.amax(x), ops.amax(wp)
ws = ops.streamk_workspace()
outs = []
for s in (None, ws, ws):
    o = torch.empty(n, hw, hw, co, device="cuda")
    ops.conv_fwd(x, wp, o, k=3, stride=1, pad=1, Co=co, tile=5, split="f16x2", amax_in=ax, amax_w=aw, streamk_ws=s)
    outs.append(o)
torch.cuda.synchronize()
print("EQUAL", bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])), "GAVEUP", int(ws.view(torch.int32)[2047]))
"""


def test_every_consumer_gives_up_and_recomputes():
    """VD_SK_TIMEOUT_TICKS=0: no consumer ever takes a hand-off - each recomputes its tile's K-prefix itself (what happens
    when the producer's workgroup is not resident yet).  Same bits; the give-up counter says the path ran; a second launch
    on the same workspace finds its counters consistent (producers published, nobody consumed)."""
    env = dict(os.environ, VD_SK_TIMEOUT_TICKS="0")
    r = subprocess.run([sys.executable, "-c", _GIVE_UP % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("EQUAL")][-1].split()
    assert line[1] == "True" and int(line[3]) > 100, line


@pytest.mark.parametrize("tile", [6, 7, 2, 1, 10])
@pytest.mark.parametrize("case", [(64, 256, 26, 512, 3), (32, 128, 38, 256, 3), (100, 256, 26, 128, 1)])
def test_bf16_kernel_bit_identical(case, tile):
    """k_conv_igemm_bf16 (bf16 inference, bf16-storage training): the same claim - outputs and the fused BatchNorm statistics
    table of a training forward bit-identical to the one-tile-per-workgroup launch"""
    from viddet_amd import ops, lib as L
    n, ci, hw, co, k = case
    g = torch.Generator(device="cuda").manual_seed(4321 + tile)
    x = torch.randn(n, hw, hw, ci, device="cuda", generator=g).to(torch.bfloat16)
    wt = torch.randn(co, k * k * ci, device="cuda", generator=g) / np.sqrt(ci * k * k)
    wb = torch.empty(co, k * k * ci, device="cuda", dtype=torch.bfloat16)
    ops.pack_weight_bf16(wt, wb, Co=co, Co_pad=co, Ci=ci, Ci_pad=ci, T=k * k)
    res = torch.randn(n, hw, hw, co, device="cuda", generator=g).to(torch.bfloat16)
    sc, sh = torch.rand(co, device="cuda", generator=g) + 0.5, torch.randn(co, device="cuda", generator=g)
    ws = ops.streamk_workspace()
    geo = dict(N=n, Hi=hw, Wi=hw, Ci=ci, Hg=hw, Wg=hw, in_stride=1, taps=ops.fwd_taps(k, k // 2), Ho=hw, Wo=hw, Co=co, ldo=co, tile=tile)
    for nohalo in ((False, True) if k == 3 else (True,)):
        outs, used = [], False
        for sk in (None, ws):
            # inference form: BN fold + LeakyReLU + residual epilogue
            y = torch.empty(n, hw, hw, co, device="cuda", dtype=torch.bfloat16)
            d = ops.conv_igemm_bf16(x, wb, y, scale=sc, shift=sh, residual=res, ldr=co, leaky=True, nohalo=nohalo, streamk_ws=sk, **geo)
            if sk is not None:
                used = bool(L.load().vd_conv_igemm_bf16_streamk(C.byref(d), 0))
            # training form: raw output + statistics rows from the fp32 accumulators
            mt = ops.conv_bf16_mtiles(n, hw, hw, ci, co, tile)
            part = torch.zeros(mt, 2 * co, device="cuda")
            z = torch.empty(n, hw, hw, co, device="cuda", dtype=torch.bfloat16)
            ops.conv_igemm_bf16(x, wb, z, stats_part=part, nohalo=nohalo, streamk_ws=sk, **geo)
            outs.append((y, z, part))
        torch.cuda.synchronize()
        if not used:
            continue
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
    assert int(ws.view(torch.int32)[2047]) == 0


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 10])
@pytest.mark.parametrize("case", [(1, 512, 19, 1024, 3), (1, 1024, 19, 512, 1), (1, 256, 38, 512, 3), (2, 128, 38, 256, 3),
                                  (1, 1024, 13, 256, 1), (1, 384, 13, 256, 1)])
def test_bf16_split_k_of_launches_with_too_few_tiles(case, tile):
    """VD_CONV_SPLITK (batch-1 detection: 24 tiles of 128 x 128 on 256 CUs): K cut into 2..8 parts per tile, summed in part
    order by the part that arrives last.  Not the one-tile launch's association, so the claims are: the fp64 oracle within
    the bf16 kernel's tolerance, the one-tile launch within one bf16 rounding of the fp32 sums, the same bits on every
    run (whoever arrives last), the statistics rows of a training-form launch, and counters back at zero."""
    from viddet_amd import ops, lib as L
    n, ci, hw, co, k = case
    g = torch.Generator(device="cuda").manual_seed(977 + tile)
    x = torch.randn(n, hw, hw, ci, device="cuda", generator=g).to(torch.bfloat16)
    wt = torch.randn(co, k * k * ci, device="cuda", generator=g) / np.sqrt(ci * k * k)
    wb = torch.empty(co, k * k * ci, device="cuda", dtype=torch.bfloat16)
    ops.pack_weight_bf16(wt, wb, Co=co, Co_pad=co, Ci=ci, Ci_pad=ci, T=k * k)
    res = torch.randn(n, hw, hw, co, device="cuda", generator=g).to(torch.bfloat16)
    sc, sh = torch.rand(co, device="cuda", generator=g) + 0.5, torch.randn(co, device="cuda", generator=g)
    ws = ops.streamk_workspace()
    geo = dict(N=n, Hi=hw, Wi=hw, Ci=ci, Hg=hw, Wg=hw, in_stride=1, taps=ops.fwd_taps(k, k // 2), Ho=hw, Wo=hw, Co=co, ldo=co, tile=tile)
    # fp64 reference on the bf16 operands: packed weight layout [co][tap][ci] -> OIHW
    x64 = x.float().permute(0, 3, 1, 2).double().cpu().numpy()
    w64 = wb.float().reshape(co, k, k, ci).permute(0, 3, 1, 2).double().cpu().numpy()
    ref = R.conv2d(x64, w64, 1, k // 2)
    S = R.conv2d(np.abs(x64), np.abs(w64), 1, k // 2)
    hit = 0
    for nohalo in ((False, True) if k == 3 else (True,)):
        z0 = torch.empty(n, hw, hw, co, device="cuda", dtype=torch.bfloat16)
        p0 = torch.zeros(ops.conv_bf16_mtiles(n, hw, hw, ci, co, tile), 2 * co, device="cuda")
        ops.conv_igemm_bf16(x, wb, z0, stats_part=p0, nohalo=nohalo, **geo)
        y0 = torch.empty_like(z0)
        ops.conv_igemm_bf16(x, wb, y0, scale=sc, shift=sh, residual=res, ldr=co, leaky=True, nohalo=nohalo, **geo)
        runs = []
        for rep in range(3):
            z = torch.empty_like(z0)
            part = torch.zeros_like(p0)
            d = ops.conv_igemm_bf16(x, wb, z, stats_part=part, nohalo=nohalo, streamk_ws=ws, splitk=True, **geo)
            form = int(L.load().vd_conv_igemm_bf16_streamk(C.byref(d), 0))
            y = torch.empty_like(z0)
            ops.conv_igemm_bf16(x, wb, y, scale=sc, shift=sh, residual=res, ldr=co, leaky=True, nohalo=nohalo, streamk_ws=ws, splitk=True, **geo)
            runs.append((z, part, y))
        torch.cuda.synchronize()
        if form != 2:
            assert form == 0 and all(torch.equal(a, b) for a, b in zip(runs[0], (z0, p0, y0)))      # flag ignored: the same launch
            continue
        hit += 1
        for r in runs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(runs[0], r)), "split-K: run-to-run bits"
        z, part, y = runs[0]
        got = z.float().permute(0, 3, 1, 2).cpu().numpy()
        # fp32 accumulation of bf16 products in K / S chains, then one bf16 rounding of the result
        assert np.all(np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 1e-5 * S + 1e-30), float(np.abs(got - ref).max())
        # against the one-tile launch: the fp32 sums differ by summation order (<= ~1e-6 sum|a||b|), so the bf16 outputs differ
        # by at most one bf16 unit, on few elements
        dz = (z.float() - z0.float()).abs()
        assert float((dz / z0.float().abs().clamp_min(1e-3)).max()) <= 2.0 ** -7 and float((dz > 0).float().mean()) < 0.02
        dy = (y.float() - y0.float()).abs()
        assert float((dy / y0.float().abs().clamp_min(1e-2)).max()) <= 2.0 ** -6
        assert torch.allclose(part, p0, rtol=2e-5, atol=2e-5 * float(p0.abs().max()))
    assert int(ws.view(torch.int32)[L.SK_HEADER_BYTES // 8:L.SK_HEADER_BYTES // 4].abs().max()) == 0      # arrival counters back at zero
    if tile in (1, 2, 3, 4, 5) and n * hw * hw <= 1500 and k * k * ci // 64 >= 8:      # (fewer than 4 K-steps per part: not cut)
        assert hit > 0, "the split-K form was expected to apply here"
