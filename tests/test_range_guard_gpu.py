"""The dynamic-range rule of the fp16-split arithmetic (VD_MATH_F16X2) - what "f32" has to mean for the
fp32 nn.Conv2D of models/definitions/layers.py:66-67.

VD_MATH_F16X2 scales a whole operand tensor by one power of two.  Elements within ~2^18 of the tensor's max-abs keep 22-23
significant bits; below that the low fp16 piece goes subnormal and what is left is an ABSOLUTE resolution of 2^-39 of the
max-abs.  An output whose whole receptive field sits that far down is then NOT within the fp32-grade bound
8 u sum|a||b| of its own operands.  Two kinds of tensors can get there:
  * sparse, saturated ones - the loss gradient dhead.  Their consumers (data / weight gradients of the three prediction
    convs) never run the fp16 split: range-exact by construction (`YOLOV3._range_exact`).
  * dense ones whose per-channel scales have spread (a BatchNorm gamma or invstd 2^17 below its neighbours).  The range guard
    (vd_range_guard over the BatchNorm vectors, `check_operand_ranges`) flags them and the plans are rebuilt with the 3-way
    bf16 split (fp32's exponent range) for their consumers.
The tests: the bound itself on operands built to hit the hole (3-way split and fp32 MFMA inside 8 u sum|a||b| everywhere,
the fp16 split inside its own documented bound and demonstrably outside the fp32-grade one), on the head gradients of a
network after 120 SGD steps, and the guard + fallback end to end.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ops as R
from tests.util import dev, nchw_to_dev_nhwc, dev_nhwc_to_nchw
from tests.test_split_math_gpu import _packed, U

pytestmark = pytest.mark.gpu


def _ratio(got, ref, S):
    """largest |error| / sum|a||b| over the outputs that have any operand mass"""
    e = np.abs(np.asarray(got, np.float64) - ref)
    m = S > 0
    return float((e[m] / S[m]).max()), e


def test_channel_scales_spanning_2_to_the_30():
    """activation channels with magnitudes 2^0 .. 2^-30 and weights that make half the outputs read ONE channel group
    only: the outputs fed by the small channels are where a per-tensor scale cannot follow"""
    from viddet_amd import ops
    n, ci, h, w, co, k = 2, 64, 13, 13, 64, 3
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, ci, h, w)) * (2.0 ** (-30.0 * np.arange(ci) / (ci - 1))).reshape(1, ci, 1, 1)
    wt = rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)
    for j in range(co // 2):                       # output j < co/2 reads channels [2j, 2j + 2) only: scales down to 2^-30
        sel = np.zeros(ci)
        sel[2 * j:2 * j + 2] = 1.0
        wt[j] *= sel.reshape(ci, 1, 1)
    xd, wp = nchw_to_dev_nhwc(x), _packed(wt, co)
    x32 = xd.permute(0, 3, 1, 2).double().cpu().numpy()
    w32 = dev(wt).double().cpu().numpy()
    ref = R.conv2d(x32, w32, 1, 1)
    S = R.conv2d(np.abs(x32), np.abs(w32), 1, 1)
    amax_a, amax_b = np.abs(x32).max(), np.abs(w32).max()
    Sb = R.conv2d(np.ones_like(x32), np.abs(w32), 1, 1)       # sum |b| over each output's taps
    Sa = R.conv2d(np.abs(x32), np.ones_like(w32), 1, 1)       # sum |a|
    K = ci * k * k
    res = {}
    for split in (False, True, "f16x2"):
        out = torch.empty(n, h, w, co, device="cuda")
        ops.conv_fwd(xd, wp, out, k=k, stride=1, pad=1, Co=co, split=split)
        torch.cuda.synchronize()
        res[split] = _ratio(dev_nhwc_to_nchw(out), ref, S)
    print("max |err| / sum|a||b|:  fp32 MFMA %.2f u   3-way bf16 split %.2f u   2-way fp16 split %.3g u" % (
        res[False][0] / U, res[True][0] / U, res["f16x2"][0] / U))
    assert res[False][0] <= K * U                                  # any fp32 summation order
    assert res[True][0] <= 8 * U + K * U / 16                      # the range-exact split: fp32-grade on EVERY output
    # the fp16 split: inside its own a-priori bound (staging resolution 2^-39 of each tensor's max-abs) ...
    bound = (8 * U + K * U / 16) * S + 2.0 ** -39 * (amax_a * Sb + amax_b * Sa)
    assert np.all(res["f16x2"][1] <= bound), float((res["f16x2"][1] / bound).max())
    # ... and outside the fp32-grade one on the outputs that read the small channels: the hole the guard exists for
    assert res["f16x2"][0] > 64 * U


def test_head_gradients_after_training_and_their_consumers():
    """120 SGD steps on one batch: the head gradients become what they are late in training - sparse, saturated, a dynamic
    range of 2^40+.  As the activation operand of a 1x1 product (the prediction conv's data gradient) the 3-way split and
    the fp32 MFMA stay inside 8 u sum|a||b| on every output; the plan runs their consumers range-exact."""
    from viddet_amd import ops, lib as L
    from viddet_amd.model import ConvNode
    from tests.test_model_gpu import _mk_net, _targets
    c, size, B = 4, 64, 4
    net, P = _mk_net(c, 71, obj_bias=-1.0)
    rng = np.random.default_rng(71)
    x = rng.standard_normal((B, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, B, c, size, 3)
    args = [dev(x), dev(gt)] + [dev(t) for t in tg]
    for _ in range(120):
        net(*args)
        net.backward()
        net.sgd_step(lr=2e-3, momentum=0.9, wd=5e-4, batch_size=B)
    net(*args)
    net.backward()
    torch.cuda.synchronize()
    tp = net._last_train
    bufs = tp['bufs']
    heads = [n for n in net.conv_nodes if n.head]
    assert set('dz:' + n.name for n in heads) <= net._range_exact
    # every launch record that reads a head gradient: no fp16 split
    ptrs = {bufs['d:' + n.dst].data_ptr() for n in heads}
    seen = 0
    for seg in tp['bwd']:
        for fname, fn, a in seg.recs:
            if fname == 'vd_conv_igemm' and a[0]._obj.in_ in ptrs:
                assert not (a[0]._obj.flags & L.MATH_F16X2), "head data gradient in the fp16 split"
                seen += 1
            if fname == 'vd_conv_wgrad' and a[0]._obj.dout in ptrs:
                assert not (a[0]._obj.flags & L.MATH_F16X2), "head weight gradient in the fp16 split"
                seen += 1
    assert seen == 6, seen
    for n in heads:
        dout = bufs['d:' + n.dst].contiguous()
        co = dout.shape[-1]
        nz = dout[dout != 0].abs()
        span = float(np.log2(float(nz.max()) / float(nz.min()))) if nz.numel() else 0.0
        print("%s after 120 steps: zeros %.1f %%, dynamic range 2^%.0f" % (n.name, 100.0 * (1 - nz.numel() / dout.numel()), span))
        d32 = dout.permute(0, 3, 1, 2).double().cpu().numpy()
        wt = rng.standard_normal((64, co, 1, 1)) * np.sqrt(1.0 / co)
        w32 = dev(wt).double().cpu().numpy()
        ref = R.conv2d(d32, w32, 1, 0)
        S = R.conv2d(np.abs(d32), np.abs(w32), 1, 0)
        wp = _packed(wt, 64)
        r = {}
        for split in (False, True, "f16x2"):
            out = torch.empty(dout.shape[0], dout.shape[1], dout.shape[2], 64, device="cuda")
            ops.conv_fwd(dout, wp, out, k=1, stride=1, pad=0, Co=64, split=split)
            torch.cuda.synchronize()
            r[split] = _ratio(dev_nhwc_to_nchw(out), ref, S)[0]
        print("   max |err| / sum|a||b|: fp32 MFMA %.2f u, 3-way bf16 %.2f u, 2-way fp16 %.3g u" % (r[False] / U, r[True] / U, r["f16x2"] / U))
        assert r[False] <= co * U and r[True] <= 8 * U + co * U / 16, (n.name, r)


def test_guard_flags_a_spread_layer_and_the_plans_fall_back():
    """a BatchNorm whose gamma has one channel 2^-30 of the others: check_operand_ranges() names the cell's output, the
    plans are rebuilt and every consumer of that tensor (forward conv, weight gradient) leaves the fp16 split; an
    untouched network flags nothing; the step still matches itself bit for bit run to run."""
    from viddet_amd import lib as L
    from tests.test_model_gpu import _mk_net, _targets
    c, size, B = 4, 64, 2
    net, P = _mk_net(c, 72, obj_bias=-1.0)
    rng = np.random.default_rng(72)
    x = rng.standard_normal((B, 3, size, size)).astype(np.float32)
    gt, tg = _targets(rng, B, c, size, 3)
    args = [dev(x), dev(gt)] + [dev(t) for t in tg]
    net(*args)
    net.backward()
    assert net.check_operand_ranges() == {}                       # He-initialised network: every channel scale O(1)
    target = [n for n in net.conv_nodes if n.name == "stages.0.4.body.0"][0]
    p = net.collect_params()[target.name + ".1.gamma"]
    g = p.data().cpu()
    g[3] = 2.0 ** -30
    p.set_data(g)
    pb = net.collect_params()[target.name + ".1.beta"]
    b = pb.data().cpu()
    b[3] = 0.0
    pb.set_data(b)
    net(*args)
    net.backward()
    flagged = net.check_operand_ranges()
    assert target.dst in flagged and flagged[target.dst] < 2.0 ** -20, flagged
    assert ('dz:' + target.name) in flagged                      # scale = gamma * invstd spreads with gamma
    assert not net._programs or all(k[0] == 'buf' for k in net._programs), "plans were not dropped"
    out1 = [t.clone() for t in net(*args)]
    net.backward()
    torch.cuda.synchronize()
    tp = net._last_train
    src_ptr = tp['bufs'][target.dst].data_ptr()
    hit = 0
    for seg in tp['fwd'] + tp['bwd']:
        for fname, fn, a in seg.recs:
            if fname == 'vd_conv_igemm' and a[0]._obj.in_ == src_ptr:
                assert not (a[0]._obj.flags & L.MATH_F16X2)
                hit += 1
            if fname == 'vd_conv_wgrad' and a[0]._obj.in_ == src_ptr:
                assert not (a[0]._obj.flags & L.MATH_F16X2)
                hit += 1
    assert hit >= 2, hit
    assert net.check_operand_ranges() == {}                       # nothing new
    out2 = [t.clone() for t in net(*args)]
    torch.cuda.synchronize()
    assert all(torch.equal(u, v) for u, v in zip(out1, out2))
    assert all(bool(torch.isfinite(t).all()) for t in out2)
