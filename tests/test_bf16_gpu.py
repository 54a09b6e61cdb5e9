"""GPU: bf16 inference path (bf16 storage + v_mfma_f32_32x32x16_bf16, fp32 accumulate/epilogue, fp32 heads).
The reference has no reduced precision, so the bound is defined here against the fp32 oracle:
  * a single conv on bf16-representable inputs is exact to fp32 accumulation error (products of bf16 are exact in fp32);
  * whole-network raw head outputs stay within 3 % of the head's max magnitude (bf16 has 8 significant bits and the
    75-layer stack re-rounds every activation), and the detections keep their class and overlap the oracle's boxes
    (IoU > 0.9; > 0.75 on the full 608x608 frame of BASELINE configs[1], whose larger logits carry larger absolute error)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import net as ON
from oracle import ops as R
from tests.util import dev, maxdiff

pytestmark = pytest.mark.gpu


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)


_BF16_CASES = [(t, sh) for t in (1, 2, 3, 4, 5, 6, 7, 8, 9)
               for sh in [(2, 64, 13, 11, 128, 3, 1, 1), (3, 128, 8, 8, 192, 1, 1, 0), (2, 64, 17, 15, 64, 3, 2, 1)]]
# narrow-output tiles (Co = 64 / 32) and the two-taps-per-K-step layout of 32-channel inputs (3x3: odd tap count ->
# half-empty last K-step; 1x1: a single half-empty K-step; stride 2)
_BF16_CASES += [(t, sh) for t in (10, 11, 13) for sh in [(2, 64, 17, 15, 64, 3, 1, 1), (2, 128, 9, 9, 64, 1, 1, 0),
                                                      (2, 32, 19, 21, 64, 3, 1, 1), (2, 32, 18, 20, 64, 3, 2, 1),
                                                      (1, 32, 9, 9, 64, 1, 1, 0), (3, 32, 40, 24, 40, 3, 1, 1)]]
# 14 / 15: the small four-wave tiles of the short-K 1x1 layers (also on the two-tap layout of 32-channel inputs)
_BF16_CASES += [(14, (2, 128, 9, 9, 64, 1, 1, 0)), (14, (2, 64, 13, 11, 128, 1, 1, 0)), (14, (3, 32, 12, 10, 64, 1, 1, 0)),
                (14, (2, 64, 17, 15, 64, 3, 1, 1)), (15, (2, 64, 21, 19, 32, 1, 1, 0)), (15, (2, 128, 12, 12, 24, 3, 1, 1))]
_BF16_CASES += [(12, (2, 64, 21, 19, 32, 1, 1, 0)), (12, (2, 64, 12, 12, 24, 3, 1, 1)), (0, (2, 64, 21, 19, 32, 1, 1, 0)),
                (0, (2, 64, 17, 15, 64, 3, 1, 1)), (0, (2, 64, 17, 15, 256, 3, 1, 1))]

# 16: the first-stage patch kernel (vd_conv_c32_bf16.hip: 32 -> 64 channels, 3x3, stride 1 / 2): maps narrower and wider than a
# 32-pixel patch, heights that are not multiples of the patch's 8 / 4 rows, odd sizes under stride 2, one full-width row of the
# 608 x 608 frame's 304-wide map; (out_f32 and shapes it does not serve fall back to the generic tiles: same assertions)
_BF16_CASES += [(16, sh) for sh in [(2, 32, 19, 21, 64, 3, 1, 1), (2, 32, 18, 20, 64, 3, 2, 1), (1, 32, 33, 70, 64, 3, 1, 1),
                                    (3, 32, 41, 67, 64, 3, 2, 1), (1, 32, 9, 304, 64, 3, 1, 1), (1, 32, 17, 608, 64, 3, 2, 1),
                                    (2, 32, 8, 32, 64, 3, 1, 1), (3, 32, 40, 24, 40, 3, 1, 1)]]

# halo-staged loop (3x3 stride 1 on the 8-wave tiles): several 64-channel chunks, maps wider than a tile's rows and narrower,
# frames ending inside a tile, the 76- and 152-wide maps of the 608x608 detection path; every case also with the flag that
# forces the generic loop
_BF16_CASES += [(t, sh) for t in (2, 3, 6, 7, 10) for sh in [(2, 192, 20, 76, 128, 3, 1, 1), (3, 128, 7, 9, 128, 3, 1, 1),
                                                            (1, 64, 9, 152, 64, 3, 1, 1), (2, 256, 19, 19, 128, 3, 1, 1)]]


@pytest.mark.parametrize("tile,shape", _BF16_CASES)
def test_conv_bf16_single_layer(tile, shape):
    from viddet_amd import ops, lib as L
    n, ci, h, w, co, k, s, p = shape
    rng = np.random.default_rng(90 + tile)
    x = _bf16_round(rng.standard_normal((n, ci, h, w)))
    wt = _bf16_round(rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k))
    scale, shift = rng.uniform(0.5, 1.5, co), rng.standard_normal(co)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    res = _bf16_round(rng.standard_normal((n, co, ho, wo)))
    z = R.conv2d(x, wt, s, p) * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
    ref = R.leaky(z) + res
    xd = torch.from_numpy(np.moveaxis(x, 1, -1).copy()).to(torch.bfloat16).cuda()
    rd = torch.from_numpy(np.moveaxis(res, 1, -1).copy()).to(torch.bfloat16).cuda()
    wp32 = torch.empty(co, k * k * ci, device="cuda")
    ops.pack_weight_fwd(dev(wt), wp32, co)
    wb = torch.empty(co, k * k * ci, dtype=torch.bfloat16, device="cuda")
    L.check(L.load().vd_pack_weight_bf16(wp32.data_ptr(), wb.data_ptr(), co, co, ci, ci, k * k, L.stream_ptr()), "pack")
    for out_f32 in (0, 1):
        out = torch.empty(n, ho, wo, co, dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
        d = L.ConvDesc()
        d.in_, d.wp, d.out = xd.data_ptr(), wb.data_ptr(), out.data_ptr()
        d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = n, h, w, ci, ho, wo, s
        ops._set_taps(d, ops.fwd_taps(k, p))
        d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.ldo, d.ldr, d.tile = 1, ho, wo, co, 1, co, co, tile
        sc, sh = dev(scale), dev(shift)
        d.scale, d.shift, d.residual = sc.data_ptr(), sh.data_ptr(), rd.data_ptr()
        for nohalo in (0, L.MATH_NOHALO):
            d.flags, d.slope = 1 | 2 | 4 | nohalo, 0.1
            out.zero_()
            L.check(L.load().vd_conv_igemm_bf16(C.byref(d), out_f32, L.stream_ptr()), "vd_conv_igemm_bf16")
            torch.cuda.synchronize()
            got = np.moveaxis(out.float().cpu().numpy(), -1, 1)
            tol = 2e-4 if out_f32 else 2e-4 + np.abs(ref).max() * 2 ** -8      # + one bf16 rounding of the output
            assert maxdiff(got, ref) < tol, (nohalo, maxdiff(got, ref), tol)


@pytest.mark.parametrize("shape", [(2, 38, 70), (1, 64, 608), (3, 41, 67), (1, 9, 33), (2, 16, 64)])
def test_stem_and_first_stride2_conv_in_one_launch(shape):
    """vd_stem_conv_c32_bf16: the stem (3 -> 32, folded BN, LeakyReLU, bf16) computed inside the stride-2 conv's patch kernel
    instead of being stored - the SAME BITS as vd_stem_conv(out_bf16) + vd_conv_igemm_bf16 (frames of odd and even sizes, maps
    narrower and wider than a patch, the 608-wide row), and the two-launch form itself against the fp64 oracle."""
    from viddet_amd import ops, lib as L
    n, h, w = shape
    rng = np.random.default_rng(700 + h + w)
    frames = rng.standard_normal((n, 3, h, w)).astype(np.float32)
    w0 = (rng.standard_normal((32, 3, 3, 3)) / np.sqrt(27)).astype(np.float32)
    s0, b0 = rng.uniform(0.5, 1.5, 32).astype(np.float32), rng.standard_normal(32).astype(np.float32)
    w1 = _bf16_round(rng.standard_normal((64, 32, 3, 3)) / np.sqrt(288))
    s1, b1 = rng.uniform(0.5, 1.5, 64).astype(np.float32), rng.standard_normal(64).astype(np.float32)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    lib = L.load()
    xd = dev(frames)
    wp0 = torch.zeros(32, 32, device="cuda")
    wp0[:, :27].copy_(dev(w0).permute(0, 2, 3, 1).reshape(32, 27))      # k = (ky * 3 + kx) * 3 + c
    sc0, sh0, sc1, sh1 = dev(s0), dev(b0), dev(s1), dev(b1)
    a = torch.empty(n, h, w, 32, dtype=torch.bfloat16, device="cuda")
    L.check(lib.vd_stem_conv(xd.data_ptr(), wp0.data_ptr(), a.data_ptr(), 32, n, h, w, sc0.data_ptr(), sh0.data_ptr(), 0.1, 1 | 2, 1,
                             None, L.stream_ptr()), "vd_stem_conv")
    wp32 = torch.empty(64, 288, device="cuda")
    ops.pack_weight_fwd(dev(w1), wp32, 64)
    wb = torch.empty(64, 288, dtype=torch.bfloat16, device="cuda")
    L.check(lib.vd_pack_weight_bf16(wp32.data_ptr(), wb.data_ptr(), 64, 64, 32, 32, 9, L.stream_ptr()), "pack")
    outs = []
    for fused in (False, True):
        out = torch.empty(n, ho, wo, 64, dtype=torch.bfloat16, device="cuda")
        d = L.ConvDesc()
        d.in_, d.wp, d.out = a.data_ptr(), wb.data_ptr(), out.data_ptr()
        d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = n, h, w, 32, ho, wo, 2
        ops._set_taps(d, ops.fwd_taps(3, 1))
        d.Kfr, d.Ho, d.Wo, d.Co, d.out_stride, d.ldo, d.ldr, d.tile = 1, ho, wo, 64, 1, 64, 64, 16
        d.scale, d.shift, d.flags, d.slope = sc1.data_ptr(), sh1.data_ptr(), 1 | 2, 0.1
        if fused:
            d.in_ = None
            L.check(lib.vd_stem_conv_c32_bf16(xd.data_ptr(), wp0.data_ptr(), sc0.data_ptr(), sh0.data_ptr(), 0.1, C.byref(d), L.stream_ptr()),
                    "vd_stem_conv_c32_bf16")
        else:
            L.check(lib.vd_conv_igemm_bf16(C.byref(d), 0, L.stream_ptr()), "vd_conv_igemm_bf16")
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    # the two-launch form against fp64 on what the device holds: bf16 stem output, bf16 weights
    a64 = np.moveaxis(a.float().cpu().numpy(), -1, 1).astype(np.float64)
    ref = R.leaky(R.conv2d(a64, w1, 2, 1) * s1.reshape(1, -1, 1, 1) + b1.reshape(1, -1, 1, 1))
    got = np.moveaxis(outs[0].float().cpu().numpy(), -1, 1)
    assert maxdiff(got, ref) < 2e-4 + np.abs(ref).max() * 2 ** -8
    # a descriptor that is not that conv is refused
    d.in_stride = 1
    assert lib.vd_stem_conv_c32_bf16(xd.data_ptr(), wp0.data_ptr(), sc0.data_ptr(), sh0.data_ptr(), 0.1, C.byref(d), L.stream_ptr()) != 0


@pytest.mark.parametrize("c,b,size,obj_bias", [(4, 2, 96, -1.0),
                                               (20, 1, 608, -3.0)])     # BASELINE configs[1]: one full 608x608 frame
def test_bf16_network_inference_close_to_fp32_oracle(c, b, size, obj_bias):
    from viddet_amd.model import yolo3_darknet53
    P = ON.init_params(c, seed=51, obj_bias=obj_bias)
    net = yolo3_darknet53(["c%d" % i for i in range(c)])
    for k, p in net.collect_params().items():
        p.set_data(torch.from_numpy(P[k].astype(np.float32)))
    rng = np.random.default_rng(51)
    x = rng.standard_normal((b, 3, size, size)).astype(np.float32)
    ids_r, sc_r, bx_r, rows_r, heads_r = ON.Net(P, c).detect(x.astype(np.float64))
    ids32, sc32, bx32 = [t.clone() for t in net(dev(x))]
    net.set_precision('bf16')
    ids, sc, bx = net(dev(x))
    torch.cuda.synchronize()
    bufs = net._programs[('infer_bf16', b, size, size)][1]
    for s_, hname in enumerate(net.head_names):
        got = bufs[hname].cpu().numpy()[..., :3 * (5 + c)]
        ref = np.moveaxis(heads_r[s_], 1, -1)
        rel = maxdiff(got, ref) / np.abs(ref).max()
        print("head", s_, "max err / max|head| =", rel)
        assert rel < 3e-2
    # detections: the top-scoring fp32 detections are found again (same class, IoU > 0.9)
    ir, br, sr = ids_r[..., 0], bx_r, sc_r[..., 0]
    ib, bb = ids.cpu().numpy()[..., 0], bx.cpu().numpy()
    for bi in range(b):
        top = [j for j in range(100) if ir[bi, j] >= 0 and sr[bi, j] > 0.3][:10]
        for j in top:
            cand = np.nonzero(ib[bi] == ir[bi, j])[0]
            assert len(cand) > 0
            a = br[bi, j]
            best = 0.0
            for q in cand:
                bq = bb[bi, q]
                iw = max(0.0, min(a[2], bq[2]) - max(a[0], bq[0])); ih = max(0.0, min(a[3], bq[3]) - max(a[1], bq[1]))
                u = (a[2] - a[0]) * (a[3] - a[1]) + (bq[2] - bq[0]) * (bq[3] - bq[1]) - iw * ih
                best = max(best, iw * ih / u if u > 0 else 0.0)
            # full-size frame: raw wh logits carry up to ~0.12 of absolute error (1.2 % of max|head|, inside the 3 % bound
            # above), i.e. e^0.12 on a box side: IoU down to ~0.8 (measured 0.815)
            assert best > (0.9 if size < 608 else 0.75), (bi, j, best)
    net.set_precision('fp32')
    again = net(dev(x))
    torch.cuda.synchronize()
    assert torch.equal(again[0], ids32) and torch.equal(again[2], bx32)


def _heads_close(net, key_fp32, key_bf16, c, tol=3e-2):
    """raw head outputs of the bf16 plan against the fp32 plan of the same shape: within `tol` of the head's max magnitude
    (the bound test_bf16_network_inference_close_to_fp32_oracle holds the plain network to, against the oracle)."""
    b32, b16 = net._programs[key_fp32], net._programs[key_bf16][1]
    for hname in net.head_names:
        ref = b32[hname][..., :3 * (5 + c)].float()
        got = b16[hname][..., :3 * (5 + c)].float()
        assert got.shape == ref.shape
        rel = float((got - ref).abs().max()) / float(ref.abs().max())
        print(hname, "bf16 vs fp32 tensors: max err / max|head| = %.4f" % rel)
        assert rel < tol, (hname, rel)


@pytest.mark.parametrize("cfg", [dict(jt="max", jp="early", bct="2"), dict(jt="mean", jp="late", bct="21"),
                                 dict(jt="cat", jp="late", bct="3"), dict(jt="cat", jp="early", bct="2")])
def test_bf16_inference_of_the_temporal_window_networks(cfg):
    """k = 3 windows in bf16 (set_precision('bf16') used to be k = 1 only): TimeDistributed backbone, temporal pooling /
    stacking, 3-D and 2+1-D neck convs - the fp32 heads of the bf16 plan against the fp32 plan (itself oracle-checked in
    tests/test_temporal_gpu.py), and the same detections classes for the confident ones."""
    from tests.test_temporal_gpu import _mk
    c, b, size, K = 3, 2, 64, 3
    net, P = _mk(cfg, c, 41)
    rng = np.random.default_rng(41)
    x = rng.standard_normal((b, K, 3, size, size)).astype(np.float32)
    ids32, sc32, bx32 = [t.clone() for t in net(dev(x))]
    net.set_precision('bf16')
    ids, sc, bx = net(dev(x))
    torch.cuda.synchronize()
    assert tuple(ids.shape) == tuple(ids32.shape)
    _heads_close(net, ('buf', b, size, size, False), ('infer_bf16', b, size, size), c)
    net.set_precision('fp32')
    again = net(dev(x))
    assert torch.equal(again[0], ids32) and torch.equal(again[2], bx32)


def test_bf16_inference_of_the_no_backbone_and_per_frame_output_networks():
    from tests.test_noback_gpu import _mk_noback
    from tests.test_temporal_out_gpu import _mk as _mk_tout, T_
    # YOLOV3_noback: three cached fp32 feature maps in, converted to bf16 NHWC on the way
    c, b, size = 4, 2, 64
    P = ON.init_params(c, seed=13, obj_bias=-1.0)
    net = _mk_noback(c, P)
    rng = np.random.default_rng(13)
    feats = [rng.standard_normal((b, ch, size // d, size // d)).astype(np.float32) for ch, d in ((256, 8), (512, 16), (1024, 32))]
    r32 = [t.clone() for t in net(*[dev(f) for f in feats])]
    net.set_precision('bf16')
    r16 = net(*[dev(f) for f in feats])
    torch.cuda.synchronize()
    assert tuple(r16[0].shape) == tuple(r32[0].shape)
    _heads_close(net, ('buf', b, size, size, False), ('infer_bf16', b, size, size), c)
    # YOLOV3Temporal with per-frame outputs (t = 5): frame selections, valid-frame (3,1,1) convs, heads on every frame
    net2, P2 = _mk_tout("21", 3, 51)
    x = rng.standard_normal((1, T_, 3, size, size)).astype(np.float32)
    a32 = [t.clone() for t in net2(dev(x))]
    net2.set_precision('bf16')
    a16 = net2(dev(x))
    torch.cuda.synchronize()
    assert tuple(a16[0].shape) == (1, T_, 100, 1) == tuple(a32[0].shape)
    _heads_close(net2, ('buf', 1, size, size, False), ('infer_bf16', 1, size, size), 3)
