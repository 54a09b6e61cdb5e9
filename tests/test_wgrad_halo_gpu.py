"""GPU parity of the halo-ring weight-gradient kernel (viddet_amd/csrc/vd_wgrad_halo.hip, VD_WGRAD_HALO in
include/viddet_hip.h): the weight gradient of the 3x3 / stride-1 / pad-1 convolutions (autograd.backward of
models/definitions/layers.py:66-67, train_yolov3.py:631) with the activation rows staged once per pixel in an LDS ring.

Checked against the fp64 oracle (oracle/ops.py conv2d_backward) with the tolerance of the other weight-gradient kernels
(2e-4 * sqrt(pixels) on O(1) operands), against the generic kernel's error level, and on the geometries that stress the
padded reduction space: maps narrower / wider than a 32-position step, images that start mid-step, rings that wrap
(128 / 256 / 512 rows), ragged channel tiles, forced split counts (including one: no slabs), and pixel ranges that end
inside an image.  bf16-stored operands (VD_STORE_BF16) are exact products with fp32 sums: same bound, no bf16 term."""
import numpy as np
import pytest
import torch

from oracle import ops as R
from tests.util import nchw_to_dev_nhwc, maxdiff
from tests.test_conv_gpu import _mk

pytestmark = pytest.mark.gpu
TOL = 2e-4
BF = torch.bfloat16

CASES = [  # n, ci, h, w, co, splits
    (4, 128, 26, 26, 256, 0),      # 128 co x two chunks (the tile of every layer with Ci % 64 == 0), ring of 128 rows
    (3, 64, 13, 13, 128, 0),       # a step (32 positions) spans more than two image rows
    (2, 32, 15, 17, 288, 3),       # one chunk: the 256-row tile (ragged second co tile), odd split count, H != W
    (2, 96, 9, 11, 288, 1),        # single split: the kernel writes dwp itself (no slabs); Ci = 96 takes the 256-row tile
    (1, 64, 52, 52, 128, 5),       # ring of 256 rows
    (1, 64, 104, 104, 160, 7),     # two chunks x 512 rows do not fit LDS: the 320-row ring that wraps by compare; ragged co
    (2, 64, 6, 5, 128, 2),         # the smallest map the kernel takes (W = 5)
    (5, 64, 19, 19, 256, 0),       # a 608-family size: the pad column makes rows of 20 positions
    (1, 32, 40, 208, 256, 0),      # the widest map (ring 512 rows = 64 + 2 * 224), 256-row tile
]


def _uses_halo(x, dout, co, mode):
    """does the library take the halo-ring kernel for this launch? (a test that silently ran the generic one proves nothing)"""
    import ctypes as C
    from viddet_amd import lib as L, ops
    d = L.WgradDesc()
    N, Hi, Wi, Ci = x.shape
    d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.Co, d.ldd = N, Hi, Wi, Ci, Hi, Wi, co, dout.shape[-1]
    d.in_stride, d.Kfr = 1, 1
    ops._set_taps(d, ops.fwd_taps(3, 1))
    if x.dtype == torch.bfloat16:
        d.flags = L.STORE_BF16 | L.MATH_BF16 | L.WGRAD_HALO
    else:
        d.flags = L.MATH_F16X2 | L.WGRAD_HALO
        d.amax_in = d.amax_dout = 1          # only tested for NULL
    return bool(L.load().vd_conv_wgrad_uses_halo(C.byref(d)))


def _ref_wgrad(x, dy, co, ci):
    _, dw = R.conv2d_backward(x, np.zeros((co, ci, 3, 3)), dy, 1, 1)
    return dw


@pytest.mark.parametrize("case", CASES)
def test_halo_wgrad_matches_oracle(case):
    from viddet_amd import ops
    n, ci, h, w, co, splits = case
    rng, x, _ = _mk(n, ci, h, w, co, 3, 31)
    dy = rng.standard_normal((n, co, h, w))
    dw_ref = _ref_wgrad(x, dy, co, ci)
    ws = torch.empty(96 << 20, dtype=torch.uint8, device="cuda")
    xd, dyd = nchw_to_dev_nhwc(x), nchw_to_dev_nhwc(dy)
    assert _uses_halo(xd, dyd, co, "f16x2h")
    err = {}
    for mode in ("f16x2h", "f16x2"):
        dwp = torch.full((co, 9 * ci), 5.0, device="cuda")
        ops.conv_wgrad(xd, dyd, dwp, ws, k=3, stride=1, pad=1, Co=co, splits=splits, split=mode)
        dw = torch.empty(co, ci, 3, 3, device="cuda")
        ops.unpack_weight(dwp, dw)
        torch.cuda.synchronize()
        err[mode] = maxdiff(dw.cpu().numpy(), dw_ref)
    assert err["f16x2h"] < TOL * np.sqrt(n * h * w), err
    assert err["f16x2h"] <= 4.0 * err["f16x2"] + 1e-6, err          # the same arithmetic in another summation order


def test_halo_wgrad_padded_dout_pitch_and_operand_scales():
    """dout rows wider than Co (ldd > Co: the head-gradient pitch rule) and operands far outside fp16's range: the
    per-tensor power-of-two scales come from the max-abs slots, as in the generic kernel."""
    from viddet_amd import ops
    n, ci, h, w, co, ldd = 2, 64, 13, 13, 128, 192
    rng, x, _ = _mk(n, ci, h, w, co, 3, 5)
    x = x * 3.0e7
    dy = rng.standard_normal((n, co, h, w)) * 2.0 ** -40
    dw_ref = _ref_wgrad(x, dy, co, ci)
    dyd = torch.full((n, h, w, ldd), float("nan"), device="cuda")     # the pad channels are never read
    dyd[..., :co] = nchw_to_dev_nhwc(dy)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    dwp = torch.empty(co, 9 * ci, device="cuda")
    assert _uses_halo(nchw_to_dev_nhwc(x), dyd, co, "f16x2h")
    ops.conv_wgrad(nchw_to_dev_nhwc(x), dyd, dwp, ws, k=3, stride=1, pad=1, Co=co, split="f16x2h")
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    ops.unpack_weight(dwp, dw)
    torch.cuda.synchronize()
    scale = 3.0e7 * 2.0 ** -40
    assert maxdiff(dw.cpu().numpy() / scale, dw_ref / scale) < TOL * np.sqrt(n * h * w)


def test_halo_wgrad_is_deterministic_and_ignored_where_it_does_not_apply():
    from viddet_amd import ops
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    # two launches, identical bits (slabs summed in slab order)
    n, ci, h, w, co = 4, 64, 26, 26, 256
    rng, x, _ = _mk(n, ci, h, w, co, 3, 77)
    xd, dyd = nchw_to_dev_nhwc(x), nchw_to_dev_nhwc(rng.standard_normal((n, co, h, w)))
    a, b = torch.empty(co, 9 * ci, device="cuda"), torch.empty(co, 9 * ci, device="cuda")
    ops.conv_wgrad(xd, dyd, a, ws, k=3, stride=1, pad=1, Co=co, split="f16x2h")
    ops.conv_wgrad(xd, dyd, b, ws, k=3, stride=1, pad=1, Co=co, split="f16x2h")
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # stride 2, 1x1 and Co < 128: the flag is ignored - bit for bit the generic kernel's result
    for (ci2, co2, k, s, p) in ((64, 128, 3, 2, 1), (128, 256, 1, 1, 0), (64, 64, 3, 1, 1), (32, 128, 3, 1, 1)):     # (last: one chunk would need the 256-row tile)
        rng, x, _ = _mk(2, ci2, 14, 14, co2, k, 3)
        ho = (14 + 2 * p - k) // s + 1
        xd, dyd = nchw_to_dev_nhwc(x), nchw_to_dev_nhwc(rng.standard_normal((2, co2, ho, ho)))
        a, b = torch.empty(co2, k * k * ci2, device="cuda"), torch.empty(co2, k * k * ci2, device="cuda")
        ops.conv_wgrad(xd, dyd, a, ws, k=k, stride=s, pad=p, Co=co2, split="f16x2h")
        ops.conv_wgrad(xd, dyd, b, ws, k=k, stride=s, pad=p, Co=co2, split="f16x2")
        torch.cuda.synchronize()
        assert torch.equal(a, b)


def _r(a):      # bf16-representable fp64 values
    return torch.from_numpy(a.astype(np.float32)).to(BF).double().numpy()


def _nhwc_b(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 1)).astype(np.float32)).to(BF).cuda()


@pytest.mark.parametrize("case", [(4, 128, 26, 26, 256, 0), (3, 64, 13, 13, 128, 0), (2, 32, 15, 17, 288, 3), (1, 64, 52, 52, 128, 5),
                                  (2, 96, 9, 11, 288, 1), (1, 64, 104, 104, 160, 7), (1, 64, 38, 208, 128, 0)])
def test_halo_wgrad_from_bf16_operands(case):
    from viddet_amd import ops
    n, ci, h, w, co, splits = case
    rng = np.random.default_rng(sum(case) + 11)
    x = _r(rng.standard_normal((n, ci, h, w)))
    dz = _r(rng.standard_normal((n, co, h, w)))
    dw_ref = _ref_wgrad(x, dz, co, ci)
    ws = torch.empty(96 << 20, dtype=torch.uint8, device="cuda")
    dwp = torch.full((co, 9 * ci), 3.0, device="cuda")
    assert _uses_halo(_nhwc_b(x), _nhwc_b(dz), co, "halo")
    ops.conv_wgrad(_nhwc_b(x), _nhwc_b(dz), dwp, ws, k=3, stride=1, pad=1, Co=co, splits=splits, split="halo")
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    ops.unpack_weight(dwp, dw)
    torch.cuda.synchronize()
    assert maxdiff(dw.cpu().numpy(), dw_ref) < TOL * np.sqrt(n * h * w)       # exact products, fp32 sums


def test_halo_wgrad_random_geometries_against_the_generic_kernel():
    """Seeded sweep over map sizes, channel counts and split counts - every ring size, both tiles, ranges that start and end
    anywhere in an image - against k_conv_wgrad on the same operands (same arithmetic, another summation order: agreement to
    a few fp32 roundings of the sum), fp32 and bf16 tensors.  A wrong ring row, a stale mirror row or a misplaced pad position
    shows up as an O(1) difference."""
    from viddet_amd import ops
    rng = np.random.default_rng(2024)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    tried = 0
    for _ in range(48):
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(2, 61)), int(rng.integers(5, 121))
        ci = int(rng.choice([32, 64, 96, 128, 192]))
        co = int(rng.choice([128, 160, 256, 288, 384]))
        splits = int(rng.choice([0, 0, 1, 2, 3, 5, 7]))
        x = torch.randn(n, h, w, ci, device="cuda")
        dy = torch.randn(n, h, w, co, device="cuda")
        for bf in (False, True):
            xx, dd = (x.to(BF), dy.to(BF)) if bf else (x, dy)
            mode_h, mode_g = ("halo", False) if bf else ("f16x2h", "f16x2")
            if not _uses_halo(xx, dd, co, mode_h):
                continue
            tried += 1
            a = torch.full((co, 9 * ci), 3.0, device="cuda")
            b = torch.full((co, 9 * ci), 4.0, device="cuda")
            ops.conv_wgrad(xx, dd, a, ws, k=3, stride=1, pad=1, Co=co, splits=splits, split=mode_h)
            ops.conv_wgrad(xx, dd, b, ws, k=3, stride=1, pad=1, Co=co, split=mode_g)
            torch.cuda.synchronize()
            err = float((a - b).abs().max())
            assert err < 2e-4 * np.sqrt(n * h * w), (n, h, w, ci, co, splits, bf, err)
    assert tried >= 40, tried
