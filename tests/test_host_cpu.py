"""CPU: host logic — C-ABI surface, tap/dgrad plans, graph structure, target generator, checkpoint IO."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import net as ON
from oracle import ops as R
from oracle import yolo as Y

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_loads_and_exports_every_declared_symbol():
    from viddet_amd import lib as L
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "viddet_hip.h")).read()
    declared = set(re.findall(r"\b(vd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), "libviddet_hip.so does not export %s" % name
        assert name in L.SIGNATURES, "no ctypes signature for %s" % name
    assert set(L.SIGNATURES) == declared
    assert lib.vd_version() >= 100
    assert lib.vd_abi_version() == L.ABI_VERSION == int(re.search(r"#define VD_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert [lib.vd_sizeof_desc(i) for i in range(4)] == [ctypes.sizeof(L.ConvDesc), ctypes.sizeof(L.WgradDesc),
                                                         ctypes.sizeof(L.HeadDesc), -1]
    # struct sizes the Python side assumes (no compute call without a GPU)
    assert ctypes.sizeof(L.HeadDesc) == 136
    assert ctypes.sizeof(L.ConvDesc) % 8 == 0 and ctypes.sizeof(L.WgradDesc) % 8 == 0


def test_abi_rejects_bad_arguments_without_touching_the_gpu():
    from viddet_amd import lib as L
    lib = L.load()
    d = L.ConvDesc()
    assert lib.vd_conv_igemm(ctypes.byref(d), None) == -1
    assert b"null" in lib.vd_last_error()
    d.in_, d.wp, d.out = 16, 16, 16
    d.Ci = 24
    assert lib.vd_conv_igemm(ctypes.byref(d), None) == -1
    assert b"multiple of 32" in lib.vd_last_error()
    assert lib.vd_sgd_momentum(None, None, None, 10, 0.1, 0.9, 0.0, 1.0, None) == -1


def test_halo_weight_gradient_dispatch_rules_without_touching_the_gpu():
    """vd_conv_wgrad_uses_halo / vd_conv_wgrad_ws_bytes are host logic (include/viddet_hip.h VD_WGRAD_HALO): which launches
    take the halo-ring kernel of vd_wgrad_halo.hip, and that the workspace query follows the kernel's own split count."""
    from viddet_amd import lib as L, ops
    lib = L.load()

    def desc(n, ci, h, w, co, k=3, stride=1, flags=L.MATH_F16X2 | L.WGRAD_HALO, amax=True, ldd=None):
        d = L.WgradDesc()
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.Co, d.ldd = n, h, w, ci, ho, wo, co, (co if ldd is None else ldd)
        d.in_stride, d.Kfr, d.flags = stride, 1, flags
        ops._set_taps(d, ops.fwd_taps(k, pad))
        if amax:
            d.amax_in = d.amax_dout = 16                    # only compared with NULL
        return d

    uses = lambda d: lib.vd_conv_wgrad_uses_halo(ctypes.byref(d))
    assert uses(desc(64, 256, 26, 26, 512)) == 1             # the Darknet-53 3x3 layers
    assert uses(desc(64, 64, 104, 104, 128)) == 1            # 104-wide map: the ring that wraps by compare
    assert uses(desc(64, 512, 13, 13, 1024, flags=L.STORE_BF16 | L.MATH_BF16 | L.WGRAD_HALO, amax=False)) == 1
    assert uses(desc(64, 256, 26, 26, 512, flags=L.MATH_F16X2)) == 0                        # flag not set
    assert uses(desc(64, 256, 26, 26, 512, flags=L.MATH_SPLIT | L.WGRAD_HALO)) == 0         # another arithmetic
    assert uses(desc(64, 256, 26, 26, 512, amax=False)) == 0                                # fp16 split without max-abs slots
    assert uses(desc(64, 256, 52, 52, 512, stride=2)) == 0                                  # stride 2
    assert uses(desc(64, 512, 26, 26, 256, k=1)) == 0                                       # 1x1
    assert uses(desc(64, 32, 208, 208, 64)) == 0                                            # fewer than 128 output channels
    assert uses(desc(64, 32, 52, 52, 128)) == 0                                             # one chunk would need the 256-row tile
    assert uses(desc(64, 32, 52, 52, 256)) == 1                                             # ... which Co = 256 fills
    assert uses(desc(2, 64, 26, 26, 132, flags=L.STORE_BF16 | L.MATH_BF16 | L.WGRAD_HALO, amax=False)) == 0   # rows not whole 16-byte loads
    assert uses(desc(64, 64, 4, 4, 128)) == 0                                               # narrower than the cursors handle
    # the workspace is the kernel's slab array: splits x Co x 9 Ci floats, a whole number of slabs, and it differs from the
    # generic kernel's (other split count) - a caller must ask with the flags of the launch
    d = desc(64, 256, 26, 26, 512)
    need = lib.vd_conv_wgrad_ws_bytes(ctypes.byref(d))
    slab = 512 * 9 * 256 * 4
    assert need > 0 and need % slab == 0 and need // slab <= 512
    d.splits = 5
    assert lib.vd_conv_wgrad_ws_bytes(ctypes.byref(d)) == 5 * slab
    d.splits = 1
    assert lib.vd_conv_wgrad_ws_bytes(ctypes.byref(d)) == 0                                 # one range: straight into dwp


@pytest.mark.parametrize("k,pad,stride,hi", [(3, 1, 1, 8), (1, 0, 1, 5), (3, 1, 2, 8), (3, 1, 2, 9)])
def test_dgrad_plans_reproduce_conv_backward(k, pad, stride, hi):
    """Execute the tap plans with plain numpy loops and compare with the oracle's conv backward."""
    from viddet_amd.ops import dgrad_plans
    rng = np.random.default_rng(0)
    ci, co, wi = 3, 4, hi + 1
    x = rng.standard_normal((1, ci, hi, wi))
    w = rng.standard_normal((co, ci, k, k))
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    dy = rng.standard_normal((1, co, ho, wo))
    ref, _ = R.conv2d_backward(x, w, dy, stride, pad)
    dx = np.full_like(x, np.nan)
    for plan in dgrad_plans(k, pad, stride, hi, wi):
        for qy in range(plan["Hg"]):
            for qx in range(plan["Wg"]):
                acc = np.zeros(ci)
                for (dyy, dxx, _), t in zip(plan["taps"], plan["tap_ids"]):
                    oy, ox = qy + dyy, qx + dxx
                    if 0 <= oy < ho and 0 <= ox < wo:
                        acc += w[:, :, t // k, t % k].T @ dy[0, :, oy, ox]
                dx[0, :, qy * stride + plan["py"], qx * stride + plan["px"]] = acc
    assert not np.isnan(dx).any(), "some input pixel is covered by no plan"
    assert np.allclose(dx, ref, atol=1e-12)


def test_graph_structure_and_flops():
    from viddet_amd.model import build_graph, ConvNode
    nodes, tensors, heads = build_graph(80)
    convs = [n for n in nodes if isinstance(n, ConvNode)]
    assert len(convs) == 75 and sum(1 for n in convs if n.bn) == 72
    assert sum(n.cout for n in convs if n.bn) == 26304                      # sum of BN channels
    names = {n.name + (".0" if n.bn else "") for n in convs}
    shapes = ON.param_shapes(80)
    assert {k.rsplit(".", 1)[0] for k in shapes if k.endswith("weight")} == names
    flops = sum(2 * n.cin * n.cout * n.k * n.k * (416 // n.div_out) ** 2 for n in convs)
    assert abs(flops / 1e9 - 65.86) < 0.02                                   # 65.86 GFLOP @416, C=80
    flops608 = sum(2 * n.cin * n.cout * n.k * n.k * (608 // n.div_out) ** 2 for n in convs)
    assert abs(flops608 / 1e9 - 140.69) < 0.05
    assert [tensors[h][2] for h in heads] == [256, 256, 256]


@pytest.mark.parametrize("seed", range(3))
def test_targets_match_oracle(seed):
    from viddet_amd.targets import prefetch_targets, synthetic_batch
    size, c = 416, 20
    x, gt, ids = synthetic_batch(3, size, c, seed, max_gt=6)
    gt = gt.astype(np.float64)
    gt[1, 3:] = -1
    ids[1, 3:] = -1                      # ragged: image 1 has 3 boxes, and nothing after the first pad is used
    gt[2, :] = -1
    ids[2, :] = -1                       # empty image
    got = prefetch_targets(size, size, gt, ids, c)
    ref = Y.prefetch_targets(size, size, [13, 26, 52], gt, ids, c)
    for g, r in zip(got, ref):
        assert g.shape == r.shape
        assert np.allclose(g, r, atol=1e-6)
    assert got[0][2].sum() == 0 and got[0][1].sum() <= 3
    assert x.shape == (3, 3, size, size) and x.dtype == np.float32


def test_targets_multihot_and_mixratio():
    from viddet_amd.targets import prefetch_targets
    gt = np.array([[[10., 10., 100., 120.]]])
    ids = np.array([[[1., 0., 1.]]])
    mix = np.array([[[0.4]]])
    got = prefetch_targets(128, 128, gt, ids, 3, mix)
    ref = Y.prefetch_targets(128, 128, [4, 8, 16], gt, ids, 3, mix)
    for g, r in zip(got, ref):
        assert np.allclose(g, r, atol=1e-6)
    assert np.isclose(got[0].max(), 0.4)


def test_params_io_roundtrip(tmp_path):
    from viddet_amd.params_io import save_params, load_params
    from collections import OrderedDict
    rng = np.random.default_rng(0)
    arrs = OrderedDict([("stages.0.0.0.weight", rng.standard_normal((32, 3, 3, 3)).astype(np.float32)),
                        ("stages.0.0.1.gamma", rng.standard_normal(32).astype(np.float32)),
                        ("yolo_outputs.0.prediction.bias", rng.standard_normal(255).astype(np.float32))])
    p = str(tmp_path / "t.params")
    save_params(p, arrs)
    back = load_params(p)
    assert list(back) == list(arrs)
    for k in arrs:
        assert back[k].dtype == np.float32 and np.array_equal(back[k], arrs[k])
    raw = open(p, "rb").read()
    assert raw[:8] == (0x112).to_bytes(8, "little")
    # the reader is strict: everything the layout does not allow fails loudly (no real MXNet file exists offline to
    # validate the layout itself - SURVEY A.5 - so at least a mismatch can never load as garbage)
    import struct

    def broken(mut, msg):
        b = bytearray(raw)
        mut(b)
        q = str(tmp_path / "bad.params")
        open(q, "wb").write(bytes(b))
        with pytest.raises(ValueError, match=msg):
            load_params(q)

    broken(lambda b: b.__setitem__(slice(0, 8), (0x113).to_bytes(8, "little")), "not an NDArray list")
    broken(lambda b: b.__setitem__(slice(8, 16), (1).to_bytes(8, "little")), "reserved")
    broken(lambda b: b.__setitem__(slice(24, 28), (0xdeadbeef).to_bytes(4, "little")), "unsupported NDArray magic")
    broken(lambda b: b.__setitem__(slice(28, 32), struct.pack("<i", 1)), "sparse")
    broken(lambda b: b.__setitem__(slice(32, 36), struct.pack("<I", 40)), "implausible rank")
    broken(lambda b: b.__setitem__(slice(36, 44), struct.pack("<q", -3)), "negative dimension")
    off_flag = 24 + 12 + 4 * 8 + 8                      # first array: magic, stype, ndim, 4 dims, dev_type, dev_id -> dtype flag
    broken(lambda b: b.__setitem__(slice(off_flag, off_flag + 4), struct.pack("<i", 99)), "unknown dtype flag")
    broken(lambda b: b.__delitem__(slice(len(b) - 7, len(b))), "implausible length|truncated")
    broken(lambda b: b.extend(b"xx"), "trailing bytes")
    broken(lambda b: b.__delitem__(slice(200, 1200)), "magic|past the end|truncated|names|length")
    with pytest.raises(ValueError):
        open(p, "wb").write(b"\0" * 64)
        load_params(p)
