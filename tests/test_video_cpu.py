"""CPU: the image side of the training augmentation chain (viddet_amd/video.py, viddet_amd/data.py) against known
answers, and the ORDER in which the chain consumes its two random sources, which is what makes a seeded run of this
pipeline take the reference's decisions (models/transforms/video.py:12-158, models/definitions/yolo/transforms.py:
199-246).  The box side (translate / constrained crop / resize / flip) is pinned by the reference-generated golden
vectors in tests/test_golden_bbox.py."""
import random

import numpy as np
import pytest

from viddet_amd import video as V
from viddet_amd.data import MEAN, SyntheticDetection, YOLO3VideoInferenceTransform, YOLO3VideoTrainTransform


class _Rec:
    """A random source that logs every draw and answers from a script (or from a real generator)."""

    def __init__(self, log, tag, real, forced=None):
        self.log, self.tag, self.real, self.forced = log, tag, real, dict(forced or {})

    def _draw(self, name, args):
        self.log.append((self.tag, name) + tuple(args))
        return getattr(self.real, name)(*args)

    def uniform(self, a, b):
        v = self._draw("uniform", (a, b))
        return self.forced.get(("uniform", a, b), v)

    def randint(self, a, b=None):
        return self._draw("randint", (a, b) if b is not None else (a,))

    def randrange(self, a):
        return self._draw("randrange", (a,))


def _rng(log, gates_open=True, seed=5):
    # (0, 1) gates forced open (> 0.5) or shut so that every parameter draw happens / none does
    g = {("uniform", 0, 1): 0.9 if gates_open else 0.1}
    return V.Rng(_Rec(log, "np", np.random.RandomState(seed), g), _Rec(log, "py", random.Random(seed)))


def test_color_distort_draw_order_and_arithmetic():
    x = np.random.default_rng(0).integers(0, 256, (2, 5, 7, 3)).astype(np.uint8)
    log = []
    rng = _rng(log)
    rng.np.forced[("uniform", -32, 32)] = 10.0
    rng.np.forced[("uniform", 0.5, 1.5)] = 1.25
    out = V.random_color_distort(x, rng=rng)
    order = log[2][2:]                                   # np.randint(0, 2): which branch
    assert log[0] == ("np", "uniform", 0, 1) and log[1] == ("np", "uniform", -32, 32) and log[2][:2] == ("np", "randint")
    coin = np.random.RandomState(5)
    coin.uniform(0, 1); coin.uniform(-32, 32)
    first = coin.randint(0, 2)
    tail = [e[:2] + e[2:] for e in log[3:]]
    con = [("np", "uniform", 0, 1), ("np", "uniform", 0.5, 1.5)]
    sat = [("np", "uniform", 0, 1), ("np", "uniform", 0.5, 1.5)]
    hue = [("np", "uniform", 0, 1), ("py", "uniform", -18, 18)]
    assert tail == (con + sat + hue if first else sat + hue + con), (first, tail)
    # the arithmetic, replayed by hand with the same parameters
    h_alpha = random.Random(5).uniform(-18, 18)
    y = x.astype(np.float32) + np.float32(10.0)

    def sat_f(v):
        gray = (v * np.array([0.299, 0.587, 0.114], np.float32)).sum(-1, keepdims=True)
        return v * np.float32(1.25) + gray * np.float32(-0.25)

    def hue_f(v):
        return v @ V.hue_matrix(h_alpha).astype(np.float32)
    y = hue_f(sat_f(y * np.float32(1.25))) if first else hue_f(sat_f(y)) * np.float32(1.25)
    assert out.dtype == np.float32 and np.allclose(out, y, rtol=1e-5, atol=1e-3)
    # all gates shut: nothing but the three... four gate draws and the coin, and the image comes back unchanged
    log2 = []
    out2 = V.random_color_distort(x, rng=_rng(log2, gates_open=False))
    assert [e[1] for e in log2] == ["uniform", "randint", "uniform", "uniform", "uniform"]
    assert np.array_equal(out2, x.astype(np.float32))


def test_hue_matrix_properties():
    assert np.abs(V.hue_matrix(0.0) - np.eye(3)).max() < 2e-3      # the published YIQ pair is inverse to 3 digits
    g = np.array([[120.0, 120.0, 120.0]])
    assert np.allclose(g @ V.hue_matrix(0.37), g, atol=0.5)        # greys have no hue


def test_random_expand_places_the_frames():
    x = np.random.default_rng(1).integers(0, 256, (3, 10, 16, 3)).astype(np.float32)
    log = []
    rng = _rng(log, seed=9)
    fill = [m * 255 for m in MEAN]
    dst, (ox, oy, ow, oh) = V.random_expand(x, fill=fill, rng=rng)
    r = random.Random(9)
    ratio = r.uniform(1, 4)
    assert (oh, ow) == (int(10 * ratio), int(16 * ratio))
    assert oy == r.randint(0, oh - 10) and ox == r.randint(0, ow - 16)
    assert [e[:2] for e in log] == [("py", "uniform"), ("py", "randint"), ("py", "randint")]
    assert dst.shape == (3, oh, ow, 3) and np.array_equal(dst[:, oy:oy + 10, ox:ox + 16], x)
    mask = np.ones((oh, ow), bool)
    mask[oy:oy + 10, ox:ox + 16] = False
    assert np.allclose(dst[0][mask], np.asarray(fill, np.float32))
    same, box = V.random_expand(x, max_ratio=1, rng=rng)
    assert same is x


def test_imresize_known_answers():
    ramp = np.array([[[0.0], [100.0]]], dtype=np.float32)                      # 1 x 2 image
    assert np.allclose(V.imresize(ramp, 4, 1, interp=1)[0, :, 0], [0, 25, 75, 100])          # OpenCV INTER_LINEAR
    assert np.allclose(V.imresize(ramp, 4, 1, interp=3)[0, :, 0], [-10.546875, 22.65625, 77.34375, 110.546875])  # a = -0.75, replicated border
    assert np.allclose(V.imresize(ramp, 4, 1, interp=0)[0, :, 0], [0, 0, 100, 100])          # floor(dst * scale)
    img = np.arange(4 * 6, dtype=np.float32).reshape(4, 6, 1)
    assert np.allclose(V.imresize(img, 3, 2, interp=2)[..., 0], img[..., 0].reshape(2, 2, 3, 2).mean(axis=(1, 3)))  # box mean
    five = np.array([[[10.0], [20.0], [30.0], [40.0], [50.0]]], dtype=np.float32)
    assert np.allclose(V.imresize(five, 2, 1, interp=2)[0, :, 0], [(10 + 20 + 0.5 * 30) / 2.5, (0.5 * 30 + 40 + 50) / 2.5])
    const = np.full((7, 9, 3), 93, np.uint8)
    for it in (0, 1, 2, 3, 4, 9):
        for (w, h) in ((5, 4), (13, 11), (9, 7), (5, 11)):
            out = V.imresize(const, w, h, interp=it)
            assert out.dtype == np.uint8 and out.shape == (h, w, 3) and np.all(out == 93), (it, w, h)
    # a linear ramp is reproduced in the interior: exactly by the bilinear kernel, to a tenth of a source step by the
    # a = -0.75 cubic and the Lanczos kernel (neither is linear-precise; OpenCV's are not either)
    lin = (np.arange(32, dtype=np.float32) * 3.0)[None, :, None].repeat(4, 0)
    for it, tol in ((1, 1e-3), (3, 0.3), (4, 0.3)):
        up = V.imresize(lin, 64, 4, interp=it)[2, 8:-8, 0]
        assert np.allclose(up, ((np.arange(64) + 0.5) * 0.5 - 0.5)[8:-8] * 3.0, atol=tol), it
    # interp 9: area when shrinking, bicubic when enlarging, bilinear when mixed; uint8 saturates
    rnd = np.random.default_rng(3).integers(0, 256, (12, 10, 3)).astype(np.uint8)
    assert np.array_equal(V.imresize(rnd, 5, 6, interp=9), V.imresize(rnd, 5, 6, interp=2))
    assert np.array_equal(V.imresize(rnd, 20, 24, interp=9), V.imresize(rnd, 20, 24, interp=3))
    assert np.array_equal(V.imresize(rnd, 20, 6, interp=9), V.imresize(rnd, 20, 6, interp=1))
    edge = np.zeros((2, 4, 1), np.uint8)
    edge[:, 2:] = 255
    assert V.imresize(edge, 16, 2, interp=3).max() == 255 and V.imresize(edge, 16, 2, interp=3).min() == 0


def test_train_transform_chain_order_and_invariants():
    ds = SyntheticDetection("voc", num_samples=3, size=(120, 90))
    img, label = ds[1]
    log = []
    tf = YOLO3VideoTrainTransform(64, 64, 20, _rng(log, seed=11))
    out = tf(img, label)
    names = [e[:2] for e in log]
    # colour distortion: 1 gate + delta, coin, 3 x (gate + parameter); then the expand gate + ratio + two offsets; the
    # crop's trials (python source) end with numpy's pick; then the interpolation code and the flip gate
    assert names[:9] == [("np", "uniform"), ("np", "uniform"), ("np", "randint")] + [("np", "uniform")] * 0 + names[3:9]
    i = 9
    assert names[i] == ("np", "uniform") and names[i + 1] == ("py", "uniform") and names[i + 2:i + 4] == [("py", "randint")] * 2
    assert names[-3:] == [("np", "randint"), ("np", "randint"), ("np", "uniform")]          # crop pick, interp, flip gate
    assert log[-2][2:] == (0, 5)
    assert all(n[0] == "py" for n in names[i + 4:-3])                                        # the crop trials
    x, obj, ctr, scl, wgt, cls, gt = out
    assert x.shape == (3, 64, 64) and x.dtype == np.float32 and gt.shape[1] == 4
    assert gt.min() >= 0 and gt.max() <= 64 and np.all(gt[:, 2] >= gt[:, 0]) and np.all(gt[:, 3] >= gt[:, 1])
    assert obj.shape == (3 * (2 * 2 + 4 * 4 + 8 * 8), 1) and int((obj > 0).sum()) >= 1
    # a private seeded pair reproduces itself and leaves the global generators alone
    np.random.seed(1); random.seed(1)
    a = YOLO3VideoTrainTransform(64, 64, 20, V.Rng.seeded(7))(img, label)
    b = YOLO3VideoTrainTransform(64, 64, 20, V.Rng.seeded(7))(img, label)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    assert np.random.uniform() == np.random.RandomState(1).uniform() and random.random() == random.Random(1).random()
    # the global pair (the reference's sources) is what Rng() draws from
    np.random.seed(3); random.seed(3)
    c = YOLO3VideoTrainTransform(64, 64, 20, V.Rng())(img, label)
    d = YOLO3VideoTrainTransform(64, 64, 20, V.Rng.seeded(3))(img, label)
    assert all(np.array_equal(u, v) for u, v in zip(c, d))
    # windows: one set of decisions for all frames; per-frame labels give stacked targets
    dsw = SyntheticDetection("vid", num_samples=2, window=3, mult_out=True, size=(80, 60))
    w = YOLO3VideoTrainTransform(64, 64, 30, V.Rng.seeded(2))(*dsw[0])
    assert w[0].shape == (3, 3, 64, 64) and w[1].shape[0] == 3 and w[-1].shape[0] == 3


def test_inference_transform_uint8_and_float_paths_agree():
    ds = SyntheticDetection("voc", num_samples=2, size=(120, 90))
    img, label = ds[0]
    xf, bf, _ = YOLO3VideoInferenceTransform(64, 64)(img, label, 0)
    xu, bu, _ = YOLO3VideoInferenceTransform(64, 64, device_normalize=True)(img, label, 0)
    assert xu.dtype == np.uint8 and xu.shape == (64, 64, 3) and np.array_equal(bf, bu)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    assert np.array_equal(xf, np.ascontiguousarray(((xu.astype(np.float32) / 255.0 - MEAN) / std).transpose(2, 0, 1)))
    assert np.array_equal(xu, V.imresize(img, 64, 64, interp=9))                       # transforms.py:332 interp = 9


def test_random_shape_loader_draws_one_shape_per_interval_and_the_same_on_every_rank():
    """train_yolov3.py:262-271 (gluoncv RandomTransformDataLoader, the reference's default training loader): a transform
    out of the list every `interval` batches, a whole batch through one of them; every rank of a data-parallel run draws
    the same sequence (its shards belong to one global batch)."""
    from viddet_amd.data import Loader
    ds = SyntheticDetection("voc", num_samples=24, size=(120, 90))
    mk = lambda seed: [YOLO3VideoTrainTransform(s, s, 20, V.Rng.seeded(seed)) for s in (32, 64, 96)]
    seqs = []
    for rank in (0, 1):
        ld = Loader(ds, mk(5 + rank), 2, train=True, shuffle=True, seed=5, rank=rank, world=2, interval=2)
        shapes = []
        for batch in ld:
            assert batch[0].shape[0] == 2 and batch[0].shape[2] == batch[0].shape[3]
            shapes.append(batch[0].shape[-1])
            g = batch[0].shape[-1] // 32
            assert batch[1].shape[1] == 3 * (g * g + 4 * g * g + 16 * g * g)          # targets of that shape's grids
        assert len(shapes) == 6 and all(shapes[i] == shapes[i - 1] for i in range(1, 6, 2)), shapes
        seqs.append(shapes)
    assert seqs[0] == seqs[1] and len(set(seqs[0])) > 1, seqs
    with pytest.raises(ValueError):
        Loader(ds, mk(0), 2, train=False)
