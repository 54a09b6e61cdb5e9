"""Image-side training augmentations of the reference's video pipeline, on NumPy arrays (k,h,w,c).

Mirrors (paths under /root/reference):
  random_expand          models/transforms/video.py:12-65
  random_color_distort   models/transforms/video.py:68-158
  imresize               gluoncv.data.transforms.image.imresize -> mx.image.imresize (called at
                         models/definitions/yolo/transforms.py:229,332; interp codes 0-4, 9, 10)

The random decisions are drawn in the reference's order from the reference's two sources - NumPy's global RNG and
Python's `random` module (the hue angle and the expand ratio / offsets use the latter) - through an `Rng` pair, so a
caller that seeds both (or passes the global modules) consumes exactly the numbers the reference's code would.

[UPSTREAM-UNVERIFIED] mx.image.imresize is an OpenCV cv::resize call; OpenCV is not available offline.  The kernels
below follow OpenCV's sampling convention (pixel centres: src = (dst + 0.5) * scale - 0.5, replicated borders; bicubic
a = -0.75; Lanczos a = 4; INTER_AREA = coverage-weighted box average when shrinking), in float arithmetic - not OpenCV's
fixed-point uint8 paths, so uint8 results can differ from cv2 by one grey level.
"""
import functools
import random as _pyrandom

import numpy as np


class Rng:
    """The two random sources the reference's transforms draw from: `np` (numpy.random module or a RandomState) and
    `py` (the random module or a random.Random).  Default: the global modules, as in the reference."""

    def __init__(self, np_rng=None, py_rng=None):
        self.np = np.random if np_rng is None else np_rng
        self.py = _pyrandom if py_rng is None else py_rng

    @classmethod
    def seeded(cls, seed):
        return cls(np.random.RandomState(seed), _pyrandom.Random(seed))


def random_expand(src, max_ratio=4, fill=0, keep_ratio=True, rng=None):
    """video.py:12-65: place the (k,h,w,c) frames on a larger canvas filled with `fill`.
    Returns (canvas, (offset_x, offset_y, new_width, new_height)).  Draws: random.uniform(1, max_ratio)
    [, a second one if not keep_ratio], random.randint(0, oh - h), random.randint(0, ow - w)."""
    rng = Rng() if rng is None else rng
    if max_ratio <= 1:
        return src, (0, 0, src.shape[1], src.shape[0])          # (:39-40, the reference's own index slip kept)
    k, h, w, c = src.shape
    ratio_x = rng.py.uniform(1, max_ratio)
    ratio_y = ratio_x if keep_ratio else rng.py.uniform(1, max_ratio)
    oh, ow = int(h * ratio_y), int(w * ratio_x)
    off_y = rng.py.randint(0, oh - h)
    off_x = rng.py.randint(0, ow - w)
    if np.isscalar(fill):
        dst = np.full((k, oh, ow, c), fill, dtype=src.dtype)
    else:
        fill = np.asarray(fill, dtype=src.dtype)
        if c != fill.size:
            raise ValueError("Channel and fill size mismatch, {} vs {}".format(c, fill.size))
        dst = np.tile(fill.reshape(1, 1, 1, c), (k, oh, ow, 1))
    dst[:, off_y:off_y + h, off_x:off_x + w, :] = src
    return dst, (off_x, off_y, ow, oh)


_TYIQ = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.321], [0.211, -0.523, 0.311]])
_ITYIQ = np.array([[1.0, 0.956, 0.621], [1.0, -0.272, -0.647], [1.0, -1.107, 1.705]])


def hue_matrix(alpha):
    """video.py:131-146: the 3x3 the frames are right-multiplied by for a hue rotation of alpha * pi."""
    u, w = np.cos(alpha * np.pi), np.sin(alpha * np.pi)
    bt = np.array([[1.0, 0.0, 0.0], [0.0, u, -w], [0.0, w, u]])
    return np.dot(np.dot(_ITYIQ, bt), _TYIQ).T


def random_color_distort(src, brightness_delta=32, contrast_low=0.5, contrast_high=1.5, saturation_low=0.5,
                         saturation_high=1.5, hue_delta=18, rng=None):
    """video.py:68-158 on frames in [0, 255]; returns float32.  Draw order: brightness gate (np.uniform) [+ delta],
    order coin (np.randint(0, 2)), then contrast / saturation / hue - or saturation / hue / contrast - each a gate
    (np.uniform(0, 1) > 0.5) followed, if taken, by its parameter (np.uniform; the hue angle from random.uniform)."""
    rng = Rng() if rng is None else rng
    src = np.asarray(src).astype(np.float32)

    def brightness(x):
        if rng.np.uniform(0, 1) > 0.5:
            x = x + np.float32(rng.np.uniform(-brightness_delta, brightness_delta))
        return x

    def contrast(x):
        if rng.np.uniform(0, 1) > 0.5:
            x = x * np.float32(rng.np.uniform(contrast_low, contrast_high))
        return x

    def saturation(x):
        if rng.np.uniform(0, 1) > 0.5:
            alpha = np.float32(rng.np.uniform(saturation_low, saturation_high))
            gray = (x * np.array([0.299, 0.587, 0.114], np.float32)).sum(axis=-1, keepdims=True)
            x = x * alpha + gray * (np.float32(1.0) - alpha)
        return x

    def hue(x):
        if rng.np.uniform(0, 1) > 0.5:
            alpha = rng.py.uniform(-hue_delta, hue_delta)
            x = np.dot(x, hue_matrix(alpha).astype(np.float32))
        return x

    src = brightness(src)
    if rng.np.randint(0, 2):
        src = hue(saturation(contrast(src)))
    else:
        src = contrast(hue(saturation(src)))
    return src.astype(np.float32)


# --------------------------------------------------------------------------------------------
# imresize
# --------------------------------------------------------------------------------------------
def _cubic_w(t, a=-0.75):
    t = np.abs(t)
    return np.where(t <= 1, ((a + 2) * t - (a + 3)) * t * t + 1, np.where(t < 2, ((a * t - 5 * a) * t + 8 * a) * t - 4 * a, 0.0))


def _lanczos_w(t, a=4):
    t = np.asarray(t, dtype=np.float64)
    out = np.where(np.abs(t) < a, np.sinc(t) * np.sinc(t / a), 0.0)
    return out


@functools.lru_cache(maxsize=256)
def _axis_taps(n_in, n_out, interp):
    """Sparse resampling operator of one axis: (idx (n_out, T) int64, w (n_out, T) float64) with
    out[d] = sum_k w[d, k] * in[idx[d, k]]; indices are clipped to the axis (replicated border: OpenCV's convention), so
    one source pixel may appear under several taps.  Cached per (n_in, n_out, interp): the training loader draws from a
    handful of shapes (train_yolov3.py:262-271)."""
    scale = n_in / n_out
    d = np.arange(n_out)
    if interp == 0:                                          # nearest: floor(dst * scale)
        return np.minimum((d * scale).astype(np.int64), n_in - 1)[:, None], np.ones((n_out, 1))
    if interp == 2 and n_out < n_in:                         # area, shrinking: coverage of [d*scale, (d+1)*scale)
        lo, hi = d * scale, np.minimum((d + 1) * scale, n_in)
        T = int(np.ceil(scale)) + 1
        j = np.floor(lo).astype(np.int64)[:, None] + np.arange(T)[None, :]
        w = np.clip(np.minimum(hi[:, None], j + 1) - np.maximum(lo[:, None], j), 0.0, None)
        w /= w.sum(axis=1, keepdims=True)
        return np.clip(j, 0, n_in - 1), w
    f = (d + 0.5) * scale - 0.5
    if interp in (1, 2):                                     # bilinear (area when enlarging = bilinear)
        taps, wf = 2, lambda t: np.maximum(0.0, 1.0 - np.abs(t))
    elif interp == 3:
        taps, wf = 4, _cubic_w
    elif interp == 4:
        taps, wf = 8, _lanczos_w
    else:
        raise ValueError("interp %r" % (interp,))
    j = (np.floor(f).astype(np.int64) - (taps // 2 - 1))[:, None] + np.arange(taps)[None, :]
    w = wf(f[:, None] - j)
    if interp == 4:
        w = w / w.sum(axis=1, keepdims=True)
    return np.clip(j, 0, n_in - 1), w


def _axis_weights(n_in, n_out, interp):
    """Dense (n_out, n_in) form of `_axis_taps` (tests; small shapes)."""
    idx, w = _axis_taps(n_in, n_out, interp)
    W = np.zeros((n_out, n_in))
    np.add.at(W, (np.arange(n_out)[:, None].repeat(idx.shape[1], 1), idx), w)
    return W


def _resample_axis(x, axis, idx, w):
    """out = sum_k w[:, k] * take(x, idx[:, k], axis): T gathers of the whole array, no (n_out, n_in) matrix."""
    shape = [1] * x.ndim
    shape[axis] = idx.shape[0]
    out = None
    for k in range(idx.shape[1]):
        t = np.take(x, idx[:, k], axis=axis) * w[:, k].reshape(shape)
        out = t if out is None else out + t
    return out


def imresize(img, w, h, interp=1, rng=None):
    """(h0,w0,c) image -> (h,w,c).  interp: 0 nearest, 1 bilinear, 2 area, 3 bicubic, 4 Lanczos (8x8), 9 = area when
    shrinking, bicubic when enlarging, bilinear otherwise, 10 = one of 0-4 at random.  uint8 in -> uint8 out (rounded,
    saturated); float in -> float32 out, unclipped."""
    img = np.asarray(img)
    h0, w0 = img.shape[:2]
    if interp == 10:
        interp = int((Rng() if rng is None else rng).np.randint(0, 5))
    if interp == 9:
        interp = 2 if (h < h0 and w < w0) else (3 if (h > h0 and w > w0) else 1)
    if (h, w) == (h0, w0):
        return img.copy()
    x = img.astype(np.float64)
    if interp == 0:
        ys = np.minimum((np.arange(h) * (h0 / h)).astype(np.int64), h0 - 1)
        xs = np.minimum((np.arange(w) * (w0 / w)).astype(np.int64), w0 - 1)
        out = x[ys][:, xs]
    else:
        # separable: the axis that shrinks more goes first (fewer elements for the second pass)
        ty, tx = _axis_taps(h0, h, interp), _axis_taps(w0, w, interp)
        if w * h0 <= h * w0:
            out = _resample_axis(_resample_axis(x, 1, *tx), 0, *ty)
        else:
            out = _resample_axis(_resample_axis(x, 0, *ty), 1, *tx)
    if img.dtype == np.uint8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out.astype(np.float32)
