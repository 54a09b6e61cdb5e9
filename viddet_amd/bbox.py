"""Host-side box utilities with the reference's names, argument meaning and quirks.

Mirrors (paths under /root/reference) — behaviour is pinned by tests/golden/bbox_golden.npz, which
was produced by running the reference's own NumPy code:
  utils/bbox.py:11-140                bbox_iou, bbox_xywh_to_xyxy, bbox_xyxy_to_xywh, bbox_clip_xyxy
  models/transforms/bbox.py:13-333    random_crop_with_constraints, crop, flip, resize, translate
All functions that take `bboxs` accept one (N,4+) array or a list of T such arrays (the frames of a
temporal window) and return the same structure.  These run on the host (data pipeline), like the
reference's.
"""
import random

import numpy as np


def bbox_iou(bbox_a, bbox_b, offset=0):
    """(N,4+) x (M,4+) -> (N,M) IoU; width = right - left + offset (utils/bbox.py:11-38)."""
    if bbox_a.shape[1] < 4 or bbox_b.shape[1] < 4:
        raise IndexError("Bounding boxes axis 1 must have at least length 4")
    lo = np.maximum(bbox_a[:, None, :2], bbox_b[None, :, :2])
    hi = np.minimum(bbox_a[:, None, 2:4], bbox_b[None, :, 2:4])
    overlap = (lo < hi).all(axis=2)
    inter = (hi - lo + offset).prod(axis=2) * overlap
    area_a = (bbox_a[:, 2:4] - bbox_a[:, :2] + offset).prod(axis=1)
    area_b = (bbox_b[:, 2:4] - bbox_b[:, :2] + offset).prod(axis=1)
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def _check4(v, what):
    if len(v) != 4:
        raise IndexError("Bounding boxes must have 4 elements, given {}".format(len(v)))


def bbox_xywh_to_xyxy(xywh):
    """(x, y, w, h) -> (xmin, ymin, xmax, ymax) with xmax = x + max(w-1, 0) (utils/bbox.py:41-69)."""
    if isinstance(xywh, (tuple, list)):
        _check4(xywh, "xywh")
        x, y, w, h = xywh
        return x, y, x + np.maximum(w - 1, 0), y + np.maximum(h - 1, 0)
    if isinstance(xywh, np.ndarray):
        if xywh.size % 4:
            raise IndexError("Bounding boxes must have n * 4 elements, given {}".format(xywh.shape))
        return np.concatenate([xywh[:, :2], xywh[:, :2] + np.maximum(0, xywh[:, 2:4] - 1)], axis=1)
    raise TypeError("Expect input xywh a list, tuple or numpy.ndarray, given {}".format(type(xywh)))


def bbox_xyxy_to_xywh(xyxy):
    """(xmin, ymin, xmax, ymax) -> (x, y, w, h) with w = xmax - xmin + 1 (utils/bbox.py:72-100)."""
    if isinstance(xyxy, (tuple, list)):
        _check4(xyxy, "xyxy")
        x1, y1, x2, y2 = xyxy
        return x1, y1, x2 - x1 + 1, y2 - y1 + 1
    if isinstance(xyxy, np.ndarray):
        if xyxy.size % 4:
            raise IndexError("Bounding boxes must have n * 4 elements, given {}".format(xyxy.shape))
        return np.concatenate([xyxy[:, :2], xyxy[:, 2:4] - xyxy[:, :2] + 1], axis=1)
    raise TypeError("Expect input xywh a list, tuple or numpy.ndarray, given {}".format(type(xyxy)))


def bbox_clip_xyxy(xyxy, width, height):
    """Clip to [0, width-1] x [0, height-1].  The ndarray form returns a FLAT concatenation
    [x1.., y1.., x2.., y2..] exactly as the reference does (utils/bbox.py:138)."""
    lim = (width - 1, height - 1, width - 1, height - 1)
    if isinstance(xyxy, (tuple, list)):
        _check4(xyxy, "xyxy")
        return tuple(np.minimum(l, np.maximum(0, v)) for v, l in zip(xyxy, lim))
    if isinstance(xyxy, np.ndarray):
        if xyxy.size % 4:
            raise IndexError("Bounding boxes must have n * 4 elements, given {}".format(xyxy.shape))
        return np.concatenate([np.minimum(l, np.maximum(0, xyxy[:, i])) for i, l in enumerate(lim)])
    raise TypeError("Expect input xywh a list, tuple or numpy.ndarray, given {}".format(type(xyxy)))


def _per_frame(bboxs, fn):
    """Apply fn to a private copy of each frame's boxes; keep the single-array / list calling form."""
    single = not isinstance(bboxs, list)
    frames = [bboxs] if single else bboxs
    done = [fn(np.array(b, copy=True)) for b in frames]
    return done[0] if single else done


def crop(bboxs, crop_box=None, allow_outside_center=True):
    """Clip boxes to crop_box=(l,t,w,h) and shift into its frame (models/transforms/bbox.py:131-197).
    As in the reference, rows that fall outside are NOT removed from the returned arrays (its mask is
    applied to a loop-local name only); callers see clipped, possibly degenerate boxes."""
    if crop_box is None:
        return _per_frame(bboxs, lambda b: b)
    if len(crop_box) != 4:
        raise ValueError("Invalid crop_box parameter, requires length 4, given {}".format(str(crop_box)))
    if all(c is None for c in crop_box):
        return _per_frame(bboxs, lambda b: b)
    l, t, w, h = crop_box
    left, top = (l or 0), (t or 0)
    region = np.array((left, top, left + (w if w else np.inf), top + (h if h else np.inf)))

    def one(b):
        b[:, :2] = np.maximum(b[:, :2], region[:2])
        b[:, 2:4] = np.minimum(b[:, 2:4], region[2:])
        b[:, :2] -= region[:2]
        b[:, 2:4] -= region[:2]
        return b

    return _per_frame(bboxs, one)


def flip(bboxs, size, flip_x=False, flip_y=False):
    """Mirror boxes inside an image of size=(width,height) (models/transforms/bbox.py:200-249)."""
    if len(size) != 2:
        raise ValueError("size requires length 2 tuple, given {}".format(len(size)))
    width, height = size

    def one(b):
        if flip_y:
            b[:, 1], b[:, 3] = height - b[:, 3], height - b[:, 1].copy()
        if flip_x:
            b[:, 0], b[:, 2] = width - b[:, 2], width - b[:, 0].copy()
        return b

    return _per_frame(bboxs, one)


def resize(bboxs, in_size, out_size):
    """Scale boxes from in_size=(w,h) to out_size=(w,h) (models/transforms/bbox.py:252-296)."""
    if len(in_size) != 2:
        raise ValueError("in_size requires length 2 tuple, given {}".format(len(in_size)))
    if len(out_size) != 2:
        raise ValueError("out_size requires length 2 tuple, given {}".format(len(out_size)))
    sx, sy = out_size[0] / in_size[0], out_size[1] / in_size[1]

    def one(b):
        b[:, 0::2][:, :2] = sx * b[:, 0::2][:, :2]
        b[:, 1::2][:, :2] = sy * b[:, 1::2][:, :2]
        return b

    return _per_frame(bboxs, one)


def translate(bboxs, x_offset=0, y_offset=0):
    """Shift boxes (models/transforms/bbox.py:299-333)."""
    def one(b):
        b[:, :2] += (x_offset, y_offset)
        b[:, 2:4] += (x_offset, y_offset)
        return b

    return _per_frame(bboxs, one)


_SSD_CONSTRAINTS = ((0.1, None), (0.3, None), (0.5, None), (0.7, None), (0.9, None), (None, 1))


def random_crop_with_constraints(bboxs, size, min_scale=0.3, max_scale=1, max_aspect_ratio=2, constraints=None,
                                 max_trial=50, py_rng=None, np_rng=None):
    """SSD-style constrained random crop (models/transforms/bbox.py:13-128).  Consumes python's `random`
    for the trials and numpy's global RNG for the final pick, in the reference's order, so that a fixed
    (random.seed, np.random.seed) pair reproduces the reference's crop.  py_rng / np_rng: private generators with the
    same methods (random.Random / numpy RandomState) instead of the two global ones."""
    random = py_rng if py_rng is not None else globals()['random']
    nprand = np_rng if np_rng is not None else np.random
    constraints = _SSD_CONSTRAINTS if constraints is None else constraints
    w, h = size
    single = not isinstance(bboxs, list)
    frames = [bboxs] if single else bboxs
    all_empty = all(len(b) == 0 for b in frames)
    candidates = [(0, 0, w, h)]
    for lo, hi in constraints:
        lo = -np.inf if lo is None else lo
        hi = np.inf if hi is None else hi
        for _ in range(max_trial):
            scale = random.uniform(min_scale, max_scale)
            ar = random.uniform(max(1 / max_aspect_ratio, scale * scale), min(max_aspect_ratio, 1 / (scale * scale)))
            ch = int(h * scale / np.sqrt(ar))
            cw = int(w * scale * np.sqrt(ar))
            top = random.randrange(h - ch)
            left = random.randrange(w - cw)
            if all_empty:
                return (frames[0] if single else frames), (left, top, cw, ch)
            region = np.array((left, top, left + cw, top + ch))[np.newaxis]
            ok = True
            for b in frames:
                iou = bbox_iou(b, region)
                if lo > iou.min() or iou.max() > hi:
                    ok = False
            if ok:
                candidates.append((left, top, cw, ch))
                break
    pick = candidates.pop(nprand.randint(0, len(candidates)))
    cropped = crop(frames, pick, allow_outside_center=False)
    return cropped, (pick[0], pick[1], pick[2], pick[3])
