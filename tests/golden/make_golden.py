"""Generates tests/golden/bbox_golden.npz by importing the reference's NumPy-only modules
(utils/bbox.py, models/transforms/bbox.py) from /root/reference in the build container.

Only the arrays travel (inputs + the reference's outputs); the reference source does not.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import random
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)          # the reference's `utils` / `models` packages must win over anything else
sys.dont_write_bytecode = True

import importlib.util


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ub = _load("utils.bbox", os.path.join(REF, "utils", "bbox.py"))
sys.modules["utils.bbox"] = ub
import types
pkg = types.ModuleType("utils"); pkg.bbox = ub; sys.modules.setdefault("utils", pkg)
tb = _load("ref_transforms_bbox", os.path.join(REF, "models", "transforms", "bbox.py"))


def boxes(rng, n, size=400.0, extra=0):
    x1 = rng.uniform(0, size * 0.8, n); y1 = rng.uniform(0, size * 0.8, n)
    w = rng.uniform(1, size * 0.5, n); h = rng.uniform(1, size * 0.5, n)
    b = np.stack([x1, y1, x1 + w, y1 + h], axis=1)
    if extra:
        b = np.concatenate([b, rng.integers(0, 20, (n, extra)).astype(np.float64)], axis=1)
    return b


out = {}
for seed in range(4):
    rng = np.random.default_rng(seed)
    a, b = boxes(rng, 64), boxes(rng, 32)
    # special rows: identical, contained, disjoint, degenerate (zero area), touching
    a[0] = b[0]
    a[1] = [10, 10, 100, 100]; b[1] = [20, 20, 50, 50]
    a[2] = [0, 0, 10, 10]; b[2] = [200, 200, 210, 210]
    a[3] = [5, 5, 5, 20]
    a[4] = [0, 0, 10, 10]; b[4] = [10, 0, 20, 10]
    out["iou_a_%d" % seed], out["iou_b_%d" % seed] = a, b
    with np.errstate(divide="ignore", invalid="ignore"):
        for off in (0, 1):
            out["iou_%d_off%d" % (seed, off)] = ub.bbox_iou(a, b, offset=off)

rng = np.random.default_rng(10)
xywh = np.abs(rng.standard_normal((16, 4))) * 50
out["xywh_in"] = xywh
out["xywh_to_xyxy"] = ub.bbox_xywh_to_xyxy(xywh.copy())
out["xyxy_to_xywh"] = ub.bbox_xyxy_to_xywh(out["xywh_to_xyxy"].copy())
out["xywh_to_xyxy_tuple"] = np.array(ub.bbox_xywh_to_xyxy((3.0, 4.0, 10.0, 0.5)))
out["xyxy_to_xywh_tuple"] = np.array(ub.bbox_xyxy_to_xywh((3.0, 4.0, 10.0, 20.0)))
clip_in = rng.uniform(-50, 500, (16, 4))
out["clip_in"] = clip_in
out["clip_out"] = ub.bbox_clip_xyxy(clip_in.copy(), 416, 320)       # note: ndarray form returns a flat hstack
out["clip_tuple"] = np.array(ub.bbox_clip_xyxy((-5.0, 10.0, 700.0, 300.0), 416, 320))

bx = boxes(rng, 12, extra=2)
lst = [boxes(rng, 5, extra=1), boxes(rng, 0, extra=1).reshape(0, 5), boxes(rng, 3, extra=1)]
out["t_single"] = bx
for i, l in enumerate(lst):
    out["t_list_%d" % i] = l
out["resize_single"] = tb.resize(bx, (400, 300), (416, 416))
for i, l in enumerate(tb.resize(lst, (400, 300), (416, 416))):
    out["resize_list_%d" % i] = l
out["flip_x_single"] = tb.flip(bx, (400, 300), flip_x=True)
out["flip_xy_single"] = tb.flip(bx, (400, 300), flip_x=True, flip_y=True)
for i, l in enumerate(tb.flip(lst, (400, 300), flip_y=True)):
    out["flip_y_list_%d" % i] = l
out["translate_single"] = tb.translate(bx, x_offset=13, y_offset=-7)
for i, l in enumerate(tb.translate(lst, x_offset=-3, y_offset=5)):
    out["translate_list_%d" % i] = l
cb = (50, 40, 200, 180)
out["crop_box"] = np.array(cb)
out["crop_single_outside"] = tb.crop(bx, cb, allow_outside_center=True)
out["crop_single_center"] = tb.crop(bx, cb, allow_outside_center=False)
for i, l in enumerate(tb.crop(lst, cb, allow_outside_center=False)):
    out["crop_list_%d" % i] = l

# random_crop_with_constraints consumes python's `random` and numpy's global RNG (bbox.py:85-93,121)
for seed in range(3):
    random.seed(seed); np.random.seed(seed)
    nb, cr = tb.random_crop_with_constraints(bx, (400, 300))
    out["rcrop_boxes_%d" % seed] = nb
    out["rcrop_crop_%d" % seed] = np.array(cr)
random.seed(5); np.random.seed(5)
lst2 = [lst[0], lst[2]]     # the reference raises on an empty array inside a list (iou.min() of size 0)
nbl, crl = tb.random_crop_with_constraints(lst2, (400, 300))
for i, l in enumerate(nbl):
    out["rcrop_list_%d" % i] = l
out["rcrop_list_crop"] = np.array(crl)

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bbox_golden.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, len(out), "arrays")
