"""CPU: the host box utilities (viddet_amd/bbox.py) and the oracle's IoU against golden vectors produced by
the reference's own NumPy code (tests/golden/make_golden.py, run in the build container)."""
import os
import random

import numpy as np
import pytest

from viddet_amd import bbox as B
from oracle import yolo as Y

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "bbox_golden.npz"))


def eq(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=0, atol=1e-12)


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("off", [0, 1])
def test_bbox_iou_golden(seed, off):
    a, b = G["iou_a_%d" % seed], G["iou_b_%d" % seed]
    with np.errstate(divide="ignore", invalid="ignore"):
        eq(B.bbox_iou(a, b, offset=off), G["iou_%d_off%d" % (seed, off)])
        eq(Y.bbox_iou_np(a, b, offset=off), G["iou_%d_off%d" % (seed, off)])     # the oracle's mAP IoU


def test_format_conversions_golden():
    eq(B.bbox_xywh_to_xyxy(G["xywh_in"].copy()), G["xywh_to_xyxy"])
    eq(B.bbox_xyxy_to_xywh(G["xywh_to_xyxy"].copy()), G["xyxy_to_xywh"])
    eq(np.array(B.bbox_xywh_to_xyxy((3.0, 4.0, 10.0, 0.5))), G["xywh_to_xyxy_tuple"])
    eq(np.array(B.bbox_xyxy_to_xywh((3.0, 4.0, 10.0, 20.0))), G["xyxy_to_xywh_tuple"])
    eq(B.bbox_clip_xyxy(G["clip_in"].copy(), 416, 320), G["clip_out"])
    eq(np.array(B.bbox_clip_xyxy((-5.0, 10.0, 700.0, 300.0), 416, 320)), G["clip_tuple"])
    with pytest.raises(IndexError):
        B.bbox_xywh_to_xyxy((1, 2, 3))
    with pytest.raises(TypeError):
        B.bbox_clip_xyxy("nope", 1, 1)


def _lst():
    return [G["t_list_%d" % i] for i in range(3)]


def test_resize_flip_translate_crop_golden():
    bx = G["t_single"]
    keep = bx.copy()
    eq(B.resize(bx, (400, 300), (416, 416)), G["resize_single"])
    for i, l in enumerate(B.resize(_lst(), (400, 300), (416, 416))):
        eq(l, G["resize_list_%d" % i])
    eq(B.flip(bx, (400, 300), flip_x=True), G["flip_x_single"])
    eq(B.flip(bx, (400, 300), flip_x=True, flip_y=True), G["flip_xy_single"])
    for i, l in enumerate(B.flip(_lst(), (400, 300), flip_y=True)):
        eq(l, G["flip_y_list_%d" % i])
    eq(B.translate(bx, x_offset=13, y_offset=-7), G["translate_single"])
    for i, l in enumerate(B.translate(_lst(), x_offset=-3, y_offset=5)):
        eq(l, G["translate_list_%d" % i])
    cb = tuple(int(v) for v in G["crop_box"])
    eq(B.crop(bx, cb, allow_outside_center=True), G["crop_single_outside"])
    eq(B.crop(bx, cb, allow_outside_center=False), G["crop_single_center"])
    for i, l in enumerate(B.crop(_lst(), cb, allow_outside_center=False)):
        eq(l, G["crop_list_%d" % i])
    assert np.array_equal(bx, keep), "inputs must not be modified in place"


@pytest.mark.parametrize("seed", range(3))
def test_random_crop_golden(seed):
    random.seed(seed); np.random.seed(seed)
    nb, cr = B.random_crop_with_constraints(G["t_single"], (400, 300))
    eq(np.array(nb), G["rcrop_boxes_%d" % seed])
    eq(np.array(cr), G["rcrop_crop_%d" % seed])


def test_random_crop_list_golden():
    lst = _lst()
    random.seed(5); np.random.seed(5)
    nbl, crl = B.random_crop_with_constraints([lst[0], lst[2]], (400, 300))
    for i, l in enumerate(nbl):
        eq(l, G["rcrop_list_%d" % i])
    eq(np.array(crl), G["rcrop_list_crop"])
