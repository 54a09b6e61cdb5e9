"""Host-side prefetch target generator (runs on the CPU in the data pipeline, as in the reference).

Mirrors YOLOV3PrefetchTargetGenerator.forward / _slice, models/definitions/yolo/yolo_target.py:31-148,
without the reference's dummy network forward: anchors, offsets and feature-map sizes are closed form
(SURVEY.md 3.1).  Output rows follow the network's training order: scales stride 32, 16, 8; inside a
scale row = (y*w + x)*3 + a.  The dense (B, sum(g^2), 9, .) intermediate of the reference is never built:
only each scale's own 3 anchors are materialised.
"""
import numpy as np

from .consts import ANCHORS, STRIDES      # (not .model: the loader's worker processes stay NumPy-only)

_OUT_ANCHORS = np.asarray(ANCHORS[::-1], dtype=np.float64).reshape(9, 2)   # stride 32 anchors first


def prefetch_targets(height, width, gt_boxes, gt_ids, num_class, gt_mixratio=None):
    """gt_boxes (B,M,4) corner boxes padded with -1; gt_ids (B,M,1) class index or (B,M,C) multi-hot.

    Returns float32 arrays: objectness (B,P,1), center_targets (B,P,2), scale_targets (B,P,2),
    weights (B,P,2), class_targets (B,P,C)."""
    # fp32 boxes and fp32 width / height / centre (the generator's inputs are fp32 NDArrays and BBoxCornerToCenter runs on
    # them, yolo_target.py:86-87), float64 from there on (NumPy scalars in the per-gt loop, :96-133): a centre within fp32
    # rounding of a cell edge goes to the cell its rounded value names
    gt_boxes = np.asarray(gt_boxes, dtype=np.float32)
    gt_ids = np.asarray(gt_ids)
    B, M = gt_boxes.shape[:2]
    grids_h = [height // s for s in STRIDES[::-1]]
    grids_w = [width // s for s in STRIDES[::-1]]
    sizes = [gh * gw * 3 for gh, gw in zip(grids_h, grids_w)]
    base = np.concatenate([[0], np.cumsum(sizes)])
    P = int(base[-1])
    obj = np.zeros((B, P, 1), np.float32)
    ctr = np.zeros((B, P, 2), np.float32)
    scl = np.zeros((B, P, 2), np.float32)
    wgt = np.zeros((B, P, 2), np.float32)
    cls = np.full((B, P, num_class), -1.0, np.float32)
    gw = gt_boxes[..., 2] - gt_boxes[..., 0]
    gh = gt_boxes[..., 3] - gt_boxes[..., 1]
    gx = (gt_boxes[..., 0] + gw / np.float32(2.0)).astype(np.float64)
    gy = (gt_boxes[..., 1] + gh / np.float32(2.0)).astype(np.float64)
    gw, gh = gw.astype(np.float64), gh.astype(np.float64)
    # zero-centred shape IoU of every gt against the 9 anchors (yolo_target.py:88-94)
    aw, ah = _OUT_ANCHORS[:, 0][None, None], _OUT_ANCHORS[:, 1][None, None]
    inter = np.maximum(0.0, np.minimum(aw, gw[..., None])) * np.maximum(0.0, np.minimum(ah, gh[..., None]))
    union = aw * ah + (gw * gh)[..., None] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(union > 0, inter / union, 0.0)
    match = iou.argmax(axis=-1)                                  # (B,M)
    valid = (gt_boxes >= 0).all(axis=-1)
    for b in range(B):
        for m in range(M):
            if not valid[b, m]:
                break                                           # the reference stops at the first padded row (:106-107)
            a9 = int(match[b, m])
            layer, a = divmod(a9, 3)
            hh, ww = grids_h[layer], grids_w[layer]
            fx, fy = gx[b, m] / width * ww, gy[b, m] / height * hh
            lx, ly = int(fx), int(fy)
            p = int(base[layer]) + (ly * ww + lx) * 3 + a
            ctr[b, p] = (fx - lx, fy - ly)
            scl[b, p] = (np.log(max(gw[b, m], 1) / _OUT_ANCHORS[a9, 0]), np.log(max(gh[b, m], 1) / _OUT_ANCHORS[a9, 1]))
            wgt[b, p] = 2.0 - gw[b, m] * gh[b, m] / width / height
            obj[b, p, 0] = 1.0 if gt_mixratio is None else gt_mixratio[b, m, 0]
            cls[b, p] = 0.0
            if gt_ids.shape[-1] == 1:
                cls[b, p, int(gt_ids[b, m, 0])] = 1.0
            else:
                cls[b, p] = gt_ids[b, m]
    return obj, ctr, scl, wgt, cls


def synthetic_batch(batch, size, num_class, seed, max_gt=8):
    """SURVEY.md 8(d) synthetic inputs: uint8 frames -> reference normalisation; M=8 boxes per image with
    centre ~U(0.1,0.9)*W and w,h ~U(16, 0.5*W), clipped, corner format, class ~U{0..C-1}."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (batch, size, size, 3), dtype=np.uint8)
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    x = ((img.astype(np.float32) / 255.0 - mean) / std).transpose(0, 3, 1, 2).copy()
    c = rng.uniform(0.1, 0.9, (batch, max_gt, 2)) * size
    wh = rng.uniform(16, 0.5 * size, (batch, max_gt, 2))
    gt = np.concatenate([np.clip(c - wh / 2, 0, size - 1), np.clip(c + wh / 2, 0, size - 1)], axis=-1)
    ids = rng.integers(0, num_class, (batch, max_gt, 1)).astype(np.float64)
    return x, gt.astype(np.float32), ids
