"""CPU oracle — YOLOv3 head, targets, loss, NMS and mAP.  TEST INFRASTRUCTURE ONLY (see oracle/ops.py).

PARITY STATUS: parity unpinned for box_nms / BBoxBatchIOU / YOLOV3Loss (MXNet/GluonCV internals,
SURVEY.md Appendix A.1-A.3 are the definitions used here); the pure-Python control flow of the
reference files cited per function is followed line by line.
"""
import numpy as np

from .ops import sigmoid

# models/definitions/yolo/wrappers.py:80-84
ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
STRIDES = [8, 16, 32]
# yolo3.py:1013-1014: outputs are built from anchors[::-1], strides[::-1] (stride 32 first)
OUT_ANCHORS = ANCHORS[::-1]
OUT_STRIDES = STRIDES[::-1]


# ---------------------------------------------------------------------------------------------
# YOLOOutputV3.hybrid_forward   models/definitions/yolo/yolo3.py:132-199
# ---------------------------------------------------------------------------------------------
def yolo_output(pred, num_class, anchors, stride, training):
    """pred: prediction-conv output (B, 3*(5+C), H, W).

    training -> (bbox (B,HW*3,4), raw_centers (B,HW,3,2), raw_scales, objness (..,1), class_pred (..,C))
    inference -> detections (B, C*HW*3, 6) rows [id, score, x1, y1, x2, y2] ordered [class][pixel][anchor]
    """
    b, _, h, w = pred.shape
    na, npred = 3, 5 + num_class
    p = pred.reshape(b, na * npred, h * w)                    # :158
    p = p.transpose(0, 2, 1).reshape(b, h * w, na, npred)     # :160
    raw_centers, raw_scales = p[..., 0:2], p[..., 2:4]
    objness, class_pred = p[..., 4:5], p[..., 5:]
    gx, gy = np.meshgrid(np.arange(w), np.arange(h))          # :67-74 offsets = (x, y)
    offsets = np.stack([gx, gy], axis=-1).reshape(1, h * w, 1, 2).astype(p.dtype)
    anc = np.asarray(anchors, dtype=p.dtype).reshape(1, 1, na, 2)
    centers = (sigmoid(raw_centers) + offsets) * stride       # :172
    scales = np.exp(raw_scales) * anc                         # :173
    conf = sigmoid(objness)
    class_score = sigmoid(class_pred) * conf                  # :175
    wh = scales / 2.0
    bbox = np.concatenate([centers - wh, centers + wh], axis=-1)   # (B,HW,3,4)
    if training:
        return bbox.reshape(b, -1, 4), raw_centers, raw_scales, objness, class_pred
    # :191-197  per-class rows
    bboxes = np.broadcast_to(bbox[None], (num_class,) + bbox.shape)
    scores = class_score.transpose(3, 0, 1, 2)[..., None]          # (C,B,HW,3,1)
    ids = np.broadcast_to(np.arange(num_class, dtype=p.dtype).reshape(-1, 1, 1, 1, 1), scores.shape)
    det = np.concatenate([ids, scores, bboxes], axis=-1)           # (C,B,HW,3,6)
    return det.transpose(1, 0, 2, 3, 4).reshape(b, -1, 6)


# ---------------------------------------------------------------------------------------------
# F.contrib.box_nms(overlap_thresh, valid_thresh=0.01, topk, id_index=0, score_index=1,
#                   coord_start=2, force_suppress=False)  yolo3.py:1197-1200   
# ---------------------------------------------------------------------------------------------
def box_nms(data, overlap_thresh=0.45, valid_thresh=0.01, topk=400):
    """data (B,N,6).  Returns (out (B,N,6) with -1 fill, kept_rows list per image = original row indices)."""
    out = np.full_like(data, -1.0)
    kept_rows = []
    for b in range(data.shape[0]):
        d = data[b]
        valid = np.nonzero(d[:, 1] > valid_thresh)[0]
        order = valid[np.argsort(-d[valid, 1], kind='stable')]     # score desc, stable in row order
        if topk > 0:
            order = order[:topk]
        boxes = d[order, 2:6]
        ids = d[order, 0]
        area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        alive = np.ones(len(order), dtype=bool)
        for i in range(len(order)):
            if not alive[i]:
                continue
            j = np.arange(i + 1, len(order))
            j = j[alive[j] & (ids[j] == ids[i])]
            if len(j) == 0:
                continue
            iw = np.maximum(0.0, np.minimum(boxes[i, 2], boxes[j, 2]) - np.maximum(boxes[i, 0], boxes[j, 0]))
            ih = np.maximum(0.0, np.minimum(boxes[i, 3], boxes[j, 3]) - np.maximum(boxes[i, 1], boxes[j, 1]))
            inter = iw * ih
            union = area[i] + area[j] - inter
            iou = np.where(union <= 0, 0.0, inter / np.where(union <= 0, 1.0, union))
            alive[j[iou > overlap_thresh]] = False
        keep = order[alive]
        out[b, :len(keep)] = d[keep]
        kept_rows.append(keep)
    return out, kept_rows


def detect_postprocess(dets_per_scale, nms_thresh=0.45, nms_topk=400, post_nms=100):
    """yolo3.py:1195-1206: concat scales, nms, slice -> ids (B,100,1), scores (B,100,1), bboxes (B,100,4), rows"""
    result = np.concatenate(dets_per_scale, axis=1)
    out, kept = box_nms(result, nms_thresh, 0.01, nms_topk)
    out = out[:, :post_nms]
    rows = np.full((out.shape[0], post_nms), -1, dtype=np.int64)
    for b, k in enumerate(kept):
        k = k[:post_nms]
        rows[b, :len(k)] = k
    return out[..., 0:1], out[..., 1:2], out[..., 2:6], rows


# ---------------------------------------------------------------------------------------------
# YOLOV3PrefetchTargetGenerator.forward   models/definitions/yolo/yolo_target.py:31-148
# ---------------------------------------------------------------------------------------------
def _shape_iou(anchors_wh, gtw, gth):
    """nd.contrib.box_iou of zero-centred anchor boxes vs zero-centred gt boxes (yolo_target.py:88-92)."""
    aw, ah = anchors_wh[:, 0][:, None], anchors_wh[:, 1][:, None]
    iw = np.maximum(0.0, np.minimum(aw, gtw[None]) )
    ih = np.maximum(0.0, np.minimum(ah, gth[None]))
    # both centred at 0: overlap extent = min(w) x min(h)
    inter = iw * ih
    union = aw * ah + (gtw * gth)[None] - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1.0), 0.0)


def prefetch_targets(img_h, img_w, grids, gt_boxes, gt_ids, num_class, gt_mixratio=None):
    """gt_boxes (B,M,4) corner, padded -1; gt_ids (B,M,1) class index (or (B,M,C) multi-hot).

    grids: feature-map sides in the reference's xs order = output order (stride 32, 16, 8).
    Returns objectness (B,P,1), center_t (B,P,2), scale_t (B,P,2), weights (B,P,2), class_t (B,P,C),
    P ordered as the network's training outputs ((y*w+x)*3+a per scale, stride 32 first).
    """
    b, m = gt_boxes.shape[0], gt_boxes.shape[1]
    all_anchors = np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1, 2) for a in OUT_ANCHORS], axis=0)  # :62
    num_anchors = np.cumsum([3, 3, 3])
    n_off = [g * g for g in grids]
    num_offsets = np.cumsum(n_off)
    _offsets = [0] + num_offsets.tolist()
    tot = int(num_offsets[-1])
    center_t = np.zeros((b, tot, 9, 2))
    scale_t = np.zeros_like(center_t)
    weights = np.zeros_like(center_t)
    objectness = np.zeros((b, tot, 9, 1))
    class_t = np.full((b, tot, 9, num_class), -1.0)                      # :83
    # The generator receives fp32 NDArrays (transforms.py:258 mx.nd.array -> float32) and BBoxCornerToCenter runs on them
    # (yolo_target.py:86-87): box coordinates are ROUNDED TO fp32 and width / height / centre are fp32 arithmetic
    # (w = x2 - x1, x = x1 + w / 2).  The per-gt loop then works on NumPy scalars taken from .asnumpy() (:96-101) with
    # Python ints, which NumPy 1.x promotes to float64 [UPSTREAM-UNVERIFIED: the NumPy of the MXNet 1.4-1.5 era; NEP 50 /
    # NumPy 2 would keep float32], so the cell index int(gtx / orig_width * width) is a float64 quotient of an fp32
    # centre.  A centre within fp32 rounding of a cell edge therefore lands where its ROUNDED value says.
    gt_boxes = np.asarray(gt_boxes, dtype=np.float32)
    gtw = gt_boxes[..., 2] - gt_boxes[..., 0]
    gth = gt_boxes[..., 3] - gt_boxes[..., 1]
    gtx = (gt_boxes[..., 0] + gtw / np.float32(2.0)).astype(np.float64)
    gty = (gt_boxes[..., 1] + gth / np.float32(2.0)).astype(np.float64)
    gtw, gth = gtw.astype(np.float64), gth.astype(np.float64)
    for bi in range(b):
        ious = _shape_iou(all_anchors, gtw[bi], gth[bi])                  # (9, M)
        matches = ious.argmax(axis=0)                                    # :94
        for mi in range(m):
            if np.prod(gt_boxes[bi, mi] >= 0) < 1:                       # :95,106-107
                break
            match = int(matches[mi])
            nlayer = int(np.nonzero(num_anchors > match)[0][0])
            height = width = grids[nlayer]
            x, y, w_, h_ = gtx[bi, mi], gty[bi, mi], gtw[bi, mi], gth[bi, mi]
            loc_x = int(x / img_w * width)
            loc_y = int(y / img_h * height)
            index = _offsets[nlayer] + loc_y * width + loc_x
            center_t[bi, index, match, 0] = x / img_w * width - loc_x
            center_t[bi, index, match, 1] = y / img_h * height - loc_y
            scale_t[bi, index, match, 0] = np.log(max(w_, 1) / all_anchors[match, 0])
            scale_t[bi, index, match, 1] = np.log(max(h_, 1) / all_anchors[match, 1])
            weights[bi, index, match, :] = 2.0 - w_ * h_ / img_w / img_h
            objectness[bi, index, match, 0] = gt_mixratio[bi, mi, 0] if gt_mixratio is not None else 1
            class_t[bi, index, match, :] = 0
            if gt_ids.shape[-1] == 1:
                class_t[bi, index, match, int(gt_ids[bi, mi, 0])] = 1
            else:
                class_t[bi, index, match, :] = gt_ids[bi, mi, :]

    def _slice(x):                                                       # :139-148
        anchors = [0] + num_anchors.tolist()
        offs = [0] + num_offsets.tolist()
        ret = []
        for i in range(3):
            yv = x[:, offs[i]:offs[i + 1], anchors[i]:anchors[i + 1], :]
            ret.append(yv.reshape(b, -1, x.shape[-1]))
        return np.concatenate(ret, axis=1)

    return _slice(objectness), _slice(center_t), _slice(scale_t), _slice(weights), _slice(class_t)


# ---------------------------------------------------------------------------------------------
# gluoncv.nn.bbox.BBoxBatchIOU ; YOLOV3DynamicTargetGeneratorSimple yolo_target.py:173-205
# ---------------------------------------------------------------------------------------------
def bbox_batch_iou(a, b, eps=1e-15):
    al, at, ar, ab = [a[..., i:i + 1] for i in range(4)]                 # (B,N,1)
    bl, bt, br, bb = [b[..., i][:, None, :] for i in range(4)]           # (B,1,M)
    iw = np.clip(np.minimum(ar, br) - np.maximum(al, bl), 0, 6.55040e+04)
    ih = np.clip(np.minimum(ab, bb) - np.maximum(at, bt), 0, 6.55040e+04)
    i = iw * ih
    area_a = (ar - al) * (ab - at)
    area_b = (br - bl) * (bb - bt)
    return i / (area_a + area_b - i + eps)


def dynamic_targets(box_preds, gt_boxes, ignore_iou_thresh=0.7):
    ious = bbox_batch_iou(box_preds, gt_boxes)
    ious_max = ious.max(axis=-1, keepdims=True)
    return (ious_max > ignore_iou_thresh) * -1.0                         # :204


# YOLOV3TargetMerger.hybrid_forward   yolo_target.py:226-281
def merge_targets(box_preds, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t, num_class,
                  ignore_iou_thresh=0.7, label_smooth=False):
    dyn_obj = dynamic_targets(box_preds, gt_boxes, ignore_iou_thresh)
    mask = obj_t > 0
    objectness = np.where(mask, obj_t, dyn_obj)
    mask2 = np.tile(mask, (1, 1, 2))
    center_targets = np.where(mask2, centers_t, 0.0)
    scale_targets = np.where(mask2, scales_t, 0.0)
    weights = np.where(mask2, weights_t, 0.0)
    mask3 = np.tile(mask, (1, 1, num_class))
    class_targets = np.where(mask3, clas_t, -1.0)
    if label_smooth:
        smooth_weight = min(1.0 / num_class, 1.0 / 40)
        class_targets = np.where(class_targets > 0.5, class_targets - smooth_weight, class_targets)
        class_targets = np.where((class_targets < -0.5) | (class_targets > 0.5), class_targets, smooth_weight)
    class_mask = mask3 * (class_targets >= 0)
    return objectness, center_targets, scale_targets, weights, class_targets, class_mask.astype(np.float64)


# ---------------------------------------------------------------------------------------------
# gluoncv.loss.YOLOV3Loss — constructed yolo3.py:994, called :1187
# ---------------------------------------------------------------------------------------------
def _bce_logits(x, z):
    return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))


def yolo3_loss(objness, box_centers, box_scales, cls_preds, objness_t, center_t, scale_t, weight_t, class_t,
               class_mask, with_grads=False):
    """All inputs (B,P,k).  Returns the 4 per-sample losses (B,), and (optionally) d(sum of all)/d(logits)."""
    denorm = float(np.prod(objness_t.shape[1:]))
    weight_t = weight_t * objness_t
    hard_objness_t = np.where(objness_t > 0, 1.0, objness_t)
    new_objness_mask = np.where(objness_t > 0, objness_t, (objness_t >= 0).astype(objness_t.dtype))

    def mean_nb(v):
        return v.reshape(v.shape[0], -1).mean(axis=1)

    obj_loss = mean_nb(_bce_logits(objness, hard_objness_t) * new_objness_mask) * denorm
    center_loss = mean_nb(_bce_logits(box_centers, center_t) * weight_t) * denorm * 2
    scale_loss = mean_nb(np.abs(box_scales - scale_t) * weight_t) * denorm * 2
    denorm_class = float(np.prod(class_t.shape[1:]))
    class_mask = class_mask * objness_t
    cls_loss = mean_nb(_bce_logits(cls_preds, class_t) * class_mask) * denorm_class
    if not with_grads:
        return obj_loss, center_loss, scale_loss, cls_loss
    # every loss is a plain masked sum, so d/dlogit is closed form
    g_obj = (sigmoid(objness) - hard_objness_t) * new_objness_mask
    g_ctr = (sigmoid(box_centers) - center_t) * weight_t
    g_scl = np.sign(box_scales - scale_t) * weight_t
    g_cls = (sigmoid(cls_preds) - class_t) * class_mask
    return (obj_loss, center_loss, scale_loss, cls_loss), (g_obj, g_ctr, g_scl, g_cls)


# ---------------------------------------------------------------------------------------------
# VOCMApMetric   metrics/pascalvoc.py:85-259
# ---------------------------------------------------------------------------------------------
def bbox_iou_np(a, b, offset=0):
    """utils/bbox.py:11-38 (vendored copy of gluoncv.utils.bbox.bbox_iou)."""
    tl = np.maximum(a[:, None, :2], b[:, :2])
    br = np.minimum(a[:, None, 2:4], b[:, 2:4])
    area_i = np.prod(br - tl + offset, axis=2) * (tl < br).all(axis=2)
    area_a = np.prod(a[:, 2:4] - a[:, :2] + offset, axis=1)
    area_b = np.prod(b[:, 2:4] - b[:, :2] + offset, axis=1)
    return area_i / (area_a[:, None] + area_b - area_i)


class VOCMApMetric:
    """metrics/pascalvoc.py:14-259: area-under-PR AP per class, mAP = nanmean."""

    def __init__(self, iou_thresh=0.5, class_names=None, class_map=None):
        self.num = len(class_names)
        self.iou_thresh = iou_thresh
        self.class_map = class_map                  # pascalvoc.py:37
        self.reset()

    def reset(self):
        self._n_pos = {}
        self._score = {}
        self._match = {}

    def update(self, pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels, gt_difficults=None):
        if gt_difficults is None:
            gt_difficults = [None for _ in gt_labels]
        for pred_bbox, pred_label, pred_score, gt_bbox, gt_label, gt_difficult in zip(
                pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels, gt_difficults):
            valid_pred = np.where(pred_label.flat >= 0)[0]                 # :116-119
            pred_bbox = pred_bbox[valid_pred, :]
            pred_label = pred_label.flat[valid_pred].astype(int)
            pred_score = pred_score.flat[valid_pred]
            if self.class_map is not None:                                  # :126-127
                gt_label = np.expand_dims(np.array([self.class_map[int(l)] for l in gt_label.flat]), axis=0)
            valid_gt = np.where(gt_label.flat >= 0)[0]
            gt_bbox = gt_bbox[valid_gt, :]
            gt_label = gt_label.flat[valid_gt].astype(int)
            if gt_difficult is None:
                gt_difficult = np.zeros(gt_bbox.shape[0])
            else:
                gt_difficult = gt_difficult.flat[valid_gt]
            for l in np.unique(np.concatenate((pred_label, gt_label)).astype(int)):
                pred_mask_l = pred_label == l
                pred_bbox_l = pred_bbox[pred_mask_l]
                pred_score_l = pred_score[pred_mask_l]
                order = pred_score_l.argsort()[::-1]                        # :137-139
                pred_bbox_l = pred_bbox_l[order]
                pred_score_l = pred_score_l[order]
                gt_mask_l = gt_label == l
                gt_bbox_l = gt_bbox[gt_mask_l]
                gt_difficult_l = gt_difficult[gt_mask_l]
                self._n_pos[l] = self._n_pos.get(l, 0) + int(np.logical_not(gt_difficult_l).sum())
                self._score.setdefault(l, []).extend(pred_score_l)
                m = self._match.setdefault(l, [])
                if len(pred_bbox_l) == 0:
                    continue
                if len(gt_bbox_l) == 0:
                    m.extend((0,) * pred_bbox_l.shape[0])
                    continue
                iou = bbox_iou_np(pred_bbox_l, gt_bbox_l)                   # no +1 offset (:162-164)
                gt_index = iou.argmax(axis=1)
                gt_index[iou.max(axis=1) < self.iou_thresh] = -1
                selec = np.zeros(gt_bbox_l.shape[0], dtype=bool)
                for gt_idx in gt_index:
                    if gt_idx >= 0:
                        if gt_difficult_l[gt_idx]:
                            m.append(-1)
                        else:
                            m.append(0 if selec[gt_idx] else 1)
                        selec[gt_idx] = True
                    else:
                        m.append(0)

    def _recall_prec(self):
        n_fg_class = max(self._n_pos.keys()) + 1 if self._n_pos else 0
        prec, rec = [None] * n_fg_class, [None] * n_fg_class
        for l in self._n_pos.keys():
            score_l = np.array(self._score[l])
            match_l = np.array(self._match[l], dtype=np.int32)
            order = score_l.argsort()[::-1]
            match_l = match_l[order]
            tp = np.cumsum(match_l == 1)
            fp = np.cumsum(match_l == 0)
            with np.errstate(divide='ignore', invalid='ignore'):
                prec[l] = tp / (fp + tp)
            if self._n_pos[l] > 0:
                rec[l] = tp / self._n_pos[l]
        return rec, prec

    @staticmethod
    def _average_precision(rec, prec):
        if rec is None or prec is None:
            return np.nan
        mrec = np.concatenate(([0.], rec, [1.]))
        mpre = np.concatenate(([0.], np.nan_to_num(prec), [0.]))
        for i in range(mpre.size - 1, 0, -1):
            mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
        i = np.where(mrec[1:] != mrec[:-1])[0]
        return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))

    def get(self):
        rec, prec = self._recall_prec()
        aps = [self._average_precision(r, p) for r, p in zip(rec, prec)]
        while len(aps) < self.num:
            aps.append(np.nan)
        m_ap = float(np.nanmean(aps)) if len(aps) else float('nan')
        if self.class_map:                                                  # :71-80
            aps = [float('nan') if self.class_map[i] < 0 else aps[self.class_map[i]] for i in range(self.num)]
        return aps, m_ap


class VOCMApMetricTemporal:
    """metrics/pascalvoc.py:262-560 keeps per-offset copies (self._n_pos[t], self._score[t], self._match[t]) of
    exactly the bookkeeping of VOCMApMetric (:381-450 repeat :116-184 with a [t] index); restated as t accumulators."""

    def __init__(self, t, iou_thresh=0.5, class_names=None):
        self.t = t
        self.m = [VOCMApMetric(iou_thresh, class_names) for _ in range(t)]

    def update(self, pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels, gt_difficults=None):
        for i in range(len(pred_bboxes)):
            for t in range(np.asarray(pred_bboxes[i]).shape[0]):
                gd = None if gt_difficults is None or gt_difficults[i] is None else [np.asarray(gt_difficults[i])[t]]
                self.m[t].update([np.asarray(pred_bboxes[i])[t]], [np.asarray(pred_labels[i])[t]],
                                 [np.asarray(pred_scores[i])[t]], [np.asarray(gt_bboxes[i])[t]],
                                 [np.asarray(gt_labels[i])[t]], gd)

    def get(self):
        return [m.get() for m in self.m]
