"""Host-side evaluation metrics with the reference's interface.

Mirrors metrics/pascalvoc.py:14-259 (VOCMApMetric: update / get / reset; area-under-PR AP, IoU >= 0.5
match to the arg-max ground truth, no +1 pixel offset — the `+= 1` is commented out at :162-164) and
mx.metric.Loss as used at train_yolov3.py:537-540,637-640 (running mean of per-sample losses).
"""
import numpy as np

from .bbox import bbox_iou


class LossMetric:
    """mx.metric.Loss: update(_, [losses]) accumulates sum and count; get() -> (name, mean)."""

    def __init__(self, name):
        self.name = name
        self.reset()

    def reset(self):
        self.sum_metric, self.num_inst, self._dev = 0.0, 0, None

    def update(self, _, preds):
        """Device tensors are summed ON the device (fp32 per call like ndarray.sum, fp64 across calls) and only read in
        get(): the training loop updates after every step, as the reference does, without a host round trip per step."""
        for p in preds if isinstance(preds, (list, tuple)) else [preds]:
            if hasattr(p, "detach") and p.is_cuda:
                s = p.detach().sum().double()
                self._dev = s if self._dev is None else self._dev + s
                self.num_inst += int(p.numel())
                continue
            a = p.detach().cpu().numpy() if hasattr(p, "detach") else np.asarray(p)
            self.sum_metric += float(a.sum())
            self.num_inst += int(a.size)

    def get(self):
        if self._dev is not None:
            self.sum_metric += float(self._dev)
            self._dev = None
        return self.name, (self.sum_metric / self.num_inst if self.num_inst else float("nan"))


class VOCMApMetric:
    def __init__(self, iou_thresh=0.5, class_names=None, class_map=None):
        """class_map (pascalvoc.py:23,37): for a model trained on other labels than the evaluation set's -
        class_map[eval class] = model class or -1 (detect_yolo3.py get_class_map)."""
        if class_names is None:
            raise ValueError("class_names is required")
        self.class_names = list(class_names)
        self.num = len(self.class_names)
        self.iou_thresh = iou_thresh
        self.class_map = class_map
        self.reset()

    def reset(self):
        # keyed by label id: with a class_map the model's label set may be larger than the evaluation set's
        from collections import defaultdict
        self._npos = defaultdict(int)
        self._scores = defaultdict(list)
        self._hits = defaultdict(list)                  # 1 = true positive, 0 = false positive, -1 = ignored

    def update(self, pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels, gt_difficults=None):
        """Arguments are lists (or batched arrays) over images, as validate() passes them (train_yolov3.py:487)."""
        n = len(pred_bboxes)
        if gt_difficults is None:
            gt_difficults = [None] * n
        for i in range(n):
            pb, pl, ps = np.asarray(pred_bboxes[i]), np.asarray(pred_labels[i]).reshape(-1), np.asarray(pred_scores[i]).reshape(-1)
            gb, gl = np.asarray(gt_bboxes[i]), np.asarray(gt_labels[i]).reshape(-1)
            if self.class_map is not None:            # pascalvoc.py:126-127: ground-truth ids -> model ids (before the -1 strip)
                gl = np.array([self.class_map[int(l)] for l in gl])
            keep_p, keep_g = pl >= 0, gl >= 0
            pb, pl, ps = pb.reshape(-1, 4)[keep_p], pl[keep_p].astype(int), ps[keep_p]
            gb, gl = gb.reshape(-1, 4)[keep_g], gl[keep_g].astype(int)
            gd = np.zeros(len(gl)) if gt_difficults[i] is None else np.asarray(gt_difficults[i]).reshape(-1)[keep_g]
            for c in np.union1d(pl, gl):
                sel_p, sel_g = pl == c, gl == c
                order = np.argsort(-ps[sel_p], kind="stable")     # pascalvoc.py:137 sorts score-descending
                boxes, scores = pb[sel_p][order], ps[sel_p][order]
                gboxes, gdiff = gb[sel_g], gd[sel_g].astype(bool)
                self._npos[c] += int((~gdiff).sum())
                self._scores[c].extend(scores.tolist())
                if len(boxes) == 0:
                    continue
                if len(gboxes) == 0:
                    self._hits[c].extend([0] * len(boxes))
                    continue
                iou = bbox_iou(boxes, gboxes)
                best = iou.argmax(axis=1)
                best[iou.max(axis=1) < self.iou_thresh] = -1
                taken = np.zeros(len(gboxes), dtype=bool)
                for g in best:
                    if g < 0:
                        self._hits[c].append(0)
                    elif gdiff[g]:
                        self._hits[c].append(-1)
                        taken[g] = True
                    else:
                        self._hits[c].append(0 if taken[g] else 1)
                        taken[g] = True

    def _ap(self, c):
        if c not in self._npos or (self._npos[c] == 0 and not self._scores[c]):
            return np.nan
        s = np.asarray(self._scores[c])
        h = np.asarray(self._hits[c], dtype=np.int64)[np.argsort(-s, kind="stable")] if len(s) else np.zeros(0, np.int64)
        tp, fp = np.cumsum(h == 1), np.cumsum(h == 0)
        if self._npos[c] == 0:
            return np.nan
        with np.errstate(divide="ignore", invalid="ignore"):
            prec = np.nan_to_num(tp / (tp + fp))
        rec = tp / self._npos[c]
        # area under the monotone precision envelope (pascalvoc.py:229-259)
        mrec = np.concatenate([[0.0], rec, [1.0]])
        mpre = np.concatenate([[0.0], prec, [0.0]])
        mpre = np.maximum.accumulate(mpre[::-1])[::-1]
        idx = np.nonzero(mrec[1:] != mrec[:-1])[0]
        return float(((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]).sum())

    def get(self):
        """-> (names + ['mAP'], values) as the reference returns (pascalvoc.py:51-66)."""
        n_lab = max([self.num] + [c + 1 for c in self._npos])      # pascalvoc.py:205: every label id seen takes part
        aps = [self._ap(c) for c in range(n_lab)]
        m_ap = float(np.nanmean(aps)) if np.any(~np.isnan(aps)) else float("nan")
        if self.class_map:
            # pascalvoc.py:71-80: report per evaluation class, mAP untouched.  (For a model class id >= the number of
            # evaluation names the reference indexes its fixed-size sum_metric list - the mAP slot or an IndexError;
            # here the mapped class's AP is reported.)
            vals = [float("nan") if self.class_map[i] < 0 else aps[self.class_map[i]] for i in range(self.num)]
        else:
            vals = aps[:self.num]
        return self.class_names + ["mAP"], vals + [m_ap]


class VOCMApMetricTemporal:
    """metrics/pascalvoc.py:262-560: one VOC mAP accumulator per frame offset of a t-frame window (--mult_out).
    update() takes (B, t, N, .) predictions and (B, t, M, .) ground truth; get() lists, offset by offset,
    '<class> t=<offset>/<t>' ... 'mAP t=<offset>/<t>' with their values."""

    def __init__(self, t, iou_thresh=0.5, class_names=None, class_map=None):
        self.t = int(t)
        self._per_t = [VOCMApMetric(iou_thresh, class_names, class_map) for _ in range(self.t)]
        self.class_names = self._per_t[0].class_names

    def reset(self):
        for m in self._per_t:
            m.reset()

    def update(self, pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels, gt_difficults=None):
        n = len(pred_bboxes)
        for i in range(n):
            pb, pl, ps = np.asarray(pred_bboxes[i]), np.asarray(pred_labels[i]), np.asarray(pred_scores[i])
            gb, gl = np.asarray(gt_bboxes[i]), np.asarray(gt_labels[i])
            gd = None if gt_difficults is None or gt_difficults[i] is None else np.asarray(gt_difficults[i])
            for t in range(pb.shape[0]):                                     # :381 for t in range(pred_bbox_t.shape[0])
                self._per_t[t].update([pb[t]], [pl[t]], [ps[t]], [gb[t]], [gl[t]], None if gd is None else [gd[t]])

    def get(self):
        names, values = [], []
        for t, m in enumerate(self._per_t):
            nm, vals = m.get()
            names += ["%s t=%d/%d" % (x, t, self.t) for x in nm]                # :326
            values += list(vals)
        return names, values


def update_metric_sharded(metric, records, group=None):
    """Validation under frame-batch data parallelism: every rank evaluated its shard of the validation set; the
    reference evaluates the WHOLE set in one process (train_yolov3.py:434-489).  `records` = this rank's
    [(sample index, det_bboxes, det_ids, det_scores, gt_bboxes, gt_ids, gt_difficults or None)], one per image.  All
    ranks exchange their records (host objects, a few KB per image) and update `metric` with every image in sample
    order, so each rank holds the single-process metric - the mAP that picks *_best.params is the full-set mAP."""
    from . import dist as vdist
    merged = [r for part in vdist.all_gather_objects(list(records), group) for r in part]
    merged.sort(key=lambda r: int(r[0]))
    for _, pb, pl, ps, gb, gl, gd in merged:
        metric.update([pb], [pl], [ps], [gb], [gl], None if gd is None else [gd])
    return len(merged)
