"""Class-hierarchy post-processing of saved detections: the WordNet-style class tree of the
combined dataset, the class map between two label sets and the hierarchical NMS of detect_yolo3.py.

Follows, under /root/reference:
  datasets/combined.py:97-156   generate_branches / get_levels / get_leaves / on_branch
  detect_yolo3.py:698-709       get_class_map
  detect_yolo3.py:712-733       iou (single pair, +1 pixel convention)
  detect_yolo3.py:736-789       hierarchical_nms
Host-side NumPy; predictions are the dict {image path: [[cls, conf, x1, y1, x2, y2], ...]} that detect() returns and
save_predictions() writes.
"""
import numpy as np

ROOT = "ROOT"


class ClassTree:
    """wn_classes: list of class ids in label order; parents: {class id: parent id or 'ROOT'} (may also hold ids that
    are not labels).  Mirrors the attributes hierarchical_nms reads from the combined dataset."""

    def __init__(self, wn_classes, parents):
        self.wn_classes = list(wn_classes)
        self.parents = dict(parents)
        self._index = {c: i for i, c in enumerate(self.wn_classes)}
        self.class_levels = self.get_levels()
        self.leaves = self.get_leaves()
        self.branches, self.branches_ind = self.generate_branches()

    def generate_branches(self):
        """combined.py:97-115: root-to-class lineage of every label (the walk stops below 'ROOT')."""
        branches = {}
        for c in self.wn_classes:
            line, p = [c], c
            while p in self.parents and self.parents[p] != ROOT:
                p = self.parents[p]
                line.append(p)
            branches[c] = line[::-1]
        ind = {self._index[c]: [self._index[a] for a in branches[c]] for c in self.wn_classes}
        return branches, ind

    def get_levels(self):
        """combined.py:117-126: number of edges up to 'ROOT'."""
        out = []
        for c in self.wn_classes:
            lvl, p = 0, c
            while p != ROOT:
                p = self.parents[p]
                lvl += 1
            out.append(lvl)
        return out

    def get_leaves(self):
        """combined.py:128-141: 1 for labels that are nobody's parent."""
        is_parent = {self.parents[c] for c in self.wn_classes}
        return [0 if c in is_parent else 1 for c in self.wn_classes]

    def on_branch(self, c1, c2):
        """combined.py:143-150: same lineage?  As in the reference the LOWER label index is taken as the candidate
        ancestor (labels are ordered parents first)."""
        if c1 == c2:
            return True
        child, parent = max(c1, c2), min(c1, c2)
        return parent in self.branches_ind[child]

    def parent_index(self, cls):
        return self._index[self.parents[self.wn_classes[cls]]]


def get_class_map(trained_on_wn, eval_on_wn):
    """detect_yolo3.py:698-709: for every evaluation class its index in the training label set, or -1."""
    pos = {}
    for i, c in enumerate(trained_on_wn):
        pos.setdefault(c, i)                      # list.index semantics: first occurrence
    return [pos.get(c, -1) for c in eval_on_wn]


def iou(bb, bbgt):
    """detect_yolo3.py:712-733: IoU of two corner boxes with the +1 pixel convention; 0 when they do not overlap."""
    iw = min(bb[2], bbgt[2]) - max(bb[0], bbgt[0]) + 1
    ih = min(bb[3], bbgt[3]) - max(bb[1], bbgt[1]) + 1
    if iw <= 0 or ih <= 0:
        return 0
    inter = iw * ih
    union = (bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) + (bbgt[2] - bbgt[0] + 1.) * (bbgt[3] - bbgt[1] + 1.) - inter
    return inter / union


def _iou_many(box, kept):
    """iou(box, k) for every row k of kept [n,4] (same arithmetic as iou(), vectorised)."""
    iw = np.minimum(box[2], kept[:, 2]) - np.maximum(box[0], kept[:, 0]) + 1
    ih = np.minimum(box[3], kept[:, 3]) - np.maximum(box[1], kept[:, 1]) + 1
    inter = iw * ih
    union = (box[2] - box[0] + 1.) * (box[3] - box[1] + 1.) + (kept[:, 2] - kept[:, 0] + 1.) * (kept[:, 3] - kept[:, 1] + 1.) - inter
    return np.where((iw > 0) & (ih > 0), inter / union, 0.0)


def hierarchical_nms(predictions, dataset, ov_thresh=0.5, conf_thresh=0.0, level_thresh=10):
    """detect_yolo3.py:736-789.  Per image, boxes are visited from the highest class index (the most leaf-like) down;
    a class deeper than `level_thresh` is lifted to its ancestor at that level; a box that overlaps (IoU > ov_thresh,
    the best such) an already kept box of the SAME lineage is absorbed - if the classes are equal the kept confidence
    becomes the max - and otherwise it is kept as a new box.  `dataset` is a ClassTree (or the combined dataset)."""
    levels = dataset.get_levels()
    parents, cls_map = dataset.parents, dataset.wn_classes
    index = {c: i for i, c in enumerate(cls_map)}
    level_thresh = max(0, level_thresh)
    out = {}
    for img_path, boxes in predictions.items():
        kept, kept_xy = [], np.zeros((0, 4))
        # stable, descending by class index (python's sorted(reverse=True) keeps the order of equal keys)
        for box in sorted(boxes, key=lambda b: b[0], reverse=True):
            cls, conf, coords = box[0], box[1], list(box[2:])
            if conf < conf_thresh:
                continue
            while levels[cls] > level_thresh:
                up = parents[cls_map[cls]]
                if up not in index:      # e.g. level_thresh = 0: no label sits at level 0; the reference's list.index raises too
                    raise ValueError("%r is not in list" % (up,))
                cls = index[up]
            hit = -1
            if kept:
                ov = _iou_many(np.asarray(coords, dtype=np.float64), kept_xy)
                cand = np.where(ov > ov_thresh)[0]
                if cand.size:
                    hit = int(cand[np.argmax(ov[cand])])           # first of the largest overlaps
            if hit < 0 or not dataset.on_branch(cls, kept[hit][0]):
                kept.append([cls, conf] + coords)
                kept_xy = np.vstack([kept_xy, np.asarray(coords, dtype=np.float64)[None]])
            elif cls == kept[hit][0]:
                kept[hit][1] = max(kept[hit][1], conf)
        out[img_path] = kept
    return out
