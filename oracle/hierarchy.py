"""CPU oracle — class-tree helpers and hierarchical NMS.  TEST INFRASTRUCTURE ONLY.
Parity status: UNPINNED by reference fixtures (the reference has no tests for this path and its module imports
mxnet/absl at load time, so it cannot be imported here); the functions below restate the cited lines one by one
in plain Python loops, and tests/test_hierarchy_cpu.py adds hand-computed known answers.

  /root/reference/datasets/combined.py:97-156     branches / levels / on_branch
  /root/reference/detect_yolo3.py:698-709         get_class_map
  /root/reference/detect_yolo3.py:712-733         iou
  /root/reference/detect_yolo3.py:736-789         hierarchical_nms
"""


class Tree:
    def __init__(self, wn_classes, parents):
        self.wn_classes, self.parents = list(wn_classes), dict(parents)
        # combined.py:97-115
        branches = {}
        for cls in self.wn_classes:
            cls_o, branch = cls, [cls]
            while cls in self.parents:
                if self.parents[cls] == 'ROOT':
                    break
                cls = self.parents[cls]
                branch.append(cls)
            branch.reverse()
            branches[cls_o] = branch
        self.branches_ind = {self.wn_classes.index(c): [self.wn_classes.index(a) for a in branches[c]]
                             for c in self.wn_classes}

    def get_levels(self):                       # combined.py:117-126
        levels = []
        for c in self.wn_classes:
            lvl, p = 0, c
            while p != 'ROOT':
                p = self.parents[p]
                lvl += 1
            levels.append(lvl)
        return levels

    def on_branch(self, c1, c2):                # combined.py:143-150
        if c1 == c2:
            return True
        return min(c1, c2) in self.branches_ind[max(c1, c2)]


def get_class_map(toc, eoc):                    # detect_yolo3.py:698-709
    return [toc.index(c) if c in toc else -1 for c in eoc]


def iou(bb, bbgt):                              # detect_yolo3.py:712-733
    ov = 0
    iw = min(bb[2], bbgt[2]) - max(bb[0], bbgt[0]) + 1
    ih = min(bb[3], bbgt[3]) - max(bb[1], bbgt[1]) + 1
    if iw > 0 and ih > 0:
        intersect = iw * ih
        ua = (bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) + (bbgt[2] - bbgt[0] + 1.) * (bbgt[3] - bbgt[1] + 1.) - intersect
        ov = intersect / ua
    return ov


def hierarchical_nms(predictions, tree, ov_thresh=0.5, conf_thresh=0.0, level_thresh=10):   # detect_yolo3.py:736-789
    levels, parents, cls_map = tree.get_levels(), tree.parents, tree.wn_classes
    n = len(cls_map)
    branch_matrix = [[tree.on_branch(i, j) for j in range(n)] for i in range(n)]
    level_thresh = max(0, level_thresh)
    new = {}
    for img_path, boxes in predictions.items():
        new[img_path] = []
        for box in sorted(boxes, key=lambda x: x[0], reverse=True):
            cls, conf, coords = box[0], box[1], list(box[2:])
            if conf < conf_thresh:
                continue
            while levels[cls] > level_thresh:
                cls = cls_map.index(parents[cls_map[cls]])
            max_ov, max_idx = 0, -1
            for idx, boxb in enumerate(new[img_path]):
                overlap = iou(coords, boxb[2:])
                if overlap > ov_thresh and overlap > max_ov:
                    max_ov, max_idx = overlap, idx
            if max_idx == -1:
                new[img_path].append([cls, conf] + coords)
            else:
                boxb = new[img_path][max_idx]
                if not branch_matrix[cls][boxb[0]]:
                    new[img_path].append([cls, conf] + coords)
                elif cls == boxb[0]:
                    new[img_path][max_idx][1] = max(new[img_path][max_idx][1], conf)
    return new
