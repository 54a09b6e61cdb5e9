"""Host logic of SURVEY §8(f) N2: class tree, class map, single-pair IoU and hierarchical NMS — the product
(viddet_amd/hierarchy.py) against the line-by-line oracle (oracle/hierarchy.py) on random trees and detections,
plus hand-computed known answers.  Parity unpinned by reference fixtures (see the oracle's header)."""
import numpy as np

from oracle import hierarchy as O
from viddet_amd import hierarchy as H


def _random_tree(rng, n):
    """Labels ordered parents-first (as the combined dataset's class list is): parent index < child index."""
    names = ["n%03d" % i for i in range(n)]
    parents = {}
    for i, c in enumerate(names):
        parents[c] = "ROOT" if i == 0 or rng.random() < 0.15 else names[int(rng.integers(0, i))]
    return names, parents


def _random_preds(rng, n_cls, n_img=4, n_box=40):
    preds = {}
    for i in range(n_img):
        rows = []
        for _ in range(int(rng.integers(0, n_box))):
            x1, y1 = rng.uniform(0, 300, 2)
            w, h = rng.uniform(5, 150, 2)
            if rows and rng.random() < 0.5:           # a jittered copy of an earlier box: overlaps are the point
                b = rows[int(rng.integers(0, len(rows)))]
                x1, y1, w, h = b[2] + rng.normal(0, 4), b[3] + rng.normal(0, 4), b[4] - b[2] + rng.normal(0, 4), b[5] - b[3] + rng.normal(0, 4)
                w, h = max(w, 1.0), max(h, 1.0)
            rows.append([int(rng.integers(0, n_cls)), float(rng.uniform(0, 1)), float(x1), float(y1), float(x1 + w), float(y1 + h)])
        preds["img%d.jpg" % i] = rows
    return preds


def test_tree_matches_oracle():
    rng = np.random.default_rng(0)
    for n in (1, 5, 30, 120):
        names, parents = _random_tree(rng, n)
        t, o = H.ClassTree(names, parents), O.Tree(names, parents)
        assert t.get_levels() == o.get_levels()
        assert t.branches_ind == o.branches_ind
        for a in range(n):
            for b in range(n):
                assert t.on_branch(a, b) == o.on_branch(a, b)


def test_tree_known_answers():
    #   animal(0) -> dog(1) -> puppy(3);  animal -> cat(2);  vehicle(4) -> car(5)
    names = ["animal", "dog", "cat", "puppy", "vehicle", "car"]
    parents = {"animal": "ROOT", "dog": "animal", "cat": "animal", "puppy": "dog", "vehicle": "ROOT", "car": "vehicle"}
    t = H.ClassTree(names, parents)
    assert t.get_levels() == [1, 2, 2, 3, 1, 2]
    assert t.get_leaves() == [0, 0, 1, 1, 0, 1]
    assert t.branches_ind[3] == [0, 1, 3] and t.branches_ind[5] == [4, 5]
    assert t.on_branch(3, 0) and t.on_branch(1, 3) and not t.on_branch(2, 3) and not t.on_branch(5, 0)


def test_class_map_and_iou():
    assert H.get_class_map(["a", "b", "c", "b"], ["c", "z", "b"]) == O.get_class_map(["a", "b", "c", "b"], ["c", "z", "b"]) == [2, -1, 1]
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.uniform(0, 50, 4); b = rng.uniform(0, 50, 4)
        a[2:] += a[:2]; b[2:] += b[:2]
        assert H.iou(list(a), list(b)) == O.iou(list(a), list(b))
    assert H.iou([0, 0, 9, 9], [0, 0, 9, 9]) == 1.0                    # 10x10 pixels with the +1 convention
    assert H.iou([0, 0, 9, 9], [10, 0, 19, 9]) == 0                    # touching pixel columns do not overlap
    assert abs(H.iou([0, 0, 9, 9], [5, 0, 14, 9]) - 50.0 / 150.0) < 1e-15


def test_hierarchical_nms_matches_oracle():
    rng = np.random.default_rng(2)
    for n_cls in (1, 6, 40):
        names, parents = _random_tree(rng, n_cls)
        t, o = H.ClassTree(names, parents), O.Tree(names, parents)
        preds = _random_preds(rng, n_cls)
        for kw in (dict(), dict(ov_thresh=0.3, conf_thresh=0.2), dict(level_thresh=1), dict(level_thresh=2, ov_thresh=0.7)):
            got, ref = H.hierarchical_nms(preds, t, **kw), O.hierarchical_nms(preds, o, **kw)
            assert got.keys() == ref.keys()
            for k in ref:
                assert got[k] == ref[k], (n_cls, kw, k)


def test_hierarchical_nms_known_answers():
    names = ["animal", "dog", "cat", "puppy"]
    parents = {"animal": "ROOT", "dog": "animal", "cat": "animal", "puppy": "dog"}
    t = H.ClassTree(names, parents)
    box = [10.0, 10.0, 60.0, 60.0]
    near = [11.0, 10.0, 61.0, 60.0]
    # a puppy box and an overlapping dog box: the dog (ancestor, lower index, visited later) is absorbed, conf untouched
    out = H.hierarchical_nms({"a": [[1, 0.9] + near, [3, 0.6] + box]}, t)
    assert out["a"] == [[3, 0.6] + box]
    # a cat box over a puppy box is a different lineage: both stay
    out = H.hierarchical_nms({"a": [[2, 0.9] + near, [3, 0.6] + box]}, t)
    assert out["a"] == [[3, 0.6] + box, [2, 0.9] + near]
    # level_thresh=2 lifts puppy (level 3) to dog: two dog boxes merge with the max confidence, first coordinates kept
    out = H.hierarchical_nms({"a": [[1, 0.9] + near, [3, 0.6] + box]}, t, level_thresh=2)
    assert out["a"] == [[1, 0.9] + box]
    # below the confidence threshold a box is dropped before anything else
    out = H.hierarchical_nms({"a": [[3, 0.1] + box]}, t, conf_thresh=0.5)
    assert out["a"] == []


def test_hierarchical_nms_level_zero_raises_like_the_reference():
    """level_thresh <= 0 asks for a label at level 0; none exists ('ROOT' is not a label), and the reference's
    cls_map.index('ROOT') raises ValueError — same error here."""
    import pytest
    names, parents = ["a", "b"], {"a": "ROOT", "b": "a"}
    with pytest.raises(ValueError):
        H.hierarchical_nms({"i": [[1, 0.5, 0.0, 0.0, 5.0, 5.0]]}, H.ClassTree(names, parents), level_thresh=-3)
    with pytest.raises(ValueError):
        O.hierarchical_nms({"i": [[1, 0.5, 0.0, 0.0, 5.0, 5.0]]}, O.Tree(names, parents), level_thresh=-3)


def test_voc_map_with_class_map_product_vs_oracle():
    """VOCMApMetric(class_map=...) (metrics/pascalvoc.py:37,71-80,126-127): ground-truth ids are mapped into the
    model's label set before matching and the per-class values are reported per evaluation class."""
    from oracle import yolo as Y
    from viddet_amd.metrics import VOCMApMetric
    rng = np.random.default_rng(5)
    eval_names = ["e0", "e1", "e2", "e3"]
    cmap = H.get_class_map(["m0", "e2", "e0", "e3"], ["e0", "e1", "e2", "e3"])
    assert cmap == [2, -1, 1, 3]
    prod = VOCMApMetric(0.5, class_names=eval_names, class_map=cmap)
    orac = Y.VOCMApMetric(0.5, class_names=eval_names, class_map=cmap)
    for _ in range(6):
        ng, npred = int(rng.integers(1, 6)), int(rng.integers(1, 12))
        g = rng.uniform(0, 80, (ng, 2)); gtb = np.concatenate([g, g + rng.uniform(10, 40, (ng, 2))], 1)
        gtl = rng.integers(0, 4, ng).astype(float)
        idx = rng.integers(0, ng, npred)
        pb = gtb[idx] + rng.normal(0, 3, (npred, 4))
        pl = np.array([cmap[int(l)] if rng.random() < 0.7 else int(rng.integers(0, 4)) for l in gtl[idx]], dtype=float)
        ps = rng.uniform(0, 1, npred)
        prod.update([pb], [pl], [ps], [gtb], [gtl])
        orac.update([pb], [pl], [ps], [gtb], [gtl])
    names, vals = prod.get()
    aps, mean = orac.get()
    assert names == eval_names + ["mAP"]
    assert np.allclose(vals[:-1], aps, equal_nan=True) and np.isclose(vals[-1], mean, equal_nan=True)
    assert np.isnan(vals[1])                                                             # e1 has no model class
    # a model with more labels than the evaluation set does not break the bookkeeping
    big = VOCMApMetric(0.5, class_names=["e0", "e1"], class_map=[5, -1])
    big.update([np.array([[0., 0., 10., 10.]])], [np.array([5.])], [np.array([0.9])], [np.array([[0., 0., 10., 10.]])], [np.array([0.])])
    assert big.get()[1][0] == 1.0 and np.isnan(big.get()[1][1])


def test_voc_map_temporal_product_vs_oracle():
    """VOCMApMetricTemporal (metrics/pascalvoc.py:262-560): per-offset accumulators over (B, t, N, .) inputs."""
    from oracle import yolo as Y
    from viddet_amd.metrics import VOCMApMetricTemporal
    rng = np.random.default_rng(9)
    T_, names = 3, ["a", "b", "c"]
    prod, orac = VOCMApMetricTemporal(T_, 0.5, class_names=names), Y.VOCMApMetricTemporal(T_, 0.5, class_names=names)
    for _ in range(4):
        B, N, M = 2, 7, 3
        g = rng.uniform(0, 60, (B, T_, M, 2)); gtb = np.concatenate([g, g + rng.uniform(8, 30, (B, T_, M, 2))], -1)
        gtl = rng.integers(-1, 3, (B, T_, M, 1)).astype(float)
        pb = np.concatenate([gtb, gtb[:, :, :1].repeat(N - M, 2) + rng.normal(0, 5, (B, T_, N - M, 4))], 2) + rng.normal(0, 1.5, (B, T_, N, 4))
        pl = rng.integers(-1, 3, (B, T_, N, 1)).astype(float)
        ps = rng.uniform(0, 1, (B, T_, N, 1))
        prod.update(pb, pl, ps, gtb, gtl)
        orac.update(pb, pl, ps, gtb, gtl)
    nm, vals = prod.get()
    assert nm[:4] == ["a t=0/3", "b t=0/3", "c t=0/3", "mAP t=0/3"] and len(nm) == len(vals) == T_ * 4
    for t, (aps, mean) in enumerate(orac.get()):
        assert np.allclose(vals[4 * t:4 * t + 3], aps, equal_nan=True) and np.isclose(vals[4 * t + 3], mean, equal_nan=True)
