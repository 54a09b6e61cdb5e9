"""--mixup on the host side (train_yolov3.py:227-229,571-581; transforms.py:264-270; yolo_target.py:124-125): the pair
blending of gluoncv's MixupDetection (restated, [UPSTREAM-UNVERIFIED]: the class is not under /root/reference), the mix
ratio's way through the training transform into the objectness targets (against the oracle's target generator), the
loader's worker processes, and a bound on the per-frame transform time (the dense-matrix imresize of round 2 took 3.5 s
per 480x640 frame)."""
import time

import numpy as np
import pytest

from oracle import yolo as Y
from viddet_amd.data import Loader, MixupDetection, SyntheticDetection, YOLO3VideoTrainTransform
from viddet_amd.targets import prefetch_targets
from viddet_amd.video import Rng, imresize


class _TwoFrames:
    classes, num_class = ["a", "b", "c"], 3

    def __init__(self):
        self.imgs = [np.full((4, 6, 3), 200, np.uint8), np.full((5, 3, 3), 100, np.uint8)]
        self.labels = [np.array([[0., 0., 3., 2., 1., 0.]]), np.array([[1., 1., 2., 4., 2., 0.], [0., 0., 1., 1., 0., 1.]])]

    def __len__(self):
        return 2

    def sample_path(self, i):
        return "x%d.jpg" % i

    def __getitem__(self, i):
        return self.imgs[i], self.labels[i]


def test_mixup_pair_blend_known_answer():
    ds = MixupDetection(_TwoFrames(), rng=np.random.RandomState(0))
    img, lab = ds[0]                                         # no mix function: the frame itself, ratio column of ones
    assert img is ds._dataset.imgs[0] and lab.shape == (1, 7) and lab[0, 6] == 1.0
    ds.set_mixup(lambda a, b: 0.25, 1.5, 1.5)
    img, lab = ds[0]
    assert img.shape == (5, 6, 3) and img.dtype == np.uint8  # the larger height and width
    assert np.all(img[:4, :3] == 125)                        # 0.25 * 200 + 0.75 * 100
    assert np.all(img[:4, 3:] == 50) and np.all(img[4, :3] == 75) and np.all(img[4, 3:] == 0)
    assert lab.shape == (3, 7) and np.allclose(lab[:, 6], [0.25, 0.75, 0.75])
    assert np.array_equal(lab[0, :6], ds._dataset.labels[0][0]) and np.array_equal(lab[1:, :6], ds._dataset.labels[1])
    ds.set_mixup(lambda: 1.7)                                # clipped to 1: no blend
    img, lab = ds[1]
    assert img is ds._dataset.imgs[1] and np.all(lab[:, 6] == 1.0)
    ds.set_mixup(lambda: -0.3)                               # clipped to 0: the other frame alone, ratios 0 / 1
    img, lab = ds[1]
    assert np.all(img[:4, :6] == 200) and np.allclose(lab[:, 6], [0.0, 0.0, 1.0])


def test_mix_ratio_reaches_the_objectness_targets_like_the_oracle():
    size, c = 96, 5
    base = SyntheticDetection("synthetic", num_samples=6, size=(120, 90), num_class=c, seed=3)
    ds = MixupDetection(base, np.random.RandomState(1).beta, 1.5, 1.5, rng=np.random.RandomState(2))
    tf = YOLO3VideoTrainTransform(size, size, c, Rng.seeded(5), augment=False, mixup=True)
    seen_frac = 0
    for i in range(6):
        img, lab = ds[i]
        assert lab.shape[1] == 7
        x, obj, ctr, scl, wgt, cls, gt = tf(img, lab)
        # what the oracle's generator makes of the boxes the transform kept (resized, maybe flipped) and their ratios:
        # the transform keeps row order, so ratios follow their rows
        ratios = None
        for flipped in (False, True):
            from viddet_amd import bbox as tbbox
            bb = tbbox.resize(lab, (img.shape[1], img.shape[0]), (size, size))
            if flipped:
                bb = tbbox.flip(bb, (size, size), flip_x=True)
            if np.allclose(bb[:, :4], gt, atol=1e-4):
                ratios = bb[:, 6:7]
                break
        assert ratios is not None
        ref = Y.prefetch_targets(size, size, [size // 32, size // 16, size // 8], gt[None].astype(np.float64), bb[None, :, 4:5], c,
                                 ratios[None])
        for a, r in zip((obj, ctr, scl, wgt, cls), ref):
            assert np.allclose(a, r[0], atol=1e-6)
        pos = obj[obj > 0]
        assert pos.size and np.all(pos <= 1.0)
        seen_frac += int(((pos > 0) & (pos < 1)).any())
        # and without the flag the same label gives hard objectness targets
        o1 = prefetch_targets(size, size, gt[None], bb[None, :, 4:5], c)[0]
        assert set(np.unique(o1)) <= {0.0, 1.0}
    assert seen_frac >= 3


def test_loader_worker_processes_and_mixup_switch():
    base = SyntheticDetection("synthetic", num_samples=12, size=(96, 80), num_class=4, seed=7)
    ds = MixupDetection(base)
    tfs = [YOLO3VideoTrainTransform(s_, s_, 4, Rng.seeded(1), mixup=True) for s_ in (64, 96)]
    ld = Loader(ds, tfs, 4, train=True, shuffle=True, seed=3, interval=1, num_workers=2)
    try:
        ds.set_mixup(np.random.beta, 1.5, 1.5)
        got = list(ld)
        assert len(got) == 3
        for bt in got:
            assert len(bt) == 7 and bt[0].shape[0] == 4 and bt[0].shape[2] in (64, 96) and bt[0].shape[2] == bt[0].shape[3]
            P = 3 * sum((bt[0].shape[2] // s_) ** 2 for s_ in (32, 16, 8))
            assert bt[1].shape == (4, P, 1) and bt[5].shape == (4, P, 4) and bt[6].shape[2] == 4
        frac = sum(int(((bt[1] > 0) & (bt[1] < 1)).any()) for bt in got)
        assert frac >= 1                                    # blended samples carry fractional objectness targets
        # every sample draws its OWN lambda from beta(1.5, 1.5): a task that shipped the parent's bound np.random.beta
        # pickled the parent's generator state with it, and every worker drew one and the same ratio for every sample
        ratios = lambda batches: {round(float(v), 6) for bt in batches for v in np.unique(bt[1]) if 0.0 < v < 1.0}
        r1 = ratios(got)
        assert len(r1) >= 8, sorted(r1)
        # a second epoch draws other ratios; the same seed in a fresh loader - with another worker count - repeats the
        # first run bit for bit (per-sample generators seeded with (seed, rank, epoch, index))
        got2 = list(ld)
        assert ratios(got2) != r1
        ld3 = Loader(ds, tfs, 4, train=True, shuffle=True, seed=3, interval=1, num_workers=3)
        try:
            again = list(ld3)
        finally:
            ld3.close()
        assert len(again) == len(got) and all(np.array_equal(a, b) for x, y in zip(again, got) for a, b in zip(x, y))
        ds.set_mixup(None)                                  # the last --no_mixup_epochs epochs: travels with the tasks
        for bt in ld:
            assert set(np.unique(bt[1])) <= {0.0, 1.0}
        # the single-process loader draws the same shapes (one generator seeded with `seed` on every rank)
        ld0 = Loader(ds, tfs, 4, train=True, shuffle=True, seed=3, interval=1)
        assert [bt[0].shape for bt in ld0] == [bt[0].shape for bt in Loader(ds, tfs, 4, train=True, shuffle=True, seed=3, interval=1, num_workers=2)]
    finally:
        ld.close()


@pytest.mark.parametrize("side", [416, 608])
def test_training_transform_time_per_frame(side):
    ds = SyntheticDetection("voc", num_samples=4, size=(640, 480), seed=11)
    tf = YOLO3VideoTrainTransform(side, side, ds.num_class, Rng.seeded(0))
    tf(*ds[0])                                               # warm the tap cache
    # (the FASTEST frame of four: the bound is about the algorithm - round 2's dense resampling matrices took 3.5 s - and must
    # not trip on a host that is busy with something else while the suite runs)
    per = []
    for i in range(4):
        t0 = time.perf_counter()
        tf(*ds[i])
        per.append(time.perf_counter() - t0)
    print("YOLO3VideoTrainTransform %d: %.3f s per 480x640 frame (fastest of 4; slowest %.3f)" % (side, min(per), max(per)))
    assert min(per) < 1.0, per
    img = ds[0][0]
    per = []
    for interp in (1, 2, 3, 4, 9):
        t0 = time.perf_counter()
        imresize(img, side, side, interp)
        per.append(time.perf_counter() - t0)
    assert min(per) < 0.5, per
