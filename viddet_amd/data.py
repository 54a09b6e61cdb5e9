"""Sample contract of the reference's datasets + the two transforms its scripts use, for synthetic frames.

Mirrors (paths under /root/reference):
  Dataset.__getitem__ -> (img HWC uint8, label (N,6) [x1,y1,x2,y2,cls,difficult])   datasets/pascalvoc.py:94-127
  YOLO3VideoTrainTransform.__call__       models/definitions/yolo/transforms.py:199-294  (the full augmentation chain)
  YOLO3VideoInferenceTransform.__call__   models/definitions/yolo/transforms.py:316-350
  batchify: Stack images/targets, Pad(-1) gt boxes                                   train_yolov3.py:252-256
The real file-system readers (VOC/COCO/DET/VID) are out of scope: no dataset files
exist offline and the headline metric is quoted on synthetic frames.  Class counts are kept.
"""
import numpy as np

from . import bbox as tbbox
from .targets import prefetch_targets
from .video import Rng, imresize, random_color_distort, random_expand

NUM_CLASSES = {"voc": 20, "coco": 80, "det": 200, "vid": 30, "comb": 285, "synthetic": 20}
MEAN = np.array([0.485, 0.456, 0.406], np.float32)     # transforms.py:167
STD = np.array([0.229, 0.224, 0.225], np.float32)      # transforms.py:168


class SyntheticDetection:
    """Deterministic synthetic dataset: uint8 frames with `max_gt` random boxes."""

    def __init__(self, name="synthetic", num_samples=64, size=(480, 360), num_class=None, max_gt=8, seed=233,
                 window=1, mult_out=False):
        """window > 1: a sample is a window of `window` frames (k,h,w,3) as the VID dataset yields with
        --window k (datasets/imgnetvid.py); its label is the centre frame's boxes, or with mult_out a list of
        per-frame box arrays (--mult_out)."""
        self.name = name
        self.num_class = NUM_CLASSES.get(name, 20) if num_class is None else num_class
        self.classes = ["class%d" % i for i in range(self.num_class)]
        self._n, self._size, self._max_gt, self._seed = num_samples, size, max_gt, seed
        self._window, self._mult_out = int(window), bool(mult_out)

    def __len__(self):
        return self._n

    def sample_path(self, idx):
        return "synthetic/s%d_%06d.jpg" % (self._seed, idx)        # the seed keeps train / val file ids apart

    def _frame(self, rng):
        w, h = self._size
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        n = int(rng.integers(1, self._max_gt + 1))
        c = rng.uniform(0.1, 0.9, (n, 2)) * (w, h)
        wh = rng.uniform(16, 0.5 * min(w, h), (n, 2))
        box = np.concatenate([np.clip(c - wh / 2, 0, (w - 1, h - 1)), np.clip(c + wh / 2, 0, (w - 1, h - 1))], axis=1)
        cls = rng.integers(0, self.num_class, (n, 1)).astype(np.float64)
        return img, np.concatenate([box, cls, np.zeros((n, 1))], axis=1)

    def __getitem__(self, idx):
        rng = np.random.default_rng(self._seed * 1000003 + idx)
        if self._window <= 1:
            return self._frame(rng)
        frames = [self._frame(rng) for _ in range(self._window)]
        imgs = np.stack([f[0] for f in frames])
        if self._mult_out:
            return imgs, [f[1] for f in frames]
        return imgs, frames[self._window // 2][1]


class SyntheticCombined(SyntheticDetection):
    """Stand-in for CombinedDetection(datasets, class_tree=True) (detect_yolo3.py:167, datasets/combined.py): the label set of
    several datasets arranged in a class tree - one group label per dataset under 'ROOT' (labels are ordered parents first,
    as combined.py's tree walk orders them), that dataset's classes below it - with the attributes the hierarchical NMS
    reads (`parents`, `wn_classes`, `get_levels()`, `on_branch()`).  Ground-truth boxes carry leaf labels."""

    def __init__(self, names, num_samples=64, classes_per_set=None, **kw):
        from .hierarchy import ClassTree, ROOT
        per = [NUM_CLASSES.get(n, 20) if classes_per_set is None else int(classes_per_set) for n in names]
        groups = ["grp_%s" % n for n in names]
        leaves = ["%s_c%d" % (n, i) for n, k in zip(names, per) for i in range(k)]
        parents = {g: ROOT for g in groups}
        for n, k in zip(names, per):
            parents.update({"%s_c%d" % (n, i): "grp_%s" % n for i in range(k)})
        super().__init__("comb", num_samples=num_samples, num_class=len(groups) + len(leaves), **kw)
        self.classes = groups + leaves
        self.tree = ClassTree(self.classes, parents)
        self.parents, self.wn_classes = self.tree.parents, self.tree.wn_classes
        self._first_leaf = len(groups)

    def get_levels(self):
        return self.tree.get_levels()

    def on_branch(self, c1, c2):
        return self.tree.on_branch(c1, c2)

    def _frame(self, rng):
        img, lab = super()._frame(rng)
        lab[:, 4] = self._first_leaf + np.mod(lab[:, 4], self.num_class - self._first_leaf)     # leaves only
        return img, lab


class MixupDetection:
    """gluoncv.data.MixupDetection as train_yolov3.py:227-229 wraps the training set with --mixup (the class is not in
    /root/reference; restated from gluoncv/data/mixup/detection.py, [UPSTREAM-UNVERIFIED]): with a mix function set
    (`set_mixup(np.random.beta, 1.5, 1.5)`, train_yolov3.py:571-581) a sample is the pixel blend
    lambda * img1 + (1 - lambda) * img2 of frame `idx` and one other frame on a canvas of the larger height / width
    (uint8 again), its label both frames' rows with a 7th column = the row's mix ratio (lambda or 1 - lambda).  lambda is
    clipped to [0, 1]; lambda >= 1 (or no mix function) returns frame `idx` with a column of ones.  Draws: the mix
    function, then np.random.choice over the other indices - numpy's global generator, as upstream (or `rng`)."""

    def __init__(self, dataset, mixup=None, *args, rng=None):
        self._dataset, self._mixup, self._mixup_args = dataset, mixup, args
        self._rng = rng                                      # None = numpy's global generator
        self.classes, self.num_class = dataset.classes, dataset.num_class

    def set_mixup(self, mixup=None, *args):
        self._mixup, self._mixup_args = mixup, args

    def __len__(self):
        return len(self._dataset)

    def sample_path(self, idx):
        return self._dataset.sample_path(idx)

    def __getitem__(self, idx):
        img1, label1 = self._dataset[idx]
        if np.ndim(img1) != 3 or isinstance(label1, (list, tuple)):
            raise NotImplementedError("MixupDetection blends single (h,w,3) frames; windows (--window k > 1) have no mixup upstream either")
        lambd = 1.0
        if self._mixup is not None:
            lambd = max(0.0, min(1.0, float(self._mixup(*self._mixup_args))))
        if lambd >= 1:
            return img1, np.hstack((label1, np.ones((label1.shape[0], 1))))
        choice = (np.random if self._rng is None else self._rng).choice
        idx2 = int(choice(np.delete(np.arange(len(self)), idx)))
        img2, label2 = self._dataset[idx2]
        height, width = max(img1.shape[0], img2.shape[0]), max(img1.shape[1], img2.shape[1])
        mix = np.zeros((height, width, 3), dtype=np.float32)
        mix[:img1.shape[0], :img1.shape[1], :] = img1.astype(np.float32) * np.float32(lambd)
        mix[:img2.shape[0], :img2.shape[1], :] += img2.astype(np.float32) * np.float32(1.0 - lambd)
        y1 = np.hstack((label1, np.full((label1.shape[0], 1), lambd)))
        y2 = np.hstack((label2, np.full((label2.shape[0], 1), 1.0 - lambd)))
        return mix.astype(np.uint8), np.vstack((y1, y2))       # astype('uint8') truncates, as mx.nd's cast does


def _to_tensor_normalize(img):
    x = img.astype(np.float32) / 255.0            # mx.nd.image.to_tensor (any input dtype is divided by 255)
    x = (x - MEAN) / STD                          # mx.nd.image.normalize
    return np.ascontiguousarray(x.transpose(2, 0, 1))


class YOLO3VideoInferenceTransform:
    """transforms.py:297-350: resize to (width,height) with interp 9 (area when shrinking / bicubic when enlarging,
    viddet_amd/video.py imresize), to_tensor, normalize; boxes resized along.  device_normalize=True keeps the resized
    frames as uint8 (H,W,3) / (k,H,W,3): the network normalises them on the GPU (vd_preprocess_u8_nchw) - the same
    arithmetic, a quarter of the bytes over PCIe."""

    def __init__(self, width, height, device_normalize=False):
        self._w, self._h, self._u8 = width, height, device_normalize

    def __call__(self, img, label, idx=0):
        h, w = img.shape[-3], img.shape[-2]
        frames = img if img.ndim == 4 else img[np.newaxis]
        ims = [imresize(f, self._w, self._h, interp=9) for f in frames]                      # transforms.py:329-333
        out = np.stack(ims) if self._u8 else np.stack([_to_tensor_normalize(im) for im in ims])
        if img.ndim != 4:
            out = out[0]
        bb = tbbox.resize(label, (w, h), (self._w, self._h))
        if isinstance(bb, (list, tuple)):                  # per-frame labels (--mult_out): (k, M, 6), -1 padded
            return out, pad_stack([np.asarray(b, dtype=np.float32) for b in bb]), idx
        return out, bb.astype(np.float32), idx


class YOLO3VideoTrainTransform:
    """transforms.py:143-294, the whole augmentation chain in the reference's order (:199-246): random colour distortion,
    random expansion (probability 0.5, canvas filled with the dataset mean) with the boxes translated, SSD-style
    constrained random crop, resize with a random interpolation (0-4), random horizontal flip (0.5), to_tensor +
    normalise, then the prefetch targets (:252-294).  One set of decisions per sample: every frame of a window gets the
    same distortion / crop / flip.  `rng`: viddet_amd.video.Rng (numpy + python generators; default = a private pair
    seeded with 0; Rng() = the global modules, exactly the reference's sources)."""

    def __init__(self, width, height, num_class, rng=None, augment=True, device_normalize=False, mixup=False):
        """mixup (transforms.py:166,264-270): the labels carry a last column of mix ratios (MixupDetection), which goes to
        the target generator as gt_mixratio -> the objectness target (yolo_target.py:124-125).  The class id is column 4,
        as in gluoncv's YOLO3DefaultTrainTransform, which this transform was derived from: the reference's own test
        `bbox.shape[-1] == 6` (transforms.py:261) sends 7-column labels down its multi-hot branch, whose 3-wide slice
        (class, difficult, ratio) its target generator cannot broadcast into C class targets (yolo_target.py:128) - the
        reference's --mixup stops there for every C != 3."""
        self._w, self._h, self._c, self._mixup = width, height, num_class, bool(mixup)
        if rng is None:
            rng = Rng.seeded(0)
        elif isinstance(rng, np.random.Generator):             # an older call form: derive the pair from the generator
            rng = Rng.seeded(int(rng.integers(0, 2 ** 31 - 1)))
        self._rng, self._augment, self._u8 = rng, augment, device_normalize

    def __call__(self, img, label):
        rng = self._rng
        frames = (img if img.ndim == 4 else img[np.newaxis])
        bb = label
        if self._augment:
            frames = random_color_distort(frames, rng=rng)                                   # :207 (float32 from here on)
            if rng.np.uniform(0, 1) > 0.5:                                                   # :210-214
                frames, expand = random_expand(frames, fill=[m * 255 for m in MEAN], rng=rng)
                bb = tbbox.translate(bb, x_offset=expand[0], y_offset=expand[1])
            k, h, w, c = frames.shape                                                        # :217-220
            bb, crop = tbbox.random_crop_with_constraints(bb, (w, h), py_rng=rng.py, np_rng=rng.np)
            x0, y0, cw, ch = crop
            frames = frames[:, y0:y0 + ch, x0:x0 + cw, :]
        k, h, w, c = frames.shape
        interp = int(rng.np.randint(0, 5)) if self._augment else 1                           # :224
        ims = [imresize(f, self._w, self._h, interp=interp) for f in frames]
        bb = tbbox.resize(bb, (w, h), (self._w, self._h))
        if rng.np.uniform(0, 1) > 0.5:                                                       # :233-236
            ims = [im[:, ::-1] for im in ims]
            bb = tbbox.flip(bb, (self._w, self._h), flip_x=True)
        if self._u8:          # device-side normalisation: round the (possibly distorted, float) frames to uint8 first
            x = np.stack([np.clip(np.rint(im), 0, 255).astype(np.uint8) for im in ims])
        else:
            x = np.stack([_to_tensor_normalize(im) for im in ims])
        if img.ndim != 4:
            x = x[0]
        bboxs = list(bb) if isinstance(bb, (list, tuple)) else [bb]       # the crop returns a list of per-frame arrays
        if len(bboxs) > 1:
            # per-frame labels (--mult_out, transforms.py:252-294): targets of every frame stacked on a leading t axis,
            # gt boxes (t, M, 4) padded with -1
            tg = [prefetch_targets(self._h, self._w, b[np.newaxis, :, :4], b[np.newaxis, :, 4:5], self._c,
                                   b[np.newaxis, :, -1:] if self._mixup else None) for b in bboxs]
            cols = [np.concatenate([t[i] for t in tg], axis=0) for i in range(5)]
            gt = pad_stack([np.asarray(b[:, :4], dtype=np.float32) for b in bboxs])
            return (x,) + tuple(cols) + (gt,)
        b0 = np.asarray(bboxs[0])                                         # :269-271 one label set: un-stacked targets
        gt = b0[np.newaxis, :, :4]
        ids = b0[np.newaxis, :, 4:5]
        obj, ctr, scl, wgt, cls = prefetch_targets(self._h, self._w, gt, ids, self._c,
                                                   b0[np.newaxis, :, -1:] if self._mixup else None)
        return x, obj[0], ctr[0], scl[0], wgt[0], cls[0], gt[0].astype(np.float32)


def feature_file_id(img_path):
    """extract_base_features.py:144-148 (non-VID branch): `<basename without its 4-char extension>`."""
    return img_path.split("/")[-1][:-4]


class FeatureDataset:
    """A dataset built with `features_dir` (datasets/pascalvoc.py with features_dir set; train_yolov3.py:177-205):
    samples become (img, f1, f2, f3, label) with the three cached backbone maps read from
    `<features_dir>/<file id>_F{1,2,3}.npy` as extract_base_features.py wrote them ((C,h,w) fp32 each)."""

    def __init__(self, dataset, features_dir):
        self.ds, self.dir = dataset, features_dir
        self.name, self.classes = dataset.name, dataset.classes
        self.num_class = getattr(dataset, "num_class", len(dataset.classes))

    def __len__(self):
        return len(self.ds)

    def sample_path(self, idx):
        return self.ds.sample_path(idx)

    def __getitem__(self, idx):
        import os
        img, label = self.ds[idx]
        fid = feature_file_id(self.ds.sample_path(idx))
        feats = [np.load(os.path.join(self.dir, "%s_F%d.npy" % (fid, i))) for i in (1, 2, 3)]
        return (img,) + tuple(feats) + (label,)


class YOLO3NBVideoTrainTransform:
    """transforms.py:353-428: the frames are only used for their size; boxes are resized to the network input and
    the prefetch targets are generated on the host; the cached features pass through."""

    def __init__(self, k, width, height, num_class):
        self._k, self._w, self._h, self._c = k, width, height, num_class

    def __call__(self, img, f1, f2, f3, label):
        h, w = img.shape[-3], img.shape[-2]                       # (h,w,c) or (k,h,w,c)  (transforms.py:399-402)
        bb = tbbox.resize(label, (w, h), (self._w, self._h))
        gt = bb[np.newaxis, :, :4]
        ids = bb[np.newaxis, :, 4:5]
        obj, ctr, scl, wgt, cls = prefetch_targets(self._h, self._w, gt, ids, self._c)
        return f1, f2, f3, obj[0], ctr[0], scl[0], wgt[0], cls[0], gt[0].astype(np.float32)


class YOLO3NBVideoInferenceTransform:
    """transforms.py:431-457: features pass through, boxes are resized to (width,height)."""

    def __init__(self, width, height):
        self._w, self._h = width, height

    def __call__(self, img, f1, f2, f3, label, idx=None):
        h, w = img.shape[-3], img.shape[-2]
        bb = tbbox.resize(label, (w, h), (self._w, self._h)).astype(np.float32)
        return (f1, f2, f3, bb) if idx is None else (f1, f2, f3, bb, idx)


def pad_stack(arrs, pad_val=-1.0):
    """Stack arrays whose leading dims are ragged (Pad(pad_val=-1) batchify, axis 0 for (M,.) labels and axis 1 for
    the (t,M,.) per-frame labels of --mult_out, train_yolov3.py:253-256,273-276)."""
    shape = tuple(max(a.shape[d] for a in arrs) for d in range(arrs[0].ndim))
    out = np.full((len(arrs),) + shape, pad_val, dtype=np.float32)
    for i, a in enumerate(arrs):
        out[(i,) + tuple(slice(0, n) for n in a.shape)] = a
    return out


class Loader:
    """Minimal single-process loader (the reference's multi-worker DataLoader only feeds the host pipeline).

    `transform` may be a LIST of transforms: then this is gluoncv's RandomTransformDataLoader (train_yolov3.py:262-271, the
    reference's DEFAULT training loader: YOLO3VideoTrainTransform at 320, 352, ... 608 pixels) - one of them is drawn
    when iteration starts and again every `interval` batches, and a whole batch goes through the same one (so a batch
    has one shape).  The draw comes from a generator seeded with `seed` alone, the same on every rank: under data
    parallelism all shards of a global batch must have one shape, as the reference's single split_and_load batch does.
    [UPSTREAM-UNVERIFIED] gluoncv draws with numpy's global generator from its worker-feeding thread, interleaved with
    the workers' own draws, so the reference's sequence of shapes is not reproducible either."""

    def __init__(self, dataset, transform, batch_size, train, shuffle=False, last_batch="rollover", seed=0,
                 rank=0, world=1, interval=10, num_workers=0):
        self.ds, self.tf, self.bs, self.train = dataset, transform, batch_size, train
        # num_workers > 0: samples are read and transformed in that many worker processes (the reference's
        # DataLoader(num_workers=...), train_yolov3.py:242-280), one batch ahead of the consumer.  Every SAMPLE's draws
        # are seeded with (seed, rank, epoch, sample index), so a run is reproducible whatever worker takes which sample;
        # num_workers = 0 keeps the single sequential stream.
        self.num_workers, self._seed, self._pool, self._epoch = max(0, int(num_workers)), seed, None, 0
        self.shuffle, self.last_batch, self._rng = shuffle, last_batch, np.random.default_rng(seed)
        self.rank, self.world = rank, world
        self.tfs = list(transform) if isinstance(transform, (list, tuple)) else None
        self.interval = max(1, int(interval))
        self._choice = np.random.RandomState(seed)
        if self.tfs is not None:
            if not self.tfs or not train:
                raise ValueError("a list of transforms is the training-time random-shape loader")
            self.tf = self.tfs[0]

    def __len__(self):
        if self.last_batch == "keep":                    # every sample of this rank's shard (validation / detection)
            n = len(range(self.rank, len(self.ds), self.world))
            return (n + self.bs - 1) // self.bs
        return (len(self.ds) // self.world) // self.bs   # training: the same number of batches on every rank

    def __iter__(self):
        idx = np.arange(len(self.ds))
        if self.shuffle:
            self._rng.shuffle(idx)
        idx = idx[self.rank::self.world]                 # frames are sharded across ranks, windows never split
        if self.num_workers > 0:
            yield from self._iter_workers(idx)
            return
        for i in range(len(self)):
            chunk = idx[i * self.bs:(i + 1) * self.bs]
            if self.tfs is not None and i % self.interval == 0:
                self.tf = self.tfs[int(self._choice.randint(len(self.tfs)))]
            yield self._collate(self._samples(chunk, self.tf))

    def _collate(self, samples):
        cols = list(zip(*samples))
        if self.train:
            # Stack every column, Pad(-1) the trailing gt boxes (train_yolov3.py:238 / :252: 8+1 columns with
            # cached features, 6+1 with frames)
            return [np.stack(c) for c in cols[:-1]] + [pad_stack(cols[-1])]
        # (data..., Pad(-1) labels, sample index): 1 data column for frames, 3 for cached features
        return tuple(np.stack(c) for c in cols[:-2]) + (pad_stack(cols[-2]), np.asarray(cols[-1]))

    def _samples(self, chunk, tf):
        if self.train:
            return [tf(*self.ds[int(j)]) for j in chunk]
        return [tf(*self.ds[int(j)], int(j)) for j in chunk]

    # ---- worker processes -----------------------------------------------------------------------------------------
    @staticmethod
    def _mix_token(ds):
        """What a task carries about the dataset's mix function (train_yolov3.py:571-581 switches it per epoch on the
        parent's copy).  A draw of numpy's global generator (`np.random.beta`, ...) travels as its NAME: the bound method
        itself would pickle the parent's generator state, and every worker would then draw the same lambda for every
        sample.  Any other callable travels as it is (it must be picklable and carry its own randomness)."""
        if not isinstance(ds, MixupDetection):
            return None
        f = ds._mixup
        if f is None:
            return ('none',)
        owner = getattr(f, "__self__", None)
        if isinstance(owner, np.random.RandomState) or owner is np.random:
            return ('np.random', f.__name__, tuple(ds._mixup_args))
        return ('callable', f, tuple(ds._mixup_args))

    def _iter_workers(self, idx):
        """The same batches as the single-process loop, produced by the pool one batch ahead.  Every sample's random
        draws come from generators seeded with (seed, rank, epoch, sample index) - `_sample_seed` - so the augmentation
        of a sample does not depend on which worker happens to transform it: a fixed `seed` reproduces the run for any
        num_workers > 0 (the single-process loop keeps its one sequential stream)."""
        if self._pool is None:
            import multiprocessing as mp
            # fresh interpreters (spawn), not forks of this process: it has the GPU open, its children must not inherit
            # that; viddet_amd.data imports NumPy only, so a worker starts in a fraction of a second
            ctx = mp.get_context("spawn")
            self._pool = ctx.Pool(self.num_workers, initializer=_worker_init,
                                  initargs=(self.ds, self.tfs if self.tfs is not None else [self.tf], self.train))
        epoch = self._epoch
        self._epoch += 1
        pending = None
        for i in range(len(self)):
            chunk = idx[i * self.bs:(i + 1) * self.bs]
            if self.tfs is not None and i % self.interval == 0:
                self.tf = self.tfs[int(self._choice.randint(len(self.tfs)))]
            ti = self.tfs.index(self.tf) if self.tfs is not None else 0
            mix = self._mix_token(self.ds)
            nxt = self._pool.map_async(_worker_sample, [(ti, int(j), mix, _sample_seed(self._seed, self.rank, epoch, int(j)))
                                                        for j in chunk])
            if pending is not None:
                yield self._collate(pending.get())
            pending = nxt
        if pending is not None:
            yield self._collate(pending.get())

    def close(self):
        if self._pool is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_WORKER = {}


def _sample_seed(seed, rank, epoch, j):
    """32-bit seed of one sample's draws: a SeedSequence hash of (seed, rank, epoch, sample index)."""
    return int(np.random.SeedSequence([int(seed) & 0xffffffff, int(rank), int(epoch), int(j)]).generate_state(1)[0])


def _worker_init(ds, tfs, train):
    _WORKER.update(ds=ds, tfs=tfs, train=train)


def _worker_sample(task):
    import random as _pyrandom
    ti, j, mix, sseed = task
    tf, ds = _WORKER["tfs"][ti], _WORKER["ds"]
    # this sample's generators: the global modules (Rng() transforms and MixupDetection draw from them) and the
    # transform's private pair
    np.random.seed(sseed)
    _pyrandom.seed(sseed)
    if getattr(tf, "_rng", None) is not None:
        tf._rng = Rng.seeded(sseed)
    if mix is not None:
        if mix[0] == 'none':
            ds.set_mixup(None)
        elif mix[0] == 'np.random':
            ds.set_mixup(getattr(np.random, mix[1]), *mix[2])      # THIS process's generator, seeded above
        else:
            ds.set_mixup(mix[1], *mix[2])
    return tf(*ds[j]) if _WORKER["train"] else tf(*ds[j], j)
