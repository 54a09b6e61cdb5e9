"""Host-side mirror of the reference's yolo3_darknet53 network object for the MI355X kernel library.

Mirrors (paths under /root/reference):
  yolo3_darknet53 factory            models/definitions/yolo/wrappers.py:9-110
  Darknet-53 2-D backbone            models/definitions/darknet/three_darknet.py:100-123,152-264
  YOLODetectionBlockV3               models/definitions/yolo/yolo3.py:218-263
  YOLOOutputV3                       models/definitions/yolo/yolo3.py:43-199
  YOLOV3T (wiring, NMS, losses)      models/definitions/yolo/yolo3.py:959-1302
  _conv2d / _upsample                models/definitions/layers.py:11-20,63-70

The network is a static list of nodes; for every (mode, batch, H, W) a *launch program* — a flat
list of pre-built C-ABI calls (descriptor structs built once) — is compiled and replayed.  All
arithmetic runs in libviddet_hip.so; torch provides device memory, streams and (for N>1 ranks)
torch.distributed collectives only.  There is no autograd tape: the backward program is the fixed
reverse schedule of this one graph.
"""
import ctypes as C
import math
import re
from collections import OrderedDict

import numpy as np
import torch

from . import lib as L
from . import ops
from .lib import ConvDesc, WgradDesc, EPI_AFFINE, EPI_LEAKY, EPI_RESIDUAL
from .ops import round_up, fwd_taps, dgrad_plans

from .consts import ANCHORS, STRIDES, BN_EPS, BN_MOMENTUM, LEAKY_SLOPE   # wrappers.py:80-84, layers.py:68-69


class Slot:
    """Late-bound pointer argument of a launch record (set right before a program runs)."""

    def __init__(self):
        self.value = None


class Program:
    """A flat replayable list of C-ABI launches; the stream is bound at run time (the last argument of
    every vd_* entry point), so a program can be replayed eagerly or inside a HIP graph capture.
    A record may name a side stream (`stream=`) and python records (`add_py`) carry event record / wait
    calls, which is how the weight-gradient GEMMs run beside the rest of the backward pass."""

    def __init__(self):
        self.recs = []
        self.meta = []          # per-record info (kind, algorithmic flops) for the roofline accounting
        self.keep = []          # descriptor structs / tensors the records point into
        self.streams = []       # per-record side stream (torch.cuda.Stream) or None = current stream

    def add(self, fname, *args, meta=None, stream=None):
        fn = getattr(L.load(), fname)
        self.recs.append((fname, fn, args))
        self.meta.append(meta)
        self.streams.append(stream)

    def add_py(self, fn):
        self.recs.append((None, fn, ()))
        self.meta.append(None)
        self.streams.append(None)

    def add_coll(self, fn):
        """a collective (SyncBN's statistics all-reduce) as a record of the program: enqueued in place between the launches
        it sits between - the replay loop never leaves the program for it - and replayed by run_timed as well, so the ranks
        of a job stay in step whichever replay form they run"""
        self.recs.append((None, fn, ()))
        self.meta.append(dict(kind='collective'))
        self.streams.append(None)

    def hold(self, *objs):
        self.keep.extend(objs)

    def run(self):
        s = L.stream_ptr()
        for (fname, fn, args), st in zip(self.recs, self.streams):
            if fname is None:
                fn()
                continue
            a = [x.value if isinstance(x, Slot) else x for x in args]
            rc = fn(*a, s if st is None else C.c_void_p(st.cuda_stream))
            if rc != 0:
                L.check(rc, fname)

    def run_timed(self, select):
        """Replay (everything on the current stream, side streams ignored so that each launch is timed alone)
        with a (start, stop) event pair around every record whose entry point is in `select`.
        Returns [(fname, meta, start, stop)]."""
        s = L.stream_ptr()
        out = []
        for (fname, fn, args), meta in zip(self.recs, self.meta):
            if fname is None:
                if meta and meta.get('kind') == 'collective':
                    fn()
                continue
            a = [x.value if isinstance(x, Slot) else x for x in args]
            if fname in select:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*a, s)
                e1.record()
                if fname in ("vd_conv_igemm", "vd_conv_wgrad"):     # which product arithmetic this record runs in
                    d = args[0]._obj
                    meta = dict(meta or {}, split=bool(d.flags & L.MATH_SPLIT), bf16=bool(d.flags & L.MATH_BF16),
                                f16x2=bool(d.flags & L.MATH_F16X2), nohalo=bool(d.flags & L.MATH_NOHALO), tile=int(getattr(d, "tile", 0)),
                                streamk=bool(fname == "vd_conv_igemm" and L.load().vd_conv_igemm_streamk(args[0])))
                    if fname == "vd_conv_wgrad":                # the halo-ring kernel (vd_wgrad_halo.hip) or the generic one
                        meta["wgrad_halo"] = bool(L.load().vd_conv_wgrad_uses_halo(args[0]))
                    if fname == "vd_conv_igemm" and meta.get("bytes"):
                        # operands the fused epilogue reads besides input / weights: the residual (or accumulated
                        # gradient) rows, and the producer's z rows of the fused BatchNorm-backward reductions
                        out_bytes = 4.0 * d.N * d.Hg * d.Wg * d.Co
                        extra = (out_bytes if d.residual else 0.0) + (out_bytes if getattr(d, "bs_part", None) else 0.0)
                        meta["bytes_epilogue_reads"] = extra
                        meta["bytes"] = meta["bytes"] + extra
                if fname == "vd_conv_igemm_bf16":
                    form = int(L.load().vd_conv_igemm_bf16_streamk(args[0], args[1]))
                    meta = dict(meta or {}, tile=int(args[0]._obj.tile), nohalo=bool(args[0]._obj.flags & L.MATH_NOHALO),
                                streamk=form == 1, splitk=form == 2)
                out.append((fname, meta, e0, e1))
            else:
                rc = fn(*a, s)
            if rc != 0:
                L.check(rc, fname)
        return out


class TuneCache(dict):
    """The plan-time autotuner's choices (launch-record signature -> arithmetic / tile), persisted.

    A choice made by timing differs between boxes and runs (launch-to-launch noise under the power limit is of the order
    of the differences between tiles), and with it the summation order of every convolution: two runs of one seed were not
    bit-identical, ranks of one job could disagree, and every process start paid the tuning again (38 of the 39.5 s of a
    driver bench run).  So the table lives in a JSON file: `VD_TUNE_CACHE` (a path), default
    viddet_amd/tune/gfx950_<hash>.json where <hash> identifies the kernel sources the choices were timed on (another
    library build never reads them).  Read on first use, written (atomically, by rank 0) whenever a plan build added
    entries; `VD_TUNE_CACHE=off` keeps it in memory only.  Under torch.distributed every rank adopts rank 0's choice for
    a new entry (`agree`), so the ranks of a job run the same kernels."""

    def __init__(self):
        super().__init__()
        self._loaded, self._dirty, self.tuned, self.hits, self._synced_world = False, False, 0, 0, 1

    @staticmethod
    def library_id():
        import hashlib
        import os
        here = os.path.dirname(os.path.abspath(__file__))
        # the conv kernels' sources (the public header is not part of it: declarations of other entry points change there
        # without touching a tile; descriptor layouts are guarded by the ABI revision)
        srcs = [os.path.join(here, "csrc", f) for f in ("vd_conv.hip", "vd_conv_igemm.h", "vd_conv_sk.hip", "vd_conv_par.hip",
                                                         "vd_conv_bf16.hip", "vd_conv_igemm_bf16.h", "vd_conv_bf16_sk.hip", "vd_conv_c32_bf16.hip",
                                                         "vd_wgrad_halo.hip", "vd_common.h")]
        h = hashlib.sha256()
        if all(os.path.exists(f) for f in srcs):
            for f in srcs:
                h.update(open(f, "rb").read())
        else:
            h.update(open(L.LIB_PATH, "rb").read())
        # library switches that change what a timed launch runs (the default keeps the hash of the sources alone)
        for sw in ("VD_WGRAD_RESERVE",):
            if os.environ.get(sw):
                h.update(("%s=%s" % (sw, os.environ[sw])).encode())
        return h.hexdigest()[:12]

    def path(self):
        import os
        p = os.environ.get("VD_TUNE_CACHE", "")
        if p.lower() in ("off", "0", "none"):
            return None
        if p:
            return p
        return os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune", "gfx950_%s.json" % self.library_id())

    @staticmethod
    def _world():
        if not torch.distributed.is_available() or not torch.distributed.is_initialized():
            return 1
        return torch.distributed.get_world_size()

    @staticmethod
    def device_arch():
        """gcnArchName of the current device without its feature suffixes ('gfx950'), None without a GPU"""
        if not torch.cuda.is_available():
            return None
        return str(torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName).split(":")[0]

    def _read_file(self):
        import ast
        import json
        import os
        out = {}
        p = self.path()
        if p is None or not os.path.exists(p):
            return out
        try:
            doc = json.load(open(p))
            if doc.get("library") not in (None, self.library_id()):
                return out                               # timed on other kernels
            arch = self.device_arch()
            if arch is not None and doc.get("device") not in (None, arch):
                return out                               # timed on another device
            for k, v in doc.get("entries", {}).items():
                out[ast.literal_eval(k)] = tuple(v) if isinstance(v, list) else v
        except (OSError, ValueError, SyntaxError) as e:   # an unreadable table is a cold start, not an error
            print("viddet_amd: ignoring tuning table %s (%s)" % (p, e), flush=True)
            return {}
        return out

    def load(self):
        """Read the table.  Under torch.distributed ONLY rank 0 reads the file and every rank adopts its entries (one
        broadcast at first use): ranks that read their own copies - another node's file system, another VD_TUNE_CACHE,
        a file rank 0 has rewritten since - would disagree on hit or miss, and the ranks that miss would enter `agree`'s
        broadcast alone.  After this every lookup gives the same answer on every rank, so `agree` is entered by all or none."""
        import os
        self._loaded = True
        world = self._world()
        self._synced_world = world
        if world > 1 and os.environ.get("VD_TUNE_AGREE", "1") != "0":
            box = [self._read_file() if torch.distributed.get_rank() == 0 else None]
            torch.distributed.broadcast_object_list(box, src=0)
            dict.clear(self)                              # rank 0's table, nothing else
            entries = box[0]
        else:
            entries = self._read_file()
        for key, v in entries.items():
            if not dict.__contains__(self, key):
                dict.__setitem__(self, key, v)

    def __contains__(self, key):
        # (a process group that came up after the first lookup: sync once more - every rank is in the same position)
        if not self._loaded or self._synced_world != self._world():
            self.load()
        hit = dict.__contains__(self, key)
        self.hits += int(hit)
        return hit

    def __setitem__(self, key, value):
        dict.__setitem__(self, key, value)
        self._dirty = True
        self.tuned += 1

    def clear(self):
        dict.clear(self)
        self._loaded = False                             # entries on disk come back on the next lookup

    def agree(self, best):
        """rank 0's choice for a new entry, on every rank (all ranks build the same plans in the same order)."""
        import os
        if os.environ.get("VD_TUNE_AGREE", "1") == "0" or not torch.distributed.is_available() or \
                not torch.distributed.is_initialized() or torch.distributed.get_world_size() == 1:
            return best
        box = [best]
        torch.distributed.broadcast_object_list(box, src=0)
        return tuple(box[0]) if isinstance(box[0], list) else box[0]

    def save(self):
        import json
        import os
        if not self._dirty:
            return
        self._dirty = False
        p = self.path()
        if p is None:
            return
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_rank() != 0:
            return
        try:
            os.makedirs(os.path.dirname(p) or ".", exist_ok=True)
            doc = {"library": self.library_id(), "device": self.device_arch() or "gfx950",
                   "entries": {repr(k): (list(v) if isinstance(v, tuple) else v) for k, v in sorted(self.items(), key=lambda kv: repr(kv[0]))}}
            tmp = "%s.%d.tmp" % (p, os.getpid())
            with open(tmp, "w") as f:
                json.dump(doc, f, indent=0)
            os.replace(tmp, p)
        except OSError as e:
            print("viddet_amd: could not write tuning table %s (%s)" % (p, e), flush=True)


_TUNE_CACHE = TuneCache()


def _parity_streams():
    """the four parity launches of a stride-2 data gradient on four streams (+1 % in same-box A/B); read when a plan is
    built, so a test can build both forms in one process"""
    return __import__('os').environ.get('VD_PARITY_STREAMS', '1') == '1'


def _fuse_bwd_s2():
    """A/B switch: fused BN-backward reductions in stride-2 data gradients (read at plan-build time)"""
    return __import__('os').environ.get('VD_FUSE_BWD_S2', '1') == '1'



def fp32_math():
    """Arithmetic of the fp32 convolution products (include/viddet_hip.h VD_MATH_SPLIT): 'native' = fp32 MFMA,
    'split' = three-way bf16 operand split on the bf16 matrix pipe (fp32-accurate), 'auto' = the
    plan-time autotuner times both per launch record and keeps the faster."""
    import os
    m = _MATH_OVERRIDE[0] or os.environ.get("VD_FP32_MATH", "auto")
    if m not in _MATH_MODES:
        raise ValueError("VD_FP32_MATH must be one of %s, got %r" % ("|".join(_MATH_MODES), m))
    return m


_MATH_OVERRIDE = [None]
# 'split2' = two-way fp16 operand split with per-tensor power-of-two scales (VD_MATH_F16X2: three MFMAs per product block
# instead of six, fp32-accurate); 'auto' times native, split and split2 per launch record
_MATH_MODES = ("native", "split", "split2", "auto", "bf16")
_ALL_MATH = L.MATH_SPLIT | L.MATH_BF16 | L.MATH_F16X2 | L.MATH_NOHALO | L.CONV_STREAMK


def _streamk_on():
    """VD_STREAMK=0: never the persistent stream-K form of k_conv_igemm (read when a plan is built).  The form changes how a
    launch is cut into workgroups, not one bit of what it computes (vd_conv_sk.hip), so this is a speed switch only."""
    return __import__('os').environ.get('VD_STREAMK', '1') != '0'


def _streamk_applies(d, fl, tile):
    """would vd_conv_igemm run this record as a stream-K grid with arithmetic `fl` and tile `tile`?"""
    if not (d.sk_ws and (fl & L.MATH_F16X2)):
        return False
    keep = (d.flags, d.tile)
    d.flags, d.tile = (d.flags & ~_ALL_MATH) | fl | L.CONV_STREAMK, tile
    try:
        return bool(L.load().vd_conv_igemm_streamk(C.byref(d)))
    finally:
        d.flags, d.tile = keep


def set_conv_math(mode):
    """Process-wide product arithmetic of the fp32-tensor convolutions for programs built from now on: None (the
    VD_FP32_MATH environment, default 'auto'), 'native', 'split', 'auto', or 'bf16' = products on bf16-rounded
    operands with fp32 accumulation (VD_MATH_BF16; the mixed-precision training arithmetic, bf16-accurate)."""
    assert mode is None or mode in _MATH_MODES
    _MATH_OVERRIDE[0] = mode


def _tile_candidates(d, math):
    """(flags bit, tile) candidates of one launch record."""
    import os
    if d.flags & L.CONV_PARITY4:            # the parity-fused stride-2 data gradient: fp16 split, the tiles it is built for
        return [(L.MATH_F16X2, t) for t in (5, 1, 6, 2, 12, 11)]
    cands = []
    if math in ("native", "auto"):
        if d.Co <= 32:
            cands += [(0, 7)]
        elif d.Co <= 64:
            cands += [(0, 6), (0, 8)]
        else:
            cands += [(0, t) for t in (1, 2, 3, 4, 5)]
    fls = []
    if math in ("split", "auto"):
        fls.append(L.MATH_SPLIT)
    if math in ("split2", "auto"):
        # the fp16 split needs the max-abs slots of both operands and has no in-load transform
        fls.append(L.MATH_F16X2 if (d.amax_in and d.amax_w and not d.in_scale) else L.MATH_SPLIT)
    if math == "bf16":
        fls.append(L.MATH_BF16)
    # the halo-staged loop (3x3 stride-1 geometry) is timed against the generic one
    if L.MATH_F16X2 in fls and os.environ.get("VD_HALO_AB", "1") == "1" and d.T == 9 and d.in_stride == 1 and d.Hg == d.Hi and d.Co > 32:
        fls.append(L.MATH_F16X2 | L.MATH_NOHALO)
    for fl in dict.fromkeys(fls):
        if d.Co <= 32:           # 256 x 32 tiles (9: 32x32x16 MFMA, 10: 16x16x32)
            cands += [(fl, t) for t in (9, 10, 13, 14)]
        elif d.Co <= 64:         # split tiles 1..4 on the 32x32x16 MFMA, 5..8 the same tiles on 16x16x32
            cands += [(fl, t) for t in (3, 4, 7, 8)] + ([(fl, t) for t in (15, 16)] if not (fl & L.MATH_F16X2 and not fl & L.MATH_NOHALO and d.T == 9 and d.in_stride == 1 and d.Hg == d.Hi) else [])
        else:
            # 11, 12: 128x128 as four waves (two workgroups per CU); never the halo loop, so only with the generic bit
            cands += [(fl, t) for t in (1, 2, 3, 4, 5, 6, 7, 8)] + ([(fl, t) for t in (11, 12)] if not (fl & L.MATH_F16X2 and not fl & L.MATH_NOHALO and d.T == 9 and d.in_stride == 1 and d.Hg == d.Hi) else [])
    return cands


def autotune_desc(d, reps=3):
    """Pick the fastest k_conv_igemm variant (product arithmetic x tile shape) for one launch descriptor (timed in
    place on its own buffers with events on the launch stream; cached per problem signature).  VD_AUTOTUNE=0 keeps
    the kernel's heuristic tile.  Tuning launches only rewrite buffers every real run rewrites first."""
    import os
    math = fp32_math()
    base = d.flags & ~_ALL_MATH
    if os.environ.get("VD_AUTOTUNE", "1") == "0":
        f16 = L.MATH_F16X2 if (d.amax_in and d.amax_w and not d.in_scale) else L.MATH_SPLIT
        d.flags = base | {"split": L.MATH_SPLIT, "split2": f16, "bf16": L.MATH_BF16}.get(math, 0)
        if d.flags & L.MATH_F16X2 and d.T >= 9 and d.sk_ws and _streamk_on():
            d.flags |= L.CONV_STREAMK                     # (the library ignores it where the form does not apply)
        if os.environ.get("VD_TILE_ALT", "0") == "1":
            # a second, equally deterministic tile set (the last candidate of the launch record's list instead of the
            # kernel's heuristic tile): other M-tile heights, hence other groupings of the BatchNorm partial sums and
            # split-K slabs - the control of tests/test_training_loop_gpu.py
            alt = [t for fl, t in _tile_candidates(d, math if math != "auto" else "native") if fl == (d.flags & _ALL_MATH)]
            if alt:
                d.tile = alt[-1]
        return
    lib = L.load()
    s = L.stream_ptr()
    key = (math, d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride, d.T, d.Co, d.out_stride, base, bool(d.in_scale),
           bool(d.stats_part), bool(d.amax_in and d.amax_w), bool(d.amax_out), d.Kfr, bool(d.bs_part))
    if key in _TUNE_CACHE:
        fl, d.tile = _TUNE_CACHE[key]
        d.flags = base | fl
        return
    cands = _tile_candidates(d, math)
    best = cands[0]
    if len(cands) > 1:
        def time_of(fl, c, n):
            d.flags, d.tile = base | fl, c
            L.check(lib.vd_conv_igemm(C.byref(d), s), 'vd_conv_igemm/tune')
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                lib.vd_conv_igemm(C.byref(d), s)
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / n
            if os.environ.get("VD_TUNE_VERBOSE") == "1":
                print("igemm tune %s math %d tile %d: %.4f ms (%d launches)" % (key[1:11], fl, c, t, n), flush=True)
            return t
        ranked = sorted((time_of(fl, c, reps), (fl, c)) for fl, c in cands)
        # the persistent stream-K form of the three fastest tiles (same bits, another cut of the launch into workgroups)
        if _streamk_on():
            sk = [(fl | L.CONV_STREAMK, c) for _, (fl, c) in ranked[:3] if _streamk_applies(d, fl, c)]
            ranked = sorted(ranked + [(time_of(fl, c, reps), (fl, c)) for fl, c in sk])
        best = ranked[0][1]
        # candidates within 4 % of the fastest are re-timed with more launches: the first pass is 3 launches each, and
        # launch-to-launch noise under the power limit is of that order
        close = [fc for t, fc in ranked if t <= 1.04 * ranked[0][0]][:3]
        if len(close) > 1:
            best = min((time_of(fl, c, 3 * reps), (fl, c)) for fl, c in close)[1]
    best = _TUNE_CACHE.agree(tuple(best))
    d.flags, d.tile = base | best[0], best[1]
    _TUNE_CACHE[key] = best


def _wgrad_halo_applies(d, flags):
    """would vd_conv_wgrad run the halo-ring kernel (vd_wgrad_halo.hip) for this record with `flags` | VD_WGRAD_HALO?
    The library ignores the flag elsewhere; this keeps such launches out of the timing loop."""
    keep = d.flags
    d.flags = flags | L.WGRAD_HALO
    try:
        return bool(L.load().vd_conv_wgrad_uses_halo(C.byref(d)))
    finally:
        d.flags = keep


def autotune_wgrad_bf16(d, ws_ptr, ws_bytes, reps=2):
    """bf16-stored operands: generic kernel vs the halo ring, timed in place (the flag is ignored where it does not apply)"""
    import os
    d.flags = L.STORE_BF16 | L.MATH_BF16
    if os.environ.get("VD_WGRAD_HALO", "1") != "1" or not _wgrad_halo_applies(d, L.STORE_BF16 | L.MATH_BF16):
        return
    if os.environ.get("VD_AUTOTUNE", "1") == "0":
        d.flags |= L.WGRAD_HALO
        return
    key = ('wgrad_bf16', d.N, d.Hi, d.Wi, d.Ci, d.Co)
    if key not in _TUNE_CACHE:
        lib = L.load()
        s = L.stream_ptr()
        best, best_t = 0, None
        for fl in (0, L.WGRAD_HALO):
            d.flags = L.STORE_BF16 | L.MATH_BF16 | fl
            L.check(lib.vd_conv_wgrad(C.byref(d), ws_ptr, ws_bytes, s), 'vd_conv_wgrad/tune')
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.vd_conv_wgrad(C.byref(d), ws_ptr, ws_bytes, s)
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1)
            if os.environ.get("VD_TUNE_VERBOSE") == "1":
                print("wgrad bf16 tune %s halo %d: %.4f ms" % (key[1:], fl, t / reps), flush=True)
            if best_t is None or t < best_t:
                best, best_t = fl, t
        _TUNE_CACHE[key] = _TUNE_CACHE.agree(best)
    d.flags = L.STORE_BF16 | L.MATH_BF16 | _TUNE_CACHE[key]


def autotune_wgrad(d, ws_ptr, ws_bytes, reps=2):
    """Product arithmetic of one weight-gradient launch record (fp32 MFMA vs the split forms), timed in place."""
    import os
    math = fp32_math()
    d.flags = 0
    if d.Co < 64 or math == "native":
        return
    if math == "bf16":
        d.flags = L.MATH_BF16
        return
    f16 = L.MATH_F16X2 if (d.amax_in and d.amax_dout and not d.in_scale) else L.MATH_SPLIT
    # the halo-ring kernel (vd_wgrad_halo.hip) where the library takes it: 3x3 / stride 1 / Co >= 128, fp16 split
    halo = L.WGRAD_HALO if (f16 == L.MATH_F16X2 and os.environ.get("VD_WGRAD_HALO", "1") == "1" and _wgrad_halo_applies(d, f16)) else 0
    if math in ("split", "split2") or os.environ.get("VD_AUTOTUNE", "1") == "0":
        d.flags = (f16 | halo) if math in ("split2", "auto") else L.MATH_SPLIT
        return
    key = ('wgrad', d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride, d.T, d.Co, d.Kfr, bool(d.in_scale), f16 | halo)
    if key not in _TUNE_CACHE:
        lib = L.load()
        s = L.stream_ptr()
        best, best_t = 0, None
        for fl in dict.fromkeys((0, L.MATH_SPLIT, f16, f16 | halo)):
            d.flags = fl
            L.check(lib.vd_conv_wgrad(C.byref(d), ws_ptr, ws_bytes, s), 'vd_conv_wgrad/tune')
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.vd_conv_wgrad(C.byref(d), ws_ptr, ws_bytes, s)
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1)
            if os.environ.get("VD_TUNE_VERBOSE") == "1":
                print("wgrad tune %s flags %d: %.4f ms" % (key[1:11], fl, t / reps), flush=True)
            if best_t is None or t < best_t:
                best, best_t = fl, t
        _TUNE_CACHE[key] = _TUNE_CACHE.agree(best)
    d.flags = _TUNE_CACHE[key]


def _tune_bf16_record(d, of32, key, tiles, halo_geo, splitk=False):
    """Time the tile variants of one vd_conv_igemm_bf16 launch record in place - generic loop, halo-staged loop where it
    exists, and the persistent stream-K form of the three fastest (VD_CONV_STREAMK: same bits) - and keep the best as
    (tile, flag bits) under `key`.  splitk (inference plans only - the form is deterministic but has another summation
    order than the one-tile launch): launches with too few tiles for the chip also time the split-K grid (VD_CONV_SPLITK)
    of every tile that has one."""
    import os
    lib = L.load()
    s = L.stream_ptr()
    mask = L.MATH_NOHALO | L.CONV_STREAMK | L.CONV_SPLITK
    base = d.flags & ~mask
    if key not in _TUNE_CACHE:
        verbose = os.environ.get("VD_TUNE_VERBOSE") == "1"

        def time_of(c, fl):
            d.tile, d.flags = c, base | fl
            L.check(lib.vd_conv_igemm_bf16(C.byref(d), of32, s), 'vd_conv_igemm_bf16/tune')
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                lib.vd_conv_igemm_bf16(C.byref(d), of32, s)
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1)
            if verbose:
                fl_ = 2.0 * d.N * d.Hg * d.Wg * d.Co * d.T * d.Ci * 3 / t / 1e9
                print("bf16 tune %s tile %d %s%s: %.3f ms %.0f TF" % (key[1:10], c, "generic" if fl & L.MATH_NOHALO else "halo",
                                                                   " split-K" if fl & L.CONV_SPLITK else " stream-K" if fl & L.CONV_STREAMK else "",
                                                                   t / 3, fl_), flush=True)
            return t
        ranked = []
        for c in tiles:
            for fl in ((0, L.MATH_NOHALO) if (halo_geo and c in (2, 3, 5, 6, 7, 10)) else (L.MATH_NOHALO,)):
                ranked.append((time_of(c, fl), (c, fl)))
        ranked.sort()
        if _streamk_on() and d.sk_ws and not of32:
            classic = list(ranked)
            for _, (c, fl) in classic[:3]:
                d.tile, d.flags = c, base | fl | L.CONV_STREAMK
                if lib.vd_conv_igemm_bf16_streamk(C.byref(d), of32) == 1:
                    ranked.append((time_of(c, fl | L.CONV_STREAMK), (c, fl | L.CONV_STREAMK)))
            if splitk and os.environ.get("VD_SPLITK", "1") != "0":
                for _, (c, fl) in classic:
                    d.tile, d.flags = c, base | fl | L.CONV_SPLITK
                    if lib.vd_conv_igemm_bf16_streamk(C.byref(d), of32) == 2:
                        ranked.append((time_of(c, fl | L.CONV_SPLITK), (c, fl | L.CONV_SPLITK)))
            ranked.sort()
        _TUNE_CACHE[key] = _TUNE_CACHE.agree(ranked[0][1] if ranked else (2, L.MATH_NOHALO))
    d.tile, d.flags = _TUNE_CACHE[key][0], base | _TUNE_CACHE[key][1]


def autotune_program(prog, reps=3):
    for (fname, fn, args) in prog.recs:
        if fname == 'vd_conv_igemm':
            autotune_desc(args[0]._obj, reps)
    _TUNE_CACHE.save()


class Parameter:
    """Gluon-Parameter-like handle (name, shape in the reference's layout, wd_mult, lr_mult, grad_req).  The three
    attributes are live, as in Gluon: `grad_req = 'null'` (wrappers.py:55-57 freeze_base) removes the parameter from the
    backward schedule and from the optimiser (no gradient, no update, no weight decay); `wd_mult` / `lr_mult` scale
    the optimiser's wd / lr for this parameter (train_yolov3.py:495-497 sets wd_mult = 0 with --no_wd)."""

    def __init__(self, net, name, shape, kind, node=None, trainable=True):
        self._net, self.name, self.shape, self.kind, self.node = net, name, tuple(shape), kind, node
        self._wd_mult, self._lr_mult = 1.0, 1.0
        self._grad_req = 'write' if trainable else 'null'
        self.storage = None      # view into an arena (device layout)
        self.grad_storage = None
        self.span = None         # [lo, hi) of the parameter arena (None: not in the arena, e.g. running statistics)

    @property
    def wd_mult(self):
        return self._wd_mult

    @wd_mult.setter
    def wd_mult(self, v):
        self._wd_mult = float(v)
        self._net._opt_ranges = None

    @property
    def lr_mult(self):
        return self._lr_mult

    @lr_mult.setter
    def lr_mult(self, v):
        self._lr_mult = float(v)
        self._net._opt_ranges = None

    @property
    def grad_req(self):
        return self._grad_req

    @grad_req.setter
    def grad_req(self, v):
        if v not in ('write', 'null'):
            raise NotImplementedError("grad_req %r: only 'write' and 'null' are built (the reference uses no 'add')" % (v,))
        if self.span is None and v != 'null':
            raise ValueError("%s is an auxiliary state (no gradient)" % self.name)
        if v != self._grad_req:
            self._grad_req = v
            self._net._grad_req_changed()

    # reference layout <-> device layout (conv weights are kept fwd-packed [Co_pad][T*Ci])
    def data(self):
        if self.kind == 'conv_weight':
            out = torch.empty(self.shape, device=self.storage.device)
            ops.unpack_weight(self.storage, out)
            return out
        if self.kind == 'stem_weight':
            co = self.shape[0]
            return self.storage.view(co, 32)[:, :27].reshape(co, 3, 3, 3).permute(0, 3, 1, 2).contiguous()
        return self.storage[:int(np.prod(self.shape))].view(self.shape).clone()

    def grad(self):
        if self.kind == 'conv_weight':
            out = torch.empty(self.shape, device=self.grad_storage.device)
            ops.unpack_weight(self.grad_storage, out)
            return out
        if self.kind == 'stem_weight':
            co = self.shape[0]
            return self.grad_storage.view(co, 32)[:, :27].reshape(co, 3, 3, 3).permute(0, 3, 1, 2).contiguous()
        return self.grad_storage[:int(np.prod(self.shape))].view(self.shape).clone()

    def set_data(self, value):
        v = torch.as_tensor(np.asarray(value.detach().cpu() if torch.is_tensor(value) else value), dtype=torch.float32)
        assert tuple(v.shape) == self.shape, "%s: shape %s != %s" % (self.name, tuple(v.shape), self.shape)
        v = v.to(self.storage.device).contiguous()
        if self.kind == 'conv_weight':
            co_pad = self.storage.numel() // (int(np.prod(self.shape[1:])))
            ops.pack_weight_fwd(v, self.storage, co_pad)
        elif self.kind == 'stem_weight':
            co = self.shape[0]
            tmp = torch.zeros(co, 32, device=v.device)
            tmp[:, :27] = v.permute(0, 2, 3, 1).reshape(co, 27)
            self.storage.copy_(tmp.view(-1))
        else:
            self.storage.zero_()
            self.storage[:v.numel()].copy_(v.view(-1))
        self._net._params_changed()


class ParameterDict(OrderedDict):
    def reset_ctx(self, ctx=None):     # train_yolov3.py:494 — parameters already live on this rank's GPU
        return None

    def select(self, pattern):
        rx = re.compile(pattern)
        out = ParameterDict()
        for k, v in self.items():
            if rx.match(k):
                out[k] = v
        return out


class ConvNode:
    def __init__(self, name, src, dst, cin, cout, k, stride, div_in, bn=True, residual=None, stem=False, head=False,
                 kd=1, fr=1):
        self.name, self.src, self.dst = name, src, dst
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.pad = k // 2
        self.kd, self.pad_d = kd, kd // 2          # temporal kernel depth (Conv3D over the K frames of a window)
        self.fr = fr                               # frames per sample carried by this node's tensors (K or 1)
        self.div_in = div_in                       # input spatial = H / div_in
        self.div_out = div_in * stride
        self.bn, self.residual, self.stem, self.head = bn, residual, stem, head
        self.co_pad = round_up(cout, 32) if head else cout
        self.ci_eff = 32 if stem else cin           # stem weights are kept as [32][32]: 27 taps x channels padded to 32
        self.T = 1 if stem else kd * k * k

    def taps(self):
        return [(0, 0, 0)] if self.stem else fwd_taps(self.k, self.pad, self.kd, self.pad_d)

    def weight_shape(self):
        if self.kd > 1 or getattr(self, 'conv3d', False):
            return (self.cout, self.cin, self.kd, self.k, self.k)
        return (self.cout, self.cin, self.k, self.k)


class UpcatNode:
    def __init__(self, name, up, route, dst, cu, cr, div_out, fr=1):
        self.name, self.up, self.route, self.dst, self.cu, self.cr, self.div_out = name, up, route, dst, cu, cr, div_out
        self.fr = fr


class PoolNode:
    """TemporalPooling over the K frames of a window (layers.py:161-205): (B*K, h, w, C) -> (B, h, w, C)."""

    def __init__(self, name, src, dst, K, type_):
        # type 0 = max, 1 = mean, 2 = 'cat' (channel stacking: dst has K * C channels)
        self.name, self.src, self.dst, self.K = name, src, dst, K
        self.type = {'max': 0, 'mean': 1, 'cat': 2}[type_]


class SelNode:
    """x.slice_axis(axis=1, begin=k0, end=k0+kc) on folded frames (yolo3_temporal.py:437-446): frames [k0, k0+kc) of every
    K-frame window of `src` -> `dst` (kc frames per window)."""

    def __init__(self, name, src, dst, K, k0, kc):
        self.name, self.src, self.dst, self.K, self.k0, self.kc = name, src, dst, K, k0, kc


class AddNode:
    """dst = a + b (yolo3_temporal.py:440,445: the per-frame stage output plus the strided 2+1-D side branch)."""

    def __init__(self, name, a, b, dst, fr):
        self.name, self.a, self.b, self.dst, self.fr = name, a, b, dst, fr


def _feature_name(f):
    if f < 15:
        return "stages.0.%d" % f
    if f < 24:
        return "stages.1.%d" % (f - 15)
    return "stages.2.%d" % (f - 24)


ROUTE_TENSORS = (('f14', 256, 8), ('f23', 512, 16), ('f28', 1024, 32))   # features[:15], [15:24], [24:] (wrappers.py:58)


def build_graph(num_class, k=1, k_join_type=None, k_join_pos=None, block_conv_type='2', noback=False,
                temporal_out=False, temporal_side=False):
    """Node list of YOLOV3T over Darknet-53 (wrappers.py:54-58,101-103; three_darknet.py:252-258;
    yolo3.py:1003-1054 wiring, :1095-1177 forward).  k>1: the backbone is TimeDistributed (K frames folded
    into the batch, layers.py:241-250); 'early' joins pool each stage output over K, 'late' joins keep K frames
    through the neck (2-D blocks per frame, or 3-D / 2+1-D convs across the K axis, yolo3.py:229-262) and pool
    the tips right before the prediction convs (:1134-1138)."""
    nodes, tensors = [], OrderedDict()     # tensors[name] = (channels, div, pitch, frames)
    K = k if k and k > 1 else 1
    # temporal_out = YOLOV3Temporal with t_out (yolo3_temporal.py:396-470, corr_d = 0): every one of the t frames runs
    # through the backbone, the detection blocks (per frame, or 3-D / 2+1-D convs across the t axis) and its OWN
    # prediction convs - no join; TimeDistributed wrappers are created inside hybrid_forward there, so parameter
    # names carry no `.model`
    # temporal_side = YOLOV3Temporal with t_out=False (yolo3_temporal.py:326-333,436-447): the three Darknet stages run on
    # 5, 3 and 1 frames of the window; strided 2+1-D side branches (convs1 / convs2: a per-frame 3x3 stride-2 conv, then a
    # (3,1,1) conv WITHOUT temporal padding, 5 -> 3 and 3 -> 1 frames) carry the neighbours' features down and are added to
    # the stage outputs; the routes are the centre frames and the neck / heads are the plain single-frame ones.
    late = K > 1 and (k_join_pos == 'late' or temporal_out)
    td_names = K > 1 and not temporal_out and not temporal_side

    def T(name, c, div, ld=None, fr=1):
        tensors[name] = (c, div, c if ld is None else ld, fr)
        return name

    def sname(f):
        nm = _feature_name(f)
        if td_names:
            head, idx = nm.rsplit(".", 1)
            nm = "%s.model.%s" % (head, idx)
        return nm

    routes = []
    if noback:
        # YOLOV3_noback (yolo3.py:1730-1840): the three backbone feature maps are the inputs (net(x1, x2, x3)); only
        # transitions / yolo_blocks / yolo_outputs exist, under the same structural names as in the full network
        assert K == 1, "YOLOV3_noback has no temporal window"
        routes = [T(nm, c_, d_) for nm, c_, d_ in ROUTE_TENSORS]
    else:
        T('in', 3, 1, fr=K)
        cur = T('f0', 32, 1, fr=K)
        nodes.append(ConvNode(sname(0), 'in', cur, 3, 32, 3, 1, 1, stem=True, fr=K))
    f = 1
    div = 1
    sfr = K                                # frames the current Darknet stage runs on
    side = None                            # output of the side branch to add to the next stage output

    def side_branch(i, src, cin, cout, d, fr):
        """convs{i} = _conv21d(channel=cout, t=3, d=3, m=cin, padding=[1,0], stride=[(1,2,2),1]) on `src` (fr frames)."""
        sp = T('cx%d.s' % i, cin, d * 2, fr=fr)
        n1 = ConvNode("convs%d.0.0" % i, src, sp, cin, cin, 3, 2, d, fr=fr)        # (1,3,3) stride (1,2,2), BN + LeakyReLU
        n1.conv3d = True
        nodes.append(n1)
        out = T('cx%d' % i, cout, d * 2, fr=fr - 2)
        n2 = ConvNode("convs%d.0.1" % i, sp, out, cin, cout, 1, 1, d * 2, kd=3, fr=fr)   # (3,1,1), no temporal padding
        n2.tvalid = True
        nodes.append(n2)
        return out

    for gi, (nlayer, ch) in enumerate(zip([] if noback else [1, 2, 8, 8, 4], [64, 128, 256, 512, 1024])):
        if temporal_side and gi in (3, 4):
            i = gi - 2                                                               # side branch 1 / 2
            if gi == 3:
                routes.append(T('r0', tensors[cur][0], div))
                nodes.append(SelNode('sel.r0', cur, 'r0', sfr, 2, 1))                # :437 the centre frame
            side = side_branch(i, cur, tensors[cur][0], ch, div, sfr)
            if gi == 3:
                nxt = T(cur + '.s', tensors[cur][0], div, fr=3)
                nodes.append(SelNode('sel.s1', cur, nxt, sfr, 1, 3))                 # :439 frames 1..3
                cur, sfr = nxt, 3
            else:
                cur, sfr = routes[1], 1                                              # :444 frame 1 of 3 (= route 1)
        nxt = T('f%d' % f, ch, div * 2, fr=sfr)
        nodes.append(ConvNode(sname(f), cur, nxt, ch // 2, ch, 3, 2, div, fr=sfr))
        cur, div, f = nxt, div * 2, f + 1
        for _ in range(nlayer):
            mid = T('f%d.m' % f, ch // 2, div, fr=sfr)
            nxt = T('f%d' % f, ch, div, fr=sfr)
            nm = sname(f)
            nodes.append(ConvNode(nm + ".body.0", cur, mid, ch, ch // 2, 1, 1, div, fr=sfr))
            nodes.append(ConvNode(nm + ".body.1", mid, nxt, ch // 2, ch, 3, 1, div, residual=cur, fr=sfr))
            cur, f = nxt, f + 1
        if temporal_side and gi in (3, 4):
            nxt = T(cur + '.a', ch, div, fr=sfr)
            nodes.append(AddNode('add.%d' % (gi - 2), cur, side, nxt, sfr))          # :440,445 x = x + cx
            cur = nxt
            if gi == 3:
                routes.append(T('r1', ch, div))
                nodes.append(SelNode('sel.r1', cur, 'r1', 3, 1, 1))                  # :441 the middle of the three
            else:
                routes.append(cur)                                                   # :446 squeeze: one frame left
        elif f in (15, 24, 29) and not temporal_side:
            routes.append(cur)
    nfr = K if late else 1                 # frames carried through the neck
    if K > 1 and not late and not temporal_side:      # 'early' join (yolo3.py:1107-1124): pool every route over K
        pooled = []
        for i, r in enumerate(routes):
            c_, d_ = tensors[r][0], tensors[r][1]
            pr = T('route%d.pool' % i, c_ * (K if k_join_type == 'cat' else 1), d_)
            nodes.append(PoolNode('pool.route%d' % i, r, pr, K, k_join_type))
            pooled.append(pr)
        routes = pooled
    # neck + heads, deepest first (yolo3.py:1013-1054, 1126-1177)
    A = 3 * (5 + num_class)
    x, xc = routes[2], tensors[routes[2]][0]
    heads = []
    conv3 = block_conv_type in ('3', '21')
    for i, c in enumerate([512, 256, 128]):
        d = 32 >> i
        if conv3:
            pre, cell = "yolo_blocks.%d" % i, ".conv"           # Conv wrapper registers `.conv` (layers.py:135-158)
        elif late and td_names:
            pre, cell = "yolo_blocks.%d.model" % i, ""           # TimeDistributed(block)
        else:
            pre, cell = "yolo_blocks.%d" % i, ""

        def add_cell(name, src, dst_name, cin, cout, ksz):
            """one Conv+BN+LeakyReLU cell of the detection block; 3x3 cells become 3x3x3 ('3') or
            (1,3,3)+(3,1,1) ('21') across the K frames when block_conv_type asks for it"""
            if conv3 and ksz == 3 and block_conv_type == '3':
                dst = T(dst_name, cout, d, fr=nfr)
                nodes.append(ConvNode(name + cell, src, dst, cin, cout, 3, 1, d, kd=3, fr=nfr))
                return dst
            if conv3 and ksz == 3:       # R(2+1)D: spatial then temporal, each with BN + LeakyReLU (layers.py:82-89)
                mid = T(dst_name + ".s", cout, d, fr=nfr)
                n1 = ConvNode(name + cell + ".0", src, mid, cin, cout, 3, 1, d, fr=nfr)
                n1.conv3d = True
                nodes.append(n1)
                dst = T(dst_name, cout, d, fr=nfr)
                n2 = ConvNode(name + cell + ".1", mid, dst, cout, cout, 1, 1, d, kd=3, fr=nfr)
                nodes.append(n2)
                return dst
            dst = T(dst_name, cout, d, fr=nfr)
            n1 = ConvNode(name + cell, src, dst, cin, cout, ksz, 1, d, fr=nfr)
            n1.conv3d = conv3              # 1x1x1 Conv3D: same arithmetic as the per-frame 1x1, 5-D weight
            nodes.append(n1)
            return dst

        for j in range(5):
            cout = c if j % 2 == 0 else 2 * c
            x = add_cell("%s.body.%d" % (pre, j), x, 'n%d.b%d' % (i, j), xc, cout, 1 if j % 2 == 0 else 3)
            xc = cout
        route = x
        tip = add_cell(pre + ".tip", route, 'n%d.tip' % i, c, 2 * c, 3)
        tipc = 2 * c
        if late and not temporal_out:                           # yolo3.py:1134-1138: join the tip over K
            tipc = 2 * c * (K if k_join_type == 'cat' else 1)
            ptip = T('n%d.tip.pool' % i, tipc, d)
            nodes.append(PoolNode('pool.tip%d' % i, tip, ptip, K, k_join_type))
            tip = ptip
        hfr = K if temporal_out else 1                          # per-frame predictions (TimeDistributed(output))
        hd = T('head%d' % i, A, d, round_up(A, 32), fr=hfr)
        nodes.append(ConvNode("yolo_outputs.%d.prediction" % i, tip, hd, tipc, A, 1, 1, d, bn=False, head=True, fr=hfr))
        heads.append(hd)
        if i < 2:
            tr = T('n%d.tr' % i, c // 2, d, fr=nfr)
            nodes.append(ConvNode("transitions.%d%s" % (i, ".model" if (late and td_names) else ""), route, tr, c, c // 2,
                                  1, 1, d, fr=nfr))
            rt = routes[1 - i]
            rc = tensors[rt][0]
            cat = T('n%d.cat' % i, c // 2 + rc, d // 2, fr=nfr)
            nodes.append(UpcatNode("upcat.%d" % i, tr, rt, cat, c // 2, rc, d // 2, fr=nfr))
            x, xc = cat, c // 2 + rc
    return nodes, tensors, heads


class YOLOV3(object):
    """The network object the reference's loops drive (net(x) / net(x, *targets), set_nms, reset_class,
    collect_params, save/load_parameters, hybridize, initialize).  k=1 (YOLOV3T with k=1 == YOLOV3)."""

    def __init__(self, classes, nms_thresh=0.45, nms_topk=400, post_nms=100, ignore_iou_thresh=0.7,
                 device="cuda", syncbn_scope=None, process_group=None, k=1, k_join_type=None, k_join_pos=None,
                 block_conv_type='2', noback=False, temporal_out=False, temporal_side=False):
        self._classes = list(classes)
        self.temporal_side = bool(temporal_side)  # YOLOV3Temporal(t_out=False): strided 2+1-D side branches, one output
        self.temporal_out = bool(temporal_out)   # YOLOV3Temporal(t_out=True): per-frame detections / losses
        self._grad_scale = 1.0                   # d(reported loss)/d(sum of per-sample losses), see _forward_train
        self.noback = bool(noback)               # YOLOV3_noback: net(x1, x2, x3[, targets]) on cached backbone features
        self._k = k if k and k > 1 else 1
        self._k_join_type, self._k_join_pos, self._block_conv_type = k_join_type, k_join_pos, block_conv_type
        self.nms_thresh, self.nms_topk, self.post_nms = nms_thresh, nms_topk, post_nms
        self._ignore_iou_thresh = ignore_iou_thresh
        self._label_smooth = False
        self._target_generator = self            # train_yolov3.py:499-500 pokes net._target_generator._label_smooth
        self.device = torch.device(device)
        self.syncbn_scope = syncbn_scope         # None | 'reference' | 'all'
        self.process_group = process_group
        self._training = False
        self._recording = False
        self._programs = {}
        self._sk_ws = {}           # stream-K hand-off workspaces by stream index (_set_streamk)
        self._range_exact = set()  # operand tensors whose consumers never run the fp16-split arithmetic (see _build)
        self._fold_dirty = True
        self._stats_version = 1          # bumped whenever the BatchNorm running statistics move
        self._weights_version = 1        # bumped whenever the weights move; each training plan packs its own data-gradient
        self._pack_stream = None         # weight layouts and remembers the version they were packed from
        self._pack_event = None
        self._graph_cache = {}
        self.use_graphs = False
        import os as _os
        self.overlap_wgrad = _os.environ.get('VD_OVERLAP', '1') != '0'   # wgrad GEMMs on a side stream (_build_train)
        self.fuse_bn_stats = _os.environ.get('VD_FUSE_STATS', '1') != '0'  # BN statistics in the conv epilogue
        self.precision = 'fp32'        # inference precision: 'fp32' | 'bf16' (set_precision)
        self.bucketed_allreduce = _os.environ.get('VD_BUCKETED', '1') != '0'
        self.alias_skip_grad = _os.environ.get('VD_ALIAS_SKIP', '1') != '0'    # skip gradients by alias, not by copy
        self.fuse_bn_bwd = _os.environ.get('VD_FUSE_BWD', '1') != '0'          # BN backward reductions in the dgrad epilogue
        # fp32 gradients per all-reduce.  The tail bucket (everything below stage 4: ~15 M parameters, the layers whose
        # gradients come last) closes with the stem and is the one collective nothing overlaps, so buckets are kept at
        # 32 MB: large enough for the ring to reach its bandwidth, small enough that the exposed tail stays ~30 MB.
        self.bucket_elems = int(float(_os.environ.get('VD_BUCKET_MB', '32')) * (1 << 18))
        self._pending_reduces = []
        self._reduced_from = 1 << 62
        self._dp_stats = {'buckets': 0, 'bucket_bytes': 0}     # all-reduce buckets queued since the last reset (bench.py)
        self._bucket_group = None      # second communicator for the gradient buckets when SyncBN collectives exist
        self._build(len(self._classes))

    # ------------------------------------------------------------------ construction
    def _build(self, num_class):
        self.num_class = num_class
        self.nodes, self.tensors, self.head_names = build_graph(num_class, self._k, self._k_join_type,
                                                                self._k_join_pos, self._block_conv_type,
                                                                noback=self.noback, temporal_out=self.temporal_out,
                                                                temporal_side=self.temporal_side)
        self._head_frames = self._k if self.temporal_out else 1
        self.input_tensors = [nm for nm, _, _ in ROUTE_TENSORS] if self.noback else ['in']
        self.conv_nodes = [n for n in self.nodes if isinstance(n, ConvNode)]
        # The loss gradient (dhead: (sigmoid - target) x mask) is sparse and saturated - after a few hundred steps most of its
        # entries sit 2^20 .. 2^30 below its largest ones, and an output of its consumers that reads only such entries (a
        # background pixel of the data gradient) would be formed from operands the fp16 split has staged with a handful of
        # bits.  Its consumers - the data and weight gradients of the three prediction convs, 2 % of the step's FLOPs - are
        # therefore range-exact BY CONSTRUCTION (3-way bf16 split / fp32 MFMA), never chosen by timing.  Every other operand
        # tensor is dense; those join this set through check_operand_ranges() when their channel scales spread too far.
        self._range_exact = set('dz:' + n.name for n in self.conv_nodes if n.head)
        self._guard = None
        # arena layout: [conv weights (fwd-packed) | bn gamma, beta, head bias]  -> wd / no_wd ranges
        off = 0
        for n in self.conv_nodes:
            n.w_off = off
            n.w_numel = n.co_pad * n.T * n.ci_eff
            off += round_up(n.w_numel, 64)
        self.n_weight = off
        for n in self.conv_nodes:
            if n.bn:
                n.gamma_off, n.beta_off = off, off + round_up(n.cout, 64)
                off += 2 * round_up(n.cout, 64)
            else:
                n.bias_off = off
                off += round_up(n.co_pad, 64)
        self.n_params = off
        dev = self.device
        self.weights = torch.zeros(off, device=dev)
        self.grads = torch.zeros(off, device=dev)
        self.momentum_buf = torch.zeros(off, device=dev)
        soff = 0
        for n in self.conv_nodes:
            if n.bn:
                n.stat_off = soff
                soff += 2 * round_up(n.cout, 64)
        self.running = torch.zeros(max(soff, 1), device=dev)
        # max-abs of every conv weight tensor (operand scale of the VD_MATH_F16X2 arithmetic): one slot set per conv,
        # refreshed by ONE launch whenever the weights moved (_refresh_wamax)
        self._wamax = torch.zeros(len(self.conv_nodes) * L.AMAX_FLOATS, device=dev)
        self._wamax_seg = torch.tensor([[n.w_off, n.w_numel] for n in self.conv_nodes], dtype=torch.int64, device=dev)
        for i, n in enumerate(self.conv_nodes):
            n.wamax = self._wamax[i * L.AMAX_FLOATS:(i + 1) * L.AMAX_FLOATS]
        self._wamax_dirty = True
        # per-node derived buffers: folded scale/shift (eval) and batch scale/shift/mean/invstd (train)
        self.aux = torch.zeros(max(1, sum(6 * round_up(n.cout, 64) for n in self.conv_nodes if n.bn)), device=dev)
        self.sums = torch.zeros(max(1, sum(4 * round_up(n.cout, 64) for n in self.conv_nodes)), dtype=torch.float64,
                                device=dev)
        aoff = doff = 0
        for n in self.conv_nodes:
            c64 = round_up(n.cout, 64)
            if n.bn:
                v = self.aux[aoff:aoff + 6 * c64]
                n.fold_scale, n.fold_shift = v[0:n.cout], v[c64:c64 + n.cout]
                n.b_scale, n.b_shift = v[2 * c64:2 * c64 + n.cout], v[3 * c64:3 * c64 + n.cout]
                n.b_mean, n.b_invstd = v[4 * c64:4 * c64 + n.cout], v[5 * c64:5 * c64 + n.cout]
                aoff += 6 * c64
            n.sums = self.sums[doff:doff + 2 * c64]
            n.sums2 = self.sums[doff + 2 * c64:doff + 4 * c64]
            doff += 4 * c64
        self._make_params()

    def _make_params(self):
        P = ParameterDict()

        def reg(name, shape, kind, node, storage, gstorage, trainable=True, off=None):
            p = Parameter(self, name, shape, kind, node, trainable)
            p.storage, p.grad_storage = storage, gstorage
            if off is not None:
                p.span = (off, off + round_up(storage.numel(), 64))
            P[name] = p
            return p

        for n in self.conv_nodes:
            wv = self.weights[n.w_off:n.w_off + n.w_numel]
            gv = self.grads[n.w_off:n.w_off + n.w_numel]
            n.wp, n.gwp = wv, gv
            if n.head:
                reg(n.name + ".weight", (n.cout, n.cin, 1, 1), 'conv_weight', n, wv, gv, off=n.w_off)
                n.bias = self.weights[n.bias_off:n.bias_off + n.co_pad]
                n.gbias = self.grads[n.bias_off:n.bias_off + n.co_pad]
                reg(n.name + ".bias", (n.cout,), 'vector', n, n.bias, n.gbias, off=n.bias_off)
            else:
                kind = 'stem_weight' if n.stem else 'conv_weight'
                reg(n.name + ".0.weight", n.weight_shape(), kind, n, wv, gv, off=n.w_off)
                n.gamma = self.weights[n.gamma_off:n.gamma_off + n.cout]
                n.beta = self.weights[n.beta_off:n.beta_off + n.cout]
                n.ggamma = self.grads[n.gamma_off:n.gamma_off + n.cout]
                n.gbeta = self.grads[n.beta_off:n.beta_off + n.cout]
                c64 = round_up(n.cout, 64)
                n.rmean = self.running[n.stat_off:n.stat_off + n.cout]
                n.rvar = self.running[n.stat_off + c64:n.stat_off + c64 + n.cout]
                reg(n.name + ".1.gamma", (n.cout,), 'vector', n, n.gamma, n.ggamma, off=n.gamma_off)
                reg(n.name + ".1.beta", (n.cout,), 'vector', n, n.beta, n.gbeta, off=n.beta_off)
                reg(n.name + ".1.running_mean", (n.cout,), 'vector', n, n.rmean, None, trainable=False)
                reg(n.name + ".1.running_var", (n.cout,), 'vector', n, n.rvar, None, trainable=False)
        self._params = P
        self._opt_ranges = None

    # ------------------------------------------------------------------ reference-style surface
    @property
    def classes(self):
        return self._classes

    def collect_params(self, select=None):
        return self._params if select is None else self._params.select(select)

    def hybridize(self, active=True):
        """The reference compiles a CachedOp here (train_yolov3.py:586); programs are always pre-compiled."""
        return None

    def set_nms(self, nms_thresh=0.45, nms_topk=400, post_nms=100):
        # yolo3.py:1208-1228
        self.nms_thresh, self.nms_topk, self.post_nms = nms_thresh, nms_topk, post_nms
        self._programs = {k: v for k, v in self._programs.items() if k[0] not in ('infer', 'infer_bf16')}
        self._graph_cache.clear()

    def initialize(self, init='uniform', seed=233, obj_bias=0.0):
        """'uniform': MXNet default Uniform(0.07) for conv weights, gamma=1, beta=0, bias=0, running mean 0 / var 1.
        'he': N(0, 2/(k*k*Cin)) conv weights (keeps synthetic activations O(1))."""
        g = torch.Generator(device='cpu').manual_seed(seed)
        for name, p in self._params.items():
            if name.endswith('weight'):
                if init == 'uniform':
                    v = (torch.rand(p.shape, generator=g) * 2 - 1) * 0.07
                else:
                    fan = int(np.prod(p.shape[1:]))
                    v = torch.randn(p.shape, generator=g) * math.sqrt(2.0 / fan)
                    if 'prediction' in name:
                        v *= 0.05     # raw box logits O(1): exp(raw_wh)*anchor stays a sane pixel size
                p.set_data(v)
            elif name.endswith('gamma') or name.endswith('running_var'):
                v = torch.ones(p.shape)
                if init == 'he' and name.endswith('gamma') and '.body.1.1.' in name and name.startswith('stages'):
                    v *= 0.3      # residual-branch gain < 1: 23 residual adds would otherwise double the activation
                                  # variance per block in eval mode (running var = 1) and saturate every logit
                p.set_data(v)
            elif name.endswith('bias'):
                v = torch.zeros(p.shape)
                if obj_bias != 0.0:
                    v.view(3, -1)[:, 4] = obj_bias
                p.set_data(v)
            else:
                p.set_data(torch.zeros(p.shape))
        self.momentum_buf.zero_()
        self.grads.zero_()

    def reset_class(self, classes, reuse_weights=None):
        """yolo3.py:1230-1302 + YOLOOutputV3.reset_class :76-129: rebuild the 3 prediction convs."""
        old_classes, old = self._classes, {k: p.data().cpu() for k, p in self._params.items()}
        old_attr = {k: (p.grad_req, p.wd_mult, p.lr_mult) for k, p in self._params.items()}
        old_npred = 5 + len(old_classes)
        if isinstance(reuse_weights, (dict, list)):
            if isinstance(reuse_weights, dict):
                m = {}
                for k, v in reuse_weights.items():
                    if isinstance(v, str):
                        if v not in old_classes:
                            raise ValueError("{} not found in old class names {}".format(v, old_classes))
                        v = old_classes.index(v)
                    elif v < 0 or v >= len(old_classes):
                        raise ValueError("Index {} out of bounds for old class names".format(v))
                    if isinstance(k, str):
                        if k not in classes:
                            raise ValueError("{} not found in new class names {}".format(k, classes))
                        k = list(classes).index(k)
                    elif k < 0 or k >= len(classes):
                        raise ValueError("Index {} out of bounds for new class names".format(k))
                    m[k] = v
                reuse_weights = m
            else:
                reuse_weights = {list(classes).index(x): old_classes.index(x) for x in reuse_weights
                                 if x in classes and x in old_classes}
        self._classes = list(classes)
        self._programs.clear()
        self._graph_cache.clear()
        self._build(len(self._classes))
        new_npred = 5 + len(self._classes)
        g = torch.Generator(device='cpu').manual_seed(233)
        for name, p in self._params.items():
            if "yolo_outputs" not in name:
                p.set_data(old[name])
                # every parameter but the rebuilt prediction convs is the same object in the reference: it keeps its
                # grad_req ('null' under freeze_base) and multipliers
                if p.span is not None:
                    p.grad_req, p.wd_mult, p.lr_mult = old_attr[name]
                continue
            new = (torch.rand(p.shape, generator=g) * 2 - 1) * 0.07 if name.endswith('weight') else torch.zeros(p.shape)
            if reuse_weights:
                od = old[name]
                for k, v in reuse_weights.items():
                    for a in range(3):
                        new[5 + k + a * new_npred] = od[5 + v + a * old_npred]
                        new[a * new_npred:a * new_npred + 5] = od[a * old_npred:a * old_npred + 5]
            p.set_data(new)

    def _grad_req_changed(self):
        """A parameter was frozen / unfrozen: the backward schedule and the optimiser ranges are rebuilt on next use."""
        self._programs = {k: v for k, v in self._programs.items() if k[0] != 'train'}
        self._opt_ranges = None

    def _node_trainable(self, n):
        """(weight trainable, any of gamma / beta / bias trainable) of a conv node."""
        P = self._params
        if n.head:
            return P[n.name + ".weight"].grad_req != 'null', P[n.name + ".bias"].grad_req != 'null'
        return (P[n.name + ".0.weight"].grad_req != 'null',
                P[n.name + ".1.gamma"].grad_req != 'null' or P[n.name + ".1.beta"].grad_req != 'null')

    def _optimizer_ranges(self):
        """Contiguous arena ranges [(lo, hi, lr_mult, wd_mult)] of the trainable parameters, adjacent parameters with
        equal multipliers merged: two ranges (weights | gamma, beta, bias) for a fully trainable network."""
        if self._opt_ranges is None:
            ps = sorted((p for p in self._params.values() if p.span is not None and p.grad_req != 'null'),
                        key=lambda p: p.span[0])
            out = []
            for p in ps:
                lo, hi = p.span
                if out and out[-1][1] == lo and out[-1][2] == p.lr_mult and out[-1][3] == p.wd_mult and \
                        (lo != self.n_weight):
                    out[-1][1] = hi
                else:
                    out.append([lo, hi, p.lr_mult, p.wd_mult])
            self._opt_ranges = [tuple(r) for r in out]
        return self._opt_ranges

    def _params_changed(self):
        self._fold_dirty = True
        self._weights_version += 1
        self._wamax_dirty = True

    def _refresh_wamax(self):
        if self._wamax_dirty:
            L.check(L.load().vd_amax_segments(self.weights.data_ptr(), self._wamax_seg.data_ptr(), len(self.conv_nodes),
                                              self._wamax.data_ptr(), L.stream_ptr()), 'vd_amax_segments')
            self._wamax_dirty = False

    # ------------------------------------------------------------------ buffers
    def _buffers(self, key, B, H, W, train):
        ck = ('buf', B, H, W, train)
        if ck in self._programs:
            return self._programs[ck]
        dev = self.device
        bufs = {}
        for name, (c, div, ld, fr) in self.tensors.items():
            if name == 'in':
                bufs['in'] = torch.empty(B * fr, 3, H, W, device=dev)      # (B,K,3,H,W) folded: frame n = b*K + k
                continue
            bufs[name] = torch.empty(B * fr, H // div, W // div, ld, device=dev)
        if self.noback:
            for nm, c_, d_ in ROUTE_TENSORS:                                   # NCHW staging of the three inputs
                bufs['in:' + nm] = torch.empty(B, c_, H // d_, W // d_, device=dev)
        # max-abs slots (operand scales of the fp16-split arithmetic): one set per activation tensor, and in training
        # per conv node for the gradient dz it consumes; zeroed by the first record of every forward program
        names = [nm for nm in self.tensors if nm != 'in'] + (['dz:' + n.name for n in self.conv_nodes] if train else [])
        bufs['amax'] = torch.zeros(len(names) * L.AMAX_FLOATS, device=dev)
        for i, nm in enumerate(names):
            bufs['amax:' + nm] = bufs['amax'][i * L.AMAX_FLOATS:(i + 1) * L.AMAX_FLOATS]
        for n in self.conv_nodes:
            if getattr(n, 'tvalid', False):        # the 'same'-padded temporal conv output the valid frames are taken from
                bufs['zf:' + n.dst] = torch.empty(B * n.fr, H // n.div_out, W // n.div_out, n.cout, device=dev)
                if train:
                    bufs['dzf:' + n.dst] = torch.empty_like(bufs['zf:' + n.dst])
                else:
                    bufs['zs:' + n.dst] = torch.empty_like(bufs[n.dst])
        if train:
            for n in self.nodes:
                if isinstance(n, PoolNode) and n.type == 0:
                    bufs['am:' + n.dst] = torch.empty(bufs[n.dst].shape, dtype=torch.int32, device=dev)
            for n in self.conv_nodes:
                if n.bn:
                    bufs['z:' + n.dst] = torch.empty_like(bufs[n.dst])
            for name in self.tensors:
                if name != 'in':
                    bufs['d:' + name] = torch.empty_like(bufs[name])
            mx = max(bufs[n.dst].numel() for n in self.conv_nodes)
            bufs['dz'] = torch.empty(mx, device=dev)
            bufs['dz2'] = torch.empty(mx, device=dev)
            bufs['tmp'] = torch.empty(mx, device=dev)
        # The plan-time autotuner times candidate kernels in place on these buffers.  Fresh allocations are zero pages, and on
        # all-zero operands the chip holds a ~20 % higher clock - for every candidate, but not equally: stand-alone the 256x128
        # tiles on the two MFMA shapes tie at 310 TF on zeros and differ by 8 % (255 vs 275) on real data.  Tune on noise.
        for nm, t in bufs.items():
            if torch.is_tensor(t) and t.dtype == torch.float32 and not nm.startswith('amax'):
                t.normal_()
        self._programs[ck] = bufs
        return bufs

    def _grid(self, H, W):
        assert H == W, "square inputs only (the reference resizes to data_shape x data_shape)"
        assert H % 32 == 0, "input side must be a multiple of 32"
        return [H // 32, H // 16, H // 8]

    # ------------------------------------------------------------------ program builders
    def _conv_desc(self, n, bufs, B, H, W, out, *, scale=None, shift=None, residual=None, leaky=False, amax_out=False):
        d = ConvDesc()
        x = bufs[n.src]
        Hi, Wi = H // n.div_in, W // n.div_in
        Ho, Wo = H // n.div_out, W // n.div_out
        d.in_, d.wp, d.out = x.data_ptr(), n.wp.data_ptr(), out.data_ptr()
        d.N, d.Hi, d.Wi, d.Ci = B * n.fr, Hi, Wi, n.ci_eff
        d.Hg, d.Wg, d.in_stride = Ho, Wo, n.stride
        ops._set_taps(d, n.taps())
        d.Kfr = n.fr if n.kd > 1 else 1
        d.Ho, d.Wo, d.Co = Ho, Wo, n.co_pad
        d.out_stride, d.out_oy, d.out_ox = 1, 0, 0
        d.ldo = d.ldr = n.co_pad
        flags = 0
        if scale is not None or shift is not None:
            flags |= EPI_AFFINE
            d.scale = scale.data_ptr() if scale is not None else None
            d.shift = shift.data_ptr() if shift is not None else None
        if leaky:
            flags |= EPI_LEAKY
        if residual is not None:
            flags |= EPI_RESIDUAL
            d.residual = residual.data_ptr()
        d.flags, d.slope = flags, LEAKY_SLOPE
        d.amax_in, d.amax_w = self._amax_or_none(bufs, n.src), n.wamax.data_ptr()
        self._set_streamk(d, 0)
        if amax_out:                                   # inference: the epilogue publishes the max-abs of what it writes
            d.amax_out = bufs['amax:' + n.dst].data_ptr()
        return d

    def _amax_or_none(self, bufs, name):
        """max-abs slots of an operand tensor for the fp16-split arithmetic, or None where that arithmetic must not be used
        for its consumers (`_range_exact`): without the slots a launch record runs the 3-way bf16 split or the fp32 MFMA,
        whose operands keep fp32's exponent range"""
        if name in self._range_exact:
            return None
        return bufs['amax:' + name].data_ptr()

    def check_operand_ranges(self, thresh=2.0 ** -16):
        """The data-driven half of the fp16 split's range rule (include/viddet_hip.h vd_range_guard).  One
        small launch over the BatchNorm vectors of the network, then a read of its flags (synchronises: call it where the
        host waits anyway - train_yolov3.py does at every log line).  A tensor whose per-channel scales spread over more
        than 1 / thresh joins `_range_exact`; the plans are dropped and rebuilt with the range-exact arithmetic for its
        consumers.  Returns {tensor name: min / max channel-scale ratio} of the newly flagged tensors."""
        bn = [n for n in self.conv_nodes if n.bn]
        if not bn:
            return {}
        if getattr(self, '_guard', None) is None:
            arr = (L.GuardItem * len(bn))()
            for i, n in enumerate(bn):
                arr[i].gamma, arr[i].beta, arr[i].scale, arr[i].C = n.gamma.data_ptr(), n.beta.data_ptr(), n.b_scale.data_ptr(), n.cout
            items = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            self._guard = (items, torch.zeros(2 * len(bn), device=self.device), torch.zeros(2 * len(bn), dtype=torch.int32, device=self.device))
        items, ratios, flags = self._guard
        L.check(L.load().vd_range_guard(items.data_ptr(), len(bn), float(thresh), ratios.data_ptr(), flags.data_ptr(), L.stream_ptr()),
                'vd_range_guard')
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size(self.process_group) > 1:
            # every rank takes the union: without SyncBN the ranks' invstd differ, and the ranks must run the same plans
            torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MAX, group=self.process_group)
        fl, ra = flags.cpu().numpy(), ratios.cpu().numpy()
        new = {}
        for i, n in enumerate(bn):
            for k, name in ((0, n.dst), (1, 'dz:' + n.name)):
                if fl[2 * i + k] and name not in self._range_exact:
                    new[name] = float(ra[2 * i + k])
        if new:
            self._range_exact.update(new)
            self._drop_plans()
        return new

    def _set_streamk(self, d, stream_idx):
        """give a conv record the stream-K hand-off workspace of the stream it runs on (0 = the program's main stream,
        1.. = the parity streams of a stride-2 data gradient): launches that may overlap never share one"""
        if not _streamk_on():
            return
        ws = self._sk_ws.get(stream_idx)
        if ws is None:
            ws = self._sk_ws[stream_idx] = ops.streamk_workspace(self.device)
        d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()

    def _syncbn_exchange(self, t):
        """the SyncBN collective of one layer and direction: sum all-reduce of its fp64 [2C] vector over the ranks"""
        def f():
            if getattr(self, '_syncbn_suppress', False):          # (bench.py times a step without the exchanges)
                return
            torch.distributed.all_reduce(t, group=self.process_group)
            if getattr(self, '_dp_stats', None) is not None:
                self._dp_stats['syncbn_exchanges'] = self._dp_stats.get('syncbn_exchanges', 0) + 1
        return f

    def _add_sel_add_fwd(self, prog, n, bufs, B):
        """Forward launches of a SelNode / AddNode (the same in inference and training)."""
        am = lambda t: bufs['amax:' + t].data_ptr()
        if isinstance(n, SelNode):
            xs, o = bufs[n.src], bufs[n.dst]
            prog.add('vd_frame_slice', xs.data_ptr(), o.data_ptr(), B, n.K, n.k0, n.kc, xs[0].numel(), 0)
            prog.add('vd_amax_merge', am(n.src), None, am(n.dst))
        else:
            a, b_, o = bufs[n.a], bufs[n.b], bufs[n.dst]
            prog.add('vd_add', a.data_ptr(), b_.data_ptr(), o.data_ptr(), o.numel())
            prog.add('vd_amax', o.data_ptr(), o.numel(), am(n.dst))

    def _add_amax_reset(self, prog, bufs):
        prog.add('vd_fill', bufs['amax'].data_ptr(), 0.0, bufs['amax'].numel())

    def _add_input_stage(self, prog, bufs, B, H, W):
        """Layout change of the network inputs: none for the frame batch (the stem kernel reads NCHW directly); the
        three cached feature maps of the no-backbone variant go NCHW -> NHWC."""
        self._add_amax_reset(prog, bufs)
        if self.noback:
            for nm, c_, d_ in ROUTE_TENSORS:
                prog.add('vd_nchw_to_nhwc', bufs['in:' + nm].data_ptr(), bufs[nm].data_ptr(), B, c_, H // d_, W // d_)
                prog.add('vd_amax', bufs[nm].data_ptr(), bufs[nm].numel(), bufs['amax:' + nm].data_ptr())

    def _add_stem(self, prog, n, bufs, B, H, W, out, *, scale=None, shift=None, leaky=False, bf16=False, stats=None):
        """vd_stem_conv: the 3 -> 32 stem straight from the NCHW batch (vd_stem.hip)."""
        flags = (EPI_AFFINE if scale is not None else 0) | (EPI_LEAKY if leaky else 0)
        prog.add('vd_stem_conv', bufs['in'].data_ptr(), n.wp.data_ptr(), out.data_ptr(), out.shape[-1], B * self._k, H, W,
                 scale.data_ptr() if scale is not None else None, shift.data_ptr() if shift is not None else None,
                 LEAKY_SLOPE, flags, 1 if bf16 else 0, stats, meta=self._flops(n, B, H, W, 'fwd'))

    def _stage_inputs(self, bufs, x):
        if self.noback:
            for (nm, _, _), t in zip(ROUTE_TENSORS, x):
                bufs['in:' + nm].copy_(t.reshape(bufs['in:' + nm].shape))
        elif x.dtype == torch.uint8:
            # uint8 frames (B,H,W,3) or windows (B,K,H,W,3) straight from the loader: /255, normalise and NHWC -> planar in
            # one kernel (transforms.py:239-245), a quarter of the host-to-device bytes
            n_, h_, w_ = bufs['in'].shape[0], bufs['in'].shape[2], bufs['in'].shape[3]
            assert x.shape[-1] == 3 and x.numel() == n_ * h_ * w_ * 3, "uint8 input must be (B[,K],H,W,3)"
            xd = x.to(self.device).contiguous()
            L.check(L.load().vd_preprocess_u8_nchw(xd.data_ptr(), bufs['in'].data_ptr(), n_, h_, w_, L.stream_ptr()),
                    'vd_preprocess_u8_nchw')
        else:
            bufs['in'].copy_(x.reshape(bufs['in'].shape))

    def _build_infer(self, B, H, W):
        bufs = self._buffers('infer', B, H, W, False)
        prog = Program()
        self._add_input_stage(prog, bufs, B, H, W)
        am = lambda t: bufs['amax:' + t].data_ptr()
        for n in self.nodes:
            if isinstance(n, UpcatNode):
                o = bufs[n.dst]
                prog.add('vd_upsample2x_concat', bufs[n.up].data_ptr(), bufs[n.route].data_ptr(), o.data_ptr(), B * n.fr,
                         o.shape[1], o.shape[2], n.cu, n.cr)
                prog.add('vd_amax_merge', am(n.up), am(n.route), am(n.dst))       # a concatenation: max of the two
                continue
            if isinstance(n, PoolNode):
                o, xs = bufs[n.dst], bufs[n.src]
                if n.type == 2:
                    prog.add('vd_temporal_cat', xs.data_ptr(), o.data_ptr(), B, n.K, xs.shape[1] * xs.shape[2], xs.shape[3], 0)
                else:
                    prog.add('vd_temporal_pool', xs.data_ptr(), o.data_ptr(), None, B, n.K, o[0].numel(), n.type)
                prog.add('vd_amax_merge', am(n.src), None, am(n.dst))             # max / mean / stacking: bounded by the source's
                continue
            if isinstance(n, (SelNode, AddNode)):
                self._add_sel_add_fwd(prog, n, bufs, B)
                continue
            if n.stem:
                self._add_stem(prog, n, bufs, B, H, W, bufs[n.dst], scale=n.fold_scale, shift=n.fold_shift, leaky=True)
                prog.add('vd_amax', bufs[n.dst].data_ptr(), bufs[n.dst].numel(), am(n.dst))
                continue
            if getattr(n, 'tvalid', False):
                # (3,1,1) conv without temporal padding = frames [1, K-1) of the 'same'-padded one; BatchNorm sees those only
                zf, zs, o = bufs['zf:' + n.dst], bufs['zs:' + n.dst], bufs[n.dst]
                d = self._conv_desc(n, bufs, B, H, W, zf)
                prog.hold(d)
                prog.add('vd_conv_igemm', C.byref(d), meta=self._flops(n, B, H, W, 'fwd'))
                prog.add('vd_frame_slice', zf.data_ptr(), zs.data_ptr(), B, n.fr, 1, n.fr - 2, zf[0].numel(), 0)
                prog.add('vd_bn_apply_leaky', zs.data_ptr(), n.fold_scale.data_ptr(), n.fold_shift.data_ptr(), None,
                         o.data_ptr(), o.numel() // n.cout, n.cout, LEAKY_SLOPE, am(n.dst))
                continue
            if n.head:
                d = self._conv_desc(n, bufs, B, H, W, bufs[n.dst], shift=n.bias)
            else:
                res = bufs[n.residual] if n.residual else None
                d = self._conv_desc(n, bufs, B, H, W, bufs[n.dst], scale=n.fold_scale, shift=n.fold_shift, residual=res,
                                    leaky=True, amax_out=True)
            prog.hold(d)
            prog.add('vd_conv_igemm', C.byref(d), meta=self._flops(n, B, H, W, 'fwd'))
        grids = self._grid(H, W)
        Bh = B * self._head_frames           # images the decode / NMS see (B*t with per-frame predictions)
        hd = ops.make_head_desc([bufs[h] for h in self.head_names], grids, round_up(3 * (5 + self.num_class), 32),
                                STRIDES[::-1], ANCHORS[::-1], Bh, self.num_class)
        P = 3 * sum(g * g for g in grids)
        # one candidate slot per row of the reference's (B, C*P, 6) tensor: box_nms (yolo3.py:1197-1202) has no cap, and an
        # untrained net (validation after epoch 0) passes valid_thresh on every row.  8 bytes x C*P per image (14.6 MB at
        # 608x608 / 80 classes) is address space, not traffic: only the rows that pass are ever written or read.
        cap = self.num_class * P
        o = dict(cand_score=torch.empty(Bh, cap, device=self.device),
                 cand_row=torch.empty(Bh, cap, dtype=torch.int32, device=self.device),
                 counts=torch.zeros(Bh, dtype=torch.int32, device=self.device),
                 ids=torch.empty(Bh, self.post_nms, 1, device=self.device),
                 scores=torch.empty(Bh, self.post_nms, 1, device=self.device),
                 bboxes=torch.empty(Bh, self.post_nms, 4, device=self.device),
                 rows=torch.empty(Bh, self.post_nms, dtype=torch.int32, device=self.device),
                 overflow=torch.zeros(Bh, dtype=torch.int32, device=self.device))
        prog.hold(hd, o)
        prog.add('vd_yolo_decode_filter', C.byref(hd), 0.01, o['cand_score'].data_ptr(), o['cand_row'].data_ptr(), cap,
                 o['counts'].data_ptr())
        prog.add('vd_nms_topk', C.byref(hd), o['cand_score'].data_ptr(), o['cand_row'].data_ptr(), cap,
                 o['counts'].data_ptr(), float(self.nms_thresh), int(self.nms_topk), int(self.post_nms),
                 o['ids'].data_ptr(), o['scores'].data_ptr(), o['bboxes'].data_ptr(), o['rows'].data_ptr(),
                 o['overflow'].data_ptr(), 4 * Bh)
        autotune_program(prog)
        return prog, bufs, o

    def _refresh_fold(self):
        if not self._fold_dirty:
            return
        for n in self.conv_nodes:
            if n.bn:
                ops.bn_fold_eval(n.gamma, n.beta, n.rmean, n.rvar, BN_EPS, n.fold_scale, n.fold_shift)
        self._fold_dirty = False

    # ------------------------------------------------------------------ bf16 inference
    def set_precision(self, precision):
        """'fp32' (reference precision) or 'bf16' (BASELINE configs[1]): bf16 storage + bf16 MFMA with fp32
        accumulation and fp32 epilogue for inference; the prediction heads stay fp32."""
        assert precision in ('fp32', 'bf16')
        self.precision = precision

    def _build_infer_bf16(self, B, H, W):
        """bf16 inference plan of every network variant: the plain net (BASELINE configs[1]), k > 1 windows (TimeDistributed
        backbone = frames folded into the batch, temporal pooling / stacking, 3-D and 2+1-D neck convs through Kfr), the
        no-backbone net (its three fp32 feature maps converted on the way in) and YOLOV3Temporal (frame selections, sums,
        valid-frame convs, per-frame heads)."""
        dev = self.device
        lib = L.load()
        BFT = torch.bfloat16
        # channel runs of 64 (one 128-byte K-step); a 32-channel map stays unpadded (two taps per K-step, vd_conv_bf16.hip)
        cp = lambda c: 32 if c == 32 else round_up(c, 64)
        bufs = {}
        for name, (c, div, ld, fr) in self.tensors.items():
            if name == 'in':
                bufs['in'] = torch.empty(B * fr, 3, H, W, device=dev)
            elif name in self.head_names:
                bufs[name] = torch.empty(B * fr, H // div, W // div, ld, device=dev)
            else:
                # zeros: padded channels are never written
                bufs[name] = torch.zeros(B * fr, H // div, W // div, cp(c), dtype=BFT, device=dev)
        prog = Program()
        packs = []
        fuse_stem = None
        if self.noback:
            # the three cached feature maps arrive as fp32 NCHW: NHWC, then one conversion each
            for nm, c_, d_ in ROUTE_TENSORS:
                bufs['in:' + nm] = torch.empty(B, c_, H // d_, W // d_, device=dev)
                bufs['f32:' + nm] = torch.empty(B, H // d_, W // d_, c_, device=dev)
                assert cp(c_) == c_
                prog.add('vd_nchw_to_nhwc', bufs['in:' + nm].data_ptr(), bufs['f32:' + nm].data_ptr(), B, c_, H // d_, W // d_)
                n_el = bufs[nm].numel()
                prog.add('vd_pack_weight_bf16', bufs['f32:' + nm].data_ptr(), bufs[nm].data_ptr(), 1, 1, n_el, n_el, 1)
        for n in self.nodes:
            if isinstance(n, UpcatNode):
                o = bufs[n.dst]
                # 16-byte-unit copy kernel: pass channel counts as if fp32 (bf16 count / 2)
                prog.add('vd_upsample2x_concat', bufs[n.up].data_ptr(), bufs[n.route].data_ptr(), o.data_ptr(), B * n.fr,
                         o.shape[1], o.shape[2], cp(n.cu) // 2, cp(n.cr) // 2)
                continue
            if isinstance(n, PoolNode):
                o, xs = bufs[n.dst], bufs[n.src]
                if n.type == 2:          # stacking the K frames' channels: a copy (bf16 channel count / 2)
                    assert xs.shape[3] % 8 == 0
                    prog.add('vd_temporal_cat', xs.data_ptr(), o.data_ptr(), B, n.K, xs.shape[1] * xs.shape[2], xs.shape[3] // 2, 0)
                else:
                    prog.add('vd_temporal_pool_bf16', xs.data_ptr(), o.data_ptr(), B, n.K, o[0].numel(), n.type)
                continue
            if isinstance(n, SelNode):
                xs, o = bufs[n.src], bufs[n.dst]
                prog.add('vd_frame_slice', xs.data_ptr(), o.data_ptr(), B, n.K, n.k0, n.kc, xs[0].numel() // 2, 0)
                continue
            if isinstance(n, AddNode):
                a_, b_, o = bufs[n.a], bufs[n.b], bufs[n.dst]
                prog.add('vd_add_bf16', a_.data_ptr(), b_.data_ptr(), o.data_ptr(), o.numel())
                continue
            if n.stem:
                # the stem and the stride-2 conv behind it as ONE launch (vd_stem_conv_c32_bf16: the stem's 32-channel map is
                # computed inside the first-stage patch kernel and never stored; same bits) where that conv is the stem's only
                # reader; VD_STEM_FUSED=0: two launches
                users = [m for m in self.nodes if isinstance(m, ConvNode) and (m.src == n.dst or m.residual == n.dst)] + \
                        [m for m in self.nodes if not isinstance(m, ConvNode) and n.dst in
                         [getattr(m, a, None) for a in ('src', 'up', 'route', 'a', 'b')]]
                import os
                fuse_stem = None
                if (os.environ.get("VD_STEM_FUSED", "1") != "0" and len(users) == 1 and isinstance(users[0], ConvNode)
                        and users[0].src == n.dst and users[0].k == 3 and users[0].kd == 1 and users[0].stride == 2
                        and users[0].cin == 32 and users[0].cout == 64 and users[0].bn and not users[0].residual
                        and not getattr(users[0], 'tvalid', False)):
                    fuse_stem = (n, users[0])
                    continue
                # fp32 master weights and BN fold on the fp32 VALU, bf16 output (vd_stem.hip)
                self._add_stem(prog, n, bufs, B, H, W, bufs[n.dst], scale=n.fold_scale, shift=n.fold_shift, leaky=True,
                               bf16=True)
                continue
            tvalid = getattr(n, 'tvalid', False)
            ci_p = cp(n.cin)
            co_p = n.co_pad if n.head else cp(n.cout)
            wb = torch.empty(co_p * n.T * ci_p, dtype=BFT, device=dev)
            packs.append((n, wb, co_p, ci_p))
            d = ConvDesc()
            self._set_streamk(d, 0)
            x = bufs[n.src]
            Hi, Wi = H // n.div_in, W // n.div_in
            Ho, Wo = H // n.div_out, W // n.div_out
            out = bufs[n.dst]
            if tvalid:
                # (3,1,1) conv without temporal padding = frames [1, K-1) of the 'same'-padded one: all frames go through
                # the conv and its (per-channel, batch-independent) BN-fold epilogue, the valid ones are sliced out
                if ('zf:' + n.dst) not in bufs:
                    bufs['zf:' + n.dst] = torch.zeros(B * n.fr, Ho, Wo, co_p, dtype=BFT, device=dev)
                out = bufs['zf:' + n.dst]
            d.in_, d.wp, d.out = x.data_ptr(), wb.data_ptr(), out.data_ptr()
            d.N, d.Hi, d.Wi, d.Ci = B * n.fr, Hi, Wi, ci_p
            d.Hg, d.Wg, d.in_stride = Ho, Wo, n.stride
            ops._set_taps(d, n.taps())
            d.Kfr = n.fr if n.kd > 1 else 1
            d.Ho, d.Wo, d.Co = Ho, Wo, co_p
            d.out_stride, d.out_oy, d.out_ox = 1, 0, 0
            d.ldo = d.ldr = out.shape[-1]
            if n.head:
                d.flags, d.shift = EPI_AFFINE, n.bias.data_ptr()
            else:
                # padded fp32 fold vectors: one pair per NODE, shared by the plans of every input shape (a per-plan pair
                # would go stale in all plans but the one that last refreshed it)
                if getattr(n, 'bf_scale', None) is None or n.bf_scale.numel() != co_p:
                    n.bf_scale, n.bf_shift = torch.zeros(co_p, device=dev), torch.zeros(co_p, device=dev)
                sc, sh = n.bf_scale, n.bf_shift
                d.flags = EPI_AFFINE | EPI_LEAKY | (EPI_RESIDUAL if n.residual else 0)
                d.scale, d.shift = sc.data_ptr(), sh.data_ptr()
                if n.residual:
                    d.residual = bufs[n.residual].data_ptr()
            d.slope = LEAKY_SLOPE
            prog.hold(d, wb)
            if fuse_stem is not None and fuse_stem[1] is n:
                st = fuse_stem[0]
                meta = self._flops(n, B, H, W, 'fwd')
                meta['flops'] += self._flops(st, B, H, W, 'fwd')['flops']
                meta['fused_stem'] = True
                d.tile = 16
                prog.add('vd_stem_conv_c32_bf16', bufs['in'].data_ptr(), st.wp.data_ptr(), st.fold_scale.data_ptr(),
                         st.fold_shift.data_ptr(), LEAKY_SLOPE, C.byref(d), meta=meta)
                continue
            prog.add('vd_conv_igemm_bf16', C.byref(d), 1 if n.head else 0, meta=self._flops(n, B, H, W, 'fwd'))
            if tvalid:
                prog.add('vd_frame_slice', out.data_ptr(), bufs[n.dst].data_ptr(), B, n.fr, 1, n.fr - 2, out[0].numel() // 2, 0)
        grids = self._grid(H, W)
        Bh = B * self._head_frames           # images the decode / NMS see (B*t with per-frame predictions)
        hd = ops.make_head_desc([bufs[h] for h in self.head_names], grids, round_up(3 * (5 + self.num_class), 32),
                                STRIDES[::-1], ANCHORS[::-1], Bh, self.num_class)
        P = 3 * sum(g * g for g in grids)
        cap = self.num_class * P          # no candidate cap (see _build_infer)
        o = dict(cand_score=torch.empty(Bh, cap, device=dev), cand_row=torch.empty(Bh, cap, dtype=torch.int32, device=dev),
                 counts=torch.zeros(Bh, dtype=torch.int32, device=dev), ids=torch.empty(Bh, self.post_nms, 1, device=dev),
                 scores=torch.empty(Bh, self.post_nms, 1, device=dev), bboxes=torch.empty(Bh, self.post_nms, 4, device=dev),
                 rows=torch.empty(Bh, self.post_nms, dtype=torch.int32, device=dev),
                 overflow=torch.zeros(Bh, dtype=torch.int32, device=dev))
        prog.hold(hd, o)
        prog.add('vd_yolo_decode_filter', C.byref(hd), 0.01, o['cand_score'].data_ptr(), o['cand_row'].data_ptr(), cap,
                 o['counts'].data_ptr())
        prog.add('vd_nms_topk', C.byref(hd), o['cand_score'].data_ptr(), o['cand_row'].data_ptr(), cap,
                 o['counts'].data_ptr(), float(self.nms_thresh), int(self.nms_topk), int(self.post_nms),
                 o['ids'].data_ptr(), o['scores'].data_ptr(), o['bboxes'].data_ptr(), o['rows'].data_ptr(),
                 o['overflow'].data_ptr(), 4 * Bh)
        return prog, bufs, o, packs

    def _refresh_bf16(self, packs):
        """Re-derive the bf16 weight images and the padded fp32 scale/shift vectors after a parameter change."""
        self._refresh_fold()
        lib = L.load()
        s = L.stream_ptr()
        for n, wb, co_p, ci_p in packs:
            L.check(lib.vd_pack_weight_bf16(n.wp.data_ptr(), wb.data_ptr(), n.co_pad, co_p, n.ci_eff, ci_p, n.T, s),
                    'vd_pack_weight_bf16')
            if n.bn:
                n.bf_scale[:n.cout].copy_(n.fold_scale)
                n.bf_shift[:n.cout].copy_(n.fold_shift)

    def _tune_bf16(self, prog):
        import os
        if os.environ.get("VD_AUTOTUNE", "1") == "0":
            return
        for (fname, fn, args) in prog.recs:
            if fname != 'vd_conv_igemm_bf16':
                continue
            d, of32 = args[0]._obj, args[1]
            base = d.flags & ~(L.MATH_NOHALO | L.CONV_STREAMK | L.CONV_SPLITK)
            key = ('bf16', d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride, d.T, d.Co, base, of32, d.Kfr)
            one = d.T == 1           # 14 / 15: small four-wave tiles for the HBM-bound 1x1 layers
            # (16: the first-stage patch kernel of vd_conv_c32_bf16.hip - 3x3, 32 -> 64 channels; the library falls back to the
            # default tile where it does not apply)
            tiles = ((10, 11, 13) + ((14,) if one else ()) + ((16,) if (d.T == 9 and d.Co == 64 and not of32) else ()) if d.Ci == 32
                     else (12, 10, 11, 13) + ((15,) if one else ()) if d.Co <= 32
                     else (10, 11, 13, 2, 3, 4, 5) + ((14,) if one else ()) if d.Co <= 64
                     else (1, 2, 3, 4, 5, 6, 7) + ((8, 9) if d.Co > 128 else ()) + ((14,) if one else ()))
            # 3x3 stride-1 'same' geometry: the halo-staged loop (8-wave tiles; the library falls back by itself where it
            # does not apply) is timed against the generic one
            halo_geo = d.T == 9 and d.in_stride == 1 and d.Hg == d.Hi and d.Ci % 64 == 0
            _tune_bf16_record(d, of32, key, tiles, halo_geo, splitk=True)
        _TUNE_CACHE.save()

    # ------------------------------------------------------------------ inference
    def _in_shape(self, x):
        """(B, H, W) of the image batch a call refers to (the no-backbone inputs are the stride-8/16/32 maps)."""
        if self.noback:
            return x[0].shape[0], x[0].shape[-2] * 8, x[0].shape[-1] * 8
        if x.dtype == torch.uint8:                  # (B[,K],H,W,3) frames, normalised on the device
            return x.shape[0], x.shape[-3], x.shape[-2]
        return x.shape[0], x.shape[-2], x.shape[-1]

    def _forward_infer(self, x):
        B, H, W = self._in_shape(x)
        assert 0 < self.nms_thresh < 1, "nms_thresh outside (0,1) (NMS disabled) is not implemented"
        if self.precision == 'bf16':
            key = ('infer_bf16', B, H, W)
            if key not in self._programs:
                built = self._build_or_evict(lambda: self._build_infer_bf16(B, H, W))
                self._refresh_bf16(built[3])
                built[1]['packs_version'] = (self._weights_version, self._stats_version)
                self._tune_bf16(built[0])
                self._programs[key] = built
            prog, bufs, o, packs = self._programs[key]
            # every plan owns its bf16 weight images: refreshed when the weights / running statistics moved since THIS
            # plan last packed them (a net-wide flag would leave the other shapes' plans stale)
            if bufs.get('packs_version') != (self._weights_version, self._stats_version):
                self._refresh_bf16(packs)
                bufs['packs_version'] = (self._weights_version, self._stats_version)
        else:
            key = ('infer', B, H, W)
            if key not in self._programs:
                self._programs[key] = self._build_or_evict(lambda: self._build_infer(B, H, W))
            prog, bufs, o = self._programs[key]
            self._refresh_fold()
        self._stage_inputs(bufs, x)
        self._refresh_wamax()
        if self.use_graphs:
            g = self._graph_cache.get(key)
            if g is None:
                prog.run()                       # warm-up outside capture (function attributes etc.)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    prog.run()
                self._graph_cache[key] = g
            g.replay()
        else:
            prog.run()
        self.last_rows, self.last_overflow = o['rows'], o['overflow']
        if self.temporal_out:                    # (B, t, 100, .) as TimeDistributed(output) stacks them
            K = self._k
            return (o['ids'].view(B, K, -1, 1), o['scores'].view(B, K, -1, 1), o['bboxes'].view(B, K, -1, 4))
        return o['ids'], o['scores'], o['bboxes']

    def extract_features(self, x):
        """extract_base_features.py:127-130: f1 = features[:15](x), f2 = features[15:24](f1), f3 = features[24:](f2) in
        inference mode (BatchNorm on running statistics) -> three NCHW fp32 tensors (B,256,H/8,W/8), (B,512,H/16,W/16),
        (B,1024,H/32,W/32), the inputs of the no-backbone network."""
        if self.noback or self._k > 1:
            raise NotImplementedError("extract_features is the per-frame Darknet-53 trunk of the full k=1 network")
        B, H, W = self._in_shape(x)
        key = ('features', B, H, W)
        if key not in self._programs:
            bufs = self._buffers('infer', B, H, W, False)
            prog = Program()
            self._add_input_stage(prog, bufs, B, H, W)
            last = max(i for i, n in enumerate(self.nodes) if isinstance(n, ConvNode) and n.dst == ROUTE_TENSORS[-1][0])
            for n in self.nodes[:last + 1]:
                if n.stem:
                    self._add_stem(prog, n, bufs, B, H, W, bufs[n.dst], scale=n.fold_scale, shift=n.fold_shift, leaky=True)
                    prog.add('vd_amax', bufs[n.dst].data_ptr(), bufs[n.dst].numel(), bufs['amax:' + n.dst].data_ptr())
                    continue
                res = bufs[n.residual] if n.residual else None
                d = self._conv_desc(n, bufs, B, H, W, bufs[n.dst], scale=n.fold_scale, shift=n.fold_shift, residual=res,
                                    leaky=True, amax_out=True)
                prog.hold(d)
                prog.add('vd_conv_igemm', C.byref(d), meta=self._flops(n, B, H, W, 'fwd'))
            autotune_program(prog)
            self._programs[key] = (prog, bufs)
        prog, bufs = self._programs[key]
        self._refresh_fold()
        self._refresh_wamax()
        self._stage_inputs(bufs, x)
        prog.run()
        return tuple(bufs[nm].permute(0, 3, 1, 2).contiguous() for nm, _, _ in ROUTE_TENSORS)

    # ------------------------------------------------------------------ training forward / backward
    def _syncbn(self, n):
        if not self.syncbn_scope or not torch.distributed.is_available() or not torch.distributed.is_initialized():
            return False
        # (VD_FORCE_DIST=1, bench.py / tests: the exchange runs on a single rank too - the RCCL call pattern of an N-rank job)
        if torch.distributed.get_world_size(self.process_group) < 2 and __import__('os').environ.get("VD_FORCE_DIST", "0") != "1":
            return False
        if self.syncbn_scope == 'all':
            return True
        # 'reference': --syncbn only reaches the stem and the 5 stride-2 convs
        return n.stem or n.stride == 2

    def _build_train(self, B, H, W):
        bufs = self._buffers('train', B, H, W, True)
        dev = self.device
        ws_bytes = 1 << 20
        for n in self.conv_nodes:
            Hi, Wi = H // n.div_in, W // n.div_in
            Ho, Wo = H // n.div_out, W // n.div_out
            if n.stem:
                ws_bytes = max(ws_bytes, int(L.load().vd_stem_wgrad_ws_bytes(B * n.fr, Hi, Wi)))
            else:
                ws_bytes = max(ws_bytes, ops.wgrad_ws_bytes(B * n.fr, Hi, Wi, n.cin, Ho, Wo, n.co_pad, n.k, n.stride,
                                                            n.pad, n.kd, n.pad_d))
            ws_bytes = max(ws_bytes, ops.bn_stats_ws_bytes(B * n.fr * Ho * Wo, n.co_pad))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size(self.process_group)
            # With SyncBN the statistics all-reduces sit on the critical path of backward; on one communicator they
            # would queue behind a gradient bucket in flight (one RCCL stream per communicator).  The buckets get
            # their own communicator (new_group is collective: every rank builds its training plan at the same point).
            if (world > 1 and self.syncbn_scope and self.bucketed_allreduce and self._bucket_group is None
                    and self.process_group is None):
                self._bucket_group = torch.distributed.new_group()

        # partial-sum table of the fused BN statistics: rows = M tiles (>= 64 rows each), 2*Cout floats per row
        smax = 16
        for n in self.conv_nodes:
            if n.bn:
                # (+ 8 rows: the four parity launches of a stride-2 data gradient round their tile counts up separately)
                # (+ 8 + 4 x 64 rows: every parity launch of every frame chunk rounds its tile count up separately)
                smax = max(smax, ((B * n.fr * (H // n.div_out) * (W // n.div_out) + 63) // 64 + 8 + 256) * 2 * n.cout)
        stats_ws = torch.empty(smax, device=dev)
        # ---- forward: list of segments; a segment is a Program or a python callable (collectives)
        fwd, seg = [], Program()
        self._add_input_stage(seg, bufs, B, H, W)

        amx = lambda t: bufs['amax:' + t].data_ptr()
        for n in self.nodes:
            if isinstance(n, UpcatNode):
                o = bufs[n.dst]
                seg.add('vd_upsample2x_concat', bufs[n.up].data_ptr(), bufs[n.route].data_ptr(), o.data_ptr(), B * n.fr,
                        o.shape[1], o.shape[2], n.cu, n.cr)
                seg.add('vd_amax_merge', amx(n.up), amx(n.route), amx(n.dst))
                continue
            if isinstance(n, PoolNode):
                o, xs = bufs[n.dst], bufs[n.src]
                if n.type == 2:
                    seg.add('vd_temporal_cat', xs.data_ptr(), o.data_ptr(), B, n.K, xs.shape[1] * xs.shape[2], xs.shape[3], 0)
                else:
                    am = bufs['am:' + n.dst].data_ptr() if n.type == 0 else None
                    seg.add('vd_temporal_pool', xs.data_ptr(), o.data_ptr(), am, B, n.K, o[0].numel(), n.type)
                seg.add('vd_amax_merge', amx(n.src), None, amx(n.dst))
                continue
            if isinstance(n, (SelNode, AddNode)):
                self._add_sel_add_fwd(seg, n, bufs, B)
                continue
            Ho, Wo = H // n.div_out, W // n.div_out
            tvalid = getattr(n, 'tvalid', False)
            M = B * (n.fr - 2 if tvalid else n.fr) * Ho * Wo
            if n.head:
                d = self._conv_desc(n, bufs, B, H, W, bufs[n.dst], shift=n.bias)
                seg.hold(d)
                seg.add('vd_conv_igemm', C.byref(d), meta=self._flops(n, B, H, W, 'fwd'))
                continue
            z = bufs['z:' + n.dst]
            if n.stem:
                # raw conv output + one row of BatchNorm partial sums per 256-pixel block (vd_stem.hip)
                nb = L.load().vd_stem_conv_blocks(B * n.fr, H, W)
                assert nb * 2 * n.cout <= stats_ws.numel(), "stats workspace too small"
                self._add_stem(seg, n, bufs, B, H, W, z, stats=stats_ws.data_ptr())
                table_rows = nb
                d = None
            else:
                d = self._conv_desc(n, bufs, B, H, W, bufs['zf:' + n.dst] if tvalid else z)
                seg.hold(d)
            if d is None:
                pass
            elif tvalid:
                # 'same'-padded temporal conv on all frames, the valid ones sliced out, statistics over those
                zf = bufs['zf:' + n.dst]
                seg.add('vd_conv_igemm', C.byref(d), meta=self._flops(n, B, H, W, 'fwd'))
                seg.add('vd_frame_slice', zf.data_ptr(), z.data_ptr(), B, n.fr, 1, n.fr - 2, zf[0].numel(), 0)
                seg.add('vd_bn_stats', z.data_ptr(), M, n.cout, n.sums.data_ptr(), ws.data_ptr(), ws_bytes)
                table_rows = None
            elif self.fuse_bn_stats:
                table_rows = None
                # BN statistics ride in the conv epilogue: one row of partial sums per M tile, reduced in fp64
                d.stats_part = stats_ws.data_ptr()
                autotune_desc(d)                                   # fixes the tile, hence the number of M tiles
                mt = L.load().vd_conv_igemm_mtiles(C.byref(d))
                assert mt * 2 * n.cout * 4 <= stats_ws.numel() * 4, "stats workspace too small"
                seg.add('vd_conv_igemm', C.byref(d), meta=self._flops(n, B, H, W, 'fwd'))
                table_rows = mt
            else:
                seg.add('vd_conv_igemm', C.byref(d), meta=self._flops(n, B, H, W, 'fwd'))
                seg.add('vd_bn_stats', z.data_ptr(), M, n.cout, n.sums.data_ptr(), ws.data_ptr(), ws_bytes)
                table_rows = None
            count = float(M)
            fin = (n.gamma.data_ptr(), n.beta.data_ptr(), BN_EPS, BN_MOMENTUM, n.rmean.data_ptr(), n.rvar.data_ptr(),
                   n.b_scale.data_ptr(), n.b_shift.data_ptr(), n.b_mean.data_ptr(), n.b_invstd.data_ptr())
            if table_rows is not None and not self._syncbn(n):
                # partial table -> fp64 sums -> scale / shift / running statistics in ONE launch (short tables)
                seg.add('vd_bn_sum_finalize', stats_ws.data_ptr(), table_rows, n.cout, n.sums.data_ptr(), count, *fin,
                        ws.data_ptr(), ws_bytes)
            else:
                if table_rows is not None:
                    seg.add('vd_bn_sum_partials', stats_ws.data_ptr(), table_rows, n.cout, n.sums.data_ptr(), ws.data_ptr(),
                            ws_bytes)
                if self._syncbn(n):
                    seg.add_coll(self._syncbn_exchange(n.sums))
                    count = float(M * world)
                seg.add('vd_bn_finalize', n.sums.data_ptr(), count, n.cout, *fin)
            res = bufs[n.residual].data_ptr() if n.residual else None
            seg.add('vd_bn_apply_leaky', z.data_ptr(), n.b_scale.data_ptr(), n.b_shift.data_ptr(), res,
                    bufs[n.dst].data_ptr(), M, n.cout, LEAKY_SLOPE, amx(n.dst),
                    meta=dict(node=n.name, bytes=4.0 * M * n.cout * (3 if n.residual else 2),
                              single_conv_consumer=n.dst in self.single_conv_consumer_tensors()))
        # loss (targets are late-bound)
        grids = self._grid(H, W)
        hd = ops.make_head_desc([bufs[h] for h in self.head_names], grids, round_up(3 * (5 + self.num_class), 32),
                                STRIDES[::-1], ANCHORS[::-1], B * self._head_frames, self.num_class)
        slots = dict(gt=Slot(), M=Slot(), obj=Slot(), ctr=Slot(), scl=Slot(), wgt=Slot(), cls=Slot(), smooth=Slot())
        losses = torch.zeros(B * self._head_frames, 4, device=dev)
        dh = (C.c_void_p * 3)(*[bufs['d:' + h].data_ptr() for h in self.head_names])
        head_node = {m.dst: m for m in self.conv_nodes if m.head}
        dha = (C.c_void_p * 3)(*[amx('dz:' + head_node[h].name) for h in self.head_names])   # max-abs of the three dhead
        lws = torch.empty(max(16, ops.yolo_loss_ws_bytes(hd)), dtype=torch.uint8, device=dev)
        seg.hold(hd, dh, dha, lws)
        seg.add('vd_yolo_loss_fwd_bwd', C.byref(hd), slots['gt'], slots['M'], slots['obj'], slots['ctr'], slots['scl'],
                slots['wgt'], slots['cls'], float(self._ignore_iou_thresh), slots['smooth'], losses.data_ptr(),
                C.byref(dh), None, C.byref(dha), lws.data_ptr(), lws.numel())
        fwd.append(seg)

        # ---- backward
        # The weight-gradient GEMMs (MFMA-bound) run on a side stream beside the main-stream chain
        # [BN backward (HBM-bound) -> data gradient]: nothing in the backward pass consumes a weight gradient,
        # so the only edges are  dz ready -> wgrad  (event) and  wgrad done -> dz scratch reuse  (event; the dz
        # scratch is double-buffered), plus one join before the optimiser.  Tails of one GEMM fill with the other.
        bwd, seg = [], Program()
        side = torch.cuda.Stream(priority=int(__import__('os').environ.get('VD_SIDE_PRIO', '0'))) if self.overlap_wgrad else None
        ws_w = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if side is not None else ws
        dz_bufs = [bufs['dz'], bufs['dz2']]
        dz_free = [None, None]             # event after which dz_bufs[i] may be overwritten
        n_dz = [0]
        last_side = [None]

        def ev_record(e, on_side):
            def f():
                e.record(side if on_side else torch.cuda.current_stream())
            return f

        def ev_wait(e, on_side):
            def f():
                (side if on_side else torch.cuda.current_stream()).wait_event(e)
            return f

        bucket_hi, bucket_acc = [self.n_weight], [0]
        written = set(self.head_names)     # gradients already produced (the loss kernel wrote d:head*)
        dgrad_packs = []                   # (node, plan, packed weight buffer) re-packed when weights change

        # d:x of a residual block input x is  dy_block (skip)  +  dgrad(first conv of the block).  The skip term is not
        # copied: it stays an alias of the block's dy until the dgrad, which reads it as its epilogue residual and
        # writes d:x out of place (saves a read + write of every block output per step).
        alias = {}
        # BatchNorm backward reductions (sum g, sum g*xhat over dy and z) ride in the epilogue of the data-gradient conv
        # that writes the FINAL dy of a BatchNorm output - the earliest forward consumer, processed last here - when
        # that conv is a single stride-1 launch; the standalone two-tensor reduction pass is then skipped.
        producers = {m.dst: m for m in self.conv_nodes if m.bn}
        consumers = {}
        for m in self.nodes:
            srcs = [m.src, m.residual] if isinstance(m, ConvNode) else ([m.up, m.route] if isinstance(m, UpcatNode) else
                                                                          ([m.a, m.b] if isinstance(m, AddNode) else [m.src]))
            for t in srcs:
                if t:
                    consumers.setdefault(t, []).append(m)
        fused_bwd = set()
        # grad_req 'null' (wrappers.py:55-57): a tensor needs a gradient only if a trainable parameter sits in its
        # producer or anywhere upstream of it.  With the backbone frozen nothing below the three route tensors does, so
        # the whole backbone drops out of the backward schedule (as MXNet's autograd prunes it) and its weight-gradient
        # launches never run.
        tgrad = {t: False for t in self.tensors}
        for m in self.nodes:
            if isinstance(m, ConvNode):
                tgrad[m.dst] = any(self._node_trainable(m)) or tgrad[m.src] or bool(m.residual and tgrad[m.residual])
            elif isinstance(m, UpcatNode):
                tgrad[m.dst] = tgrad[m.up] or tgrad[m.route]
            elif isinstance(m, AddNode):
                tgrad[m.dst] = tgrad[m.a] or tgrad[m.b]
            else:
                tgrad[m.dst] = tgrad[m.src]
        wtrain = [m for m in self.conv_nodes if self._node_trainable(m)[0]]
        first_wtrain = wtrain[0] if wtrain else None       # its weight gradient is the last one backward produces

        def materialize(name):
            if name in alias:
                src = alias.pop(name)
                seg.add('vd_bn_apply_leaky', src.data_ptr(), self._ones(src.shape[-1]).data_ptr(),
                        self._zeros(src.shape[-1]).data_ptr(), None, bufs['d:' + name].data_ptr(),
                        src.numel() // src.shape[-1], src.shape[-1], 1.0, None)

        def grad_into(name, numel, can_alias=False):
            """Return (dst_ptr, accumulate?) for a producer of d:name."""
            if name in written:
                if not can_alias:
                    materialize(name)
                return bufs['d:' + name], True
            written.add(name)
            return bufs['d:' + name], False

        for n in reversed(self.nodes):
            if isinstance(n, UpcatNode):
                if not tgrad[n.dst]:
                    continue
                dout = bufs['d:' + n.dst]
                dup_p = drt_p = None                       # NULL = that half is not needed (frozen upstream)
                acc_r = False
                if tgrad[n.up]:
                    dup, acc_u = grad_into(n.up, 0)
                    assert not acc_u
                    dup_p = dup.data_ptr()
                if tgrad[n.route]:
                    drt, acc_r = grad_into(n.route, 0)
                    drt_p = drt.data_ptr()
                if acc_r:
                    tmp = bufs['tmp'][:drt.numel()]
                    seg.add('vd_upsample2x_concat_bwd', dout.data_ptr(), dup_p, tmp.data_ptr(), B * n.fr,
                            dout.shape[1], dout.shape[2], n.cu, n.cr)
                    seg.add('vd_add', drt.data_ptr(), tmp.data_ptr(), drt.data_ptr(), drt.numel())
                elif dup_p or drt_p:
                    seg.add('vd_upsample2x_concat_bwd', dout.data_ptr(), dup_p, drt_p, B * n.fr,
                            dout.shape[1], dout.shape[2], n.cu, n.cr)
                continue
            if isinstance(n, SelNode):
                if not tgrad[n.src]:
                    continue
                assert n.dst in written, n.name
                materialize(n.dst)
                dout = bufs['d:' + n.dst]
                dsrc, acc = grad_into(n.src, 0)
                target = bufs['tmp'][:dsrc.numel()] if acc else dsrc
                seg.add('vd_frame_slice', dout.data_ptr(), target.data_ptr(), B, n.K, n.k0, n.kc, dsrc[0].numel(), 1)
                if acc:
                    seg.add('vd_add', dsrc.data_ptr(), target.data_ptr(), dsrc.data_ptr(), dsrc.numel())
                continue
            if isinstance(n, AddNode):
                if not tgrad[n.dst]:
                    continue
                assert n.dst in written, n.name
                materialize(n.dst)
                dout = bufs['d:' + n.dst]
                for t_ in (n.a, n.b):
                    if not tgrad[t_]:
                        continue
                    dsrc, acc = grad_into(t_, 0)
                    if acc:
                        seg.add('vd_add', dsrc.data_ptr(), dout.data_ptr(), dsrc.data_ptr(), dsrc.numel())
                    else:                      # a copy (the identity form of the BatchNorm apply kernel)
                        seg.add('vd_bn_apply_leaky', dout.data_ptr(), self._ones(dout.shape[-1]).data_ptr(),
                                self._zeros(dout.shape[-1]).data_ptr(), None, dsrc.data_ptr(),
                                dout.numel() // dout.shape[-1], dout.shape[-1], 1.0, None)
                continue
            if isinstance(n, PoolNode):
                if not tgrad[n.src]:
                    continue
                dout = bufs['d:' + n.dst]
                assert n.dst in written, n.name
                dsrc, acc = grad_into(n.src, 0)
                am = bufs['am:' + n.dst].data_ptr() if n.type == 0 else None
                target = bufs['tmp'][:dsrc.numel()] if acc else dsrc
                if n.type == 2:
                    seg.add('vd_temporal_cat', dout.data_ptr(), target.data_ptr(), B, n.K, dsrc.shape[1] * dsrc.shape[2],
                            dsrc.shape[3], 1)
                else:
                    seg.add('vd_temporal_pool_bwd', dout.data_ptr(), am, target.data_ptr(), B, n.K, dout[0].numel(), n.type)
                if acc:
                    seg.add('vd_add', dsrc.data_ptr(), target.data_ptr(), dsrc.data_ptr(), dsrc.numel())
                continue
            Hi, Wi = H // n.div_in, W // n.div_in
            Ho, Wo = H // n.div_out, W // n.div_out
            tvalid = getattr(n, 'tvalid', False)
            M = B * (n.fr - 2 if tvalid else n.fr) * Ho * Wo
            if not tgrad[n.dst]:
                continue                   # frozen, and nothing trainable upstream: no gradient is needed here
            w_train, v_train = self._node_trainable(n)
            dy = bufs['d:' + n.dst]
            assert n.dst in written, n.name
            materialize(n.dst)
            if n.head:
                dz = dy
                # bias gradient = per-channel sum of dz (reuses the BN column-sum kernels)
                seg.add('vd_bn_stats', dz.data_ptr(), M, n.co_pad, n.sums.data_ptr(), ws.data_ptr(), ws_bytes)
                seg.add('vd_bn_param_grads', n.sums.data_ptr(), n.co_pad, bufs['tmp'].data_ptr(), n.gbias.data_ptr())
            else:
                if n.residual and tgrad[n.residual]:
                    dres, acc = grad_into(n.residual, 0)
                    if acc:
                        seg.add('vd_add', dres.data_ptr(), dy.data_ptr(), dres.data_ptr(), dy.numel())
                    else:
                        alias[n.residual] = dy     # first producer of the skip gradient: it IS dy (no copy, see alias)
                        if not self.alias_skip_grad:
                            materialize(n.residual)
                z = bufs['z:' + n.dst]
                slot = n_dz[0] % 2
                n_dz[0] += 1
                dz = dz_bufs[slot][:M * n.cout].view(-1, Ho, Wo, n.cout)
                if n.name not in fused_bwd:
                    seg.add('vd_bn_bwd_reduce', z.data_ptr(), dy.data_ptr(), n.b_scale.data_ptr(), n.b_shift.data_ptr(),
                            n.b_mean.data_ptr(), n.b_invstd.data_ptr(), M, n.cout, LEAKY_SLOPE, n.sums2.data_ptr(),
                            ws.data_ptr(), ws_bytes)
                if n.name not in fused_bwd:      # (fused: the gamma/beta gradients came with the table reduction)
                    seg.add('vd_bn_param_grads', n.sums2.data_ptr(), n.cout, n.ggamma.data_ptr(), n.gbeta.data_ptr())
                count = float(M)
                if self._syncbn(n):
                    seg.add_coll(self._syncbn_exchange(n.sums2))
                    count = float(M * world)
                if side is not None and dz_free[slot] is not None:
                    seg.add_py(ev_wait(dz_free[slot], False))       # the wgrad that read this scratch has finished
                seg.add('vd_bn_bwd_apply', z.data_ptr(), dy.data_ptr(), n.b_scale.data_ptr(), n.b_shift.data_ptr(),
                        n.b_mean.data_ptr(), n.b_invstd.data_ptr(), n.sums2.data_ptr(), count, M, n.cout, LEAKY_SLOPE,
                        dz.data_ptr(), amx('dz:' + n.name))
                if tvalid:
                    # the gradient of the frame slice: dz of the valid frames, zeros on the two border frames; the conv's
                    # weight / data gradients then are those of the 'same'-padded conv
                    dzf = bufs['dzf:' + n.dst]
                    seg.add('vd_frame_slice', dz.data_ptr(), dzf.data_ptr(), B, n.fr, 1, n.fr - 2, dzf[0].numel(), 1)
                    dz = dzf
            # weight gradient straight into the gradient arena (same fwd-packed layout as the weights)
            if n.stem and w_train:
                # both operands straight from global memory: the NCHW batch and dz (vd_stem.hip)
                wargs = ('vd_stem_wgrad', bufs['in'].data_ptr(), dz.data_ptr(), n.co_pad, n.gwp.data_ptr(), B * n.fr, Hi, Wi)
                if side is not None:
                    e_ready, e_done = torch.cuda.Event(), torch.cuda.Event()
                    seg.add_py(ev_record(e_ready, False))
                    seg.add_py(ev_wait(e_ready, True))
                    seg.add(*wargs, ws_w.data_ptr(), ws_bytes, meta=self._flops(n, B, H, W, 'wgrad'), stream=side)
                    seg.add_py(ev_record(e_done, True))
                    seg.hold(e_ready, e_done)
                    dz_free[slot] = e_done
                    last_side[0] = e_done
                else:
                    seg.add(*wargs, ws.data_ptr(), ws_bytes, meta=self._flops(n, B, H, W, 'wgrad'))
                bucket_acc[0] += n.w_numel
                if self.bucketed_allreduce:            # the stem is the first conv: its bucket closes the arena
                    seg.add_py(self._bucket_launcher(n.w_off, bucket_hi[0], side))
                    bucket_hi[0], bucket_acc[0] = n.w_off, 0
                continue
            if n.stem:
                continue
            if w_train:
                wd_ = WgradDesc()
                xin = bufs[n.src]
                wd_.in_, wd_.dout, wd_.dwp = xin.data_ptr(), dz.data_ptr(), n.gwp.data_ptr()
                wd_.N, wd_.Hi, wd_.Wi, wd_.Ci = B * n.fr, Hi, Wi, n.ci_eff
                wd_.Hg, wd_.Wg, wd_.Co, wd_.ldd = Ho, Wo, n.co_pad, n.co_pad
                wd_.in_stride = n.stride
                ops._set_taps(wd_, n.taps())
                wd_.Kfr, wd_.splits = (n.fr if n.kd > 1 else 1), 0
                wd_.amax_in, wd_.amax_dout = self._amax_or_none(bufs, n.src), self._amax_or_none(bufs, 'dz:' + n.name)
                autotune_wgrad(wd_, ws.data_ptr(), ws_bytes)
                seg.hold(wd_)
                if side is not None:
                    e_ready, e_done = torch.cuda.Event(), torch.cuda.Event()
                    seg.add_py(ev_record(e_ready, False))
                    seg.add_py(ev_wait(e_ready, True))
                    seg.add('vd_conv_wgrad', C.byref(wd_), ws_w.data_ptr(), ws_bytes, meta=self._flops(n, B, H, W, 'wgrad'),
                            stream=side)
                    seg.add_py(ev_record(e_done, True))
                    seg.hold(e_ready, e_done)
                    if not n.head:
                        dz_free[slot] = e_done
                    last_side[0] = e_done
                else:
                    seg.add('vd_conv_wgrad', C.byref(wd_), ws.data_ptr(), ws_bytes, meta=self._flops(n, B, H, W, 'wgrad'))
                # bucketed gradient all-reduce, overlapped with the rest of the backward pass: weight gradients complete
                # in reverse arena order on the stream that runs the wgrad GEMMs, so every time a bucket (~32 MB) of the arena tail
                # is final an async all-reduce of that contiguous range is queued behind them (RCCL syncs with that
                # stream); allreduce_grads() later waits for the handles and reduces the small gamma/beta/bias range.
                bucket_acc[0] += n.w_numel
                if self.bucketed_allreduce and (bucket_acc[0] >= self.bucket_elems or n is first_wtrain):
                    lo, hi = n.w_off, bucket_hi[0]
                    seg.add_py(self._bucket_launcher(lo, hi, side))
                    bucket_hi[0], bucket_acc[0] = lo, 0
            if n.stem or n.src in self.input_tensors or not tgrad[n.src]:     # no gradient flows into the network inputs,
                continue                                                       # nor below the last trainable parameter
            # data gradient into d:src
            dsrc, acc = grad_into(n.src, 0, can_alias=True)
            res_src = alias.pop(n.src) if n.src in alias else dsrc      # the skip gradient, still living in the block's dy
            plans = dgrad_plans(n.k, n.pad, n.stride, Hi, Wi, n.kd, n.pad_d)
            pm = producers.get(n.src)
            # the producer's BatchNorm-backward reductions ride in this data gradient's epilogue when it is the launch (or, for a
            # stride-2 conv, the four parity launches) that completes dy of the producer's output
            fuse_m = pm if (self.fuse_bn_bwd and pm is not None and (len(plans) == 1 or (n.kd == 1 and _fuse_bwd_s2())) and
                            consumers[n.src][0] is n and pm.fr == n.fr) else None
            bs_rows = 0
            # ---- a 3x3 / stride-2 data gradient as ONE launch (VD_CONV_PARITY4, vd_conv_par.hip) instead of four parity
            # launches that each gather and stage the whole dz: rows = positions of dz's grid, columns = (parity class,
            # channel), taps = the four offsets of a 2x2 window, zero (offset, class) weight blocks skipped.
            # VD_S2_FUSED=0 keeps the four launches (also: odd maps, other arithmetics, range-exact operands, pinned kernels).
            import os as _os
            fused_s2 = (len(plans) == 4 and n.k == 3 and n.stride == 2 and n.pad == 1 and n.kd == 1 and Hi % 2 == 0 and Wi % 2 == 0 and
                        _os.environ.get('VD_S2_FUSED', '1') == '1' and n.cin % 32 == 0 and
                        # (measured per layer, batch 64 / 416x416, four launches -> one: 32 channels 1.51 -> 1.12 ms, 64: 0.80 ->
                        # 0.80, 128: 0.55 -> 0.66, 256: 0.53 -> 0.59, 512: 0.47 -> 0.53 - the one launch stages all sixteen
                        # (offset, class) weight blocks for nine useful ones, which only pays where a parity launch has too few
                        # output channels to fill a tile; VD_S2_FUSED_MAX_CIN moves the limit)
                        n.cin <= int(_os.environ.get('VD_S2_FUSED_MAX_CIN', '64')) and
                        self._amax_or_none(bufs, 'dz:' + n.name) is not None and
                        (fp32_math() == 'split2' or (fp32_math() == 'auto' and _os.environ.get('VD_AUTOTUNE', '1') != '0')))
            if fused_s2:
                wp4 = torch.empty(16 * n.cin * n.co_pad, device=dev)
                dgrad_packs.append((n, dict(fused_s2=True), wp4))
                d = ConvDesc()
                d.in_, d.wp, d.out = dz.data_ptr(), wp4.data_ptr(), dsrc.data_ptr()
                d.N, d.Hi, d.Wi, d.Ci = B * n.fr, Ho, Wo, n.co_pad
                d.Hg, d.Wg, d.in_stride = Ho, Wo, 1
                ops._set_taps(d, ops.PARITY4_TAPS)
                d.Kfr, d.Ho, d.Wo, d.Co = 1, Hi, Wi, 4 * n.cin
                d.out_stride, d.out_oy, d.out_ox = 2, 0, 0
                d.ldo = d.ldr = n.cin
                d.par_cin, d.par_mask = n.cin, ops.PARITY4_MASK   # (offset, class) blocks that hold a kernel tap
                d.flags, d.slope = (EPI_RESIDUAL if acc else 0) | L.MATH_F16X2 | L.CONV_PARITY4, LEAKY_SLOPE
                d.amax_in, d.amax_w = self._amax_or_none(bufs, 'dz:' + n.name), n.wamax.data_ptr()
                if acc:
                    d.residual = res_src.data_ptr()
                seg.hold(d, wp4)
                if fuse_m is not None:
                    d.bs_z = bufs['z:' + fuse_m.dst].data_ptr()
                    d.bs_scale, d.bs_shift = fuse_m.b_scale.data_ptr(), fuse_m.b_shift.data_ptr()
                    d.bs_mean, d.bs_invstd = fuse_m.b_mean.data_ptr(), fuse_m.b_invstd.data_ptr()
                    d.bs_part, d.bs_slope = stats_ws.data_ptr(), LEAKY_SLOPE
                autotune_desc(d)                                   # fixes the tile, hence the rows of the partial table
                if fuse_m is not None:
                    mt = L.load().vd_conv_igemm_mtiles(C.byref(d))
                    assert mt * 2 * fuse_m.cout * 4 <= stats_ws.numel() * 4, "stats workspace too small"
                    seg.add('vd_fill', stats_ws.data_ptr(), 0.0, mt * 2 * fuse_m.cout)     # column tiles narrower than Cin fill part of a row
                seg.add('vd_conv_igemm', C.byref(d), meta=dict(
                    kind='dgrad', node=n.name, k=n.k, stride=n.stride, fused_s2=True,
                    flops=2.0 * n.cin * n.cout * 9 * Ho * Wo * B * n.fr, bytes=self._flops(n, B, H, W, 'dgrad')['bytes']))
                if fuse_m is not None:
                    seg.add('vd_bn_sum_param_grads', stats_ws.data_ptr(), mt, fuse_m.cout, fuse_m.sums2.data_ptr(),
                            fuse_m.ggamma.data_ptr(), fuse_m.gbeta.data_ptr(), ws.data_ptr(), ws_bytes)
                    fused_bwd.add(fuse_m.name)
                continue
            # Frame chunks of a stride-2 data gradient (VD_S2_CHUNK_MB > 0; default 0 = one launch per parity class): each of
            # the four parity launches streams the WHOLE incoming gradient dz - 709 MB at 208 x 208 x 64 channels, batch 64 - and
            # is HBM-bound on it (isolated: 5.9 TB/s on the one-tap class).  Cut into chunks of frames whose dz fits the 256 MB
            # Infinity Cache beside the outputs, the second to fourth class of a chunk find dz there: one HBM read instead of four.
            NF = B * n.fr
            nchunks = 1
            # (measured, same box, batch 64 / 416x416: off 1022.0 frames/s, 192 MB 1021.9, 96 MB 1018.6, 48 MB 991.5 - the
            # four launches already share most of dz through the L2s / Infinity Cache when they run side by side on their
            # four streams; the default stays off, the switch and its parity test stay as the record of the experiment)
            chunk_mb = float(__import__('os').environ.get('VD_S2_CHUNK_MB', '0'))
            if len(plans) == 4 and n.kd == 1 and chunk_mb > 0:
                dz_mb = 4.0 * NF * Ho * Wo * n.co_pad / 1e6
                nchunks = max(1, min(NF, int(math.ceil(dz_mb / chunk_mb))))
            fchunk = (NF + nchunks - 1) // nchunks
            nchunks = (NF + fchunk - 1) // fchunk
            wpks = []
            for plan in plans:
                assert plan['taps'], "a parity class without taps would leave its gradient unwritten"
                wpk = torch.empty(n.cin * len(plan['taps']) * n.co_pad, device=dev)
                dgrad_packs.append((n, plan, wpk))
                wpks.append(wpk)
            in_img, out_img = Ho * Wo * n.co_pad * 4, Hi * Wi * n.cin * 4          # bytes per frame of dz / of d:src
            for ch in range(nchunks):
              f0 = ch * fchunk
              fn_ = min(NF, f0 + fchunk) - f0
              # the four parity launches of a stride-2 data gradient write disjoint pixels and read the same dz: run them
              # side by side on their own streams (VD_PARITY_STREAMS=1) so that they share dz in the L2s and fill each other's tails
              par = None
              if _parity_streams() and len(plans) == 4 and self.overlap_wgrad:
                if getattr(self, '_par_streams', None) is None:
                    self._par_streams = [torch.cuda.Stream() for _ in range(3)]
                par = self._par_streams
                e_fork = torch.cuda.Event()
                seg.add_py(lambda e=e_fork: e.record(torch.cuda.current_stream()))
                for st in par:
                    seg.add_py(lambda e=e_fork, st=st: st.wait_event(e))
              for pi, plan in enumerate(plans):
                wpk = wpks[pi]
                d = ConvDesc()
                d.in_, d.wp, d.out = dz.data_ptr() + f0 * in_img, wpk.data_ptr(), dsrc.data_ptr() + f0 * out_img
                d.N, d.Hi, d.Wi, d.Ci = fn_, Ho, Wo, n.co_pad
                d.Hg, d.Wg, d.in_stride = plan['Hg'], plan['Wg'], 1
                ops._set_taps(d, plan['taps'])
                d.Kfr = n.fr if n.kd > 1 else 1
                d.Ho, d.Wo, d.Co = Hi, Wi, n.cin
                d.out_stride, d.out_oy, d.out_ox = n.stride, plan['py'], plan['px']
                d.ldo = d.ldr = n.cin
                d.flags, d.slope = (EPI_RESIDUAL if acc else 0), LEAKY_SLOPE
                d.amax_in, d.amax_w = self._amax_or_none(bufs, 'dz:' + n.name), n.wamax.data_ptr()
                self._set_streamk(d, pi if (par is not None) else 0)
                if acc:
                    d.residual = res_src.data_ptr() + f0 * out_img
                seg.hold(d, wpk)
                nplans = n.stride * n.stride
                if fuse_m is not None:
                    d.bs_z = bufs['z:' + fuse_m.dst].data_ptr() + f0 * out_img
                    d.bs_scale, d.bs_shift = fuse_m.b_scale.data_ptr(), fuse_m.b_shift.data_ptr()
                    d.bs_mean, d.bs_invstd = fuse_m.b_mean.data_ptr(), fuse_m.b_invstd.data_ptr()
                    d.bs_part, d.bs_slope = stats_ws.data_ptr() + bs_rows * 2 * fuse_m.cout * 4, LEAKY_SLOPE
                    autotune_desc(d)                               # fixes the tile, hence the number of M tiles
                    bs_rows += L.load().vd_conv_igemm_mtiles(C.byref(d))
                    mt = bs_rows
                    assert mt * 2 * fuse_m.cout * 4 <= stats_ws.numel() * 4, "stats workspace too small"
                seg.add('vd_conv_igemm', C.byref(d), meta=dict(
                    kind='dgrad', node=n.name, k=n.k, stride=n.stride,
                    flops=2.0 * n.cin * n.cout * len(plan['taps']) * plan['Hg'] * plan['Wg'] * fn_,
                    bytes=self._flops(n, B, H, W, 'dgrad')['bytes'] / nplans * fn_ / NF),
                    stream=(par[pi - 1] if (par is not None and pi > 0) else None))
                if par is not None and pi == len(plans) - 1:
                    for st in par:                                  # join before anything reads d:src or the partial table
                        e_join = torch.cuda.Event()
                        seg.add_py(lambda e=e_join, st=st: e.record(st))
                        seg.add_py(lambda e=e_join: torch.cuda.current_stream().wait_event(e))
                if fuse_m is not None and pi == len(plans) - 1 and ch == nchunks - 1:
                    seg.add('vd_bn_sum_param_grads', stats_ws.data_ptr(), mt, fuse_m.cout, fuse_m.sums2.data_ptr(),
                            fuse_m.ggamma.data_ptr(), fuse_m.gbeta.data_ptr(), ws.data_ptr(), ws_bytes)
                    fused_bwd.add(fuse_m.name)
        if side is not None and last_side[0] is not None:
            seg.add_py(ev_wait(last_side[0], False))          # join: the optimiser / all-reduce see every gradient
        seg.hold(ws_w, side, stats_ws)
        bwd.append(seg)
        for sg in fwd + bwd:
            if isinstance(sg, Program):
                autotune_program(sg)
        _TUNE_CACHE.save()
        return dict(fwd=fwd, bwd=bwd, bufs=bufs, slots=slots, losses=losses, dgrad_packs=dgrad_packs, ws=ws)


    # ------------------------------------------------------------------ bf16-STORAGE training (BASELINE configs[4])
    def set_storage(self, storage):
        """Storage type of the TRAINING activations and their gradients: 'fp32' (default: the reference's arithmetic,
        train_yolov3.py:623-636) or 'bf16' - conv outputs, cell outputs and both gradient families are bf16 tensors, the
        convolutions run one bf16 MFMA per product block with fp32 accumulation (vd_conv_igemm_bf16 / VD_STORE_BF16),
        BatchNorm statistics come from the fp32 accumulators; master weights, weight gradients, BatchNorm vectors, head
        logits, losses and the optimiser stay fp32.  No reference counterpart: judged against the fp32 oracle at a stated
        bf16 tolerance (tests/test_bf16_train_gpu.py, test_model_gpu.py)."""
        if storage not in ('fp32', 'bf16'):
            raise ValueError("storage must be 'fp32' or 'bf16'")
        if storage == 'bf16' and (self._k > 1 or self.noback or self.temporal_out or getattr(self, 'temporal_side', False)):
            raise NotImplementedError("bf16-storage training is built for the single-frame yolo3_darknet53 network")
        self.storage = storage

    def _tune_bf16_desc(self, d, of32):
        """Tile (and halo / generic loop) of one vd_conv_igemm_bf16 launch record, timed in place; persisted like the others."""
        import os
        if os.environ.get("VD_AUTOTUNE", "1") == "0":
            return
        base = d.flags & ~(L.MATH_NOHALO | L.CONV_STREAMK)
        key = ('bf16', d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride, d.T, d.Co, base, of32, d.out_stride, bool(d.stats_part),
               bool(d.bs_part))
        # 14 / 15: the small four-wave tiles (64x64 / 128x32, four or five workgroups per CU) for the HBM-bound 1x1 layers
        one = d.T == 1
        # (16: the first-stage patch kernel, forward with fused statistics; falls back to the default tile where it does not apply)
        tiles = ((10, 11, 13) + ((14,) if one else ()) + ((16,) if (d.T == 9 and d.Co == 64 and d.stats_part and not d.bs_part and not of32) else ())
                 if d.Ci == 32 else (12, 10, 11, 13) + ((15,) if one else ()) if d.Co <= 32
                 else (10, 11, 13, 2, 3, 4, 5) + ((14,) if one else ()) if d.Co <= 64
                 else (1, 2, 3, 4, 5, 6, 7) + ((8, 9) if (d.Co > 128 and not d.bs_part) else ()) + ((14,) if one else ()))
        halo_geo = d.T == 9 and d.in_stride == 1 and d.Hg == d.Hi and d.Ci % 64 == 0 and d.out_stride == 1
        _tune_bf16_record(d, of32, key, tiles, halo_geo)

    def _build_train_bf16(self, B, H, W):
        """The training plan of `_build_train` on bf16 activation / gradient tensors (set_storage('bf16')).  Same schedule:
        forward with the BatchNorm statistics in the conv epilogues, loss, then the fixed reverse pass with the
        weight-gradient GEMMs on a side stream and the bucketed gradient all-reduce queued behind them."""
        dev, BFT = self.device, torch.bfloat16
        lib = L.load()
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size(self.process_group)
            # (as in _build_train: the gradient buckets get their own communicator when SyncBN's statistics all-reduces sit
            # on the critical path of backward)
            if (world > 1 and self.syncbn_scope and self.bucketed_allreduce and self._bucket_group is None
                    and self.process_group is None):
                self._bucket_group = torch.distributed.new_group()

        hb = lambda c: c if (c == 32 or c % 64 == 0) else round_up(c, 64)      # head pitch: the data gradient's K dimension
        # algorithmic bytes of a launch: every operand tensor once, at 2 bytes per element
        fl = lambda n_, kind: dict(self._flops(n_, B, H, W, kind), bytes=self._flops(n_, B, H, W, kind)['bytes'] / 2)
        # ---- buffers
        bufs = {'in': torch.empty(B, 3, H, W, device=dev)}
        for name, (c, div, ld, fr) in self.tensors.items():
            if name == 'in':
                continue
            if name in self.head_names:
                bufs[name] = torch.zeros(B, H // div, W // div, hb(ld), device=dev)              # fp32 logits (pad columns stay 0)
                bufs['d:' + name] = torch.zeros(B, H // div, W // div, hb(ld), dtype=BFT, device=dev)
            else:
                bufs[name] = torch.empty(B, H // div, W // div, c, dtype=BFT, device=dev)
                bufs['d:' + name] = torch.empty(B, H // div, W // div, c, dtype=BFT, device=dev)
        for n in self.conv_nodes:
            if n.bn:
                bufs['z:' + n.dst] = torch.empty_like(bufs[n.dst])
        mx = max(bufs[n.dst].numel() for n in self.conv_nodes if not n.head)
        mx = max([mx] + [bufs[h].numel() for h in self.head_names])
        bufs['dz'], bufs['dz2'], bufs['tmp'] = [torch.empty(mx, dtype=BFT, device=dev) for _ in range(3)]
        for nm, t in bufs.items():                     # tune on noise, not on zero pages (see _buffers)
            if torch.is_tensor(t) and nm not in self.head_names and not nm.startswith('d:yolo') and t.dtype == BFT:
                t.copy_(torch.randn(min(t.numel(), 1 << 22), device=dev).to(BFT).repeat((t.numel() >> 22) + 1)[:t.numel()].view(t.shape))
        ws_bytes = 1 << 20
        for n in self.conv_nodes:
            Hi, Wi = H // n.div_in, W // n.div_in
            Ho, Wo = H // n.div_out, W // n.div_out
            if n.stem:
                ws_bytes = max(ws_bytes, int(lib.vd_stem_wgrad_ws_bytes(B, Hi, Wi)))
            else:
                wd_ = WgradDesc()
                wd_.N, wd_.Hi, wd_.Wi, wd_.Ci, wd_.Hg, wd_.Wg, wd_.Co, wd_.ldd = B, Hi, Wi, n.cin, Ho, Wo, n.co_pad, n.co_pad
                ops._set_taps(wd_, n.taps())
                for fl_ in (0, L.WGRAD_HALO):          # the halo-ring kernel picks its own split count
                    wd_.in_stride, wd_.Kfr, wd_.flags = n.stride, 1, L.STORE_BF16 | L.MATH_BF16 | fl_
                    ws_bytes = max(ws_bytes, int(lib.vd_conv_wgrad_ws_bytes(C.byref(wd_))))
            ws_bytes = max(ws_bytes, ops.bn_stats_ws_bytes(B * Ho * Wo, hb(n.co_pad)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        smax = 16
        for n in self.conv_nodes:
            if n.bn:
                smax = max(smax, ((B * (H // n.div_out) * (W // n.div_out) + 63) // 64 + 8) * 2 * n.cout)
        stats_ws = torch.empty(smax, device=dev)

        packs = []                                     # (kind, node, plan, fp32 scratch, bf16 image, rows, K, K_pad, T)
        # forward images: the conv weights live fwd-packed [co_pad][T * Ci] in the arena and need no padding here (every Ci
        # is 32 or a multiple of 64), so ONE conversion of the arena's weight range makes all of them
        wb_arena = torch.empty(self.n_weight, dtype=BFT, device=dev)
        packs.append(('arena', None, None, None, wb_arena, 1, self.n_weight, self.n_weight, 1))
        fwd, seg = [], Program()
        for n in self.nodes:
            if isinstance(n, UpcatNode):
                o = bufs[n.dst]
                seg.add('vd_upsample2x_concat', bufs[n.up].data_ptr(), bufs[n.route].data_ptr(), o.data_ptr(), B,
                        o.shape[1], o.shape[2], n.cu // 2, n.cr // 2)          # a copy: two bf16 = one 4-byte word
                continue
            if not isinstance(n, ConvNode):
                raise NotImplementedError("bf16-storage training: node type %s" % type(n).__name__)
            Hi, Wi = H // n.div_in, W // n.div_in
            Ho, Wo = H // n.div_out, W // n.div_out
            M = B * Ho * Wo
            if n.stem:
                z = bufs['z:' + n.dst]
                nb = lib.vd_stem_conv_blocks(B, H, W)
                assert nb * 2 * n.cout <= stats_ws.numel()
                self._add_stem(seg, n, bufs, B, H, W, z, bf16=True, stats=stats_ws.data_ptr())
                table_rows = nb
            else:
                wb = wb_arena[n.w_off:n.w_off + n.w_numel]
                assert n.w_numel == n.co_pad * n.T * n.cin and n.w_off % 8 == 0
                d = ConvDesc()
                self._set_streamk(d, 0)
                out = bufs[n.dst] if n.head else bufs['z:' + n.dst]
                d.in_, d.wp, d.out = bufs[n.src].data_ptr(), wb.data_ptr(), out.data_ptr()
                d.N, d.Hi, d.Wi, d.Ci, d.Hg, d.Wg, d.in_stride = B, Hi, Wi, n.cin, Ho, Wo, n.stride
                ops._set_taps(d, n.taps())
                d.Kfr, d.Ho, d.Wo, d.Co = 1, Ho, Wo, n.co_pad
                d.out_stride, d.out_oy, d.out_ox, d.slope = 1, 0, 0, LEAKY_SLOPE
                d.ldo = d.ldr = out.shape[-1]
                seg.hold(d, wb)
                if n.head:
                    d.flags, d.shift = EPI_AFFINE, n.bias.data_ptr()
                    self._tune_bf16_desc(d, 1)
                    seg.add('vd_conv_igemm_bf16', C.byref(d), 1, meta=fl(n, 'fwd'))
                    continue
                d.stats_part = stats_ws.data_ptr()
                self._tune_bf16_desc(d, 0)
                table_rows = lib.vd_conv_igemm_bf16_mtiles(C.byref(d))
                assert table_rows * 2 * n.cout <= stats_ws.numel(), "stats workspace too small"
                seg.add('vd_conv_igemm_bf16', C.byref(d), 0, meta=fl(n, 'fwd'))
                z = out
            fin = (n.gamma.data_ptr(), n.beta.data_ptr(), BN_EPS, BN_MOMENTUM, n.rmean.data_ptr(), n.rvar.data_ptr(),
                   n.b_scale.data_ptr(), n.b_shift.data_ptr(), n.b_mean.data_ptr(), n.b_invstd.data_ptr())
            if self._syncbn(n):
                # SyncBN (train_yolov3.py:347-354): the fp64 [sum x, sum x^2] of the fp32 accumulators, summed over the
                # ranks, then one finalize on the global count - the same exchange unit as the fp32-storage plan
                seg.add('vd_bn_sum_partials', stats_ws.data_ptr(), table_rows, n.cout, n.sums.data_ptr(), ws.data_ptr(), ws_bytes)
                seg.add_coll(self._syncbn_exchange(n.sums))
                seg.add('vd_bn_finalize', n.sums.data_ptr(), float(M * world), n.cout, *fin)
            else:
                seg.add('vd_bn_sum_finalize', stats_ws.data_ptr(), table_rows, n.cout, n.sums.data_ptr(), float(M), *fin,
                        ws.data_ptr(), ws_bytes)
            res = bufs[n.residual].data_ptr() if n.residual else None
            seg.add('vd_bn_apply_leaky_bf16', z.data_ptr(), n.b_scale.data_ptr(), n.b_shift.data_ptr(), res,
                    bufs[n.dst].data_ptr(), M, n.cout, LEAKY_SLOPE,
                    meta=dict(node=n.name, bytes=2.0 * M * n.cout * (3 if n.residual else 2), single_conv_consumer=False))
        grids = self._grid(H, W)
        ldh = bufs[self.head_names[0]].shape[-1]
        hd = ops.make_head_desc([bufs[h] for h in self.head_names], grids, ldh, STRIDES[::-1], ANCHORS[::-1], B, self.num_class)
        slots = dict(gt=Slot(), M=Slot(), obj=Slot(), ctr=Slot(), scl=Slot(), wgt=Slot(), cls=Slot(), smooth=Slot())
        losses = torch.zeros(B, 4, device=dev)
        dh = (C.c_void_p * 3)(*[bufs['d:' + h].data_ptr() for h in self.head_names])
        lws = torch.empty(max(16, ops.yolo_loss_ws_bytes(hd)), dtype=torch.uint8, device=dev)
        seg.hold(hd, dh, lws)
        seg.add('vd_yolo_loss_fwd_bwd_bf16', C.byref(hd), slots['gt'], slots['M'], slots['obj'], slots['ctr'], slots['scl'],
                slots['wgt'], slots['cls'], float(self._ignore_iou_thresh), slots['smooth'], losses.data_ptr(),
                C.byref(dh), None, lws.data_ptr(), lws.numel())
        fwd.append(seg)

        # ---- backward (the schedule of _build_train: wgrad GEMMs on a side stream, double-buffered dz scratch, skip
        # gradients by alias, bucketed all-reduce behind the weight gradients)
        bwd, seg = [], Program()
        side = torch.cuda.Stream(priority=int(__import__('os').environ.get('VD_SIDE_PRIO', '0'))) if self.overlap_wgrad else None
        ws_w = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if side is not None else ws
        dz_bufs, dz_free, n_dz, last_side = [bufs['dz'], bufs['dz2']], [None, None], [0], [None]

        def ev_record(e, on_side):
            return lambda: e.record(side if on_side else torch.cuda.current_stream())

        def ev_wait(e, on_side):
            return lambda: (side if on_side else torch.cuda.current_stream()).wait_event(e)

        bucket_hi, bucket_acc = [self.n_weight], [0]
        written, alias = set(self.head_names), {}
        # BatchNorm backward reductions in the epilogue of the data gradient that completes dy of a BatchNorm output (the
        # earliest forward consumer, processed last here), as in _build_train: the two-tensor reduction pass is skipped
        producers = {m.dst: m for m in self.conv_nodes if m.bn}
        consumers = {}
        for m in self.nodes:
            for t in ([m.src, m.residual] if isinstance(m, ConvNode) else [m.up, m.route]):
                if t:
                    consumers.setdefault(t, []).append(m)
        fused_bwd = set()
        tgrad = {t: False for t in self.tensors}
        for m in self.nodes:
            if isinstance(m, ConvNode):
                tgrad[m.dst] = any(self._node_trainable(m)) or tgrad[m.src] or bool(m.residual and tgrad[m.residual])
            else:
                tgrad[m.dst] = tgrad[m.up] or tgrad[m.route]
        wtrain = [m for m in self.conv_nodes if self._node_trainable(m)[0]]
        first_wtrain = wtrain[0] if wtrain else None
        ones = torch.ones(2048, device=dev)
        zeros = torch.zeros(2048, device=dev)

        def materialize(name):
            if name in alias:
                src = alias.pop(name)
                seg.add('vd_bn_apply_leaky_bf16', src.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None,
                        bufs['d:' + name].data_ptr(), src.numel() // src.shape[-1], src.shape[-1], 1.0)

        def grad_into(name, can_alias=False):
            if name in written:
                if not can_alias:
                    materialize(name)
                return bufs['d:' + name], True
            written.add(name)
            return bufs['d:' + name], False

        for n in reversed(self.nodes):
            if isinstance(n, UpcatNode):
                if not tgrad[n.dst]:
                    continue
                dout = bufs['d:' + n.dst]
                dup_p = drt_p = None
                acc_r = False
                if tgrad[n.up]:
                    dup, acc_u = grad_into(n.up)
                    assert not acc_u
                    dup_p = dup.data_ptr()
                if tgrad[n.route]:
                    drt, acc_r = grad_into(n.route)
                    drt_p = drt.data_ptr()
                if acc_r:
                    tmp = bufs['tmp'][:drt.numel()]
                    seg.add('vd_upsample2x_concat_bwd_bf16', dout.data_ptr(), dup_p, tmp.data_ptr(), B, dout.shape[1], dout.shape[2],
                            n.cu, n.cr)
                    seg.add('vd_add_bf16', drt.data_ptr(), tmp.data_ptr(), drt.data_ptr(), drt.numel())
                elif dup_p or drt_p:
                    seg.add('vd_upsample2x_concat_bwd_bf16', dout.data_ptr(), dup_p, drt_p, B, dout.shape[1], dout.shape[2], n.cu, n.cr)
                continue
            Hi, Wi = H // n.div_in, W // n.div_in
            Ho, Wo = H // n.div_out, W // n.div_out
            M = B * Ho * Wo
            if not tgrad[n.dst]:
                continue
            w_train, v_train = self._node_trainable(n)
            dy = bufs['d:' + n.dst]
            assert n.dst in written, n.name
            materialize(n.dst)
            slot = 0
            if n.head:
                dz, ldd = dy, dy.shape[-1]
                if getattr(n, 'sums_b', None) is None or n.sums_b.numel() != 2 * ldd:
                    n.sums_b = torch.zeros(2 * ldd, dtype=torch.float64, device=dev)
                seg.add('vd_bn_stats_bf16', dz.data_ptr(), M, ldd, n.sums_b.data_ptr(), ws.data_ptr(), ws_bytes)
                seg.add('vd_bn_param_grads', n.sums_b.data_ptr(), n.co_pad, bufs['tmp'].data_ptr(), n.gbias.data_ptr())
                # (dgamma slot of the kernel = scratch: bufs['tmp'] is bf16 storage, n.co_pad floats fit in it)
            else:
                if n.residual and tgrad[n.residual]:
                    dres, acc = grad_into(n.residual)
                    if acc:
                        seg.add('vd_add_bf16', dres.data_ptr(), dy.data_ptr(), dres.data_ptr(), dy.numel())
                    else:
                        alias[n.residual] = dy
                        if not self.alias_skip_grad:
                            materialize(n.residual)
                z = bufs['z:' + n.dst]
                slot = n_dz[0] % 2
                n_dz[0] += 1
                dz, ldd = dz_bufs[slot][:M * n.cout].view(-1, Ho, Wo, n.cout), n.cout
                if n.name not in fused_bwd:      # (fused: sums and gamma / beta gradients came with the table reduction)
                    seg.add('vd_bn_bwd_reduce_bf16', z.data_ptr(), dy.data_ptr(), n.b_scale.data_ptr(), n.b_shift.data_ptr(),
                            n.b_mean.data_ptr(), n.b_invstd.data_ptr(), M, n.cout, LEAKY_SLOPE, n.sums2.data_ptr(), ws.data_ptr(), ws_bytes)
                    seg.add('vd_bn_param_grads', n.sums2.data_ptr(), n.cout, n.ggamma.data_ptr(), n.gbeta.data_ptr())
                count = float(M)
                if self._syncbn(n):          # [sum g, sum g xhat] over the ranks (the local gamma / beta gradients came first)
                    seg.add_coll(self._syncbn_exchange(n.sums2))
                    count = float(M * world)
                if side is not None and dz_free[slot] is not None:
                    seg.add_py(ev_wait(dz_free[slot], False))
                seg.add('vd_bn_bwd_apply_bf16', z.data_ptr(), dy.data_ptr(), n.b_scale.data_ptr(), n.b_shift.data_ptr(),
                        n.b_mean.data_ptr(), n.b_invstd.data_ptr(), n.sums2.data_ptr(), count, M, n.cout, LEAKY_SLOPE, dz.data_ptr())
            if w_train:
                if n.stem:
                    wargs = ('vd_stem_wgrad_bf16', bufs['in'].data_ptr(), dz.data_ptr(), n.co_pad, n.gwp.data_ptr(), B, Hi, Wi)
                else:
                    wd_ = WgradDesc()
                    wd_.in_, wd_.dout, wd_.dwp = bufs[n.src].data_ptr(), dz.data_ptr(), n.gwp.data_ptr()
                    wd_.N, wd_.Hi, wd_.Wi, wd_.Ci = B, Hi, Wi, n.cin
                    wd_.Hg, wd_.Wg, wd_.Co, wd_.ldd = Ho, Wo, n.co_pad, ldd
                    wd_.in_stride, wd_.Kfr, wd_.splits, wd_.flags = n.stride, 1, 0, L.STORE_BF16 | L.MATH_BF16
                    ops._set_taps(wd_, n.taps())
                    autotune_wgrad_bf16(wd_, ws.data_ptr(), ws_bytes)
                    seg.hold(wd_)
                    wargs = ('vd_conv_wgrad', C.byref(wd_))
                if side is not None:
                    e_ready, e_done = torch.cuda.Event(), torch.cuda.Event()
                    seg.add_py(ev_record(e_ready, False))
                    seg.add_py(ev_wait(e_ready, True))
                    seg.add(*wargs, ws_w.data_ptr(), ws_bytes, meta=fl(n, 'wgrad'), stream=side)
                    seg.add_py(ev_record(e_done, True))
                    seg.hold(e_ready, e_done)
                    if not n.head:
                        dz_free[slot] = e_done
                    last_side[0] = e_done
                else:
                    seg.add(*wargs, ws.data_ptr(), ws_bytes, meta=fl(n, 'wgrad'))
                bucket_acc[0] += n.w_numel
                if self.bucketed_allreduce and (bucket_acc[0] >= self.bucket_elems or n is first_wtrain or n.stem):
                    lo, hi = n.w_off, bucket_hi[0]
                    seg.add_py(self._bucket_launcher(lo, hi, side))
                    bucket_hi[0], bucket_acc[0] = lo, 0
            if n.stem or n.src in self.input_tensors or not tgrad[n.src]:
                continue
            dsrc, acc = grad_into(n.src, can_alias=True)
            res_src = alias.pop(n.src) if n.src in alias else dsrc
            kp = dz.shape[-1]                                   # K dimension of the data gradient (head: the padded pitch)
            pm = producers.get(n.src)
            # (off by default in this mode, VD_FUSE_BWD_BF16=1: measured on one box, 416 / batch 64, 1992 frames/s with the
            # stand-alone reduction passes against 1958 fused - the HBM-bound pass runs beside the side stream's MFMA-bound
            # weight gradients, the fused epilogue lengthens the data gradients on the critical path)
            fuse_m = pm if (__import__('os').environ.get('VD_FUSE_BWD_BF16', '0') == '1' and pm is not None and
                            consumers[n.src][0] is n) else None
            bs_rows = 0
            plans = dgrad_plans(n.k, n.pad, n.stride, Hi, Wi, 1, 0)
            for pi, plan in enumerate(plans):
                assert plan['taps']
                T = len(plan['taps'])
                w32 = None
                wbd = torch.empty(n.cin * T * kp, dtype=BFT, device=dev)
                packs.append(('dgrad', n, plan, w32, wbd, n.cin, n.co_pad, kp, T))
                d = ConvDesc()
                self._set_streamk(d, 0)
                d.in_, d.wp, d.out = dz.data_ptr(), wbd.data_ptr(), dsrc.data_ptr()
                d.N, d.Hi, d.Wi, d.Ci = B, Ho, Wo, kp
                d.Hg, d.Wg, d.in_stride = plan['Hg'], plan['Wg'], 1
                ops._set_taps(d, plan['taps'])
                d.Kfr, d.Ho, d.Wo, d.Co = 1, Hi, Wi, n.cin
                d.out_stride, d.out_oy, d.out_ox = n.stride, plan['py'], plan['px']
                d.ldo = d.ldr = n.cin
                d.flags, d.slope = (EPI_RESIDUAL if acc else 0), LEAKY_SLOPE
                if acc:
                    d.residual = res_src.data_ptr()
                seg.hold(d, wbd)
                if fuse_m is not None:
                    d.bs_z = bufs['z:' + fuse_m.dst].data_ptr()
                    d.bs_scale, d.bs_shift = fuse_m.b_scale.data_ptr(), fuse_m.b_shift.data_ptr()
                    d.bs_mean, d.bs_invstd = fuse_m.b_mean.data_ptr(), fuse_m.b_invstd.data_ptr()
                    d.bs_part, d.bs_slope = stats_ws.data_ptr() + bs_rows * 2 * fuse_m.cout * 4, LEAKY_SLOPE
                self._tune_bf16_desc(d, 0)
                if fuse_m is not None:
                    bs_rows += lib.vd_conv_igemm_bf16_mtiles(C.byref(d))
                    assert bs_rows * 2 * fuse_m.cout <= stats_ws.numel(), "stats workspace too small"
                seg.add('vd_conv_igemm_bf16', C.byref(d), 0, meta=dict(
                    kind='dgrad', node=n.name, k=n.k, stride=n.stride,
                    flops=2.0 * n.cin * n.cout * T * plan['Hg'] * plan['Wg'] * B,
                    bytes=fl(n, 'dgrad')['bytes'] / (n.stride * n.stride)))
                if fuse_m is not None and pi == len(plans) - 1:
                    seg.add('vd_bn_sum_param_grads', stats_ws.data_ptr(), bs_rows, fuse_m.cout, fuse_m.sums2.data_ptr(),
                            fuse_m.ggamma.data_ptr(), fuse_m.gbeta.data_ptr(), ws.data_ptr(), ws_bytes)
                    fused_bwd.add(fuse_m.name)
        if side is not None and last_side[0] is not None:
            seg.add_py(ev_wait(last_side[0], False))
        seg.hold(ws_w, side, stats_ws, ones, zeros)
        bwd.append(seg)
        _TUNE_CACHE.save()
        return dict(fwd=fwd, bwd=bwd, bufs=bufs, slots=slots, losses=losses, dgrad_packs=[], packs_bf16=packs, ws=ws, storage='bf16')

    def _refresh_train_bf16(self, tp, overlap=False):
        """bf16 images of the weights (forward layout and every data-gradient tap plan) after the weights moved; on the pack
        stream beside the forward pass when `overlap` (the forward images are needed first: they are packed first and the
        main stream waits for them, the data-gradient images only gate backward())."""
        if tp.get('dgrad_version') == self._weights_version:
            return
        lib = L.load()

        def pack(kinds):
            s_ = L.stream_ptr()
            for kind, n, plan, w32, wb, rows, K, Kp, T in tp['packs_bf16']:
                if kind not in kinds:
                    continue
                if kind == 'arena':
                    L.check(lib.vd_pack_weight_bf16(self.weights.data_ptr(), wb.data_ptr(), rows, rows, K, Kp, T, s_), 'vd_pack_weight_bf16')
                else:
                    arr = (C.c_int32 * T)(*plan['tap_ids'])          # straight to bf16 [cin][T * K pitch]
                    L.check(lib.vd_pack_weight_dgrad_bf16(n.wp.data_ptr(), wb.data_ptr(), K, Kp, n.cin, 1, n.k, n.k, arr, T, 1, s_),
                            'vd_pack_weight_dgrad_bf16')
        pack(('arena',))
        ev = None
        if overlap and self.overlap_wgrad:
            if self._pack_stream is None:
                self._pack_stream = torch.cuda.Stream()
            self._pack_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._pack_stream):
                pack(('dgrad',))
                ev = torch.cuda.Event()
                ev.record(self._pack_stream)
        else:
            pack(('dgrad',))
        self._pack_event = ev
        tp['dgrad_version'] = self._weights_version

    @staticmethod
    def _flops(n, B, H, W, kind):
        """Algorithmic FLOPs of one conv launch: 2*Cin*Cout*k*k*Ho*Wo per image."""
        px_in = B * n.fr * (H // n.div_in) * (W // n.div_in)
        px_out = B * n.fr * (H // n.div_out) * (W // n.div_out)
        return dict(kind=kind, node=n.name, k=n.k, stride=n.stride,
                    flops=2.0 * n.cin * n.cout * n.kd * n.k * n.k * px_out,
                    # algorithmic bytes: every operand tensor once (input, output/gradient, weights), fp32
                    bytes=4.0 * (px_in * n.cin + px_out * n.cout + n.cout * n.cin * n.kd * n.k * n.k))

    def _ones(self, c):
        if not hasattr(self, '_const'):
            self._const = (torch.ones(1024, device=self.device), torch.zeros(1024, device=self.device))
        return self._const[0]

    def _zeros(self, c):
        self._ones(c)
        return self._const[1]

    def _refresh_dgrad(self, tp, overlap=False):
        """Re-pack the data-gradient weight layout after the weights moved. With overlap=True (start of a training
        step) the 72 small pack launches run on a side stream beside the forward pass; backward() waits on the event."""
        if tp.get('storage') == 'bf16':
            return self._refresh_train_bf16(tp, overlap)
        if tp.get('dgrad_version') == self._weights_version:
            return
        ev = None
        if overlap and self.overlap_wgrad:
            if self._pack_stream is None:
                self._pack_stream = torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            self._pack_stream.wait_stream(cur)          # the optimiser's weight update is on the main stream
            with torch.cuda.stream(self._pack_stream):
                for n, plan, wpk in tp['dgrad_packs']:
                    self._pack_dgrad(n, plan, wpk)
                ev = torch.cuda.Event()
                ev.record(self._pack_stream)
        else:
            for n, plan, wpk in tp['dgrad_packs']:
                self._pack_dgrad(n, plan, wpk)
        self._pack_event = ev
        tp['dgrad_version'] = self._weights_version

    @staticmethod
    def _pack_dgrad(n, plan, wpk):
        if plan.get('fused_s2'):            # the one-launch form of a stride-2 data gradient (VD_CONV_PARITY4)
            ops.pack_weight_dgrad_s2(n.wp, wpk, Co=n.co_pad, Co_pad=n.co_pad, Ci=n.cin)
        else:
            ops.pack_weight_dgrad(n.wp, wpk, Co=n.co_pad, Co_pad=n.co_pad, Ci=n.cin, kd=n.kd, kh=n.k, kw=n.k,
                                  tap_ids=plan['tap_ids'], src_packed=True)

    @staticmethod
    def _run_segments(segs):
        for s in segs:
            if isinstance(s, Program):
                s.run()
            else:
                s()

    def single_conv_consumer_tensors(self):
        """Outputs of Conv+BN+LeakyReLU cells (no residual) that are read by exactly one node, a convolution: the cells whose
        forward BatchNorm apply could move into the consumer's operand gather (bench.py reports their share)."""
        if getattr(self, '_scc', None) is None:
            use = {}
            for n in self.nodes:
                srcs = []
                if isinstance(n, ConvNode):
                    srcs = [(n.src, True)] + ([(n.residual, False)] if n.residual else [])
                else:
                    srcs = [(getattr(n, a), False) for a in ('src', 'up', 'route', 'a', 'b') if getattr(n, a, None)]
                for t, conv in srcs:
                    use.setdefault(t, []).append(conv)
            self._scc = {n.dst for n in self.conv_nodes if n.bn and not n.residual and not n.stem
                         and use.get(n.dst) == [True] and n.dst not in [r[0] for r in ROUTE_TENSORS]}
        return self._scc

    def _drop_plans(self, keep=None):
        """Forget every cached plan (programs, activation / gradient buffers, graphs) except `keep`, and return their
        memory to the driver."""
        for k in [k for k in self._programs if k != keep]:
            del self._programs[k]
        self._graph_cache.clear()
        self._last_train = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    def _build_or_evict(self, build):
        """Build a plan; when it does not fit beside the cached ones, drop those and build again.  The retry runs AFTER the
        except block: inside it the live exception's traceback still holds the failed build's frame - its half-allocated
        buffers - so empty_cache() could not return them and the second build needed room for one and a half plans."""
        try:
            return build()
        except torch.cuda.OutOfMemoryError:
            pass
        self._drop_plans(keep=None)
        return build()

    def _forward_train(self, x, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t):
        B, H, W = self._in_shape(x)
        bf16s = getattr(self, 'storage', 'fp32') == 'bf16'
        key = ('train_bf16' if bf16s else 'train', B, H, W)
        if key not in self._programs:
            # every input shape owns its plan and buffers (random-shape training visits ten); when the next one does not
            # fit beside the others, drop those and build again - their kernel choices stay in the tuning cache
            self._programs[key] = self._build_or_evict(lambda: (self._build_train_bf16 if bf16s else self._build_train)(B, H, W))
        tp = self._programs[key]
        f32 = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        gt, obj, ctr, scl, wgt, cls = [f32(t) for t in (gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)]
        if self.temporal_out:
            # yolo3_temporal.py:521-533: frame t of window b is matched against ITS targets (args sliced on axis 1);
            # folded here as image b*t_len + t, the order the frames already have
            K = self._k
            for t in (gt, obj, ctr, scl, wgt, cls):
                if t.dim() < 3 or t.shape[0] != B or t.shape[1] != K:
                    raise ValueError("per-frame targets are (B,%d,...) tensors, got %s" % (K, tuple(t.shape)))
            gt, obj, ctr, scl, wgt, cls = [t.reshape((B * K,) + tuple(t.shape[2:])) for t in (gt, obj, ctr, scl, wgt, cls)]
        tp['live'] = (gt, obj, ctr, scl, wgt, cls)
        s = tp['slots']
        s['gt'].value, s['M'].value = gt.data_ptr(), int(gt.shape[1])
        s['obj'].value, s['ctr'].value, s['scl'].value = obj.data_ptr(), ctr.data_ptr(), scl.data_ptr()
        s['wgt'].value, s['cls'].value = wgt.data_ptr(), cls.data_ptr()
        s['smooth'].value = 1 if self._label_smooth else 0
        self._stage_inputs(tp['bufs'], x)
        self._refresh_wamax()
        self._refresh_dgrad(tp, overlap=True)
        self._run_segments(tp['fwd'])
        self._fold_dirty = True            # running stats moved
        self._stats_version += 1
        self._last_train = tp
        L_ = tp['losses']
        if self.temporal_out:
            # :535 [F.mean(F.concat(*l, dim=0)) for l in losses]: four scalars, means over the B*t per-frame losses.
            # backward() computes the gradient of the SUM of the per-frame losses; the 1/(B*t) of the mean is a
            # common factor of the whole gradient and is applied with the optimiser's rescale (sgd_step)
            self._grad_scale = 1.0 / float(L_.shape[0])
            m = L_.mean(dim=0)
            return m[0], m[1], m[2], m[3]
        self._grad_scale = 1.0
        return L_[:, 0], L_[:, 1], L_[:, 2], L_[:, 3]

    def backward(self):
        """autograd.backward(sum_losses) (train_yolov3.py:631): gradient of the sum of all four losses over the
        local batch wrt every parameter, written into the gradient arena."""
        tp = self._last_train
        self._refresh_dgrad(tp)            # only if the weights moved between the forward and this call
        if self._pack_event is not None:
            torch.cuda.current_stream().wait_event(self._pack_event)
            self._pack_event = None
        self._run_segments(tp['bwd'])

    # ------------------------------------------------------------------ call protocol
    def __call__(self, x, *args):
        if self.noback:
            # YOLOV3_noback.hybrid_forward(x1, x2, x3, *args) (yolo3.py:1782): three feature maps, then the targets
            if len(args) < 2:
                raise TypeError("the no-backbone network is called as net(x1, x2, x3[, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t])")
            feats, args = (x, args[0], args[1]), args[2:]
            for t, (nm, c_, d_) in zip(feats, ROUTE_TENSORS):
                if t.dim() != 4 or t.shape[1] != c_:
                    raise ValueError("expected a (B,%d,H/%d,W/%d) feature map, got %s" % (c_, d_, d_, tuple(t.shape)))
            if len(args) == 0:
                return self._forward_infer(feats)
            if len(args) != 6:
                raise TypeError("training call takes (x1, x2, x3, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)")
            return self._forward_train(feats, *args)
        if x.dtype == torch.uint8:
            if x.dim() != (5 if self._k > 1 else 4) or x.shape[-1] != 3 or (self._k > 1 and x.shape[1] != self._k):
                raise ValueError("expected uint8 frames (B,%sH,W,3), got %s" % ("%d," % self._k if self._k > 1 else "", tuple(x.shape)))
        elif self._k > 1:
            if x.dim() != 5 or x.shape[1] != self._k or x.shape[2] != 3:
                raise ValueError("expected a (B,%d,3,H,W) window batch, got %s" % (self._k, tuple(x.shape)))
        elif x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected a (B,3,H,W) batch, got %s" % (tuple(x.shape),))
        if len(args) == 0:
            return self._forward_infer(x)
        if len(args) != 6:
            raise TypeError("training call takes (x, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)")
        return self._forward_train(x, *args)

    # ------------------------------------------------------------------ optimiser / DP hooks
    def _dp_active(self):
        import os
        return torch.distributed.is_available() and torch.distributed.is_initialized() and \
            (torch.distributed.get_world_size(self.process_group) > 1 or os.environ.get("VD_FORCE_DIST") == "1")

    def _bucket_launcher(self, lo, hi, side):
        def f():
            if not self._dp_active() or getattr(self, '_dp_suppress', False):     # (bench.py times a collective-free backward)
                return
            self._dp_stats['buckets'] += 1
            self._dp_stats['bucket_bytes'] += 4 * (hi - lo)
            st = side if side is not None else torch.cuda.current_stream()
            with torch.cuda.stream(st):
                h = torch.distributed.all_reduce(self.grads[lo:hi], group=self._bucket_group or self.process_group,
                                                 async_op=True)
            self._pending_reduces.append(h)
            self._reduced_from = min(self._reduced_from, lo)
        return f

    def allreduce_grads(self):
        """kvstore-'local' replacement (train_yolov3.py:530): RCCL sum all-reduce of the flat gradient arena.
        With bucketing (default) the conv-weight range was already queued in ~32 MB pieces during backward();
        here the handles are awaited and the remaining small range (gamma, beta, head bias) is reduced."""
        if not self._dp_active() or getattr(self, '_dp_suppress', False):
            return
        wt = [m.w_off for m in self.conv_nodes if self._node_trainable(m)[0]]
        wt_lo = min(wt) if wt else self.n_weight             # frozen prefix (freeze_base): zeros, never reduced
        if self._pending_reduces:
            for h in self._pending_reduces:
                h.wait()
            self._pending_reduces = []
            lo = self._reduced_from
            self._reduced_from = self.n_params
            if lo > wt_lo:                               # anything before the first bucket (not expected)
                torch.distributed.all_reduce(self.grads[wt_lo:lo], group=self.process_group)
            torch.distributed.all_reduce(self.grads[self.n_weight:], group=self.process_group)
        else:
            torch.distributed.all_reduce(self.grads[wt_lo:], group=self.process_group)

    def sgd_step(self, lr, momentum, wd, batch_size, no_wd=False):
        """gluon.Trainer('sgd').step(batch_size) (train_yolov3.py:527-530,634) over the trainable parameters only
        (grad_req 'null' => no update and no weight decay, wrappers.py:55-57), with each parameter's lr_mult / wd_mult;
        `no_wd=True` is shorthand for wd_mult = 0 on gamma / beta / bias (--no_wd, :495-497)."""
        rescale = self._grad_scale / float(batch_size)
        nw = self.n_weight
        for lo, hi, lr_mult, wd_mult in self._optimizer_ranges():
            wd_ = 0.0 if (no_wd and lo >= nw) else wd * wd_mult
            ops.sgd_momentum(self.weights[lo:hi], self.grads[lo:hi], self.momentum_buf[lo:hi], lr * lr_mult, momentum,
                             wd_, rescale)
        self._params_changed()

    # ------------------------------------------------------------------ checkpoints
    def state_arrays(self):
        return OrderedDict((k, p.data().cpu().numpy()) for k, p in self._params.items())

    def save_parameters(self, path):
        from .params_io import save_params
        save_params(path, self.state_arrays())

    def load_parameters(self, path, allow_missing=False, ignore_extra=False):
        from .params_io import load_params
        arrs = load_params(path)
        for k, p in self._params.items():
            if k in arrs:
                p.set_data(torch.from_numpy(np.ascontiguousarray(arrs[k], dtype=np.float32)))
            elif not allow_missing and not re.search(r"(anchor|offset)", k):
                raise KeyError("parameter %s missing in %s" % (k, path))
        if not ignore_extra:
            extra = [k for k in arrs if k not in self._params and not re.search(r"(anchor|offset)", k)]
            if extra:
                raise KeyError("unexpected parameters in %s: %s" % (path, extra[:5]))


def yolo3_darknet53(classes, pretrained_base=False, norm_layer=None, norm_kwargs=None, freeze_base=False,
                    k=None, k_join_type=None, k_join_pos=None, block_conv_type='2', temporal=False, t_out=False,
                    corr_d=None, **kwargs):
    """wrappers.py:9-110 -> YOLOV3T (yolo3.py:959-1054).  norm_layer='syncbn' (the reference passes
    SyncBatchNorm) selects the SyncBN collective.  k>1 builds the temporal-window variants; `t_out=True`
    (--temp --mult_out) builds YOLOV3Temporal with per-frame outputs (yolo3_temporal.py:286-555, t = k = 5)."""
    k = 1 if k is None else int(k)
    if temporal or t_out:                                         # wrappers.py:96-98
        if corr_d:
            raise NotImplementedError("YOLOV3Temporal with a correlation branch (corr_d) is outside the built scope")
        assert k == 5, "Currently only support t=5 but will increase to more later"        # yolo3_temporal.py:399
        assert block_conv_type in ('2', '3', '21')
        scope = (norm_kwargs or {}).get('scope', 'all') if norm_layer == 'syncbn' else None
        if not t_out:
            # --temp without --mult_out (yolo3_temporal.py:326-333,436-447): strided 2+1-D side branches, ONE output for the
            # centre frame.  The detection blocks get the squeezed (B,C,h,w) routes, so only the 2-D blocks are defined
            # (a 3-D block would be handed a 4-D tensor in the reference as well).
            if block_conv_type != '2':
                raise NotImplementedError("YOLOV3Temporal(t_out=False) squeezes the frame axis before the detection blocks: "
                                          "block_conv_type must be '2'")
            net = YOLOV3(classes, syncbn_scope=scope, k=k, temporal_side=True, **kwargs)
        else:
            net = YOLOV3(classes, syncbn_scope=scope, k=k, block_conv_type=block_conv_type, temporal_out=True, **kwargs)
        if freeze_base:
            for name, p in net.collect_params('stages.*').items():
                p.grad_req = 'null'
        return net
    # yolo3.py:978-985
    if block_conv_type in ('3', '21'):
        assert k > 1, "k must be greater than 1 to use 3D or 2+1D convolutions"
        assert k_join_pos == 'late', "only 'late' pooling can be used when using 3D or 2+1D convolutions"
        assert k_join_type is not None, "please specify a k_join_type: max, mean, or cat"
    assert block_conv_type in ('2', '3', '21')
    assert k_join_type in [None, 'max', 'mean', 'cat']
    assert k_join_pos in [None, 'early', 'late']
    if k > 1:
        if k_join_type is None or k_join_pos is None:
            raise NotImplementedError("k>1 needs k_join_type (max|mean|cat) and k_join_pos (early|late)")
    scope = None
    if norm_layer == 'syncbn':
        scope = (norm_kwargs or {}).get('scope', 'all')
    net = YOLOV3(classes, syncbn_scope=scope, k=k, k_join_type=k_join_type, k_join_pos=k_join_pos,
                 block_conv_type=block_conv_type, **kwargs)
    if freeze_base:                          # wrappers.py:55-57: every Darknet parameter leaves the gradient / update
        for name, p in net.collect_params('stages.*').items():
            p.grad_req = 'null'
    return net


def yolo3_no_backbone(classes, norm_layer=None, norm_kwargs=None, **kwargs):
    """wrappers.py:133-161 -> YOLOV3_noback (yolo3.py:1730-1870): neck + heads on cached Darknet-53 feature maps,
    called as net(x1, x2, x3[, targets]) with x1 (B,256,H/8,W/8), x2 (B,512,H/16,W/16), x3 (B,1024,H/32,W/32).
    Parameter names are the transitions.* / yolo_blocks.* / yolo_outputs.* subset of the full network's."""
    scope = None
    if norm_layer == 'syncbn':
        scope = (norm_kwargs or {}).get('scope', 'all')
    return YOLOV3(classes, syncbn_scope=scope, noback=True, **kwargs)
