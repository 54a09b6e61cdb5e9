#!/usr/bin/env python
"""Run YOLOv3 (Darknet-53) detection on MI355X — drop-in for the reference's detect_yolo3.py entry point.

Follows /root/reference/detect_yolo3.py: flags :41-118, detect() :198-272 (inference loop, clip, keep rows with
id >= 0, normalise boxes by W, collect [id, score, x1, y1, x2, y2] per image path), save_predictions :275-330
(one `path,id,score,x1,y1,x2,y2` text file per image), load_predictions/evaluate :333-448,659-695 (VOC mAP; the
reference's `sid=` keyword bug at :693 is not reproduced), main :792-939 (net build :871-892).
The network underneath is viddet_amd.model.YOLOV3 (hand-written HIP kernels).  Frames shard across ranks with
no collective on the data path (inference = replicas only); the per-image box lists are gathered to rank 0, which writes
the prediction files and evaluates.  Visualisation / worst-video / COCO+VID metrics are out of scope.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from viddet_amd import dist as vdist
from viddet_amd.data import SyntheticDetection, SyntheticCombined, YOLO3VideoInferenceTransform, Loader
from viddet_amd.metrics import VOCMApMetric
from viddet_amd.hierarchy import ClassTree, get_class_map, hierarchical_nms, iou  # noqa: F401  (detect_yolo3.py:698-789)
from viddet_amd.model import yolo3_darknet53
from train_yolov3 import _list, _bool


def parse_flags(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    A = ap.add_argument
    A("--model_path", default="yolo3_darknet53_voc_best.params")
    A("--network", default="darknet53")
    A("--dataset", type=_list, default=["voc"])
    A("--trained_on", default="")
    A("--save_prefix", default="0001")
    A("--save_dir", default="results")
    A("--metrics", type=_list, default=["voc", "coco"])
    A("--batch_size", type=int, default=1)
    A("--data_shape", type=int, default=416)
    A("--detection_threshold", type=float, default=0.5)
    A("--max_do", type=int, default=-1)
    A("--every", type=float, default=25)
    A("--window", type=_list, default=["1", "1"])
    A("--k_join_type", default=None)
    A("--k_join_pos", default=None)
    A("--block_conv_type", default="2")
    A("--rnn_pos", default=None)
    A("--corr_pos", default=None)
    A("--corr_d", type=int, default=4)
    A("--motion_stream", default=None)
    A("--stream_gating", default=None)
    A("--conv_types", type=_list, default=["2"] * 6)
    A("--h_join_type", default=None)
    A("--hier", type=_list, default=["1"] * 5)
    A("--mult_out", type=_bool, nargs="?", const=True, default=False)
    A("--temp", type=_bool, nargs="?", const=True, default=False)
    A("--visualise", type=_bool, nargs="?", const=True, default=False)
    A("--per_frame_metric", type=_bool, nargs="?", const=True, default=False)
    A("--worst_video_path", default=None)
    A("--display_gt", type=_bool, nargs="?", const=True, default=True)
    A("--model_agnostic", type=_bool, nargs="?", const=True, default=False)
    A("--metric_agnostic", type=_bool, nargs="?", const=True, default=False)
    A("--gpus", type=_list, default=["0"])
    A("--num_workers", type=int, default=8)
    A("--new_model", type=_bool, nargs="?", const=True, default=False)
    A("--offset", type=int, default=0)
    A("--hier_level", type=int, default=10)
    A("--synthetic_samples", type=int, default=32)
    A("--precision", default="fp32", choices=["fp32", "bf16"],
      help="(no reference counterpart) bf16: bf16 storage + bf16 MFMA inference, fp32 heads (net.set_precision, BASELINE configs[1])")
    A("--synthetic_classes", type=int, default=None, help="classes per dataset of the synthetic combined set (default: the datasets' own counts)")
    A("--random_init", type=_bool, nargs="?", const=True, default=False,
      help="skip load_parameters (no checkpoint available offline)")
    return ap.parse_args(argv)


def detect(net, dataset, loader, max_do=-1):
    """detect_yolo3.py:198-272."""
    net.set_nms(nms_thresh=0.45, nms_topk=400)
    boxes = dict()
    if max_do < 0:
        max_do = len(dataset)
    c = 0
    for x, _label, sidxs in loader:
        ids, scores, bboxes = net(torch.from_numpy(x).cuda())
        W = x.shape[-2] if x.dtype == np.uint8 else x.shape[-1]           # uint8 frames are (B,H,W,3)
        ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        bboxes = np.clip(bboxes.cpu().numpy(), 0, W)                       # :228 clip to image size
        for id_, score, box, sidx in zip(ids, scores, bboxes, sidxs):
            file = dataset.sample_path(int(sidx))
            valid = np.where(id_.flat >= 0)[0]                             # :255 boxes that have a class
            box = box[valid, :] / W                                        # :257 normalise boxes
            for i_, b_, s_ in zip(id_.flat[valid].astype(int), box, score.flat[valid]):
                boxes.setdefault(file, []).append([i_, s_] + list(b_))
        c += x.shape[0]
        if c > max_do:
            break
    return boxes


def save_predictions(save_dir, dataset, boxes, overwrite=True, max_do=-1):
    """detect_yolo3.py:275-330: one text file per image, lines `path,id,score,x1,y1,x2,y2`."""
    os.makedirs(save_dir, exist_ok=True)
    n = len(dataset) if max_do < 0 else min(max_do, len(dataset))
    for idx in range(n):
        img_path = dataset.sample_path(idx)
        file_id = os.path.split(img_path)[1].split(".")[0]
        out = os.path.join(save_dir, file_id + ".txt")
        if os.path.exists(out) and not overwrite:
            continue
        with open(out, "w") as f:
            for box in boxes.get(img_path, []):
                f.write("{},{},{},{},{},{},{}\n".format(img_path, box[0], box[1], box[2], box[3], box[4], box[5]))


def load_predictions(save_dir, dataset, max_do=-1):
    """detect_yolo3.py:333-400 (plain, non-agnostic branch)."""
    boxes = dict()
    n = len(dataset) if max_do < 0 else min(max_do, len(dataset))
    for idx in range(n):
        img_path = dataset.sample_path(idx)
        file_id = os.path.split(img_path)[1].split(".")[0]
        p = os.path.join(save_dir, file_id + ".txt")
        if not os.path.exists(p):
            continue
        with open(p) as f:
            for line in f:
                v = line.rstrip().split(",")
                boxes.setdefault(v[0], []).append([int(v[1])] + [float(t) for t in v[2:7]])
    return boxes


def evaluate(metrics, dataset, predictions, data_shape):
    """detect_yolo3.py:659-695: feed saved predictions and (resized, normalised) ground truth to the metrics."""
    tf = YOLO3VideoInferenceTransform(data_shape, data_shape)
    for idx in range(len(dataset)):
        img, label = dataset[idx]
        _, gt, _ = tf(img, label, idx)
        gt_boxes = gt[:, :4] / data_shape
        pred = np.asarray(predictions.get(dataset.sample_path(idx), np.zeros((0, 6))), dtype=np.float64).reshape(-1, 6)
        for m in metrics:
            m.update([pred[:, 2:6]], [pred[:, 0]], [pred[:, 1]], [gt_boxes], [gt[:, 4]], [gt[:, 5]])
    return [m.get() for m in metrics]


def main(argv=None):
    FLAGS = parse_flags(argv)
    FLAGS.window = [int(s) for s in FLAGS.window]
    if FLAGS.window[0] == 1:
        FLAGS.k_join_type = FLAGS.k_join_pos = None
    # accepted for command-line compatibility, refused when they would change the result (never silently ignored):
    # research variants, visualisation, the VID metric's options, evaluation on another dataset's class list
    for flag in ("temp", "mult_out", "new_model", "motion_stream", "rnn_pos", "corr_pos", "visualise", "model_agnostic",
                 "metric_agnostic", "offset", "per_frame_metric", "worst_video_path", "trained_on"):
        v = getattr(FLAGS, flag)
        if v and not (isinstance(v, str) and not v.strip()):
            raise NotImplementedError("--%s is outside the yolo3_darknet53 hot path" % flag)
    rank, world = vdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("detect_yolo3.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    name = FLAGS.dataset[0]
    if len(FLAGS.dataset) > 1:          # detect_yolo3.py:166-167: several datasets = the combined set with its class tree
        dataset = SyntheticCombined(FLAGS.dataset, num_samples=FLAGS.synthetic_samples, classes_per_set=FLAGS.synthetic_classes)
    else:
        dataset = SyntheticDetection(name, num_samples=FLAGS.synthetic_samples)
    # frames travel as uint8 and are normalised on the device (vd_preprocess_u8_nchw: the transform's own arithmetic)
    loader = Loader(dataset, YOLO3VideoInferenceTransform(FLAGS.data_shape, FLAGS.data_shape, device_normalize=True),
                    FLAGS.batch_size, train=False, last_batch="keep", rank=rank, world=world)
    # detect_yolo3.py:871-892
    net = yolo3_darknet53(dataset.classes, pretrained_base=False, k=FLAGS.window[0], k_join_type=FLAGS.k_join_type,
                          k_join_pos=FLAGS.k_join_pos, block_conv_type=FLAGS.block_conv_type)
    if FLAGS.random_init:
        net.initialize(init="he", obj_bias=-2.0)
    else:
        net.load_parameters(FLAGS.model_path)
    net.set_precision(FLAGS.precision)
    save_dir = os.path.join(FLAGS.save_dir, FLAGS.save_prefix, "pred")
    boxes = detect(net, dataset, loader, FLAGS.max_do)
    if world > 1:
        # frames are sharded over the ranks (replicas, no collective on the data path); the per-image box lists (host
        # objects) are merged so that ONE rank writes every file - a rank must never write an (empty) file for an image
        # another rank detected on - and evaluates the whole set as the reference's single process does
        merged = dict()
        for part in vdist.all_gather_objects(boxes):
            merged.update(part)
        boxes = merged
        if rank != 0:
            return None
    save_predictions(save_dir, dataset, boxes, max_do=FLAGS.max_do)
    if "voc" in FLAGS.metrics:
        preds = load_predictions(save_dir, dataset, FLAGS.max_do)
        if len(FLAGS.dataset) > 1:                                              # detect_yolo3.py:898-899 (class-tree sets)
            preds = hierarchical_nms(preds, dataset, level_thresh=FLAGS.hier_level)
        (names, values), = evaluate([VOCMApMetric(iou_thresh=0.5, class_names=dataset.classes)], dataset, preds,
                                    FLAGS.data_shape)
        print("{}={:.4f}".format(names[-1], values[-1]))
        return names, values
    return None


if __name__ == "__main__":
    main()
