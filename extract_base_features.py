#!/usr/bin/env python
"""Cache Darknet-53 backbone features on MI355X — drop-in for the reference's extract_base_features.py.

Follows /root/reference/extract_base_features.py: flags :34-54, get_dataloader :106-112, extract :115-160
(f1 = features[:15](x), f2 = features[15:24](f1), f3 = features[24:](f2), saved per sample as
`<file id>_F1.npy`, `_F2.npy`, `_F3.npy`, each (C,h,w) fp32) and main :163-199 (output under
models/<network>/<save_dir>/<dataset>).  The trunk runs through viddet_amd.model.YOLOV3.extract_features (the
HIP conv kernels in inference mode); `train_yolov3.py --features_dir <that directory>` then trains the neck and
heads on the cached maps (yolo3_no_backbone).  The reference loads ImageNet-pretrained darknet53 weights from the
GluonCV model zoo; offline, weights come from --model_path (a checkpoint of the full network) or --random_init.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from viddet_amd.data import SyntheticDetection, YOLO3VideoInferenceTransform, Loader, feature_file_id
from viddet_amd.model import yolo3_darknet53


def parse_flags(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    A = ap.add_argument
    A("--network", default="darknet53", help="Base network name: darknet53")
    A("--dataset", default="voc", help="Dataset (synthetic frames stand in offline; the class count is kept)")
    A("--save_dir", default="features", help="Directory name under models/<network>/ to save the features in")
    A("--batch_size", type=int, default=1)
    A("--data_shape", type=int, default=416)
    A("--frames", type=float, default=0.04)
    A("--gpus", default="0")
    A("--num_workers", type=int, default=8)
    A("--model_path", default="", help="checkpoint of the full yolo3_darknet53 network to take the trunk from")
    A("--random_init", action="store_true", help="seeded random trunk (no pretrained weights exist offline)")
    A("--seed", type=int, default=233)
    A("--synthetic_samples", type=int, default=128)
    A("--dataset_seed", type=int, default=None, help="seed of the synthetic dataset (train uses --seed, val --seed+1)")
    return ap.parse_args(argv)


def extract(save_dir, net, dataset, loader):
    """extract_base_features.py:115-160."""
    n = 0
    for x, _, sidx in loader:
        f1, f2, f3 = [f.cpu().numpy() for f in net.extract_features(torch.from_numpy(x).cuda())]
        for i, idx in enumerate(sidx):
            fid = feature_file_id(dataset.sample_path(int(idx)))
            np.save(os.path.join(save_dir, fid + "_F1.npy"), f1[i])
            np.save(os.path.join(save_dir, fid + "_F2.npy"), f2[i])
            np.save(os.path.join(save_dir, fid + "_F3.npy"), f3[i])
            n += 1
    return n


def main(argv=None):
    FLAGS = parse_flags(argv)
    if FLAGS.network != "darknet53":
        raise NotImplementedError("Backbone CNN model {} not implemented.".format(FLAGS.network))
    if not torch.cuda.is_available():
        raise SystemExit("extract_base_features.py needs an MI355X: the HIP path has no CPU fallback")
    dataset = SyntheticDetection(FLAGS.dataset, num_samples=FLAGS.synthetic_samples,
                                 seed=FLAGS.seed if FLAGS.dataset_seed is None else FLAGS.dataset_seed)
    batch_size = min(FLAGS.batch_size, len(dataset))                   # :170-172 fix for tiny datasets
    loader = Loader(dataset, YOLO3VideoInferenceTransform(FLAGS.data_shape, FLAGS.data_shape), batch_size, train=False,
                    last_batch="keep")
    net = yolo3_darknet53(dataset.classes, pretrained_base=False)
    if FLAGS.model_path:
        net.load_parameters(FLAGS.model_path)
    elif FLAGS.random_init:
        net.initialize(init="he", seed=FLAGS.seed)
    else:
        raise SystemExit("give --model_path or --random_init (the GluonCV model zoo is not reachable offline)")
    if FLAGS.dataset in ("voc", "coco", "det", "vid"):                 # :190-194
        save_dir = os.path.join("models", FLAGS.network, FLAGS.save_dir, FLAGS.dataset)
    else:
        save_dir = os.path.join("models", FLAGS.network, FLAGS.save_dir)
    os.makedirs(save_dir, exist_ok=True)
    n = extract(save_dir, net, dataset, loader)
    print("saved features of %d samples to %s" % (n, save_dir))
    return save_dir


if __name__ == "__main__":
    main()
