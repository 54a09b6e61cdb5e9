#!/usr/bin/env python
"""Train YOLOv3 (Darknet-53) on MI355X — drop-in for the reference's train_yolov3.py entry point.

Flag names, defaults, loop semantics, log-line formats and checkpoint naming follow
/root/reference/train_yolov3.py (flags :45-164, get_dataset :167-231, save_params :289-309, resume :312-329,
validate :434-489, train :492-680, main :683-777).  What is underneath is new: one process per GPU
(launch N ranks with `python -m torch.distributed.run --nproc-per-node N train_yolov3.py ...`; the
reference's `--gpus 0,1,..` single-process device loop is replaced by ranks), the network is
viddet_amd.model.YOLOV3 (hand-written HIP kernels, fixed backward schedule), gradients are summed with
one RCCL all-reduce and `--syncbn` becomes a SyncBN collective.

absl is not installed here: argparse re-creates the same flag surface (list flags take comma lists).
`--features_dir` trains the neck/heads on cached backbone features (yolo3_no_backbone; features written by
extract_base_features.py); `--window 5 --temp --mult_out` builds YOLOV3Temporal with per-frame outputs.  Variants
outside the built scope (--temp without --mult_out, --motion_stream, --new_model, --hier, --rnn_pos,
--corr_pos) are accepted and rejected with NotImplementedError like the reference's own guards.
"""
import argparse
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from viddet_amd import dist as vdist
from viddet_amd.data import (SyntheticDetection, MixupDetection, YOLO3VideoTrainTransform, YOLO3VideoInferenceTransform, Loader,
                             FeatureDataset, YOLO3NBVideoTrainTransform, YOLO3NBVideoInferenceTransform)
from viddet_amd.metrics import VOCMApMetric, VOCMApMetricTemporal, LossMetric
from viddet_amd.model import yolo3_darknet53, yolo3_no_backbone
from viddet_amd.schedule import LRScheduler, LRSequential
from viddet_amd.video import Rng


def _list(s):
    return [v for v in str(s).replace(" ", "").split(",") if v != ""]


def _bool(v):
    return str(v).lower() in ("1", "true", "t", "yes", "y")


def parse_flags(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    A = ap.add_argument
    A("--network", default="darknet53")
    A("--dataset", type=_list, default=["voc"])
    A("--dataset_val", type=_list, default=[])
    A("--trained_on", default="")
    A("--save_prefix", default="0001")
    A("--log_interval", type=int, default=100)
    A("--save_interval", type=int, default=-10)
    A("--val_interval", type=int, default=1)
    A("--resume", default="")
    A("--nd_only", type=_bool, nargs="?", const=True, default=False)
    A("--batch_size", type=int, default=64)
    A("--epochs", type=int, default=200)
    A("--start_epoch", type=int, default=0)
    A("--data_shape", type=int, default=416)
    A("--lr", type=float, default=0.001)
    A("--lr_mode", default="step")
    A("--lr_decay", type=float, default=0.1)
    A("--lr_decay_period", type=int, default=0)
    A("--lr_decay_epoch", type=_list, default=["160", "180"])
    A("--warmup_epochs", type=int, default=0)
    A("--momentum", type=float, default=0.9)
    A("--wd", type=float, default=0.0005)
    A("--pretrained_cnn", type=_bool, nargs="?", const=True, default=True)
    A("--syncbn", type=_bool, nargs="?", const=True, default=False)
    A("--syncbn_scope", default="all", choices=["all", "reference"],
      help="'reference' = only the stem and stride-2 convs, which is all the reference's --syncbn reaches")
    A("--no_random_shape", type=_bool, nargs="?", const=True, default=False)
    A("--random_shape_interval", type=int, default=10,
      help="batches between two draws of the training shape (the reference hard-codes interval=10, train_yolov3.py:270)")
    A("--no_wd", type=_bool, nargs="?", const=True, default=False)
    A("--mixup", type=_bool, nargs="?", const=True, default=False)
    A("--no_mixup_epochs", type=int, default=20)
    A("--label_smooth", type=_bool, nargs="?", const=True, default=False)
    A("--freeze_base", type=_bool, nargs="?", const=True, default=False)
    A("--allow_empty", type=_bool, nargs="?", const=True, default=True)
    A("--mult_out", type=_bool, nargs="?", const=True, default=False)
    A("--temp", type=_bool, nargs="?", const=True, default=False)
    A("--gpus", type=_list, default=["0"])
    A("--num_workers", type=int, default=-1)
    A("--new_model", type=_bool, nargs="?", const=True, default=False)
    A("--num_samples", type=int, default=-1)
    A("--every", type=float, default=25)
    A("--window", type=_list, default=["1", "1"])
    A("--seed", type=int, default=233)
    A("--features_dir", default=None)
    A("--k_join_type", default=None)
    A("--k_join_pos", default=None)
    A("--block_conv_type", default="2")
    A("--rnn_pos", default=None)
    A("--corr_pos", default=None)
    A("--corr_d", type=int, default=0)
    A("--motion_stream", default=None)
    A("--stream_gating", default=None)
    A("--conv_types", type=_list, default=["2"] * 6)
    A("--h_join_type", default=None)
    A("--hier", type=_list, default=["1"] * 5)
    A("--max_epoch_time", type=int, default=-1)
    A("--synthetic_samples", type=int, default=128, help="size of the synthetic stand-in dataset")
    A("--storage", default="fp32", choices=["fp32", "bf16"],
      help="(no reference counterpart) bf16: training activations and gradients in bf16, fp32 statistics / losses / master "
           "weights - net.set_storage('bf16'), BASELINE configs[4]; single-frame yolo3_darknet53 only")
    return ap.parse_args(argv)


FLAGS = None


def get_dataset(dataset_name, dataset_val_name, save_prefix=""):
    """train_yolov3.py:167-231 — dataset readers are out of scope offline; synthetic frames stand in with the
    same class count and sample contract."""
    name = dataset_name[0] if len(dataset_name) == 1 else "comb"
    win = dict(window=int(FLAGS.window[0]), mult_out=FLAGS.mult_out)      # :196-206 VID windows / per-frame labels
    train_ds = SyntheticDetection(name, num_samples=FLAGS.synthetic_samples, seed=FLAGS.seed, **win)
    val_ds = SyntheticDetection(name, num_samples=max(8, FLAGS.synthetic_samples // 4), seed=FLAGS.seed + 1, **win)
    if FLAGS.features_dir is not None:                 # :177-205 datasets built with features_dir
        train_ds, val_ds = FeatureDataset(train_ds, FLAGS.features_dir), FeatureDataset(val_ds, FLAGS.features_dir)
    if FLAGS.mult_out:                                 # :207-210
        val_metric = VOCMApMetricTemporal(t=int(FLAGS.window[0]), iou_thresh=0.5, class_names=val_ds.classes)
    else:
        val_metric = VOCMApMetric(iou_thresh=0.5, class_names=val_ds.classes)
    if FLAGS.num_samples < 0:
        FLAGS.num_samples = len(train_ds)
    if FLAGS.mixup:                                    # :227-229
        if int(FLAGS.window[0]) > 1 or FLAGS.features_dir is not None:
            raise NotImplementedError("--mixup blends single frames (gluoncv MixupDetection): not with --window k > 1 / --features_dir")
        train_ds = MixupDetection(train_ds)
    return train_ds, val_ds, val_metric


def get_dataloader(train_dataset, val_dataset, data_shape, batch_size, rank, world):
    """train_yolov3.py:234-286; per-rank batch = batch_size / world (split_and_load, :603-606)."""
    w = h = data_shape
    per_rank = batch_size // world
    if FLAGS.features_dir is not None:                 # :238-250 the input is pre-saved features
        train_loader = Loader(train_dataset, YOLO3NBVideoTrainTransform(FLAGS.window[0], w, h, train_dataset.num_class),
                              per_rank, train=True, shuffle=True, seed=FLAGS.seed, rank=rank, world=world)
        val_loader = Loader(val_dataset, YOLO3NBVideoInferenceTransform(w, h), per_rank, train=False,
                            last_batch="discard", rank=rank, world=world)
        return train_loader, val_loader
    rng = Rng.seeded(FLAGS.seed + rank)
    if FLAGS.no_random_shape:                          # :258-262
        tf = YOLO3VideoTrainTransform(w, h, train_dataset.num_class, rng, mixup=FLAGS.mixup)
    else:                                              # :263-271 the default: a random side of 320 ... 608 every 10 batches
        tf = [YOLO3VideoTrainTransform(x * 32, x * 32, train_dataset.num_class, rng, mixup=FLAGS.mixup) for x in range(10, 20)]
    train_loader = Loader(train_dataset, tf, per_rank, train=True, shuffle=True, seed=FLAGS.seed, rank=rank, world=world,
                          interval=FLAGS.random_shape_interval, num_workers=FLAGS.num_workers)
    # validation frames travel as uint8 and are normalised on the device (same arithmetic, a quarter of the bytes)
    val_loader = Loader(val_dataset, YOLO3VideoInferenceTransform(w, h, device_normalize=True), per_rank, train=False,
                        last_batch="keep", rank=rank, world=world)
    return train_loader, val_loader


def save_params(net, best_map, current_map, epoch, save_interval, prefix):
    """train_yolov3.py:289-309 (same file names, rolling delete when save_interval < 0)."""
    current_map = float(current_map)
    if current_map > best_map[0]:
        best_map[0] = current_map
        net.save_parameters("{:s}_best.params".format(prefix))
        with open(prefix + "_best_map.log", "a") as f:
            f.write("{:04d}:\t{:.4f}\n".format(epoch, current_map))
    if save_interval > 0 and epoch % save_interval == 0:
        net.save_parameters("{:s}_{:04d}.params".format(prefix, epoch))
    if save_interval < 0:
        net.save_parameters("{:s}_{:04d}.params".format(prefix, epoch))
        if epoch % -save_interval == 0:
            for d in range(max(0, epoch + save_interval + 1), epoch):
                p = "{:s}_{:04d}.params".format(prefix, d)
                if os.path.exists(p):
                    os.remove(p)


def resume(net, resume_path, start_epoch):
    """train_yolov3.py:312-329."""
    if start_epoch == -1:
        files = sorted(f for f in os.listdir(resume_path.strip()) if "_0" in f and ".params" in f)
        resume_file = files[-1]
        start_epoch = int(resume_file[:-7].split("_")[-1]) + 1
        net.load_parameters(os.path.join(resume_path.strip(), resume_file))
    else:
        net.load_parameters(resume_path.strip())
    return start_epoch


def get_net(classes, rank_world):
    """train_yolov3.py:332-431 ('ours' definition, darknet53 only)."""
    if FLAGS.network != "darknet53":
        raise NotImplementedError("Backbone CNN model {} not implemented.".format(FLAGS.network))
    for flag in ("new_model", "motion_stream", "rnn_pos", "corr_pos"):
        if getattr(FLAGS, flag):
            raise NotImplementedError("--%s selects a research variant outside the yolo3_darknet53 hot path" % flag)
    k = int(FLAGS.window[0])
    if FLAGS.features_dir is not None:                 # :335-342
        net = yolo3_no_backbone(classes, norm_layer="syncbn" if FLAGS.syncbn and rank_world[1] > 1 else None,
                                norm_kwargs={"scope": FLAGS.syncbn_scope})
        net.initialize(init="he", seed=FLAGS.seed)
        start_epoch = FLAGS.start_epoch
        if FLAGS.resume.strip():
            start_epoch = resume(net, FLAGS.resume, FLAGS.start_epoch)
        return net, start_epoch
    net = yolo3_darknet53(classes, pretrained_base=False,
                          norm_layer="syncbn" if FLAGS.syncbn and rank_world[1] > 1 else None,
                          norm_kwargs={"scope": FLAGS.syncbn_scope}, freeze_base=FLAGS.freeze_base,
                          k=k, k_join_type=FLAGS.k_join_type, k_join_pos=FLAGS.k_join_pos,
                          block_conv_type=FLAGS.block_conv_type, temporal=FLAGS.temp, t_out=FLAGS.mult_out)   # :348-360
    net.initialize(init="he", seed=FLAGS.seed)
    if FLAGS.storage == "bf16":
        net.set_storage("bf16")                       # raises for the window / per-frame-output variants
    start_epoch = FLAGS.start_epoch
    if FLAGS.resume.strip():
        start_epoch = resume(net, FLAGS.resume, FLAGS.start_epoch)
    return net, start_epoch


def validate(net, val_data, eval_metric, data_shape):
    """train_yolov3.py:434-489.  With N ranks each one runs the network on its shard of the validation set and the
    per-image detections are exchanged, so the metric (on every rank) covers the whole set in sample order exactly as
    the reference's single process does; only the forward passes are sharded."""
    from viddet_amd.metrics import update_metric_sharded
    eval_metric.reset()
    net.set_nms(nms_thresh=0.45, nms_topk=400)
    records = []
    for batch in val_data:
        label, sidxs = batch[-2], batch[-1]
        if FLAGS.features_dir is not None:             # :444-461 net(x1, x2, x3)
            ids, scores, bboxes = net(*[torch.from_numpy(f).cuda() for f in batch[:3]])
        else:
            ids, scores, bboxes = net(torch.from_numpy(batch[0]).cuda())
        det_ids, det_scores = ids.cpu().numpy(), scores.cpu().numpy()
        # :458/:477 clip to "the last dim of batch[0]" - the image width, or (as in the reference) the width of the
        # stride-8 feature map when the batch holds cached features
        width = batch[0].shape[-2] if batch[0].dtype == np.uint8 else batch[0].shape[-1]     # (B,H,W,3) uint8 frames
        det_bboxes = np.clip(bboxes.cpu().numpy(), 0, width)
        for j in range(det_ids.shape[0]):
            records.append((int(sidxs[j]), det_bboxes[j], det_ids[j], det_scores[j], label[j][..., :4], label[j][..., 4:5],
                            label[j][..., 5:6] if label.shape[-1] > 5 else None))
    validate.last_count = update_metric_sharded(eval_metric, records)
    return eval_metric.get()


def train(net, train_data, train_dataset, val_data, eval_metric, save_prefix, start_epoch, num_samples, rank, world):
    """train_yolov3.py:492-680."""
    net.collect_params().reset_ctx(None)
    if FLAGS.no_wd:                                                               # :495-497
        for k, v in net.collect_params(".*beta|.*gamma|.*bias").items():
            v.wd_mult = 0.0
    if FLAGS.label_smooth:
        net._target_generator._label_smooth = True
    if FLAGS.lr_decay_period > 0:
        lr_decay_epoch = list(range(FLAGS.lr_decay_period, FLAGS.epochs, FLAGS.lr_decay_period))
    else:
        lr_decay_epoch = FLAGS.lr_decay_epoch
    lr = FLAGS.lr
    tmp = []
    for e in lr_decay_epoch:                                                      # :507-514
        if int(e) <= start_epoch:
            lr = lr * FLAGS.lr_decay
        else:
            tmp.append(int(e) - start_epoch - FLAGS.warmup_epochs)
    lr_decay_epoch = tmp
    num_batches = max(1, num_samples // FLAGS.batch_size)
    lr_scheduler = LRSequential([
        LRScheduler("linear", base_lr=0, target_lr=lr, nepochs=FLAGS.warmup_epochs, iters_per_epoch=num_batches),
        LRScheduler(FLAGS.lr_mode, base_lr=lr, nepochs=FLAGS.epochs - FLAGS.warmup_epochs - start_epoch,
                    iters_per_epoch=num_batches, step_epoch=lr_decay_epoch, step_factor=FLAGS.lr_decay, power=2)])
    obj_metrics, center_metrics = LossMetric("ObjLoss"), LossMetric("BoxCenterLoss")
    scale_metrics, cls_metrics = LossMetric("BoxScaleLoss"), LossMetric("ClassLoss")
    logger = logging.getLogger()
    logger.setLevel(logging.INFO if rank == 0 else logging.WARNING)
    log_file_path = save_prefix + "_train.log"
    log_dir = os.path.dirname(log_file_path)
    if log_dir and not os.path.exists(log_dir):
        os.makedirs(log_dir, exist_ok=True)
    if rank == 0:
        logger.addHandler(logging.FileHandler(log_file_path))
    logger.info("Start training from [Epoch {}]".format(start_epoch))
    best_map = [0]
    if FLAGS.resume.strip() and os.path.exists(save_prefix + "_best_map.log"):
        with open(save_prefix + "_best_map.log") as f:
            best_map = [float(f.readlines()[-1].split()[1])]
    num_update = 0
    for epoch in range(start_epoch, FLAGS.epochs + 1):
        if FLAGS.mixup:                                 # :571-581 beta(1.5, 1.5) blends, switched off for the last epochs
            if epoch >= FLAGS.epochs - FLAGS.no_mixup_epochs:
                train_dataset.set_mixup(None)
            else:
                train_dataset.set_mixup(np.random.beta, 1.5, 1.5)
        st = tic = btic = time.time()
        i = -1
        batch_size = FLAGS.batch_size
        for i, batch in enumerate(train_data):
            if FLAGS.max_epoch_time > 0 and (time.time() - st) / 60 > FLAGS.max_epoch_time:
                logger.info("Max epoch time of %d minutes reached after completing %d%% of epoch. "
                            "Moving on to next epoch" % (FLAGS.max_epoch_time, int(100 * (i / num_batches))))
                break
            dv = [torch.from_numpy(b).cuda() for b in batch]
            batch_size = dv[0].shape[0] * world
            if FLAGS.features_dir is not None:
                # net(x1, x2, x3, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)  (:595-617)
                obj_loss, center_loss, scale_loss, cls_loss = net(dv[0], dv[1], dv[2], dv[8], *dv[3:8])
            else:
                # net(x, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)  (:625)
                obj_loss, center_loss, scale_loss, cls_loss = net(dv[0], dv[6], *dv[1:6])
            net.backward()                                  # autograd.backward(sum_losses) (:631)
            net.allreduce_grads()                           # kvstore reduce inside trainer.step (:634)
            num_update += 1
            cur_lr = lr_scheduler(num_update)
            net.sgd_step(cur_lr, FLAGS.momentum, FLAGS.wd, batch_size)
            # :637-640 every step (running means since the start of training, never reset), summed on the device
            obj_metrics.update(0, [obj_loss]); center_metrics.update(0, [center_loss])
            scale_metrics.update(0, [scale_loss]); cls_metrics.update(0, [cls_loss])
            if FLAGS.log_interval and not (i + 1) % FLAGS.log_interval:
                (n1, l1), (n2, l2) = obj_metrics.get(), center_metrics.get()
                (n3, l3), (n4, l4) = scale_metrics.get(), cls_metrics.get()
                logger.info("[Epoch {}][Batch {}/{}], LR: {:.2E}, Speed: {:.3f} samples/sec, {}={:.3f}, {}={:.3f}, "
                            "{}={:.3f}, {}={:.3f}".format(epoch, i, num_batches, cur_lr,
                                                          batch_size / (time.time() - btic), n1, l1, n2, l2, n3, l3, n4, l4))
                # the log line has synchronised: look at the operand ranges of the fp16-split arithmetic now (one small
                # launch; a tensor whose channel scales have spread too far gets the range-exact arithmetic from here on)
                flagged = net.check_operand_ranges()
                if flagged:
                    logger.info("range guard: %d tensor(s) moved to the range-exact arithmetic (%s)" % (
                        len(flagged), ", ".join("%s %.1e" % kv for kv in sorted(flagged.items())[:4])))
            btic = time.time()
        torch.cuda.synchronize()
        (n1, l1), (n2, l2) = obj_metrics.get(), center_metrics.get()
        (n3, l3), (n4, l4) = scale_metrics.get(), cls_metrics.get()
        logger.info("[Epoch {}] Training cost: {:.3f}, {}={:.3f}, {}={:.3f}, {}={:.3f}, {}={:.3f}".format(
            epoch, (time.time() - tic), n1, l1, n2, l2, n3, l3, n4, l4))
        if not (epoch + 1) % FLAGS.val_interval:
            nsamp = (i + 1) * batch_size
            logger.info("End Epoch {}: # samples: {}, seconds: {}, samples/sec: {:.2f}".format(
                epoch, nsamp, time.time() - st, nsamp / (time.time() - st)))
            st = time.time()
            map_name, mean_ap = validate(net, val_data, eval_metric, FLAGS.data_shape)
            nval = validate.last_count
            logger.info("End Val: # samples: {}, seconds: {}, samples/sec: {:.2f}".format(
                nval, time.time() - st, nval / (time.time() - st)))
            val_msg = "\n".join(["{}={}".format(k, v) for k, v in zip(map_name, mean_ap)])
            logger.info("[Epoch {}] Validation: \n{}".format(epoch, val_msg))
            current_map = float(mean_ap[-1]) if not np.isnan(mean_ap[-1]) else 0.0
        else:
            current_map = 0.0
        if rank == 0:
            save_params(net, best_map, current_map, epoch, FLAGS.save_interval, save_prefix)


def main(argv=None):
    global FLAGS
    FLAGS = parse_flags(argv)
    FLAGS.window = [int(s) for s in FLAGS.window]
    if FLAGS.num_workers < 0:                           # :694-695 (here capped: the transforms are NumPy, not OpenCV threads)
        import multiprocessing
        FLAGS.num_workers = min(multiprocessing.cpu_count(), 8)
    if FLAGS.window[0] > 1:
        assert "vid" in FLAGS.dataset, "If using window size >1 you can only use the vid dataset"
    else:
        FLAGS.k_join_type = None
        FLAGS.k_join_pos = None
    np.random.seed(FLAGS.seed)                          # gutils.random.seed (:697): numpy, python's random, the framework
    import random as _pyrandom
    _pyrandom.seed(FLAGS.seed)
    torch.manual_seed(FLAGS.seed)
    rank, world = vdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("train_yolov3.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    train_dataset, val_dataset, eval_metric = get_dataset(FLAGS.dataset, FLAGS.dataset_val, FLAGS.save_prefix)
    save_dir = os.path.join("models", "experiments", FLAGS.save_prefix)            # :713-723
    # One process per GPU instead of the reference's single process: rank 0 alone looks at the directory BEFORE any rank
    # creates it, every rank takes rank 0's decision (all stop together, none runs on into a collective), and the
    # directory is made only after that exchange - a fresh run can no longer be refused because a peer got there first.
    refuse = vdist.broadcast_object(bool(rank == 0 and os.path.exists(save_dir) and FLAGS.save_prefix != "0000"
                                         and not FLAGS.resume.strip()), src=0)
    if refuse:
        raise SystemExit("{} exists so won't overwrite and restart training. You can resume training by using "
                         "--resume path_to_params_file".format(save_dir))
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    vdist.barrier()
    save_prefix = os.path.join(save_dir, "yolo3_" + FLAGS.network + "_" + "_".join(FLAGS.dataset))
    # train_yolov3.py:708-729: --trained_on builds (and --resume loads) the network with THAT dataset's classes, then the
    # prediction convs are reset to the training set's (rows of classes present in both are kept)
    trained_classes = train_dataset.classes
    if FLAGS.trained_on.strip():
        trained_classes = SyntheticDetection(FLAGS.trained_on.strip(), num_samples=1).classes
    net, start_epoch = get_net(trained_classes, (rank, world))
    if FLAGS.trained_on.strip():
        net.reset_class(train_dataset.classes)
    train_data, val_data = get_dataloader(train_dataset, val_dataset, FLAGS.data_shape, FLAGS.batch_size, rank, world)
    try:
        train(net, train_data, train_dataset, val_data, eval_metric, save_prefix, start_epoch, FLAGS.num_samples, rank, world)
    finally:
        for ld in (train_data, val_data):                # the loaders' worker pools (not left to __del__)
            if ld is not None and hasattr(ld, "close"):
                ld.close()
    return net


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main()
