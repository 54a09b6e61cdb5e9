#!/usr/bin/env python
"""bench.py — headline benchmark of the yolo3_darknet53 hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--mode train|detect] [--batch B] [--size S] [--classes C]

mode train  (default; BASELINE.json configs[2]): yolo3_darknet53_coco, per-GPU batch 64, 416x416, fp32,
            one step = forward + 4 losses + backward + gradient all-reduce (N>1) + SGD-momentum update.
mode detect (BASELINE.json configs[1] shape): forward + decode + NMS at 608x608.
Inputs are synthetic, resident in HBM before the timed region; weights random-init (He).
For N>1 the driver launches one rank per GPU with torch.distributed.run; ranks shard frames (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0   # split-operand fp32 products: 6 bf16 MFMAs per fp32 product block
PEAK_F16X2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0   # two-way fp16 split: 3 f16 MFMAs per product block (f16 = bf16 rate)
PEAKS = {"native": PEAK_FP32_MFMA_TFLOPS, "split": PEAK_SPLIT_TFLOPS, "f16x2": PEAK_F16X2_TFLOPS}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="train", choices=["train", "detect"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 64 train / 32 detect)")
    ap.add_argument("--size", type=int, default=0, help="input side (default 416 train / 608 detect)")
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-detect", action="store_true", help="skip the detect (608x608, bf16) sub-line of the default training bench")
    ap.add_argument("--no-native", action="store_true", help="skip the VD_FP32_MATH=native sub-line of the training bench")
    ap.add_argument("--syncbn", default=None, choices=[None, "all", "reference"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="bf16 = bf16 storage/MFMA inference (detect mode only)")
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16"],
                    help="training only: bf16 = bf16 activations and gradients (net.set_storage('bf16'); BASELINE configs[4])")
    ap.add_argument("--graphs", action="store_true", help="replay the inference program as a captured HIP graph")
    ap.add_argument("--window", type=int, default=1, help="frames per sample (k); 3 with the join flags = BASELINE configs[3]")
    ap.add_argument("--k-join-type", default="max", choices=["max", "mean", "cat"])
    ap.add_argument("--k-join-pos", default="late", choices=["early", "late"])
    ap.add_argument("--block-conv-type", default="2", choices=["2", "3", "21"])
    ap.add_argument("--detail", default=None, help="write a per-launch conv table (JSON) to this path")
    ap.add_argument("--obj-bias", type=float, default=None,
                    help="objectness bias of the random-init heads (default: calibrated so ~2%% of score rows pass)")
    return ap.parse_args()


def cpu_baseline(mode, size, classes):
    """The network on torch-CPU operators (oneDNN, all host threads; oracle/torch_cpu.py) timed on whole steps of a small
    batch of the same workload for ~20 s: a port - NOT MXNet, which cannot be installed or shipped; MXNet's CPU
    backend is the same library family (MKL-DNN)."""
    from oracle import torch_cpu as TC
    batch = 8 if mode == "train" else 8
    fps, n, threads = TC.baseline(mode, size, classes, batch, budget_s=14.0)
    what = "fwd + 4 losses + bwd + SGD-momentum" if mode == "train" else "fwd + decode + NMS"
    out = {"value": round(fps, 3), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": "%d step(s) of %d frames %dx%d, %s, fp32, torch-CPU operators (oneDNN) + the NumPy oracle's target merge / "
                     "decode; not MXNet" % (n, batch, size, size, what)}
    # the two configurations BASELINE.md section 3 names, a few seconds each (same port, same threads)
    f0, n0, _ = TC.baseline("train", 416, 20, 4, budget_s=6.0)
    f1, n1, _ = TC.baseline("detect", 608, 80, 1, budget_s=4.0)
    out["baseline_md_configs"] = {
        "configs0_voc_train_b4_416_fwd_bwd_sgd": {"value": round(f0, 3), "unit": "frames/s", "steps": n0},
        "configs1_shape_detect_b1_608_fwd_decode_nms": {"value": round(f1, 3), "unit": "frames/s", "steps": n1}}
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    # developer rehearsal of the world>1 logic on a ONE-GPU box: every rank on cuda:0, gloo instead of RCCL (RCCL
    # refuses two ranks on one device).  Not a measurement mode.
    rehearse = os.environ.get("VD_REHEARSE_SHARED_GPU", "0") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    force_dist = os.environ.get("VD_FORCE_DIST", "0") == "1"      # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        if rehearse:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    from viddet_amd.model import yolo3_darknet53
    from viddet_amd.targets import synthetic_batch, prefetch_targets

    train = a.mode == "train"
    B = a.batch or (64 if train else 32)
    S = a.size or (416 if train else 608)
    C = a.classes
    classes = ["c%d" % i for i in range(C)]
    K = max(1, a.window)
    net = yolo3_darknet53(classes, norm_layer="syncbn" if a.syncbn else None,
                          norm_kwargs={"scope": a.syncbn} if a.syncbn else None,
                          **(dict(k=K, k_join_type=a.k_join_type, k_join_pos=a.k_join_pos,
                                  block_conv_type=a.block_conv_type) if K > 1 else {}))
    # He init keeps synthetic activations O(1); objectness bias negative so only a few % of the C*P score rows
    # pass valid_thresh, as with a trained net.  Same seed on every rank => identical replicas.
    net.initialize(init="he", seed=233, obj_bias=-4.0)
    if train and a.storage == "bf16":
        a.dtype = "bf16"
        net.set_storage("bf16")
    elif a.dtype == "bf16":
        if train:
            # mixed-precision training arithmetic (BASELINE configs[4] family): fp32 tensors, accumulation and optimiser;
            # convolution products on bf16-rounded operands (VD_MATH_BF16).  NOT the configs[2] headline, which is fp32.
            from viddet_amd.model import set_conv_math
            set_conv_math("bf16")
        else:
            net.set_precision("bf16")
    x_np, gt_np, ids_np = synthetic_batch(B * K, S, C, 233 + rank)
    if K > 1:                      # B windows of K frames; the labels are the centre frame's (one gt set per window)
        x_np = x_np.reshape(B, K, 3, S, S)
        gt_np, ids_np = gt_np.reshape(B, K, -1, 4)[:, K // 2], ids_np.reshape(B, K, -1, 1)[:, K // 2]
    x = torch.from_numpy(x_np).cuda()
    if train:
        tg = prefetch_targets(S, S, gt_np, ids_np, C)
        gt = torch.from_numpy(gt_np).cuda()
        tgd = [torch.from_numpy(t).cuda() for t in tg]
        lr, mom, wd = 1e-3, 0.9, 5e-4      # train_yolov3.py:78,91-94 defaults

        def step():
            net(x, gt, *tgd)
            net.backward()
            net.allreduce_grads()
            net.sgd_step(lr, mom, wd, batch_size=B * world)
    else:
        # a trained net rejects most score rows; choose the objectness bias so that ~2 % of the C*P rows
        # pass valid_thresh=0.01, and report the bias and the measured pass fraction
        P = 3 * ((S // 32) ** 2 + (S // 16) ** 2 + (S // 8) ** 2)
        best = None
        for bias in ([a.obj_bias] if a.obj_bias is not None else [-3.0 - 0.25 * i for i in range(13)]):
            for i in range(3):
                p = net.collect_params()["yolo_outputs.%d.prediction.bias" % i]
                v = p.data().cpu()
                v.view(3, -1)[:, 4] = bias
                p.set_data(v)
            net(x)
            torch.cuda.synchronize()
            key = ("infer_bf16" if a.dtype == "bf16" else "infer", B, S, S)
            frac = float(net._programs[key][2]["counts"].float().mean()) / (C * P)
            if best is None or abs(frac - 0.02) < abs(best[1] - 0.02):
                best = (bias, frac)
        for i in range(3):
            p = net.collect_params()["yolo_outputs.%d.prediction.bias" % i]
            v = p.data().cpu()
            v.view(3, -1)[:, 4] = best[0]
            p.set_data(v)
        pass_info = {"obj_bias": best[0], "pass_fraction": round(best[1], 5)}
        net.use_graphs = bool(a.graphs)

        def step():
            net(x)

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    if os.environ.get("VD_MAIN_PRIO"):          # developer A/B: the whole step on a stream of this priority (-1 = high)
        _ms = torch.cuda.Stream(priority=int(os.environ["VD_MAIN_PRIO"]))
        _ms.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(_ms)
    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1 or force_dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    fps = world * B * K * a.steps / dt            # frames (a window of k frames counts k; windows/s is in config)

    # ---- phase split (one extra step with device syncs between phases; not part of the timed region)
    phases = None
    if train:      # every rank runs it (allreduce_grads is a collective); rank 0 reports
        def timed(fn):
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t) * 1e3, 3)
        net._dp_stats = {"buckets": 0, "bucket_bytes": 0}
        phases = {"fwd_loss_ms": timed(lambda: net(x, gt, *tgd)), "bwd_ms": timed(net.backward),
                  "allreduce_ms": timed(net.allreduce_grads),
                  "sgd_ms": timed(lambda: net.sgd_step(lr, mom, wd, batch_size=B * world))}
        if a.syncbn and (world > 1 or force_dist):
            # SyncBN: what the per-layer statistics exchanges cost the step - the same two phases with the exchanges
            # suppressed (every rank alike; the statistics are then local, which only this diagnostic step sees)
            net._syncbn_suppress = True
            f0, b0 = timed(lambda: net(x, gt, *tgd)), timed(net.backward)
            net._syncbn_suppress = False
            net._dp_stats["syncbn_exchanges"] = 0
            f1, b1 = timed(lambda: net(x, gt, *tgd)), timed(net.backward)
            phases["syncbn"] = {"scope": a.syncbn, "exchanges_per_step": net._dp_stats.get("syncbn_exchanges", 0),
                                "fwd_bwd_ms": round(f1 + b1, 3), "fwd_bwd_without_exchanges_ms": round(f0 + b0, 3),
                                "syncbn_exposed_ms": round(max(0.0, f1 + b1 - f0 - b0), 3)}
        if world > 1 or force_dist:
            # Diagnostics of the gradient exchange (the driver computes scaling efficiency itself from `value`): how much of
            # the all-reduce the backward pass does NOT hide, and what the fabric delivers on the whole arena.
            #   bwd_ms above = backward with the bucketed all-reduces queued behind the weight-gradient stream (device-
            #   synchronised, so it includes their completion); bwd_local_ms = the same backward with the collectives
            #   suppressed; exposed = (bwd + tail all-reduce) - local.  allreduce_arena_ms = ONE all-reduce of the whole
            #   gradient arena on an idle GPU; bus GB/s = 2 (N-1)/N x bytes / time (ring convention; 7 xGMI links x ~153 GB/s
            #   per GPU is the per-link bound to compare with).
            dp = dict(net._dp_stats)
            net(x, gt, *tgd)
            net._dp_suppress = True
            phases["bwd_local_ms"] = timed(net.backward)
            net._dp_suppress = False
            phases["allreduce_exposed_ms"] = round(max(0.0, phases["bwd_ms"] + phases["allreduce_ms"] - phases["bwd_local_ms"]), 3)
            nbytes = 4 * net.grads.numel()
            torch.distributed.all_reduce(net.grads)          # warm the communicator on this size
            t_ar = min(timed(lambda: torch.distributed.all_reduce(net.grads)) for _ in range(3))
            phases["allreduce_arena_ms"] = t_ar
            phases["allreduce"] = {"arena_mb": round(nbytes / 1e6, 1), "buckets_per_step": dp["buckets"],
                                   "bucket_mb": round(dp["bucket_bytes"] / max(1, dp["buckets"]) / 1e6, 1),
                                   "bus_gb_s": round(2.0 * (world - 1) / max(1, world) * nbytes / (t_ar * 1e-3) / 1e9, 1),
                                   "algo_gb_s": round(nbytes / (t_ar * 1e-3) / 1e9, 1)}
            net.grads.zero_()                                  # the diagnostic all-reduces scaled the arena; the step is over

    # ---- roofline of the dominant kernel (k_conv_igemm: forward + data-gradient convs), measured live with
    # events on the launch stream over one more step
    roof = None
    extra = {}
    if True:       # every rank replays (keeps ranks in lock-step if SyncBN collectives are in the programs)
        recs = []
        if train:
            net(x, gt, *tgd)
            tp = net._last_train
            net._refresh_dgrad(tp)
            for seg in tp["fwd"] + tp["bwd"]:
                if hasattr(seg, "run_timed"):
                    got = seg.run_timed({"vd_conv_igemm", "vd_conv_igemm_bf16", "vd_conv_wgrad", "vd_bn_apply_leaky", "vd_bn_apply_leaky_bf16"})
                    recs += [({"vd_conv_igemm_bf16": "vd_conv_igemm", "vd_bn_apply_leaky_bf16": "vd_bn_apply_leaky"}.get(f_, f_), m, e0, e1)
                             for (f_, m, e0, e1) in got]
                else:
                    seg()
        else:
            prog = net._programs[("infer_bf16" if a.dtype == "bf16" else "infer", B, S, S)][0]
            recs = prog.run_timed({"vd_conv_igemm", "vd_conv_igemm_bf16", "vd_stem_conv_c32_bf16", "vd_yolo_decode_filter", "vd_nms_topk"})
            torch.cuda.synchronize()
            # the HBM-bound tail of the detect path (north_star: achieved GB/s on the decode / NMS kernels): algorithmic
            # bytes = the three fp32 head maps read once (B * sum(g^2) * ldh * 4; SURVEY 8d: 7.73 MB/frame at 608 / C = 80)
            # + B * 100 * 6 * 4 written, at the calibrated pass fraction above
            ldh = 32 * ((3 * (5 + C) + 31) // 32)
            head_bytes = 4.0 * B * K * sum((S // st) ** 2 for st in (32, 16, 8)) * ldh
            for f_, _, e0, e1 in recs:
                if f_ in ("vd_yolo_decode_filter", "vd_nms_topk"):
                    ms = e0.elapsed_time(e1)
                    key = "decode_filter" if f_ == "vd_yolo_decode_filter" else "nms"
                    extra[key] = {"ms": round(ms, 4)}
                    if key == "decode_filter":
                        extra[key].update(algorithmic_mb=round(head_bytes / 1e6, 2), gb_s=round(head_bytes / ms / 1e6, 1),
                                          frac_of_hbm_peak=round(head_bytes / ms / 1e6 / 8000.0, 4))
                    else:
                        cand = float(net._programs[("infer_bf16" if a.dtype == "bf16" else "infer", B, S, S)][2]["counts"].float().sum())
                        extra[key].update(candidates=int(cand), note="one workgroup (one CU) per image: radix select of the top 400 "
                                          "(12 + 12 + 8 bit digits), bitonic sort, 400 x 400 IoU bitmask by ballots, chunked sweep; "
                                          "reads 8 B per candidate (%.2f MB) - bound by one CU's issue rate, not an HBM kernel; "
                                          "phases: tools/nms_probe.sh" % (8.0 * cand / 1e6))
            if "decode_filter" in extra and "nms" in extra:
                t = extra["decode_filter"]["ms"] + extra["nms"]["ms"]
                extra["decode_plus_nms"] = {"ms": round(t, 4), "gb_s": round((head_bytes + B * K * 2400.0) / t / 1e6, 1),
                                            "algorithmic_mb_per_frame": round((head_bytes / (B * K) + 2400.0) / 1e6, 3)}
            nsplit = sum(1 for (f_, m, e0, e1) in recs if f_ == "vd_conv_igemm_bf16" and m and m.get("splitk"))
            if nsplit:
                extra["splitk_launches"] = nsplit       # VD_CONV_SPLITK: launches with too few tiles for the chip, cut along K
            # (the fused stem + stride-2 conv launch counts as a conv launch: its meta carries both layers' FLOPs)
            recs = [("vd_conv_igemm", m, e0, e1) for (f_, m, e0, e1) in recs if f_ in ("vd_conv_igemm", "vd_conv_igemm_bf16", "vd_stem_conv_c32_bf16")]
        torch.cuda.synchronize()
        # the stand-alone forward BatchNorm+LeakyReLU passes, and the part of them that belongs to cells whose output feeds
        # ONE convolution only (what fusing the pass into the consumer's gather could remove)
        bn = [(m, e0.elapsed_time(e1)) for f_, m, e0, e1 in recs if f_ == "vd_bn_apply_leaky" and m]
        recs = [r for r in recs if r[0] != "vd_bn_apply_leaky"]
        if bn:
            extra["bn_apply_leaky"] = {
                "launches": len(bn), "ms": round(sum(t for _, t in bn), 3),
                "gb_s": round(sum(m["bytes"] for m, _ in bn) / sum(t for _, t in bn) / 1e6, 0),
                "single_conv_consumer_launches": sum(1 for m, _ in bn if m["single_conv_consumer"]),
                "single_conv_consumer_ms": round(sum(t for m, t in bn if m["single_conv_consumer"]), 3)}
        agg = {}
        detail = []
        for fname, meta, e0, e1 in recs:
            ms = e0.elapsed_time(e1)
            detail.append(dict(fn=fname, ms=round(ms, 4), tflops=round(meta["flops"] / (ms * 1e-3) / 1e12, 2), **meta))
            key = fname
            v = agg.setdefault(key, [0.0, 0.0, 0])
            v[0] += meta["flops"]
            v[1] += ms
            v[2] += 1
            if fname == "vd_conv_igemm" and meta["k"] == 3 and meta["stride"] == 1 and meta["kind"] == "fwd":
                w = agg.setdefault("igemm_3x3s1_fwd", [0.0, 0.0, 0])
                w[0] += meta["flops"]; w[1] += ms; w[2] += 1
        ig = agg["vd_conv_igemm"]
        ach = ig[0] / (ig[1] * 1e-3) / 1e12
        if a.dtype == "bf16":
            peak = PEAK_BF16_MFMA_TFLOPS
            kname = ("k_conv_igemm_bf16 (bf16 tensors, v_mfma_f32_32x32x16_bf16, fp32 accumulate; forward + data gradients)"
                     if (train and a.storage == "bf16") else
                     "k_conv_igemm (fp32 tensors, one bf16 MFMA term per product block)" if train else
                     "k_conv_igemm_bf16 (v_mfma_f32_32x32x16_bf16)")
            math = None
        else:
            # fp32 convolutions run in one of three product arithmetics, chosen per launch record by the plan-time
            # autotuner: the fp32 MFMA (peak 157.3 TFLOP/s), the 3-way bf16 operand split (6 bf16 MFMAs per product
            # block: peak = dense bf16 2500 / 6) or the 2-way fp16 split (3 f16 MFMAs: 2500 / 3).  The roofline of the mix
            # is the time an ideal machine needs, sum_i flops_i / peak_i; `peak` is the equivalent blended rate.
            ideal_ms, math = 0.0, {"f16x2": [0.0, 0.0, 0], "split": [0.0, 0.0, 0], "native": [0.0, 0.0, 0]}
            for fname, meta, e0, e1 in recs:
                if fname != "vd_conv_igemm":
                    continue
                cls = "f16x2" if meta.get("f16x2") else ("split" if meta.get("split") else "native")
                ideal_ms += meta["flops"] / (PEAKS[cls] * 1e12) * 1e3
                m_ = math[cls]
                m_[0] += meta["flops"]; m_[1] += e0.elapsed_time(e1); m_[2] += 1
            peak = ig[0] / (ideal_ms * 1e-3) / 1e12
            kname = ("k_conv_igemm: fp32 in/out/accumulate; products as a 2-way fp16 split x 3 v_mfma_f32_*_f16 (%d launches), "
                     "an exact 3-way bf16 split x 6 v_mfma_f32_*_bf16 (%d launches) or v_mfma_f32_32x32x2_f32 (%d launches)"
                     % (math["f16x2"][2], math["split"][2], math["native"][2]))
            math = {k_: {"launches": v[2], "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2) if v[1] > 0 else None,
                         "ms": round(v[1], 3), "peak": round(PEAKS[k_], 1)}
                    for k_, v in math.items()}
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2),
                "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": None, "launches": ig[2], "avg_launch_ms": round(ig[1] / ig[2], 4),
                "algorithmic_gflop_per_launch": round(ig[0] / ig[2] / 1e9, 3)}
        # launches the plan runs as persistent stream-K grids (VD_CONV_STREAMK: bit-identical results, another cut of the
        # launch into workgroups; the autotuner takes the form per launch record where it is faster)
        sk_recs = [(m, e0.elapsed_time(e1)) for f_, m, e0, e1 in recs if f_ == "vd_conv_igemm" and m.get("streamk")]
        # hand-off polls that gave up (the consumer then recomputes its tile's K-prefix: correct, wasted work) over the whole
        # run so far, all workspaces: warm-up + timed steps + the diagnostic replays - beside a side stream's long-lived
        # weight-gradient workgroups a persistent grid is not always resident at once
        gave_up = sum(int(w.view(torch.int32)[2047]) for w in getattr(net, "_sk_ws", {}).values())
        roof["streamk"] = {"launches": len(sk_recs), "ms": round(sum(t for _, t in sk_recs), 3), "polls_given_up_total": gave_up,
                           "tflops": round(sum(m["flops"] for m, _ in sk_recs) / max(1e-9, sum(t for _, t in sk_recs)) / 1e9, 2) if sk_recs else None,
                           "switch": "VD_STREAMK=%s" % os.environ.get("VD_STREAMK", "1")}
        if math is not None:
            roof["by_math"] = math
            # not part of `frac`: what a bare f16-MFMA K loop of the kernel's shape sustains on this chip with random
            # operands (the power limit lowers the clock as the multipliers toggle) - measured once, not live
            if math["f16x2"]["tflops"]:
                roof["sustained_reference"] = {
                    "f16_mfma_tflops_random_operands": 1470.0, "f16_mfma_tflops_zero_operands": 2250.0,
                    "fp32_equivalent_tflops": round(1470.0 / 3, 1),
                    "f16x2_launches_frac_of_it": round(math["f16x2"]["tflops"] / (1470.0 / 3), 4),
                    "source": "profiles/r02_mfma_ceiling.txt (tools/mfma_ceiling.hip): a round-2 measurement on another box, "
                              "copied here as a yardstick - not measured in this run"}
            roof["frac_of_fp32_mfma_peak"] = round(ach / PEAK_FP32_MFMA_TFLOPS, 4)
        roof["algorithmic_mb_per_launch"] = round(sum(m["bytes"] for f_, m, _, _ in recs if f_ == "vd_conv_igemm") / ig[2] / 1e6, 1)
        # HBM traffic of the same kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of this very
        # command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), committed under profiles/
        try:
            if train and B == 64 and S == 416 and C == 80:
                import glob
                pmf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_train_b64_416_pmc_traffic.json")))[-1]
                pm = json.load(open(pmf))
                roof["traffic"] = round(pm["k_conv_igemm"]["hbm_mb_corrected"], 1)
                roof["traffic_unit"] = ("MB per launch leaving the L2s (2 x FETCH_SIZE + WRITE_SIZE over the final step's launches; "
                                        "MALL hits count as fetches, so this bounds HBM bytes from above)")
                # NOT measured in this run: PMC passes need rocprofv3 around the process (tools/pmc_traffic.py); the number
                # is read from the newest committed pass of this very command
                roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc passes of `python bench.py`, committed; not measured live)" % os.path.basename(pmf)
        except Exception:
            pass
        for k, v in agg.items():
            if k != "vd_conv_igemm" and v[1] > 0:
                extra[k] = {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2), "ms": round(v[1], 3), "launches": v[2]}
        if "vd_conv_wgrad" in extra:
            # the weight gradients by kernel and geometry: k_conv_wgrad_halo (3x3 stride 1, Co >= 128) against the generic
            # kernel's 3x3 stride-1 leftovers, the 1x1 layers (HBM-bound: two activation tensors per 2 Ci Co FLOPs a pixel)
            # and the stride-2 layers
            cls = {}
            for r in detail:
                if r["fn"] != "vd_conv_wgrad":
                    continue
                key = "halo_3x3s1" if r.get("wgrad_halo") else ("generic_3x3s1" if (r["k"] == 3 and r["stride"] == 1) else
                                                                 "generic_1x1" if r["k"] == 1 else "generic_3x3s2")
                c = cls.setdefault(key, [0.0, 0.0, 0, 0.0])
                c[0] += r["flops"]; c[1] += r["ms"]; c[2] += 1; c[3] += r.get("bytes", 0.0)
            extra["vd_conv_wgrad"]["by_kernel"] = {
                k_: {"launches": c[2], "ms": round(c[1], 3), "tflops": round(c[0] / (c[1] * 1e-3) / 1e12, 1),
                     "algorithmic_gb_s": round(c[3] / (c[1] * 1e-3) / 1e9, 0)} for k_, c in cls.items()}
            # roofline of the halo-ring kernel, priced like k_conv_igemm: the dense MFMA peak of its arithmetic (three f16 MFMAs
            # per fp32 product block, or one bf16 MFMA on bf16 tensors); traffic = bytes leaving the L2s per launch from the
            # committed PMC passes of this very command
            h = extra["vd_conv_wgrad"]["by_kernel"].get("halo_3x3s1")
            if h:
                pk = PEAK_BF16_MFMA_TFLOPS if (a.dtype == "bf16" or a.storage == "bf16") else PEAKS["f16x2"]
                h["roofline"] = {"bound": "mfma", "achieved": h["tflops"], "peak": round(pk, 1), "unit": "TFLOP/s",
                                 "frac": round(h["tflops"] / pk, 4), "traffic": None}
                try:
                    if train and B == 64 and S == 416 and C == 80 and a.storage != "bf16":
                        import glob
                        pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_train_b64_416_pmc_traffic.json")))[-1]))
                        h["roofline"]["traffic"] = round(pm["k_conv_wgrad_halo"]["hbm_mb_corrected"], 1)
                except Exception:
                    pass
        extra["conv_ms_per_step"] = round(sum(v[1] for k, v in agg.items() if k.startswith("vd_")), 3)
        if a.detail:
            with open(a.detail, "w") as f:
                json.dump(detail, f, indent=0)

    # ---- the strict reading of "fp32": the same step with every product on the fp32 MFMA (v_mfma_f32_32x32x2_f32, an exact
    # fma chain; VD_FP32_MATH=native), a few steps, beside the headline - which runs the fp32-grade split arithmetics
    if roof is not None and train and a.dtype == "f32" and world == 1 and not force_dist and K == 1 and not a.no_native \
            and os.environ.get("VD_FP32_MATH", "auto") == "auto":
        from viddet_amd.model import set_conv_math
        del net
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        set_conv_math("native")
        try:
            net = yolo3_darknet53(classes)
            net.initialize(init="he", seed=233, obj_bias=-4.0)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            dtn = (time.perf_counter() - t0) / 4
            roof["native_fp32"] = {"frames_per_s": round(B / dtn, 2), "ms_per_step": round(dtn * 1e3, 3), "steps": 4,
                                   "model_tflops": round(B / dtn * 197.3 / 1e3, 2) if (S == 416 and C == 80) else None,
                                   "note": "VD_FP32_MATH=native: every conv product on v_mfma_f32_32x32x2_f32 (peak 157.3 TFLOP/s)"}
        finally:
            set_conv_math(None)

    # ---- the metric's second half ("...; detect fps", BASELINE configs[1]: 608x608, bf16, detect_yolo3.py path) beside the
    # training number of the default run: the same script as a child process (its own plans and calibration), its line
    # embedded here - frames/s and the decode / NMS kernels' ms and GB/s at the calibrated pass fraction
    detect = None
    if train and rank == 0 and world == 1 and not force_dist and not a.no_detect and K == 1 and a.dtype == "f32" \
            and a.batch == 0 and a.size == 0:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", "detect", "--dtype", "bf16", "--steps", "20",
                                "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
            dj = json.loads(r.stdout.strip().splitlines()[-1])
            detect = {"metric": dj["metric"], "value": dj["value"], "unit": dj["unit"], "ms_per_step": dj["ms_per_step"],
                      "dtype": dj["dtype"], "config": dj["config"]["workload"], "score_filter": dj["config"]["score_filter"],
                      "roofline": {k: dj["roofline"][k] for k in ("kernel", "achieved", "peak", "frac", "launches") if k in dj["roofline"]},
                      "kernels": {k: dj["kernels"][k] for k in ("decode_filter", "nms", "decode_plus_nms", "igemm_3x3s1_fwd")
                                  if k in dj["kernels"]}}
        except Exception as e:  # noqa: BLE001  (a diagnostic sub-line must never take the headline down)
            detect = {"error": repr(e)[:200]}

    # ---- north_star's second training size ("synthetic 416x416 and 608x608"): fp32, 80 classes, 608x608, batch 32 - the same
    # script as a child process (its own plans), frames/s and the dominant kernel's rate embedded here
    train_608 = None
    if train and rank == 0 and world == 1 and not force_dist and not a.no_detect and K == 1 and a.dtype == "f32" \
            and a.batch == 0 and a.size == 0:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--size", "608", "--batch", "32", "--steps", "10", "--warmup", "3",
                                "--no-cpu-baseline", "--no-native", "--no-detect"], capture_output=True, text=True, timeout=900)
            tj = json.loads(r.stdout.strip().splitlines()[-1])
            train_608 = {"metric": tj["metric"], "value": tj["value"], "unit": tj["unit"], "ms_per_step": tj["ms_per_step"],
                         "dtype": tj["dtype"], "config": tj["config"]["workload"],
                         "model_tflops": round(tj["value"] * 421.4 / 1e3, 2),       # 421.4 GFLOP per frame at 608 / C = 80
                         "roofline": {k: tj["roofline"][k] for k in ("achieved", "peak", "frac", "launches", "avg_launch_ms", "streamk")
                                      if k in tj["roofline"]},
                         "kernels": {k: tj["kernels"][k] for k in ("igemm_3x3s1_fwd", "vd_conv_wgrad", "bn_apply_leaky") if k in tj["kernels"]}}
            train_608["kernels"].get("vd_conv_wgrad", {}).pop("by_kernel", None)
        except Exception as e:  # noqa: BLE001
            train_608 = {"error": repr(e)[:200]}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.mode, S, C)

    if world > 1 or force_dist:
        torch.distributed.barrier()
    if rank == 0:
        gflop = 197.3 if (train and S == 416 and C == 80 and K == 1) else None
        out = {
            "metric": ("frames/sec (%dx%d) yolo3_darknet53 fwd+bwd" % (S, S)) if train else
                      ("detect fps (%dx%d) yolo3_darknet53 fwd+decode+NMS" % (S, S)),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (("temporal-YOLO (k=%d frame windows) training, %d windows/GPU, %dx%d, fp32, fwd+bwd+"
                                     "SGD-momentum (BASELINE configs[3] family)" % (K, B, S, S)) if (train and K > 1) else
                                    ("yolo3_darknet53 training, %d classes, batch %d/GPU, %dx%d, %s "
                                     "(fp32 accumulate / BN statistics / loss / master weights / SGD), fwd+bwd+SGD-momentum (BASELINE configs[4]%s)"
                                     % (C, B, S, S, "bf16 STORAGE: activations and gradients bf16, bf16 MFMA" if a.storage == "bf16" else
                                        "bf16 conv products on fp32 tensors",
                                        "; its per-GPU shape" if (C == 285 and S == 608 and B == 32) else
                                        " arithmetic on the configs[2] shape" if (C == 80 and S == 416) else " arithmetic")) if a.dtype == "bf16" else
                                    "yolo3_darknet53_coco training, batch %d/GPU, %dx%d, fp32, fwd+bwd+SGD-momentum "
                                    "(BASELINE configs[2]; the reference trains with SGD, not Adam)" % (B, S, S)) if train
                       else ("yolo3_darknet53 inference (detect_yolo3.py path), batch %d/GPU, %dx%d, %s" % (B, S, S, a.dtype)),
                       "classes": C, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "window": (None if K == 1 else {"k": K, "k_join_type": a.k_join_type, "k_join_pos": a.k_join_pos,
                                                       "block_conv_type": a.block_conv_type,
                                                       "windows_per_s": round(fps / K, 2),
                                                       "note": "BASELINE configs[3] family: batch = windows per GPU"}),
                       "syncbn": a.syncbn, "score_filter": (None if train else pass_info),
                       "hip_graph": bool(a.graphs) and not train,
                       "fp32_math": (None if a.dtype == "bf16" else
                                     "VD_FP32_MATH=%s: tensors, accumulation and epilogues fp32; conv products on the fp32 "
                                     "MFMA, as an exact 3-way bf16 operand split (6 bf16 MFMAs) or as a 2-way fp16 split "
                                     "with per-tensor power-of-two scales (3 f16 MFMAs); measured error vs fp64 <= the "
                                     "fp32 MFMA's for both; per launch by the autotuner"
                                     % os.environ.get("VD_FP32_MATH", "auto"))},
            "roofline": roof, "cpu_baseline": cpu, "kernels": extra, "phases": phases,
        }
        if detect is not None:
            out["detect"] = detect
        if train_608 is not None:
            out["train_608"] = train_608
        if gflop:
            out["model_tflops"] = round(fps * gflop / 1e3, 2)
        print(json.dumps(out))
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
