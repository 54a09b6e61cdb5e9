"""Frame-batch data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  Replaces the reference's single-process multi-device loop:

  gluon.utils.split_and_load(batch, ctx_list)           train_yolov3.py:603-606  -> shard_range()
  kvstore='local' gradient aggregation in trainer.step  train_yolov3.py:527-530  -> allreduce_sum_() on the flat arena
  gluon.contrib.nn.SyncBatchNorm(num_devices=...)       train_yolov3.py:347-354  -> allreduce_sum_() on the fp64 [2C] sums
  trainer.step(batch_size): rescale_grad = 1/batch_size train_yolov3.py:634      -> rescale by the GLOBAL batch

Only the two real exchange steps of the path use a collective; inference shards frames with none.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return rank(), world_size()
    if backend is None:
        # VD_DIST_BACKEND=gloo: rehearse N ranks on ONE GPU (RCCL refuses two ranks on a device); not a measurement mode
        backend = os.environ.get("VD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank(), world_size()


def active(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def shard_range(n, r, w):
    """Even split of n frames over w ranks along axis 0 (split_and_load with even_split=True requires
    divisibility, train_yolov3.py:603; validate() uses even_split=False: the first n % w ranks get one more)."""
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def allreduce_sum_(t, group=None):
    """In-place sum over ranks; a no-op for a single rank."""
    if active(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def bucketed_allreduce_sum_(flat, bucket_elems, group=None, async_op=False):
    """Sum-all-reduce a flat arena in contiguous buckets (xGMI is point-to-point: a few large messages
    keep every link busy; the default bucket is 64 MiB).  Returns the work handles when async_op."""
    handles = []
    if not active(group):
        return handles
    n = flat.numel()
    for lo in range(0, n, bucket_elems):
        h = dist.all_reduce(flat[lo:min(n, lo + bucket_elems)], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            handles.append(h)
    return handles


def all_gather_objects(obj, group=None):
    """[obj of rank 0, obj of rank 1, ...] on every rank (host objects: detections, metric records); [obj] for one rank."""
    if not active(group):
        return [obj]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def broadcast_object(obj, src=0, group=None):
    """Rank `src`'s host object on every rank (a decision every rank must take identically)."""
    if not active(group):
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def barrier(group=None):
    if active(group):
        dist.barrier(group=group)
