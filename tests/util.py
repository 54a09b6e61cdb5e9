"""Test plumbing: layout conversion between the oracle's NCHW/OIHW numpy arrays and the NHWC
device tensors the C-ABI takes.  (torch is used to move bytes only.)"""
import numpy as np
import torch


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def nchw_to_dev_nhwc(x, pad_c=None):
    x = np.moveaxis(x, 1, -1)
    if pad_c is not None and pad_c > x.shape[-1]:
        z = np.zeros(x.shape[:-1] + (pad_c,), dtype=x.dtype)
        z[..., :x.shape[-1]] = x
        x = z
    return dev(x)


def dev_nhwc_to_nchw(t, c=None):
    a = t.detach().cpu().double().numpy()
    if c is not None:
        a = a[..., :c]
    return np.moveaxis(a, -1, 1)


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def device_leaky_masks(net, bufs):
    """LeakyReLU branch decisions the device took in its last TRAINING forward: sign of z*scale+shift per conv
    cell, evaluated in fp32 exactly as the kernels do, as {cell name: bool (N,C,H,W)}."""
    from viddet_amd.model import ConvNode
    out = {}
    for n in net.nodes:
        if isinstance(n, ConvNode) and n.bn:
            z = bufs['z:' + n.dst].cpu().numpy()
            # the kernels evaluate v*scale+shift as ONE fma (hipcc contracts): its sign is the sign of the exact
            # value, which float64 arithmetic on the fp32 operands reproduces
            u = z.astype(np.float64) * n.b_scale.cpu().numpy().astype(np.float64) + \
                n.b_shift.cpu().numpy().astype(np.float64)
            out[n.name] = np.moveaxis(u > 0, -1, 1)
    return out


def check_masks_differ_only_at_ties(onet_natural_pre, masks, band=5e-5):
    """The supplied decisions may differ from the oracle's own only where the oracle's |pre-activation| is tiny
    (below `band`: fp32 round-off of the device's conv stack around an exact zero crossing; the largest seen on the
    full-size 416x416 frame is 1.2e-5, deep in the neck, where 70 layers of fp32 round-off have accumulated)."""
    nflip, worst = 0, 0.0
    for name, u in onet_natural_pre.items():
        diff = (u > 0) != masks[name]
        if diff.any():
            worst = max(worst, float(np.abs(u[diff]).max()))
            assert np.abs(u[diff]).max() < band, (name, float(np.abs(u[diff]).max()))
            nflip += int(diff.sum())
    print("leaky branch ties: %d flips, largest |pre-activation| among them %.2e (band %.0e)" % (nflip, worst, band))
    return nflip


def assert_rows_match(rows_dev, rows_ref, scores_ref, tie=1e-6):
    """Post-NMS row indices of the device equal the oracle's.  Exact equality is the rule; the only tolerated
    difference is the ORDER of detections whose oracle scores are closer than `tie` (fp32 cannot rank scores that
    differ in the 7th digit: 0.19326594 vs 0.19326585 was the case that motivated this).  Each run of near-tied
    reference scores must hold the same set of rows on both sides.  Returns, per image, the device rank that holds
    reference rank j (identity when the indices are identical), for comparing ids / scores / boxes rank by rank."""
    rows_dev, rows_ref = np.asarray(rows_dev, dtype=np.int64), np.asarray(rows_ref, dtype=np.int64)
    perm = np.tile(np.arange(rows_ref.shape[1]), (rows_ref.shape[0], 1))     # device rank holding reference rank j
    if np.array_equal(rows_dev, rows_ref):
        return perm
    sc = np.asarray(scores_ref, dtype=np.float64).reshape(rows_ref.shape)
    for i in range(rows_ref.shape[0]):
        n, j = rows_ref.shape[1], 0
        while j < n:
            e = j + 1
            while e < n and rows_ref[i, e] >= 0 and abs(sc[i, e] - sc[i, e - 1]) < tie:
                e += 1
            a, b = rows_dev[i, j:e], rows_ref[i, j:e]
            if not np.array_equal(a, b):
                assert sorted(a.tolist()) == sorted(b.tolist()), "post-NMS indices differ (image %d, ranks %d..%d): %s vs %s" % (i, j, e - 1, a, b)
                perm[i, j:e] = j + np.array([a.tolist().index(v) for v in b.tolist()])
            j = e
    return perm


def take_ranks(t, perm):
    """Device output (B, N, .) reordered so that rank j holds what the reference has at rank j (assert_rows_match)."""
    a = np.asarray(t.cpu().numpy() if hasattr(t, "cpu") else t)
    return np.take_along_axis(a, perm[:, :, None], axis=1)


def boxes_close(got, ref, abs_tol=1e-3, rel_tol=1e-5):
    """Decoded boxes (..., 4) against the fp64 oracle: |error| <= 1e-3 px + 1e-5 of the box's scale (its largest |coordinate|
    or side).  The boxes of these random-init fixtures are up to 800 px wide (exp() of unclipped raw sizes) and a corner is
    centre -+ side / 2, so a corner's fp32 round-off follows the box's size, not the corner's own value: measured 1.05e-3 px
    on a 100 px box, 2.8e-3 px on a 680 px one (5e-6 of the scale).  The bound is 2x that at every scale - and north_star's
    1e-3 px for boxes of ordinary size."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.maximum(np.abs(ref).max(axis=-1), np.maximum(np.abs(ref[..., 2] - ref[..., 0]), np.abs(ref[..., 3] - ref[..., 1])))
    bad = np.abs(got - ref) > (abs_tol + rel_tol * scale)[..., None]
    assert not bad.any(), "boxes differ: worst %.3e px (box scale %.1f px)" % (
        float(np.abs(got - ref)[bad].max()), float(np.broadcast_to(scale[..., None], ref.shape)[bad][np.argmax(np.abs(got - ref)[bad])]))
    return True
