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


def check_masks_differ_only_at_ties(onet_natural_pre, masks, band=2e-4):
    """The supplied decisions may differ from the oracle's own only where the oracle's |pre-activation| is tiny."""
    nflip = 0
    for name, u in onet_natural_pre.items():
        diff = (u > 0) != masks[name]
        if diff.any():
            assert np.abs(u[diff]).max() < band, (name, float(np.abs(u[diff]).max()))
            nflip += int(diff.sum())
    return nflip
