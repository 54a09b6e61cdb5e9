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
