"""CPU oracle — primitive operators.  TEST INFRASTRUCTURE ONLY.

NumPy (float64 by default) restatement of the operators the reference's yolo3_darknet53 path
composes.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this;
the product path (viddet_amd/) never does.

PARITY STATUS: **parity unpinned** for everything that executes inside MXNet / GluonCV in the
reference (Convolution, BatchNorm, LeakyReLU, box_nms, BBoxBatchIOU, YOLOV3Loss, SGD): those
packages are not vendored in /root/reference, cannot be imported in the build container
(ModuleNotFoundError) and the reference has no tests or golden vectors.  The semantics below follow
the reference call sites cited per function plus SURVEY.md Appendix A.  The NumPy-only reference
modules (utils/bbox.py, models/transforms/bbox.py) ARE pinned: see tests/golden/.

Tensors follow the reference's layouts (NCHW activations, OIHW weights).
"""
import numpy as np


# ---------------------------------------------------------------------------------------------
# models/definitions/layers.py:66-67  nn.Conv2D(channel, kernel, strides, padding, use_bias=False)
# ---------------------------------------------------------------------------------------------
def _im2col(x, kh, kw, stride, pad):
    n, c, h, w = x.shape
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (w + 2 * pad - kw) // stride + 1
    xp = np.zeros((n, c, h + 2 * pad, w + 2 * pad), dtype=x.dtype)
    xp[:, :, pad:pad + h, pad:pad + w] = x
    cols = np.empty((n, c, kh, kw, ho, wo), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, :, ky, kx] = xp[:, :, ky:ky + stride * ho:stride, kx:kx + stride * wo:stride]
    return cols, ho, wo


def conv2d(x, w, stride=1, pad=0, bias=None):
    """x (N,C,H,W), w (O,C,kh,kw) -> (N,O,Ho,Wo).  Cross-correlation, zero padding (MXNet Convolution)."""
    o, c, kh, kw = w.shape
    cols, ho, wo = _im2col(x, kh, kw, stride, pad)
    y = np.tensordot(w.reshape(o, -1), cols.reshape(x.shape[0], c * kh * kw, ho * wo), axes=([1], [1]))
    y = y.transpose(1, 0, 2).reshape(x.shape[0], o, ho, wo)
    if bias is not None:
        y = y + bias.reshape(1, -1, 1, 1)
    return y


def conv2d_backward(x, w, dy, stride=1, pad=0):
    """Gradients of conv2d wrt x and w (autograd.backward, train_yolov3.py:631)."""
    n, c, h, wd = x.shape
    o, _, kh, kw = w.shape
    cols, ho, wo = _im2col(x, kh, kw, stride, pad)
    dy2 = dy.reshape(n, o, ho * wo)
    cols2 = cols.reshape(n, c * kh * kw, ho * wo)
    # dw[o,k] = sum_n sum_p dy[n,o,p] cols[n,k,p];  dcols[n,k,p] = sum_o w[o,k] dy[n,o,p]   (per-frame BLAS products)
    dw = np.tensordot(dy2, cols2, axes=([0, 2], [0, 2])).reshape(w.shape)
    dcols = np.matmul(w.reshape(o, -1).T[None], dy2).reshape(n, c, kh, kw, ho, wo)
    dxp = np.zeros((n, c, h + 2 * pad, wd + 2 * pad), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            dxp[:, :, ky:ky + stride * ho:stride, kx:kx + stride * wo:stride] += dcols[:, :, ky, kx]
    return dxp[:, :, pad:pad + h, pad:pad + wd], dw


# ---------------------------------------------------------------------------------------------
# models/definitions/layers.py:73-79  nn.Conv3D (NCDHW, OIDHW), temporal stride 1
# ---------------------------------------------------------------------------------------------
def conv3d(x, w, pad_d, pad, stride=1):
    """x (N,C,D,H,W), w (O,C,kd,kh,kw); spatial stride `stride`, temporal stride 1."""
    n, c, d, h, wd = x.shape
    o, _, kd, kh, kw = w.shape
    out = None
    xp = np.zeros((n, c, d + 2 * pad_d, h, wd), dtype=x.dtype)
    xp[:, :, pad_d:pad_d + d] = x
    do = d + 2 * pad_d - kd + 1
    for kz in range(kd):
        frames = xp[:, :, kz:kz + do]                                  # (n,c,do,h,w)
        f2 = frames.transpose(0, 2, 1, 3, 4).reshape(n * do, c, h, wd)
        y = conv2d(f2, w[:, :, kz], stride, pad)
        out = y if out is None else out + y
    ho, wo = out.shape[2], out.shape[3]
    return out.reshape(n, do, o, ho, wo).transpose(0, 2, 1, 3, 4)


def conv3d_backward(x, w, dy, pad_d, pad, stride=1):
    """Gradients of conv3d wrt x and w (temporal stride 1)."""
    n, c, d, h, wd = x.shape
    o, _, kd, kh, kw = w.shape
    do = d + 2 * pad_d - kd + 1
    xp = np.zeros((n, c, d + 2 * pad_d, h, wd), dtype=x.dtype)
    xp[:, :, pad_d:pad_d + d] = x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    ho, wo = dy.shape[3], dy.shape[4]
    dy2 = dy.transpose(0, 2, 1, 3, 4).reshape(n * do, o, ho, wo)
    for kz in range(kd):
        f2 = xp[:, :, kz:kz + do].transpose(0, 2, 1, 3, 4).reshape(n * do, c, h, wd)
        dx2, dwk = conv2d_backward(f2, w[:, :, kz], dy2, stride, pad)
        dw[:, :, kz] = dwk
        dxp[:, :, kz:kz + do] += dx2.reshape(n, do, c, h, wd).transpose(0, 2, 1, 3, 4)
    return dxp[:, :, pad_d:pad_d + d], dw


# ---------------------------------------------------------------------------------------------
# models/definitions/layers.py:68  BatchNorm(epsilon=1e-5, momentum=0.9)   
# ---------------------------------------------------------------------------------------------
def bn_train(x, gamma, beta, eps=1e-5):
    """Batch statistics over (N,H,W); biased variance.  Returns y, mean, var."""
    axes = (0,) + tuple(range(2, x.ndim))
    mean = x.mean(axis=axes)
    var = x.var(axis=axes)
    shp = (1, -1) + (1,) * (x.ndim - 2)
    y = (x - mean.reshape(shp)) / np.sqrt(var.reshape(shp) + eps) * gamma.reshape(shp) + beta.reshape(shp)
    return y, mean, var


def bn_running_update(running, batch, momentum=0.9):
    return running * momentum + batch * (1.0 - momentum)


def bn_eval(x, gamma, beta, rmean, rvar, eps=1e-5):
    shp = (1, -1) + (1,) * (x.ndim - 2)
    return (x - rmean.reshape(shp)) / np.sqrt(rvar.reshape(shp) + eps) * gamma.reshape(shp) + beta.reshape(shp)


def bn_train_backward(x, gamma, mean, var, dy, eps=1e-5):
    """Returns dx, dgamma, dbeta for training-mode BN."""
    axes = (0,) + tuple(range(2, x.ndim))
    shp = (1, -1) + (1,) * (x.ndim - 2)
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean.reshape(shp)) * invstd.reshape(shp)
    dbeta = dy.sum(axis=axes)
    dgamma = (dy * xhat).sum(axis=axes)
    m = x.size / x.shape[1]
    dx = (gamma * invstd).reshape(shp) * (dy - dbeta.reshape(shp) / m - xhat * dgamma.reshape(shp) / m)
    return dx, dgamma, dbeta


# models/definitions/layers.py:69  nn.LeakyReLU(0.1)
# `pos` (optional) = externally supplied branch decisions.  The parity tests pass the DEVICE's own x>0 decisions
# so that an element whose pre-activation is ~1e-7 (sign decided by fp32 vs fp64 rounding; about one such
# element per training step of the whole net) does not turn into a spurious 1-2 % gradient difference.  The
# tests assert that supplied and natural decisions differ only where |x| < 1e-5.
def leaky(x, slope=0.1, pos=None):
    return np.where((x > 0) if pos is None else pos, x, x * slope)


def leaky_backward(x, dy, slope=0.1, pos=None):
    return np.where((x > 0) if pos is None else pos, dy, dy * slope)


# models/definitions/layers.py:11-20  _upsample: repeat along W then H
def upsample2x(x):
    return x.repeat(2, axis=-1).repeat(2, axis=-2)


def upsample2x_backward(dy):
    n, c, h, w = dy.shape
    return dy.reshape(n, c, h // 2, 2, w // 2, 2).sum(axis=(3, 5))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ---------------------------------------------------------------------------------------------
# models/definitions/layers.py:161-205 TemporalPooling ; :208-264 TimeDistributed is a reshape
# ---------------------------------------------------------------------------------------------
def temporal_pool(x, type_):
    """x (B,K,...) -> (B,...)"""
    return x.max(axis=1) if type_ == 'max' else x.mean(axis=1)


# ---------------------------------------------------------------------------------------------
# gluon.Trainer('sgd', {'wd','momentum'}) .step(batch_size)   train_yolov3.py:527-530,634
# ---------------------------------------------------------------------------------------------
def sgd_momentum(w, g, mom, lr, momentum, wd, rescale):
    mom_new = momentum * mom - lr * (rescale * g + wd * w)
    return w + mom_new, mom_new


# transforms.py:229-245 (mean/std :167-168): to_tensor (/255, HWC->CHW) then normalize
MEAN = np.array([0.485, 0.456, 0.406])
STD = np.array([0.229, 0.224, 0.225])


def preprocess_u8(img_hwc_u8):
    x = img_hwc_u8.astype(np.float64) / 255.0
    x = (x - MEAN) / STD
    return np.moveaxis(x, -1, -3)
