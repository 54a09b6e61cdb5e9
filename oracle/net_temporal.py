"""CPU oracle — YOLOV3T with a temporal window k>1 (the config-4 "temporal-conv block over stacked frame
features").  TEST INFRASTRUCTURE ONLY (parity-status statement: oracle/ops.py).

Follows, under /root/reference:
  layers.py:208-264  TimeDistributed (reshape1: fold K into the batch; BatchNorm therefore sees B*K frames)
  layers.py:161-205  TemporalPooling (max / mean over K)
  layers.py:73-89, 135-158  _conv3d / _conv21d / Conv
  yolo3.py:229-262   YOLODetectionBlockV3 with conv_type '3' / '21' (swapaxes to (B,C,K,h,w), Conv3D, swap back)
  yolo3.py:1016-1054 which blocks get TimeDistributed; :1105-1124 early join; :1134-1138 late join of the tips;
  yolo3.py:1167-1177 transition -> upsample -> concat on the channel axis (dim 2 of (B,K,C,h,w))
Activations are kept folded as (B*K, C, h, w) with frame index b*K + k; 3-D convs unfold to (B,C,K,h,w).
"""
from collections import OrderedDict

import numpy as np

from . import ops as R
from . import yolo as Y
from .net import Var, Net, backward, stage_names


def temporal_names(k, k_join_pos, bct):
    """Structural parameter-name prefixes of the k>1 variants (TimeDistributed registers `.model`, Conv `.conv`)."""
    late = k_join_pos == 'late'
    nm0 = stage_names()
    if k_join_pos == 'tout':
        # YOLOV3Temporal(t_out=True) (yolo3_temporal.py:313-343): TimeDistributed wrappers are made inside
        # hybrid_forward, so nothing registers a `.model` child; 3-D / 2+1-D block cells keep the Conv wrapper's `.conv`
        return nm0, (lambda i: ("yolo_blocks.%d" % i, ".conv" if bct in ('3', '21') else "")), (lambda i: "transitions.%d" % i)

    def stage(f):
        head, idx = nm0(f).rsplit(".", 1)
        return "%s.model.%s" % (head, idx)

    def block(i):
        if bct in ('3', '21'):
            return "yolo_blocks.%d" % i, ".conv"
        if late:
            return "yolo_blocks.%d.model" % i, ""
        return "yolo_blocks.%d" % i, ""

    def transition(i):
        return "transitions.%d%s" % (i, ".model" if late else "")

    return stage, block, transition


def param_shapes(num_class, k, k_join_pos, bct, k_join_type='max'):
    S = OrderedDict()
    cm = k if k_join_type == 'cat' else 1           # channel multiplier of a 'cat' join
    stage, block, transition = temporal_names(k, k_join_pos, bct)

    def cell(name, cin, cout, ksz, kd=None):
        S[name + ".0.weight"] = (cout, cin, ksz, ksz) if kd is None else (cout, cin, kd, ksz, ksz)
        for t in ("gamma", "beta", "running_mean", "running_var"):
            S[name + ".1." + t] = (cout,)

    cell(stage(0), 3, 32, 3)
    f = 1
    for nlayer, ch in zip([1, 2, 8, 8, 4], [64, 128, 256, 512, 1024]):
        cell(stage(f), ch // 2, ch, 3)
        f += 1
        for _ in range(nlayer):
            cell(stage(f) + ".body.0", ch, ch // 2, 1)
            cell(stage(f) + ".body.1", ch // 2, ch, 3)
            f += 1
    A = 3 * (5 + num_class)
    early = k_join_pos != 'late'
    cin = 1024 * (cm if early else 1)
    for i, c in enumerate([512, 256, 128]):
        pre, cl = block(i)

        def neck_cell(name, ci, co, ksz):
            if bct == '3':
                cell(name + cl, ci, co, ksz, kd=ksz)              # kernel (k,k,k): 1x1x1 or 3x3x3
            elif bct == '21':
                if ksz == 3:
                    cell(name + cl + ".0", ci, co, 3, kd=1)       # (1,3,3)
                    S[name + cl + ".1.0.weight"] = (co, co, 3, 1, 1)   # (3,1,1)
                    for t in ("gamma", "beta", "running_mean", "running_var"):
                        S[name + cl + ".1.1." + t] = (co,)
                else:
                    cell(name + cl, ci, co, 1, kd=1)
            else:
                cell(name, ci, co, ksz)

        x = cin
        for j in range(5):
            co = c if j % 2 == 0 else 2 * c
            neck_cell("%s.body.%d" % (pre, j), x, co, 1 if j % 2 == 0 else 3)
            x = co
        neck_cell(pre + ".tip", c, 2 * c, 3)
        S["yolo_outputs.%d.prediction.weight" % i] = (A, 2 * c * (1 if early else cm), 1, 1)
        S["yolo_outputs.%d.prediction.bias" % i] = (A,)
        if i < 2:
            cell(transition(i), c, c // 2, 1)
            cin = c // 2 + [512, 256][i] * (cm if early else 1)
    return S


def init_params(num_class, k, k_join_pos, bct, seed=0, obj_bias=0.0, k_join_type='max'):
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for key, shp in param_shapes(num_class, k, k_join_pos, bct, k_join_type).items():
        if key.endswith("weight"):
            P[key] = rng.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[1:]))
            if "prediction" in key:
                P[key] *= 0.05
        elif key.endswith("gamma"):
            P[key] = rng.uniform(0.2, 0.4, shp) if ".body.1.1." in key and key.startswith("stages") else rng.uniform(0.8, 1.2, shp)
        elif key.endswith("running_var"):
            P[key] = rng.uniform(0.8, 1.2, shp)
        elif key.endswith("bias"):
            b = rng.standard_normal(shp) * 0.1
            b.reshape(3, -1)[:, 4] += obj_bias
            P[key] = b
        else:
            P[key] = rng.standard_normal(shp) * 0.1
    return OrderedDict((kk, v.astype(np.float32).astype(np.float64)) for kk, v in P.items())


class TemporalNet(Net):
    def __init__(self, P, num_class, k, k_join_type, k_join_pos, bct='2'):
        super().__init__(P, num_class)
        assert k > 1 and k_join_type in ('max', 'mean', 'cat') and k_join_pos in ('early', 'late')
        if bct in ('3', '21'):
            assert k_join_pos == 'late'                               # yolo3.py:980
        self.k, self.jt, self.jp, self.bct = k, k_join_type, k_join_pos, bct
        self.argmax_override, self.argmax_natural = {}, {}

    # ---- TemporalPooling on folded frames: (B*K,C,h,w) -> (B,C,h,w)
    def pool(self, x, name=None):
        K = self.k
        v5 = x.v.reshape((-1, K) + x.v.shape[1:])
        if self.jt == 'max':
            am = v5.argmax(axis=1)
            self.argmax_natural[name] = (am, v5)
            if name in self.argmax_override:         # the device's winner (differs only at exact-ish ties; cf. ops.leaky)
                am = self.argmax_override[name]
            y = np.take_along_axis(v5, am[:, None], axis=1)[:, 0]
        elif self.jt == 'cat':                      # yolo3.py:1108,1136 F.reshape(x,(0,-3,-2)): (B,K,C,h,w)->(B,K*C,h,w)
            y = v5.reshape((v5.shape[0], K * v5.shape[2]) + v5.shape[3:])
        else:
            y = v5.mean(axis=1)

        def bw(g):
            if self.jt == 'max':
                d5 = np.zeros_like(v5)
                np.put_along_axis(d5, am[:, None], g[:, None], axis=1)
            elif self.jt == 'cat':
                d5 = g.reshape(v5.shape)
            else:
                d5 = np.broadcast_to(g[:, None] / K, v5.shape).copy()
            x.acc(d5.reshape(x.v.shape))

        return Var(y, (x,), bw)

    # ---- _conv3d cell on folded frames (layers.py:73-79 + yolo3.py:256-262 swapaxes)
    def cell3d(self, name, x, train, pad_d, pad, stride=1, out_k=None):
        """out_k: frames per window of the OUTPUT when the temporal conv is not padded (kd - 1 fewer than the input's);
        stride: spatial stride (the (1,2,2) of the side branches)."""
        P, K = self.P, self.k
        Ko = K if out_k is None else out_k
        w = P[name + ".0.weight"]
        n, c, h, wd = x.v.shape
        x5 = x.v.reshape(n // K, K, c, h, wd).transpose(0, 2, 1, 3, 4)          # (B,C,K,h,w)
        z = R.conv3d(x5, w, pad_d, pad, stride)
        ho, wo = z.shape[3], z.shape[4]
        no = (n // K) * Ko
        gamma, beta = P[name + ".1.gamma"], P[name + ".1.beta"]
        if train:
            u, mean, var = R.bn_train(z, gamma, beta)
            self.new_running[name + ".1.running_mean"] = R.bn_running_update(P[name + ".1.running_mean"], mean)
            self.new_running[name + ".1.running_var"] = R.bn_running_update(P[name + ".1.running_var"], var)
        else:
            u = R.bn_eval(z, gamma, beta, P[name + ".1.running_mean"], P[name + ".1.running_var"])
        pos = self.mask_override.get(name)          # given folded (B*K,C,h,w); unfold like the data
        if pos is not None:
            pos = pos.reshape(n // K, Ko, -1, ho, wo).transpose(0, 2, 1, 3, 4)
        if train:
            self.pre[name] = u.transpose(0, 2, 1, 3, 4).reshape(no, -1, ho, wo)
        y5 = R.leaky(u, pos=pos)
        y = y5.transpose(0, 2, 1, 3, 4).reshape(no, -1, ho, wo)

        def bw(g):
            g5 = g.reshape(n // K, Ko, -1, ho, wo).transpose(0, 2, 1, 3, 4)
            du = R.leaky_backward(u, g5, pos=pos)
            dz, dgamma, dbeta = R.bn_train_backward(z, gamma, mean, var, du)
            dx5, dw = R.conv3d_backward(x5, w, dz, pad_d, pad, stride)
            self.G[name + ".0.weight"] = dw
            self.G[name + ".1.gamma"] = dgamma
            self.G[name + ".1.beta"] = dbeta
            x.acc(dx5.transpose(0, 2, 1, 3, 4).reshape(n, c, h, wd))

        out = Var(y, (x,), bw if train else None)
        self.vars[name] = out
        return out

    def neck_cell(self, name, cl, x, ksz, train):
        if self.bct == '3':
            return self.cell3d(name + cl, x, train, ksz // 2, ksz // 2)
        if self.bct == '21':
            if ksz == 3:
                y = self.cell3d(name + cl + ".0", x, train, 0, 1)            # (1,3,3), padding (0,1,1)
                return self.cell3d(name + cl + ".1", y, train, 1, 0)         # (3,1,1), padding (1,0,0)
            return self.cell3d(name + cl, x, train, 0, 0)
        return self.cell(name, x, ksz, 1, train)

    def features(self, x_bk, train):
        """x_bk: (B,K,3,H,W).  Returns the three raw head Vars (B, A, g, g)."""
        stage, block, transition = temporal_names(self.k, self.jp, self.bct)
        b, K = x_bk.shape[:2]
        x = self.cell(stage(0), Var(x_bk.reshape((b * K,) + x_bk.shape[2:])), 3, 1, train)
        f = 1
        routes = []
        for nlayer, ch in zip([1, 2, 8, 8, 4], [64, 128, 256, 512, 1024]):
            x = self.cell(stage(f), x, 3, 2, train)
            f += 1
            for _ in range(nlayer):
                m = self.cell(stage(f) + ".body.0", x, 1, 1, train)
                x = self.cell(stage(f) + ".body.1", m, 3, 1, train, residual=x)
                f += 1
            if f in (15, 24, 29):
                routes.append(x)
        late = self.jp == 'late'
        if not late:
            routes = [self.pool(r, 'pool.route%d' % i) for i, r in enumerate(routes)]   # yolo3.py:1107-1124
        heads = []
        x = routes[2]
        for i in range(3):
            pre, cl = block(i)
            for j in range(5):
                x = self.neck_cell("%s.body.%d" % (pre, j), cl, x, 1 if j % 2 == 0 else 3, train)
            tip = self.neck_cell(pre + ".tip", cl, x, 3, train)
            if late:
                tip = self.pool(tip, 'pool.tip%d' % i)                       # :1134-1138
            heads.append(self.head(i, tip, train))
            if i < 2:
                t = self.cell(transition(i), x, 1, 1, train)
                x = self.upcat(t, routes[1 - i])
        return heads


class TemporalOutNet(TemporalNet):
    """YOLOV3Temporal with t_out=True, corr_d=0 (yolo3_temporal.py:396-470, 492-536): the t frames of a window run
    through TimeDistributed(stages) (BatchNorm over B*t frames), the detection blocks (TimeDistributed for conv 2,
    3-D / 2+1-D convs across t otherwise), TimeDistributed(transitions), and TimeDistributed(output): one set of
    predictions PER FRAME.  Heads come back folded as (B*t, A, g, g), frame index b*t_len + t; Net.detect /
    Net.train_step on targets folded the same way give the per-frame detections and the per-frame losses, whose
    mean over all B*t values is what the reference returns (:535)."""

    def __init__(self, P, num_class, t, bct='2'):
        Net.__init__(self, P, num_class)
        assert t == 5                                                      # :399
        self.k, self.jt, self.jp, self.bct = t, None, 'tout', bct
        self.argmax_override, self.argmax_natural = {}, {}

    def features(self, x_bk, train):
        stage, block, transition = temporal_names(self.k, 'tout', self.bct)
        b, K = x_bk.shape[:2]
        x = self.cell(stage(0), Var(x_bk.reshape((b * K,) + x_bk.shape[2:])), 3, 1, train)
        f = 1
        routes = []
        for nlayer, ch in zip([1, 2, 8, 8, 4], [64, 128, 256, 512, 1024]):
            x = self.cell(stage(f), x, 3, 2, train)
            f += 1
            for _ in range(nlayer):
                m = self.cell(stage(f) + ".body.0", x, 1, 1, train)
                x = self.cell(stage(f) + ".body.1", m, 3, 1, train, residual=x)
                f += 1
            if f in (15, 24, 29):
                routes.append(x)
        heads = []
        x = routes[2]
        for i in range(3):
            pre, cl = block(i)
            for j in range(5):
                x = self.neck_cell("%s.body.%d" % (pre, j), cl, x, 1 if j % 2 == 0 else 3, train)
            tip = self.neck_cell(pre + ".tip", cl, x, 3, train)
            heads.append(self.head(i, tip, train))                        # TimeDistributed(output): per frame
            if i < 2:
                t = self.cell(transition(i), x, 1, 1, train)
                x = self.upcat(t, routes[1 - i])
        return heads


def side_param_shapes(num_class):
    """Parameters of YOLOV3Temporal(t_out=False, conv=2) (yolo3_temporal.py:326-343): the k = 1 network's, plus the two
    _conv21d side branches convs1 / convs2 (layers.py:82-89: a (1,3,3) and a (3,1,1) Conv3D cell each)."""
    from .net import param_shapes as base_shapes
    S = OrderedDict(base_shapes(num_class))
    for i, (m, ch) in ((1, (256, 512)), (2, (512, 1024))):
        S["convs%d.0.0.0.weight" % i] = (m, m, 1, 3, 3)
        S["convs%d.0.1.0.weight" % i] = (ch, m, 3, 1, 1)
        for j, c in ((0, m), (1, ch)):
            for t in ("gamma", "beta", "running_mean", "running_var"):
                S["convs%d.0.%d.1.%s" % (i, j, t)] = (c,)
    return S


def side_init_params(num_class, seed=0, obj_bias=0.0):
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for key, shp in side_param_shapes(num_class).items():
        if key.endswith("weight"):
            P[key] = rng.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[1:]))
            if "prediction" in key:
                P[key] *= 0.05
        elif key.endswith("gamma"):
            P[key] = rng.uniform(0.2, 0.4, shp) if ".body.1.1." in key and key.startswith("stages") else rng.uniform(0.8, 1.2, shp)
        elif key.endswith("running_var"):
            P[key] = rng.uniform(0.8, 1.2, shp)
        elif key.endswith("bias"):
            b = rng.standard_normal(shp) * 0.1
            b.reshape(3, -1)[:, 4] += obj_bias
            P[key] = b
        else:
            P[key] = rng.standard_normal(shp) * 0.1
    return OrderedDict((kk, v.astype(np.float32).astype(np.float64)) for kk, v in P.items())


class TemporalSideNet(TemporalNet):
    """YOLOV3Temporal with t_out=False (yolo3_temporal.py:436-447): TimeDistributed(stages[0]) on the 5 frames; route 0 =
    the centre frame; convs1 (a per-frame (1,3,3) stride-(1,2,2) conv cell, then a (3,1,1) conv cell WITHOUT temporal padding:
    5 -> 3 frames) is added to TimeDistributed(stages[1]) of frames 1..3; route 1 = the middle of those; convs2 takes the 3
    frames to 1 and is added to stages[2] of that middle frame = route 2.  Neck and heads are the single-frame ones."""

    def __init__(self, P, num_class, t=5):
        Net.__init__(self, P, num_class)
        assert t == 5                                                      # :399
        self.k, self.jt, self.jp, self.bct = t, None, 'side', '2'
        self.argmax_override, self.argmax_natural = {}, {}

    @staticmethod
    def frames(x, K, k0, kc):
        """slice_axis(axis=1, begin=k0, end=k0+kc) on folded frames (B*K,C,h,w) -> (B*kc,C,h,w)."""
        v5 = x.v.reshape((-1, K) + x.v.shape[1:])
        y = v5[:, k0:k0 + kc].reshape((-1,) + x.v.shape[1:])

        def bw(g):
            d5 = np.zeros_like(v5)
            d5[:, k0:k0 + kc] = g.reshape((-1, kc) + x.v.shape[1:])
            x.acc(d5.reshape(x.v.shape))
        return Var(y, (x,), bw)

    @staticmethod
    def add(a, b):
        def bw(g):
            a.acc(g)
            b.acc(g)
        return Var(a.v + b.v, (a, b), bw)

    def side(self, i, x, K, train):
        """_conv21d(channel, t=3, d=3, m, padding=[1,0], stride=[(1,2,2),1]) on K folded frames -> K - 2 frames."""
        self.k = K                                      # cell3d unfolds with self.k frames per window
        y = self.cell3d("convs%d.0.0" % i, x, train, 0, 1, stride=2)          # (1,3,3), pad (0,1,1), stride (1,2,2)
        z = self.cell3d("convs%d.0.1" % i, y, train, 0, 0, out_k=K - 2)       # (3,1,1), pad (0,0,0)
        self.k = 5
        return z

    def features(self, x_bk, train):
        nm = stage_names()
        b, K = x_bk.shape[:2]
        x = self.cell(nm(0), Var(x_bk.reshape((b * K,) + x_bk.shape[2:])), 3, 1, train)
        f = 1
        routes = []
        for gi, (nlayer, ch) in enumerate(zip([1, 2, 8, 8, 4], [64, 128, 256, 512, 1024])):
            side = None
            if gi == 3:
                routes.append(self.frames(x, 5, 2, 1))                     # :437
                side = self.side(1, x, 5, train)                           # :438
                x = self.frames(x, 5, 1, 3)                                # :439
            elif gi == 4:
                side = self.side(2, x, 3, train)                           # :443
                x = routes[1]                                              # :444 x.slice_axis(1, 1, 2)
            x = self.cell(nm(f), x, 3, 2, train)
            f += 1
            for _ in range(nlayer):
                m = self.cell(nm(f) + ".body.0", x, 1, 1, train)
                x = self.cell(nm(f) + ".body.1", m, 3, 1, train, residual=x)
                f += 1
            if side is not None:
                x = self.add(x, side)                                      # :440,445
                routes.append(self.frames(x, 3, 1, 1) if gi == 3 else x)   # :441 / :446-447
        heads = []
        x = routes[2]
        for i in range(3):
            for j in range(5):
                x = self.cell("yolo_blocks.%d.body.%d" % (i, j), x, 1 if j % 2 == 0 else 3, 1, train)
            tip = self.cell("yolo_blocks.%d.tip" % i, x, 3, 1, train)
            heads.append(self.head(i, tip, train))
            if i < 2:
                t = self.cell("transitions.%d" % i, x, 1, 1, train)
                x = self.upcat(t, routes[1 - i])
        return heads
