"""CPU oracle for the frozen Inception-v3 front end up to Mixed_7c  --  TEST INFRASTRUCTURE ONLY.

Restates nets/inception_v3.py:29-416 (inception_v3_base, depth_multiplier 1) under the arg scope of
nets/inception_utils.py:31-71 as used by the sampler and the trainer (rllab/sampler/base.py:121-129,
scripts/train_script.py:104-114: is_training=False): every slim.conv2d is conv (no bias) -> batch norm with moving
statistics, epsilon 0.001, center but no scale (slim.batch_norm defaults) -> ReLU.  Variables per conv:
<scope>/weights [kh,kw,cin,cout], <scope>/BatchNorm/{beta, moving_mean, moving_variance} [cout].

Pinned against what the reference's own test holds for this path: the end-point shapes at 299x299 and the total of
21,802,784 model variables (nets/inception_v3_test.py:87-104, :112-120).  Numerically it is cross-checked against an
independent torch statement (tests/_torch_ref.py); the pretrained checkpoint is not in the reference tree, so there are
no golden activations ("parity unpinned" beyond structure, as for the rest of oracle/).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

BN_EPS = 0.001          # inception_utils.py:34


def _same(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def _windows(x, kh, kw, s, padding, fill=0.0):
    N, H, W, C = x.shape
    if padding == "SAME":
        Ho, pt, pb = _same(H, kh, s)
        Wo, pl, pr = _same(W, kw, s)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=fill)
    else:
        Ho, Wo = (H - kh) // s + 1, (W - kw) // s + 1
    sN, sH, sW, sC = x.strides
    return np.lib.stride_tricks.as_strided(x, (N, Ho, Wo, kh, kw, C), (sN, s * sH, s * sW, sH, sW, sC), writeable=False)


class Net:
    """Walks the graph once; in 'spec' mode records (name, shape) of every variable, in 'run' mode computes."""

    def __init__(self, params=None):
        self.p, self.specs = params, []

    def conv(self, x, cout, k, scope, stride=1, padding="SAME"):
        kh, kw = k
        cin = x.shape[-1] if self.p is not None else x[-1]
        if self.p is None:
            self.specs += [(f"{scope}/weights", (kh, kw, cin, cout)), (f"{scope}/BatchNorm/beta", (cout,)),
                           (f"{scope}/BatchNorm/moving_mean", (cout,)), (f"{scope}/BatchNorm/moving_variance", (cout,))]
            H, W = x[1], x[2]
            if padding == "SAME":
                return (x[0], -(-H // stride), -(-W // stride), cout)
            return (x[0], (H - kh) // stride + 1, (W - kw) // stride + 1, cout)
        w = self.p[f"{scope}/weights"]
        win = _windows(x, kh, kw, stride, padding)
        N, Ho, Wo = win.shape[:3]
        y = (win.reshape(N * Ho * Wo, -1) @ w.reshape(-1, cout)).reshape(N, Ho, Wo, cout)
        mean, var, beta = (self.p[f"{scope}/BatchNorm/{n}"] for n in ("moving_mean", "moving_variance", "beta"))
        return np.maximum((y - mean) / np.sqrt(var + BN_EPS) + beta, 0)

    def max_pool(self, x, stride, padding):
        if self.p is None:
            return (x[0], (x[1] - 3) // stride + 1, (x[2] - 3) // stride + 1, x[3]) if padding == "VALID" else x
        return _windows(x, 3, 3, stride, padding, fill=-np.inf).max(axis=(3, 4))

    def avg_pool(self, x):
        """slim.avg_pool2d(net, [3,3]) with stride 1, SAME: TF divides by the number of taps inside the image."""
        if self.p is None:
            return x
        s = _windows(x, 3, 3, 1, "SAME").sum(axis=(3, 4))
        cnt = _windows(np.ones((1,) + x.shape[1:3] + (1,), x.dtype), 3, 3, 1, "SAME").sum(axis=(3, 4))
        return s / cnt

    def concat(self, xs):
        if self.p is None:
            return xs[0][:3] + (sum(x[3] for x in xs),)
        return np.concatenate(xs, axis=3)


def graph(net: Net, x, final="Mixed_7c"):
    """inception_v3.py:93-416.  Returns OrderedDict of end points."""
    ep = OrderedDict()
    S = "InceptionV3/"

    def done(name, val):
        ep[name] = val
        return name == final

    x = net.conv(x, 32, (3, 3), S + "Conv2d_1a_3x3", stride=2, padding="VALID")
    if done("Conv2d_1a_3x3", x): return ep
    x = net.conv(x, 32, (3, 3), S + "Conv2d_2a_3x3", padding="VALID")
    if done("Conv2d_2a_3x3", x): return ep
    x = net.conv(x, 64, (3, 3), S + "Conv2d_2b_3x3")
    if done("Conv2d_2b_3x3", x): return ep
    x = net.max_pool(x, 2, "VALID")
    if done("MaxPool_3a_3x3", x): return ep
    x = net.conv(x, 80, (1, 1), S + "Conv2d_3b_1x1", padding="VALID")
    if done("Conv2d_3b_1x1", x): return ep
    x = net.conv(x, 192, (3, 3), S + "Conv2d_4a_3x3", padding="VALID")
    if done("Conv2d_4a_3x3", x): return ep
    x = net.max_pool(x, 2, "VALID")
    if done("MaxPool_5a_3x3", x): return ep

    # 35x35 blocks (:140-213); the scope names of Mixed_5c's 5x5 branch are irregular in the reference (:170-173)
    for name, pool_c, b1 in (("Mixed_5b", 32, ("Conv2d_0a_1x1", "Conv2d_0b_5x5")),
                             ("Mixed_5c", 64, ("Conv2d_0b_1x1", "Conv_1_0c_5x5")),
                             ("Mixed_5d", 64, ("Conv2d_0a_1x1", "Conv2d_0b_5x5"))):
        P = S + name + "/"
        b0 = net.conv(x, 64, (1, 1), P + "Branch_0/Conv2d_0a_1x1")
        t = net.conv(x, 48, (1, 1), P + "Branch_1/" + b1[0])
        t = net.conv(t, 64, (5, 5), P + "Branch_1/" + b1[1])
        u = net.conv(x, 64, (1, 1), P + "Branch_2/Conv2d_0a_1x1")
        u = net.conv(u, 96, (3, 3), P + "Branch_2/Conv2d_0b_3x3")
        u = net.conv(u, 96, (3, 3), P + "Branch_2/Conv2d_0c_3x3")
        v = net.conv(net.avg_pool(x), pool_c, (1, 1), P + "Branch_3/Conv2d_0b_1x1")
        x = net.concat([b0, t, u, v])
        if done(name, x): return ep

    P = S + "Mixed_6a/"                                                        # :216-232
    b0 = net.conv(x, 384, (3, 3), P + "Branch_0/Conv2d_1a_1x1", stride=2, padding="VALID")
    t = net.conv(x, 64, (1, 1), P + "Branch_1/Conv2d_0a_1x1")
    t = net.conv(t, 96, (3, 3), P + "Branch_1/Conv2d_0b_3x3")
    t = net.conv(t, 96, (3, 3), P + "Branch_1/Conv2d_1a_1x1", stride=2, padding="VALID")
    x = net.concat([b0, t, net.max_pool(x, 2, "VALID")])
    if done("Mixed_6a", x): return ep

    for name, c in (("Mixed_6b", 128), ("Mixed_6c", 160), ("Mixed_6d", 160), ("Mixed_6e", 192)):   # :235-346
        P = S + name + "/"
        b0 = net.conv(x, 192, (1, 1), P + "Branch_0/Conv2d_0a_1x1")
        t = net.conv(x, c, (1, 1), P + "Branch_1/Conv2d_0a_1x1")
        t = net.conv(t, c, (1, 7), P + "Branch_1/Conv2d_0b_1x7")
        t = net.conv(t, 192, (7, 1), P + "Branch_1/Conv2d_0c_7x1")
        u = net.conv(x, c, (1, 1), P + "Branch_2/Conv2d_0a_1x1")
        u = net.conv(u, c, (7, 1), P + "Branch_2/Conv2d_0b_7x1")
        u = net.conv(u, c, (1, 7), P + "Branch_2/Conv2d_0c_1x7")
        u = net.conv(u, c, (7, 1), P + "Branch_2/Conv2d_0d_7x1")
        u = net.conv(u, 192, (1, 7), P + "Branch_2/Conv2d_0e_1x7")
        v = net.conv(net.avg_pool(x), 192, (1, 1), P + "Branch_3/Conv2d_0b_1x1")
        x = net.concat([b0, t, u, v])
        if done(name, x): return ep

    P = S + "Mixed_7a/"                                                        # :349-369
    b0 = net.conv(x, 192, (1, 1), P + "Branch_0/Conv2d_0a_1x1")
    b0 = net.conv(b0, 320, (3, 3), P + "Branch_0/Conv2d_1a_3x3", stride=2, padding="VALID")
    t = net.conv(x, 192, (1, 1), P + "Branch_1/Conv2d_0a_1x1")
    t = net.conv(t, 192, (1, 7), P + "Branch_1/Conv2d_0b_1x7")
    t = net.conv(t, 192, (7, 1), P + "Branch_1/Conv2d_0c_7x1")
    t = net.conv(t, 192, (3, 3), P + "Branch_1/Conv2d_1a_3x3", stride=2, padding="VALID")
    x = net.concat([b0, t, net.max_pool(x, 2, "VALID")])
    if done("Mixed_7a", x): return ep

    for name, b1b in (("Mixed_7b", "Conv2d_0b_3x1"), ("Mixed_7c", "Conv2d_0c_3x1")):     # :371-416
        P = S + name + "/"
        b0 = net.conv(x, 320, (1, 1), P + "Branch_0/Conv2d_0a_1x1")
        t = net.conv(x, 384, (1, 1), P + "Branch_1/Conv2d_0a_1x1")
        t = net.concat([net.conv(t, 384, (1, 3), P + "Branch_1/Conv2d_0b_1x3"), net.conv(t, 384, (3, 1), P + "Branch_1/" + b1b)])
        u = net.conv(x, 448, (1, 1), P + "Branch_2/Conv2d_0a_1x1")
        u = net.conv(u, 384, (3, 3), P + "Branch_2/Conv2d_0b_3x3")
        u = net.concat([net.conv(u, 384, (1, 3), P + "Branch_2/Conv2d_0c_1x3"), net.conv(u, 384, (3, 1), P + "Branch_2/Conv2d_0d_3x1")])
        v = net.conv(net.avg_pool(x), 192, (1, 1), P + "Branch_3/Conv2d_0b_1x1")
        x = net.concat([b0, t, u, v])
        if done(name, x): return ep
    return ep


def param_specs():
    net = Net()
    graph(net, (1, 299, 299, 3))
    return net.specs


def endpoint_shapes(H, W, batch=1):
    return OrderedDict((k, tuple(v)) for k, v in graph(Net(), (batch, H, W, 3)).items())


def init_params(seed, dtype=np.float64):
    """Synthetic stand-in for the absent checkpoint: variance-scaling weights (inception_utils.py:66), BN statistics
    with spread so that folding them matters."""
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shape in param_specs():
        if name.endswith("weights"):
            fan_in = shape[0] * shape[1] * shape[2]
            p[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(dtype)
        elif name.endswith("moving_variance"):
            p[name] = rng.uniform(0.5, 1.5, shape).astype(dtype)
        else:
            p[name] = (rng.standard_normal(shape) * 0.1).astype(dtype)
    return p


def forward(params, images, final="Mixed_7c"):
    """images f32 [N,H,W,3] in [-1,1] (base.py:116-119 preprocessing) -> end points."""
    return graph(Net(params), images, final)
