"""An EAGER STAND-IN for the handful of `tf.*` names the reference's live model classes touch -- NOT TensorFlow, and it pins
nothing about TensorFlow's op semantics (the oracle stays "parity unpinned").

Why it exists ("missing 1"): TensorFlow 1.x cannot be installed in the build container, so the reference's own
`ContextSkipNew.build` / `ContextAEReal.build` / `ContextAEInception2.build` (gym/envs/mujoco/arm_shaping.py:1272-1354, :1611-1684,
:1792-1894) have never been EXECUTED here; the oracle is the builder's reading of them.  With this module behind
`import tensorflow`, tests/golden/check_reference_wiring.py runs the reference's Python unmodified (loaded from /root/reference at
run time, build container only -- nothing of the reference travels or is stored) and compares every fetch and every parameter
gradient with oracle/: which tensor feeds which layer, which layers carry lrelu, the concat orders, the variable scopes / names /
shapes / sharing, where the dropout sites sit, how the loss terms are combined -- the WIRING -- become "the reference's code,
executed" instead of "read".

What the ops below compute is this file's own statement of TF's published rules, written independently of oracle/ (torch float64 +
autograd, no shared code): SAME padding = ceil(n / s) outputs with the odd pad element at the END; conv2d_transpose = the gradient of
conv2d(., w, SAME) with respect to its input (TF's literal definition: `conv2d_backprop_input`), obtained here from torch autograd
of the forward conv rather than from an index formula; l2_loss = sum(x^2) / 2; dropout = x * mask / keep_prob with the masks handed
in by the caller in call order.  So a disagreement between this file and oracle/ is either a wiring error or an op-rule error in
one of two independent statements; an agreement is NOT evidence about TensorFlow itself.

Variable scoping follows TF1's rules strictly enough to catch sharing mistakes: get_variable on an existing name without
`reuse` raises, with `reuse` on a missing name raises; `reuse` is inherited by nested scopes.  Values come from the dict given
to `install(values)` (the oracle's parameter inventory, keyed by TF variable name): a name or shape the oracle does not hold raises.
"""
import contextlib
import sys
import types

import torch
import torch.nn.functional as F

DT = torch.float64


class Shape(list):
    def as_list(self):
        return list(self)


class Tensor:
    """A torch tensor behind the few methods / operators the reference uses on tf.Tensor."""

    def __init__(self, t):
        self.t = t

    def get_shape(self):
        return Shape(int(d) for d in self.t.shape)

    def __getitem__(self, i):
        return Tensor(self.t[i])

    def __add__(self, o):
        return Tensor(self.t + _raw(o))

    __radd__ = __add__

    def __sub__(self, o):
        return Tensor(self.t - _raw(o))

    def __rsub__(self, o):
        return Tensor(_raw(o) - self.t)

    def __mul__(self, o):
        return Tensor(self.t * _raw(o))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return Tensor(self.t / _raw(o))

    def __pow__(self, e):
        return Tensor(self.t ** e)

    def __neg__(self):
        return Tensor(-self.t)

    def numpy(self):
        return self.t.detach().numpy()


def _raw(x):
    return x.t if isinstance(x, Tensor) else x


# ------------------------------------------------------------------------------------------------ variables and scopes
class _State:
    def __init__(self, values):
        self.values = values          # TF variable name -> numpy array (the oracle's inventory)
        self.vars = {}                # created variables: name -> leaf torch tensor
        self.stack = []               # open scopes: [name, reuse]
        self.dropout_masks = None     # iterator of numpy arrays (mask / keep_prob), consumed in call order
        self.dropout_calls = 0
        self.get_variable_calls = []  # (name, created?) in call order


_S = None


class _Scope:
    def __init__(self, name):
        self.name = name
        self.entry = None

    def __enter__(self):
        inherited = bool(_S.stack and _S.stack[-1][1])
        self.entry = [self.name, inherited]
        _S.stack.append(self.entry)
        return self

    def __exit__(self, *exc):
        assert _S.stack.pop() is self.entry
        return False

    def reuse_variables(self):
        self.entry[1] = True


def variable_scope(name_or_scope, *a, **k):
    assert isinstance(name_or_scope, str), "the live classes only open scopes by name"
    return _Scope(name_or_scope)


def get_variable(name, shape=None, dtype=None, initializer=None):
    full = "/".join([s[0] for s in _S.stack] + [name])
    reuse = bool(_S.stack and _S.stack[-1][1])
    shape = [int(d) for d in shape]
    if full in _S.vars:
        if not reuse:
            raise ValueError(f"Variable {full} already exists, disallowed. Did you mean to set reuse=True?")
        v = _S.vars[full]
        assert list(v.shape) == shape, (full, list(v.shape), shape)
        _S.get_variable_calls.append((full, False))
        return Tensor(v)
    if reuse:
        raise ValueError(f"Variable {full} does not exist, or was not created with tf.get_variable().")
    if full not in _S.values:
        raise KeyError(f"the reference creates variable {full} {shape}, which the oracle's inventory does not hold")
    val = _S.values[full]
    assert list(val.shape) == shape, f"{full}: the reference asks for shape {shape}, the oracle holds {list(val.shape)}"
    v = torch.tensor(val, dtype=DT, requires_grad=True)
    _S.vars[full] = v
    _S.get_variable_calls.append((full, True))
    return Tensor(v)


def _initializer(*a, **k):
    return None


# ------------------------------------------------------------------------------------------------------------------ ops
def _same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _conv2d_same_nchw(x, w_oihw, sh, sw):
    kh, kw = w_oihw.shape[2:]
    pt, pb = _same_pads(x.shape[2], kh, sh)
    pl, pr = _same_pads(x.shape[3], kw, sw)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w_oihw, stride=(sh, sw))


def conv2d(input, filter, strides, padding):
    assert padding == "SAME" and strides[0] == 1 and strides[3] == 1
    x = _raw(input).permute(0, 3, 1, 2)                      # NHWC -> NCHW
    w = _raw(filter).permute(3, 2, 0, 1)                     # HWIO -> OIHW
    return Tensor(_conv2d_same_nchw(x, w, strides[1], strides[2]).permute(0, 2, 3, 1))


def conv2d_transpose(value, filter, output_shape, strides, padding="SAME"):
    """The gradient of conv2d(y, filter, strides, SAME) with respect to y, contracted with `value` -- taken from autograd of the
    forward convolution (create_graph: it stays differentiable in `value` and `filter`).  filter is [kh, kw, out_c, in_c]:
    as a forward filter HWIO its I is the transpose's OUTPUT channels."""
    assert padding == "SAME" and strides[0] == 1 and strides[3] == 1
    x = _raw(value).permute(0, 3, 1, 2)
    w = _raw(filter).permute(3, 2, 0, 1)                     # [kh,kw,out_c,in_c] -> O = in_c, I = out_c
    n, h, wd, c = [int(d) for d in output_shape]
    y = torch.zeros((n, c, h, wd), dtype=DT, requires_grad=True)
    fwd = _conv2d_same_nchw(y, w, strides[1], strides[2])
    assert fwd.shape == x.shape, f"conv2d_transpose: output_shape {output_shape} is inconsistent with the input {tuple(x.shape)}"
    (g,) = torch.autograd.grad(fwd, y, grad_outputs=x, create_graph=True)
    return Tensor(g.permute(0, 2, 3, 1))


def bias_add(value, bias):
    return Tensor(_raw(value) + _raw(bias))


def reshape(tensor, shape):
    return Tensor(_raw(tensor).reshape([int(d) for d in shape]))


def maximum(a, b):
    return Tensor(torch.maximum(_raw(a), _raw(b)))


def matmul(a, b):
    return Tensor(_raw(a) @ _raw(b))


def concat(values, axis):
    return Tensor(torch.cat([_raw(v) for v in values], dim=axis))


def reduce_mean(x, axis=None):
    return Tensor(_raw(x).mean() if axis is None else _raw(x).mean(dim=axis))


def moments(x, axes):
    t = _raw(x)
    m = t.mean(dim=list(axes))
    return Tensor(m), Tensor(((t - m) ** 2).mean(dim=list(axes)))


def l2_loss(t):
    return Tensor((_raw(t) ** 2).sum() / 2)


def dropout(x, keep_prob):
    _S.dropout_calls += 1
    if keep_prob == 1.0:
        return x                                             # TF returns x itself for keep_prob == 1
    assert _S.dropout_masks is not None, "keep_prob < 1 needs the caller's masks"
    m = next(_S.dropout_masks)
    assert tuple(m.shape) == tuple(_raw(x).shape), (m.shape, tuple(_raw(x).shape))
    return Tensor(_raw(x) * torch.tensor(m, dtype=DT))


def placeholder(value):
    """Not tf.placeholder's signature: the check feeds the value directly (eager)."""
    return Tensor(torch.tensor(value, dtype=DT))


# ------------------------------------------------------------------------------------------------------------- install
class _Inert:
    """tf.contrib.slim / tf.contrib.layers: the reference only names their attributes at import (default arguments such as
    slim.softmax, nets/inception_v3.py:425); none of the three live classes calls them."""

    def __getattr__(self, name):
        def refuse(*a, **k):
            raise NotImplementedError(f"tf_standin: tf.contrib.*.{name} is outside the path under check")
        return refuse


def _module(values):
    tf = types.ModuleType("tensorflow")
    tf.__doc__ = "tests/golden/tf_standin.py -- NOT TensorFlow"
    tf.float32 = "float32"
    tf.variable_scope, tf.get_variable = variable_scope, get_variable
    tf.truncated_normal_initializer = tf.random_normal_initializer = tf.constant_initializer = _initializer
    tf.reshape, tf.maximum, tf.matmul, tf.concat, tf.reduce_mean = reshape, maximum, matmul, concat, reduce_mean
    tf.nn = types.SimpleNamespace(conv2d=conv2d, conv2d_transpose=conv2d_transpose, bias_add=bias_add, moments=moments,
                                  l2_loss=l2_loss, dropout=dropout)
    tf.contrib = types.SimpleNamespace(slim=_Inert(), layers=_Inert())
    tf.gfile = types.ModuleType("tensorflow.gfile")
    return tf


@contextlib.contextmanager
def install(values):
    """`import tensorflow` resolves to the stand-in inside the block; yields the state (created variables, call log)."""
    global _S
    saved = {k: sys.modules.get(k) for k in ("tensorflow", "tensorflow.gfile")}
    prev, _S = _S, _State(values)
    tf = _module(values)
    sys.modules["tensorflow"], sys.modules["tensorflow.gfile"] = tf, tf.gfile
    try:
        yield _S
    finally:
        _S = prev
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
