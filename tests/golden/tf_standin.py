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
        self.lenient = None           # tuple of name prefixes whose unknown variables are created as zeros (sub-graphs outside the path)
        self.extra_vars = []          # ... and recorded here
        self.arg_scopes = [{}]        # slim.arg_scope stack: {function name: default kwargs}


_S = None


class _Scope:
    """tf.variable_scope(name_or_scope, default_name=None, values=None, reuse=None).  A scope OBJECT captured by `with ... as sc` and passed
    back in re-enters the SAME absolute name (TF1: nets/inception_v3.py:478-483 hands its scope to inception_v3_base), a string nests."""

    def __init__(self, name_or_scope, default_name=None, reuse=None):
        self.arg, self.default_name, self.reuse_kw = name_or_scope, default_name, reuse
        self.full = None
        self.entry = None

    def __enter__(self):
        parent_full = _S.stack[-1][0] if _S.stack else ""
        inherited = bool(_S.stack and _S.stack[-1][1])
        if isinstance(self.arg, _Scope):
            full = self.arg.full                                  # absolute
        else:
            name = self.arg if self.arg is not None else self.default_name
            assert isinstance(name, str), "variable_scope needs a name, a scope object or a default_name"
            full = parent_full + "/" + name if parent_full else name
        self.full = full
        self.entry = [full, inherited or bool(self.reuse_kw)]
        _S.stack.append(self.entry)
        return self

    def __exit__(self, *exc):
        assert _S.stack.pop() is self.entry
        return False

    def reuse_variables(self):
        self.entry[1] = True


def variable_scope(name_or_scope, default_name=None, values=None, reuse=None):
    return _Scope(name_or_scope, default_name, reuse)


def get_variable(name, shape=None, dtype=None, initializer=None, **kw):
    prefix = _S.stack[-1][0] if _S.stack else ""
    full = prefix + "/" + name if prefix else name
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
        if _S.lenient is not None and full.startswith(_S.lenient):  # variables of a sub-graph outside the path under check: zeros, recorded
            v = torch.zeros(shape, dtype=DT)
            _S.vars[full] = v
            _S.extra_vars.append((full, tuple(shape)))
            return Tensor(v)
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
    assert padding in ("SAME", "VALID") and strides[0] == 1 and strides[3] == 1
    x = _raw(input).permute(0, 3, 1, 2)                      # NHWC -> NCHW
    w = _raw(filter).permute(3, 2, 0, 1)                     # HWIO -> OIHW
    if padding == "VALID":
        return Tensor(F.conv2d(x, w, stride=(strides[1], strides[2])).permute(0, 2, 3, 1))
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


def concat(values=None, axis=None, **kw):
    if isinstance(values, int):                              # (the pre-1.0 positional order concat(axis, values) is not used by the files under check)
        values, axis = axis, values
    return Tensor(torch.cat([_raw(v) for v in values], dim=axis))


def reduce_mean(x, axis=None):
    return Tensor(_raw(x).mean() if axis is None else _raw(x).mean(dim=axis))


def moments(x, axes):
    t = _raw(x)
    m = t.mean(dim=list(axes))
    return Tensor(m), Tensor(((t - m) ** 2).mean(dim=list(axes)))


def l2_loss(t):
    return Tensor((_raw(t) ** 2).sum() / 2)


def nn_dropout(x, keep_prob):
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


# --------------------------------------------------------------------------------------------------------- tf.contrib.slim
# The handful of slim layers nets/inception_v3.py and nets/inception_utils.py use, with slim's arg_scope mechanics (defaults per function,
# nested scopes merge, a captured scope can be re-applied).  Same caveat as above: this file's own statement of slim's published
# behaviour (conv2d = conv -> normalizer_fn OR biases -> activation_fn; batch_norm in inference mode = (x - moving_mean) /
# sqrt(moving_variance + epsilon) [* gamma] + beta with variables beta / gamma / moving_mean / moving_variance under <scope>/BatchNorm;
# SAME average pooling divides by the number of taps inside the image), not slim.
def _two(v):
    return [int(v), int(v)] if isinstance(v, int) else [int(v[0]), int(v[1])]


def _scoped(key):
    def deco(fn):
        def wrapper(*a, **k):
            merged = dict(_S.arg_scopes[-1].get(key, {}))
            merged.update(k)
            return fn(*a, **merged)
        wrapper.__name__ = key
        wrapper._slim_key = key
        return wrapper
    return deco


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
    cur = {k: dict(v) for k, v in _S.arg_scopes[-1].items()}
    if isinstance(list_ops_or_scope, dict):
        assert not kwargs
        for k, v in list_ops_or_scope.items():
            cur.setdefault(k, {}).update(v)
    else:
        for op in list_ops_or_scope:
            cur.setdefault(op._slim_key, {}).update(kwargs)
    _S.arg_scopes.append(cur)
    try:
        yield cur
    finally:
        _S.arg_scopes.pop()


def relu(x):
    return Tensor(torch.relu(_raw(x)))


@_scoped("batch_norm")
def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, is_training=True, updates_collections=None, scope=None, **kw):
    assert not is_training, "the path under check runs the front end with is_training=False (rllab/sampler/base.py:122-127)"
    c = int(_raw(inputs).shape[-1])
    with variable_scope(scope, "BatchNorm"):
        beta = _raw(get_variable("beta", [c])) if center else 0.0
        gamma = _raw(get_variable("gamma", [c])) if scale else 1.0
        mean = _raw(get_variable("moving_mean", [c]))
        var = _raw(get_variable("moving_variance", [c]))
    return Tensor((_raw(inputs) - mean) / torch.sqrt(var + epsilon) * gamma + beta)


@_scoped("conv2d")
def slim_conv2d(inputs, num_outputs, kernel_size, stride=1, padding="SAME", activation_fn=relu, normalizer_fn=None, normalizer_params=None,
                weights_initializer=None, weights_regularizer=None, biases_initializer=None, scope=None, **kw):
    kh, kw_ = _two(kernel_size)
    sh, sw = _two(stride)
    cin = int(_raw(inputs).shape[-1])
    with variable_scope(scope, "Conv"):
        w = get_variable("weights", [kh, kw_, cin, int(num_outputs)])
        out = conv2d(inputs, w, [1, sh, sw, 1], padding)
        if normalizer_fn is not None:
            out = normalizer_fn(out, **(normalizer_params or {}))
        else:
            out = Tensor(_raw(out) + _raw(get_variable("biases", [int(num_outputs)])))
        if activation_fn is not None:
            out = activation_fn(out)
    return out



def _pool_pads(n, k, s, padding):
    if padding == "VALID":
        return 0, 0
    return _same_pads(n, k, s)


@_scoped("max_pool2d")
def max_pool2d(inputs, kernel_size, stride=2, padding="VALID", scope=None, **kw):
    kh, kw_ = _two(kernel_size)
    sh, sw = _two(stride)
    x = _raw(inputs).permute(0, 3, 1, 2)
    pt, pb = _pool_pads(x.shape[2], kh, sh, padding)
    pl, pr = _pool_pads(x.shape[3], kw_, sw, padding)
    x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return Tensor(F.max_pool2d(x, (kh, kw_), stride=(sh, sw)).permute(0, 2, 3, 1))


@_scoped("avg_pool2d")
def avg_pool2d(inputs, kernel_size, stride=2, padding="VALID", scope=None, **kw):
    kh, kw_ = _two(kernel_size)
    sh, sw = _two(stride)
    x = _raw(inputs).permute(0, 3, 1, 2)
    pt, pb = _pool_pads(x.shape[2], kh, sh, padding)
    pl, pr = _pool_pads(x.shape[3], kw_, sw, padding)
    ones = F.pad(torch.ones((1, 1) + tuple(x.shape[2:]), dtype=DT), (pl, pr, pt, pb))
    xs = F.avg_pool2d(F.pad(x, (pl, pr, pt, pb)), (kh, kw_), stride=(sh, sw)) * (kh * kw_)
    cnt = F.avg_pool2d(ones, (kh, kw_), stride=(sh, sw)) * (kh * kw_)           # taps inside the image
    return Tensor((xs / cnt).permute(0, 2, 3, 1))


@_scoped("dropout")
def dropout(inputs, keep_prob=0.5, is_training=True, scope=None, **kw):
    assert not is_training
    return inputs


def softmax(logits, scope=None):
    return Tensor(torch.softmax(_raw(logits), dim=-1))


def squeeze(x, axis=None, name=None):
    t = _raw(x)
    for a in sorted(axis or [], reverse=True):
        t = t.squeeze(a)
    return Tensor(t)


def _slim_namespace():
    ns = types.SimpleNamespace(arg_scope=arg_scope, conv2d=slim_conv2d, batch_norm=batch_norm, max_pool2d=max_pool2d, avg_pool2d=avg_pool2d,
                               dropout=dropout, softmax=softmax, l2_regularizer=lambda *a, **k: None,
                               variance_scaling_initializer=lambda *a, **k: None, ops=types.SimpleNamespace())
    def _fc(*a, **k):
        raise NotImplementedError("slim.fully_connected is outside the path under check")
    ns.fully_connected = _scoped("fully_connected")(_fc)
    return ns


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
                                  l2_loss=l2_loss, dropout=nn_dropout, relu=relu)
    tf.squeeze = squeeze
    tf.GraphKeys = types.SimpleNamespace(UPDATE_OPS="update_ops")
    tf.contrib = types.SimpleNamespace(slim=_slim_namespace(), layers=_Inert())
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
