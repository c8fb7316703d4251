"""Host side of the frozen Inception-v3 front end of mode 'oursinception'.

The reference computes   inception_v3.inception_v3(images, num_classes=1001, is_training=False)[1]['Mixed_7c']
(rllab/sampler/base.py:122-127, scripts/train_script.py:104-111) and feeds the feature maps to ContextAEInception2.
`InceptionFrontend` is that fetch on the MI355X: it lays the layer table below out as an op list (slim.conv2d = conv +
folded batch norm + ReLU; pools; tf.concat = adjacent channel slices of one buffer), hands it to libctxtrans
(ctx_cnn_*), folds a checkpoint's batch-norm statistics into filters and biases, and runs frames through it.

Layer table: nets/inception_v3.py:93-416 at depth_multiplier 1 (the only way the reference calls it).  Variable names
follow the reference's scopes, including the three irregular ones (Mixed_5c/Branch_1 'Conv2d_0b_1x1' and
'Conv_1_0c_5x5', :170-173; the 3x3 stride-2 convs named '..._1a_1x1' in Mixed_6a, :218-227), so a TF checkpoint
converted to .npz loads by name.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import CnnBuf, CnnOp, CtxError

BN_EPS = 0.001   # nets/inception_utils.py:34 (slim.batch_norm: center=True, scale=False)

# ("conv", scope, cout, (kh, kw), stride, padding) | ("max",) | ("avg",) | ("fork", [chain, chain]) = concat of two chains
V, S = "VALID", "SAME"


def _c(scope, cout, k, stride=1, padding=S):
    return ("conv", scope, cout, k, stride, padding)


STEM = [("Conv2d_1a_3x3", _c("Conv2d_1a_3x3", 32, (3, 3), 2, V)), ("Conv2d_2a_3x3", _c("Conv2d_2a_3x3", 32, (3, 3), 1, V)),
        ("Conv2d_2b_3x3", _c("Conv2d_2b_3x3", 64, (3, 3))), ("MaxPool_3a_3x3", ("max",)),
        ("Conv2d_3b_1x1", _c("Conv2d_3b_1x1", 80, (1, 1), 1, V)), ("Conv2d_4a_3x3", _c("Conv2d_4a_3x3", 192, (3, 3), 1, V)),
        ("MaxPool_5a_3x3", ("max",))]


def _block35(pool_c, b1):
    return [[_c("Conv2d_0a_1x1", 64, (1, 1))],
            [_c(b1[0], 48, (1, 1)), _c(b1[1], 64, (5, 5))],
            [_c("Conv2d_0a_1x1", 64, (1, 1)), _c("Conv2d_0b_3x3", 96, (3, 3)), _c("Conv2d_0c_3x3", 96, (3, 3))],
            [("avg",), _c("Conv2d_0b_1x1", pool_c, (1, 1))]]


def _block17(c):
    return [[_c("Conv2d_0a_1x1", 192, (1, 1))],
            [_c("Conv2d_0a_1x1", c, (1, 1)), _c("Conv2d_0b_1x7", c, (1, 7)), _c("Conv2d_0c_7x1", 192, (7, 1))],
            [_c("Conv2d_0a_1x1", c, (1, 1)), _c("Conv2d_0b_7x1", c, (7, 1)), _c("Conv2d_0c_1x7", c, (1, 7)),
             _c("Conv2d_0d_7x1", c, (7, 1)), _c("Conv2d_0e_1x7", 192, (1, 7))],
            [("avg",), _c("Conv2d_0b_1x1", 192, (1, 1))]]


def _block8(b1b):
    return [[_c("Conv2d_0a_1x1", 320, (1, 1))],
            [_c("Conv2d_0a_1x1", 384, (1, 1)), ("fork", [[_c("Conv2d_0b_1x3", 384, (1, 3))], [_c(b1b, 384, (3, 1))]])],
            [_c("Conv2d_0a_1x1", 448, (1, 1)), _c("Conv2d_0b_3x3", 384, (3, 3)),
             ("fork", [[_c("Conv2d_0c_1x3", 384, (1, 3))], [_c("Conv2d_0d_3x1", 384, (3, 1))]])],
            [("avg",), _c("Conv2d_0b_1x1", 192, (1, 1))]]


BLOCKS = OrderedDict([
    ("Mixed_5b", _block35(32, ("Conv2d_0a_1x1", "Conv2d_0b_5x5"))),
    ("Mixed_5c", _block35(64, ("Conv2d_0b_1x1", "Conv_1_0c_5x5"))),
    ("Mixed_5d", _block35(64, ("Conv2d_0a_1x1", "Conv2d_0b_5x5"))),
    ("Mixed_6a", [[_c("Conv2d_1a_1x1", 384, (3, 3), 2, V)],
                  [_c("Conv2d_0a_1x1", 64, (1, 1)), _c("Conv2d_0b_3x3", 96, (3, 3)), _c("Conv2d_1a_1x1", 96, (3, 3), 2, V)],
                  [("max",)]]),
    ("Mixed_6b", _block17(128)), ("Mixed_6c", _block17(160)), ("Mixed_6d", _block17(160)), ("Mixed_6e", _block17(192)),
    ("Mixed_7a", [[_c("Conv2d_0a_1x1", 192, (1, 1)), _c("Conv2d_1a_3x3", 320, (3, 3), 2, V)],
                  [_c("Conv2d_0a_1x1", 192, (1, 1)), _c("Conv2d_0b_1x7", 192, (1, 7)), _c("Conv2d_0c_7x1", 192, (7, 1)),
                   _c("Conv2d_1a_3x3", 192, (3, 3), 2, V)],
                  [("max",)]]),
    ("Mixed_7b", _block8("Conv2d_0b_3x1")), ("Mixed_7c", _block8("Conv2d_0c_3x1")),
])


def _pad32(c):
    return (c + 31) // 32 * 32


class _Layout:
    """Turns the table into buffers + ops.  A 'tensor' is (buffer id, h, w, real channels[, first channel inside the buffer]).

    merge_heads: the 1x1 stride-1 convolutions that open several branches of a block all read the block's input; they are emitted
    as ONE GEMM with their filters side by side along cout (Mixed_5b-5d 64+48+64, Mixed_6b-6e 192+c+c, Mixed_7a 192+192,
    Mixed_7b/7c 320+384+448 -- nets/inception_v3.py:140-213, 236-364, 368-416): Branch_0's columns land in the block's concat
    buffer, the others in channel slots of one temporary buffer that the branches' next convolutions read as slices.  The
    variables keep their names, shapes and creation order; only their place in the weight blob changes (conv['ld'], ['col0'])."""

    def __init__(self, H, W, merge_heads=True):
        self.merge_heads = merge_heads
        self.bufs, self.ops, self.convs, self.endpoints = [], [], [], OrderedDict()
        self.woff = 0
        self.lane = 0                   # branch index inside a block: independent branches may overlap on the device
        x = self.new_buf(H, W, 3)
        for name, step in STEM:
            x = self.apply(x, step, "InceptionV3/")
            self.endpoints[name] = x
        for name, branches in BLOCKS.items():
            x = self.block(x, branches, f"InceptionV3/{name}/")
            self.endpoints[name] = x
        self.out = x

    def new_buf(self, h, w, c):
        self.bufs.append((h, w, _pad32(c)))
        return (len(self.bufs) - 1, h, w, c)

    @staticmethod
    def out_grid(x, k, stride, padding):
        h, w = x[1], x[2]
        if padding == S:
            return -(-h // stride), -(-w // stride)
        return (h - k[0]) // stride + 1, (w - k[1]) // stride + 1

    def step_shape(self, x, step):
        if step[0] == "conv":
            return self.out_grid(x, step[3], step[4], step[5]) + (step[2],)
        if step[0] == "max":
            return self.out_grid(x, (3, 3), 2, V) + (x[3],)
        if step[0] == "avg":
            return (x[1], x[2], x[3])
        parts = [self.chain_shape(x, ch) for ch in step[1]]
        return parts[0][:2] + (sum(p[2] for p in parts),)

    def chain_shape(self, x, chain):
        for st in chain:
            h, w, c = self.step_shape(x, st)
            x = (None, h, w, c)
        return x[1:]

    def apply(self, x, step, prefix, dst=None):
        """Emits `step` reading tensor x; writes into dst = (buffer id, channel offset) or a fresh buffer."""
        h, w, c = self.step_shape(x, step)
        if step[0] == "fork":
            if dst is None:
                dst = (self.new_buf(h, w, c)[0], 0)
            off = dst[1]
            for chain in step[1]:
                self.chain(x, chain, prefix, (dst[0], off))
                off += self.chain_shape(x, chain)[2]
            return (dst[0], h, w, c)
        if dst is None:
            dst = (self.new_buf(h, w, c)[0], 0)
        sliced = len(x) > 4                                   # a channel slot of a merged-heads buffer
        src_c = _pad32(x[3]) if sliced else self.bufs[x[0]][2]
        if step[0] == "conv":
            _, scope, cout, k, stride, padding = step
            nw = k[0] * k[1] * src_c * cout
            op = dict(kind=_lib.CTX_CNN_CONV, src=x[0], dst=dst[0], dst_ch0=dst[1], kh=k[0], kw=k[1], stride=stride, same=int(padding == S),
                      cout=cout, w_off=self.woff, b_off=self.woff + nw, lane=self.lane,
                      src_ch0=x[4] if sliced else 0, src_c=src_c if sliced else 0, nconvs=1)
            self.convs.append(dict(scope=prefix + scope, k=k, cin=x[3], cin_pad=src_c, cout=cout, w_off=self.woff, b_off=self.woff + nw, grid=(h, w)))
            self.woff += (nw + cout + 3) // 4 * 4
        else:
            if sliced:
                raise ValueError("pools read whole buffers")
            if x[3] != src_c:
                raise ValueError("pooling a channel-padded tensor into a slice would copy the padding")
            op = dict(kind=_lib.CTX_CNN_MAXPOOL if step[0] == "max" else _lib.CTX_CNN_AVGPOOL, src=x[0], dst=dst[0], dst_ch0=dst[1],
                      kh=3, kw=3, stride=2 if step[0] == "max" else 1, same=int(step[0] == "avg"), cout=0, w_off=0, b_off=0, lane=self.lane)
        self.ops.append(op)
        return (dst[0], h, w, c)

    def chain(self, x, chain, prefix, dst, start=0):
        for i, st in enumerate(chain):
            if i < start:
                continue
            x = self.apply(x, st, prefix, dst if i == len(chain) - 1 else None)
        return x

    def block(self, x, branches, prefix):
        shapes = [self.chain_shape(x, br) for br in branches]
        h, w = shapes[0][:2]
        out = self.new_buf(h, w, sum(s[2] for s in shapes))
        offs = [sum(s[2] for s in shapes[:bi]) for bi in range(len(branches))]
        # ---- sibling 1x1 heads on the block input -> one GEMM
        heads = [bi for bi, br in enumerate(branches) if br[0][0] == "conv" and br[0][3] == (1, 1) and br[0][4] == 1]
        head_out, head_conv = {}, {}
        if self.merge_heads and len(heads) >= 2:
            src_c = self.bufs[x[0]][2]
            direct = [bi for bi in heads if len(branches[bi]) == 1][:1]                    # lands in the concat buffer itself (Branch_0)
            slotted = [bi for bi in heads if bi not in direct]
            cols, col = {}, 0
            for bi in direct:
                cols[bi] = col
                col += branches[bi][0][2]
            nsplit = col
            slot0 = col
            for bi in slotted:
                cols[bi] = col
                col += _pad32(branches[bi][0][2])
            ntot = col
            tmp = self.new_buf(x[1], x[2], ntot - slot0) if slotted else None
            nw = src_c * ntot
            op = dict(kind=_lib.CTX_CNN_CONV, src=x[0], kh=1, kw=1, stride=1, same=1, cout=ntot, w_off=self.woff, b_off=self.woff + nw, lane=0,
                      src_ch0=0, src_c=0, nconvs=len(heads), scopes=[f"{prefix}Branch_{bi}/" + branches[bi][0][1] for bi in heads])
            if direct and slotted:
                op.update(dst=out[0], dst_ch0=offs[direct[0]], nsplit=nsplit, dst2=tmp[0], dst2_ch0=0)
            elif slotted:
                op.update(dst=tmp[0], dst_ch0=0)
            else:
                op.update(dst=out[0], dst_ch0=offs[direct[0]])
            self.ops.append(op)
            for bi in heads:
                cout = branches[bi][0][2]
                head_conv[bi] = dict(scope=f"{prefix}Branch_{bi}/" + branches[bi][0][1], k=(1, 1), cin=x[3], cin_pad=src_c, cout=cout,
                                     w_off=self.woff, b_off=self.woff + nw, ld=ntot, col0=cols[bi], grid=(x[1], x[2]))
                head_out[bi] = (out[0], x[1], x[2], cout) if bi in direct else (tmp[0], x[1], x[2], cout, cols[bi] - slot0)
            self.woff += (nw + ntot + 3) // 4 * 4
        for bi, (br, sh) in enumerate(zip(branches, shapes)):
            self.lane = bi % 4
            if bi in head_out:
                self.convs.append(head_conv[bi])                       # the variable inventory keeps the reference's creation order
                self.chain(head_out[bi], br, f"{prefix}Branch_{bi}/", (out[0], offs[bi]), start=1)
            else:
                self.chain(x, br, f"{prefix}Branch_{bi}/", (out[0], offs[bi]))
        self.lane = 0
        return out


class InceptionFrontend:
    """frames -> Mixed_7c feature maps on one MI355X.  `max_images` bounds one device pass (larger batches are chunked)."""

    def __init__(self, H=125, W=125, max_images=75, device=0, precision="f32", stream=None, merge_heads=None):
        self._lib = _lib.load()
        self.H, self.W = H, W
        if merge_heads is None:
            import os
            merge_heads = os.environ.get("CTX_CNN_MERGE", "1") != "0"
        lay = _Layout(H, W, merge_heads)
        # the output buffer must be the last one for the C side: re-number so that it is
        order = [i for i in range(len(lay.bufs)) if i != lay.out[0]] + [lay.out[0]]
        remap = {old: new for new, old in enumerate(order)}
        self._bufs = [lay.bufs[i] for i in order]
        self._ops = [dict(op, src=remap[op["src"]], dst=remap[op["dst"]], dst2=remap[op["dst2"]] if op.get("nsplit") else 0) for op in lay.ops]
        self.convs, self.weight_floats = lay.convs, lay.woff
        self.endpoints = OrderedDict((k, (remap[v[0]],) + v[1:]) for k, v in lay.endpoints.items())
        self.out_shape = lay.out[1:]                       # (h, w, 2048)
        self.max_images = max_images
        bufs = (CnnBuf * len(self._bufs))(*[CnnBuf(*b) for b in self._bufs])
        ops = (CnnOp * len(self._ops))(*[CnnOp(o["kind"], o["src"], o["dst"], o["dst_ch0"], o["kh"], o["kw"], o["stride"], o["same"], o["cout"], o["lane"],
                                               o["w_off"], o["b_off"], o.get("src_ch0", 0), o.get("src_c", 0), o.get("nsplit", 0), o.get("dst2", 0),
                                               o.get("dst2_ch0", 0), 0) for o in self._ops])
        from .translator import Translator
        self._h = ctypes.c_void_p()
        rc = self._lib.ctx_cnn_create(bufs, len(self._bufs), ops, len(self._ops), self.weight_floats, max_images,
                                      Translator.PRECISIONS[precision], device, ctypes.c_void_p(stream or 0), ctypes.byref(self._h))
        if rc != _lib.CTX_OK:
            msg = self._lib.ctx_cnn_last_error(None)
            self._h = ctypes.c_void_p()
            raise CtxError(rc, msg.decode() if msg else "")

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ctx_cnn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        if rc != _lib.CTX_OK:
            msg = self._lib.ctx_cnn_last_error(self._h)
            raise CtxError(rc, msg.decode() if msg else "")

    # ------------------------------------------------------------------ variables
    def variable_specs(self):
        """[(tf_variable_name, shape)] in graph order: what the restorer of train_script.py:109-110 restores."""
        out = []
        for c in self.convs:
            out += [(c["scope"] + "/weights", c["k"] + (c["cin"], c["cout"]))]
            out += [(c["scope"] + "/BatchNorm/" + n, (c["cout"],)) for n in ("beta", "moving_mean", "moving_variance")]
        return out

    def set_variables(self, tree):
        """tree: {tf_variable_name: array}.  Folds the batch norm (inference form) into filter and bias and uploads."""
        blob = np.zeros(self.weight_floats, np.float32)
        for c in self.convs:
            w = np.asarray(tree[c["scope"] + "/weights"], np.float64)
            beta, mean, var = (np.asarray(tree[c["scope"] + "/BatchNorm/" + n], np.float64) for n in ("beta", "moving_mean", "moving_variance"))
            if w.shape != c["k"] + (c["cin"], c["cout"]):
                raise ValueError(f"{c['scope']}/weights: expected {c['k'] + (c['cin'], c['cout'])}, got {w.shape}")
            scale = 1.0 / np.sqrt(var + BN_EPS)
            if "ld" in c:                                     # a 1x1 head inside a merged filter [cin_pad][ld]: columns col0 .. col0 + cout
                m = blob[c["w_off"]:c["w_off"] + c["cin_pad"] * c["ld"]].reshape(c["cin_pad"], c["ld"])
                m[:c["cin"], c["col0"]:c["col0"] + c["cout"]] = (w * scale)[0, 0]
                blob[c["b_off"] + c["col0"]:c["b_off"] + c["col0"] + c["cout"]] = beta - mean * scale
                continue
            wp = np.zeros(c["k"] + (c["cin_pad"], c["cout"]))
            wp[:, :, :c["cin"], :] = w * scale
            blob[c["w_off"]:c["w_off"] + wp.size] = wp.reshape(-1)
            blob[c["b_off"]:c["b_off"] + c["cout"]] = beta - mean * scale
        self._ck(self._lib.ctx_cnn_set_weights(self._h, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), blob.size))

    def load(self, path):
        with np.load(path) as z:
            self.set_variables({k: z[k] for k in z.files})

    def init_synthetic(self, seed=0):
        """Stand-in for the checkpoint the reference restores (absent from its tree): variance-scaling filters
        (inception_utils.py:66) and batch-norm statistics with some spread.  Returns the variables."""
        rng = np.random.default_rng(seed)
        tree = OrderedDict()
        for name, shape in self.variable_specs():
            if name.endswith("weights"):
                tree[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))).astype(np.float32)
            elif name.endswith("moving_variance"):
                tree[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            else:
                tree[name] = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        self.set_variables(tree)
        return tree

    # ------------------------------------------------------------------ the fetch
    def features(self, frames_u8):
        """uint8 frames [n,H,W,3] -> Mixed_7c [n,h,w,2048] (host arrays; PCIe both ways)."""
        fr = np.ascontiguousarray(frames_u8)
        if fr.dtype != np.uint8 or fr.ndim != 4 or fr.shape[1:] != (self.H, self.W, 3):
            raise ValueError(f"frames must be uint8 [n,{self.H},{self.W},3], got {fr.dtype} {fr.shape}")
        out = np.empty((fr.shape[0],) + tuple(self.out_shape), np.float32)
        self._ck(self._lib.ctx_cnn_forward_u8(self._h, fr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), fr.shape[0],
                                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out

    def flops_per_image(self):
        """Algorithmic FLOPs of one image's pass (2 per multiply-add, real channel counts)."""
        return float(sum(2.0 * c["grid"][0] * c["grid"][1] * c["k"][0] * c["k"][1] * c["cin"] * c["cout"] for c in self.convs))

    def features_u8_dev(self, frames_u8):
        """uint8 frames [n,H,W,3] (n <= max_images) -> integer DEVICE address of Mixed_7c [n,h,w,2048].  Asynchronous on the handle's
        stream; the frames array is kept alive by this object until the next call / sync."""
        fr = np.ascontiguousarray(frames_u8)
        if fr.dtype != np.uint8 or fr.ndim != 4 or fr.shape[1:] != (self.H, self.W, 3):
            raise ValueError(f"frames must be uint8 [n,{self.H},{self.W},3], got {fr.dtype} {fr.shape}")
        self._pending = fr                                    # the upload reads it in stream order
        d_out = ctypes.c_void_p()
        self._ck(self._lib.ctx_cnn_forward_u8_dev(self._h, fr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), fr.shape[0], ctypes.byref(d_out)))
        return d_out.value

    def output(self, n):
        """Host copy of the last forward's Mixed_7c maps [n,h,w,2048] (synchronises)."""
        out = np.empty((n,) + tuple(self.out_shape), np.float32)
        self._ck(self._lib.ctx_cnn_read_buffer(self._h, len(self._bufs) - 1, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out

    def features_dev(self, d_frames_f32, n):
        """Device f32 frames [n,H,W,3] in [-1,1] (integer address) -> integer device address of [n,h,w,2048].  Asynchronous."""
        d_out = ctypes.c_void_p()
        self._ck(self._lib.ctx_cnn_forward_dev(self._h, ctypes.c_void_p(d_frames_f32), n, ctypes.byref(d_out)))
        return d_out.value

    def endpoint(self, name, n):
        """Activations of a named end point after the last forward over n images (tests)."""
        bid, h, w, c = self.endpoints[name]
        cp = self._bufs[bid][2]
        out = np.empty((n, h, w, cp), np.float32)
        self._ck(self._lib.ctx_cnn_read_buffer(self._h, bid, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out[..., :c]

    def profile(self, n, iters=5):
        """[(scope or pool kind, kernel shape, output grid, ms)] per op for the n images of the last forward (measurement)."""
        ms = (ctypes.c_float * len(self._ops))()
        self._ck(self._lib.ctx_cnn_profile(self._h, n, iters, ms, len(self._ops)))
        convs = iter([c for c in self.convs if "ld" not in c])
        by_scope = {c["scope"]: c for c in self.convs}
        out = []
        for o, t in zip(self._ops, ms):
            h, w, _ = self._bufs[o["dst"]]
            if o["kind"] == _lib.CTX_CNN_CONV and o.get("nconvs", 1) > 1:                     # merged 1x1 heads of a block
                cs = [by_scope[sc] for sc in o["scopes"]]
                flops = sum(2.0 * n * h * w * c["cin"] * c["cout"] for c in cs)
                out.append((cs[0]["scope"].rsplit("/", 2)[0] + "/{1x1 heads}", f"1x1 s1 SAME {cs[0]['cin']}->" + "+".join(str(c["cout"]) for c in cs),
                            (h, w), float(t), flops))
            elif o["kind"] == _lib.CTX_CNN_CONV:
                c = next(convs)
                flops = 2.0 * n * h * w * c["k"][0] * c["k"][1] * c["cin"] * c["cout"]
                out.append((c["scope"], f"{c['k'][0]}x{c['k'][1]} s{o['stride']} {'SAME' if o['same'] else 'VALID'} {c['cin']}->{c['cout']}", (h, w), float(t), flops))
            else:
                out.append(("maxpool" if o["kind"] == _lib.CTX_CNN_MAXPOOL else "avgpool", "3x3", (h, w), float(t), 0.0))
        return out

    @property
    def stream(self):
        return self._lib.ctx_cnn_stream(self._h)

    def sync(self):
        self._ck(self._lib.ctx_cnn_sync(self._h))
