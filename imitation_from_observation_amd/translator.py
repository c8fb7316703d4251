"""Python host of the translator: a thin ctypes layer over libctxtrans.so.

`Translator.translate(obs_src, obs_tgt0) -> (pred_frame, feat)` is the call the reference's reward
hook makes as `sess.run([model.translated_z, model.out], {image: [src, [ctx]*B, [ctx]*B]})`
(rllab/sampler/base.py:216-218); `encode` is `sess.run([model.input_z, image_trans], ...)`
(base.py:234-235); `train_step` / `evaluate` are scripts/train_script.py:163 / :176.  All arithmetic
happens in the HIP kernels; this file only moves numpy buffers across the C ABI.
"""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import CtxConfig, CtxError  # noqa: F401

_FP = ctypes.POINTER(ctypes.c_float)
_UP = ctypes.POINTER(ctypes.c_uint8)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _u8(a, shape=None):
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError(f"expected uint8 frames, got {a.dtype}")
    a = np.ascontiguousarray(a)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _fp(a):
    return a.ctypes.data_as(_FP)


class _ResultPool:
    """Result arrays of the inference fetches, recycled.  A result above glibc's mmap threshold (32 MB: B >= 683 frames of 64x64x3 f32)
    is a FRESH mapping every time numpy allocates it, whose pages fault in while the device copy lands -- 4.4 of the 8.5 ms of an
    `encode` call at B = 1000 (profiles/archive/round2_b_encode_cliff.txt).  The pool hands out an array it made before as soon as nobody else holds it
    any more (its reference count is back to the pool's own), so a caller that drops or overwrites the previous result -- the reward
    hook's loop -- gets warm pages, and a caller that keeps results gets fresh arrays exactly as before: no aliasing either way.
    Only arrays of >= `min_bytes` are pooled (small ones come from malloc's free lists and are warm anyway)."""

    def __init__(self, min_bytes=1 << 20, keep=4):
        self.min_bytes, self.keep, self._free, self._idle = min_bytes, keep, {}, None

    def get(self, shape, dtype=np.float32):
        import sys
        shape = tuple(int(v) for v in shape)
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if nbytes < self.min_bytes:
            return np.empty(shape, dtype)
        lst = self._free.setdefault((shape, np.dtype(dtype).str), [])
        # What an array nobody else holds counts here -- the list, the loop variable, getrefcount's argument on CPython 3.10; fewer on
        # interpreters that borrow references -- is MEASURED on a probe in the same loop shape, not assumed: anything above it is a
        # caller (or a view of one)
        if self._idle is None:
            probe = [np.empty(1, np.uint8)]
            for a in probe:
                self._idle = sys.getrefcount(a)
        for a in lst:
            if sys.getrefcount(a) == self._idle:
                return a
        a = np.empty(shape, dtype)
        if len(lst) < self.keep:
            lst.append(a)
        return a


def _up(a):
    return a.ctypes.data_as(_UP)


class Translator:
    """One ContextSkipNew model (gym/envs/mujoco/arm_shaping.py:1260-1354) resident on one MI355X.

    Thread-compatible, not thread-safe -- like the single tf.Session it replaces.
    """

    VARIANTS = {"skipnew": _lib.CTX_VARIANT_SKIPNEW, "real": _lib.CTX_VARIANT_REAL, "inception2": _lib.CTX_VARIANT_INCEPTION2}
    PRECISIONS = {"f32": _lib.CTX_PREC_F32, "bf16x3": _lib.CTX_PREC_BF16X3}

    def __init__(self, H=64, W=64, df_dim=64, featsize=1024, max_batch=256, device=0, stream=None, arena_ptr=None,
                 variant="skipnew", precision=None, C=3, strides=None, kernels=None, filters=None, keep_prob=None, ablation_type="None"):
        """variant "skipnew": ContextSkipNew (sampler names push/reach/strike/throw); "real": ContextAEReal
        (names real/sweep; pass H=36, W=64, featsize=100 -- df_dim is ignored, rllab/sampler/base.py:134-137);
        "inception2": ContextAEInception2 on Mixed_7c feature maps (mode 'oursinception'; pass the feature grid as H, W
        and C=2048; float inputs only: translate_f32 / encode_f32 / train_step / evaluate; `strides`, `kernels`, `filters` are the
        constructor lists of ContextAEInception2 (arm_shaping.py:1787-1803), None = the sampler's [1,2,1,2] / [3,3,3,3] /
        [16d,16d,8d,8d]).
        keep_prob (variant "real" only): tf.nn.dropout keep probability of the training graph (arm_shaping.py:1637-1661;
        ablations_code/ablations.py:544 feeds 0.5); None / 1: none.  ablation_type: which terms Adam minimises
        (ablations.py:175-182): "None" = recon1 + recon2 + simloss, "L2" = recon1 + recon2, "L2L3" = recon1, "L1" = recon2 + simloss."""
        precision = precision or os.environ.get("CTX_PRECISION", "f32")     # "f32" (exact) | "bf16x3" (split-bf16 products)
        self._lib = _lib.load()
        self.variant, self.precision = variant, precision
        self.cfg = self.make_config(variant, H, W, C, df_dim, featsize, max_batch, precision, strides, kernels, filters, keep_prob, ablation_type)
        self.keep_prob, self.ablation_type = keep_prob, ablation_type
        self.H, self.W, self.C, self.df_dim, self.featsize, self.max_batch = H, W, C, df_dim, featsize, max_batch
        self.device = device
        self._h = ctypes.c_void_p()
        rc = self._lib.ctx_create_ex(ctypes.byref(self.cfg), device, ctypes.c_void_p(stream or 0),
                                     ctypes.c_void_p(arena_ptr or 0), ctypes.byref(self._h))
        if rc != _lib.CTX_OK:
            msg = self._lib.ctx_last_error(None)
            self._h = ctypes.c_void_p()
            raise CtxError(rc, msg.decode() if msg else "")
        self.n_params = int(self._lib.ctx_param_total(self._h))
        self._pool = _ResultPool()

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ctx_destroy(self._h)
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
        _lib.check(self._lib, self._h, rc)

    @staticmethod
    def make_config(variant, H, W, C, df_dim, featsize, max_batch, precision="f32", strides=None, kernels=None, filters=None,
                    keep_prob=None, ablation_type="None"):
        cfg = CtxConfig(Translator.VARIANTS[variant], H, W, C, df_dim, featsize, max_batch, Translator.PRECISIONS[precision or "f32"])
        for name, lst in (("strides", strides), ("kernels", kernels), ("filters", filters)):
            if lst is not None:
                if variant != "inception2":
                    raise ValueError(f"{name}: only ContextAEInception2 takes them (arm_shaping.py:1787)")
                if len(lst) != 4:
                    raise ValueError(f"{name} must have 4 entries")
                setattr(cfg, name, (ctypes.c_int32 * 4)(*[int(v) for v in lst]))
        if keep_prob is not None:
            cfg.keep_prob = float(keep_prob)
        if ablation_type not in _lib.LOSS_ABLATIONS:
            raise ValueError(f"ablation_type must be one of {sorted(_lib.LOSS_ABLATIONS)}")
        cfg.loss_terms = _lib.LOSS_ABLATIONS[ablation_type]
        return cfg

    @staticmethod
    def param_total(H=64, W=64, df_dim=64, featsize=1024, variant="skipnew", C=3, strides=None, kernels=None, filters=None):
        cfg = Translator.make_config(variant, H, W, C, df_dim, featsize, 1, "f32", strides, kernels, filters)
        return int(_lib.load().ctx_param_total_for(ctypes.byref(cfg)))

    @staticmethod
    def arena_floats(H=64, W=64, df_dim=64, featsize=1024, variant="skipnew", C=3, strides=None, kernels=None, filters=None):
        cfg = Translator.make_config(variant, H, W, C, df_dim, featsize, 1, "f32", strides, kernels, filters)
        return int(_lib.load().ctx_arena_bytes(ctypes.byref(cfg))) // 4

    # ------------------------------------------------------------------ tuning switches (include/ctxtrans.h: per-handle options)
    @staticmethod
    def option_names():
        lib = _lib.load()
        return [lib.ctx_option_name(i).decode() for i in range(lib.ctx_option_count())]

    def get_option(self, name):
        v = ctypes.c_int()
        self._ck(self._lib.ctx_get_option(self._h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def set_option(self, name, value):
        """This handle's switch `name` (e.g. "overlap", "balance", "wconvt"; the environment variable CTX_<NAME> is its default at create)."""
        self._ck(self._lib.ctx_set_option(self._h, name.encode(), int(value)))

    # ------------------------------------------------------------------ parameters (tf.train.Saver)
    def param_info(self):
        """[(tf_variable_name, shape, offset)] in arena order."""
        out = []
        for i in range(self._lib.ctx_param_count(self._h)):
            name, ndim, off = ctypes.c_char_p(), ctypes.c_int(), ctypes.c_int64()
            shape = (ctypes.c_int64 * 4)()
            self._ck(self._lib.ctx_param_info(self._h, i, ctypes.byref(name), ctypes.byref(ndim), shape, ctypes.byref(off)))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim.value)), int(off.value)))
        return out

    def init_params(self, seed=0):
        self._ck(self._lib.ctx_init_params(self._h, ctypes.c_uint64(seed)))

    def set_params_flat(self, flat):
        flat = _f32(flat, (self.n_params,))
        self._ck(self._lib.ctx_set_params(self._h, _fp(flat), flat.size))

    def get_params_flat(self):
        flat = np.empty(self.n_params, np.float32)
        self._ck(self._lib.ctx_get_params(self._h, _fp(flat), flat.size))
        return flat

    def get_grads_flat(self):
        flat = np.empty(self.n_params, np.float32)
        self._ck(self._lib.ctx_get_grads(self._h, _fp(flat), flat.size))
        return flat

    def _split(self, flat):
        return OrderedDict((n, flat[o:o + int(np.prod(s))].reshape(s)) for n, s, o in self.param_info())

    def set_params(self, tree, prefix=""):
        """tree: {tf_variable_name: array}; `prefix` strips e.g. 'contextmodel/' (train_script.py:120)."""
        flat = np.empty(self.n_params, np.float32)
        for n, s, o in self.param_info():
            a = np.asarray(tree[prefix + n], np.float32)
            if tuple(a.shape) != s:
                raise ValueError(f"{n}: expected {s}, got {a.shape}")
            flat[o:o + a.size] = a.reshape(-1)
        self.set_params_flat(flat)

    def get_params(self):
        return self._split(self.get_params_flat())

    def get_grads(self):
        return self._split(self.get_grads_flat())

    def get_adam_state(self):
        m = np.empty(self.n_params, np.float32)
        v = np.empty(self.n_params, np.float32)
        step = ctypes.c_int64()
        self._ck(self._lib.ctx_get_adam_state(self._h, _fp(m), _fp(v), m.size, ctypes.byref(step)))
        return m, v, int(step.value)

    def set_adam_state(self, m, v, step):
        m, v = _f32(m, (self.n_params,)), _f32(v, (self.n_params,))
        self._ck(self._lib.ctx_set_adam_state(self._h, _fp(m), _fp(v), m.size, int(step)))

    @staticmethod
    def checkpoint_file(path):
        """The file a checkpoint path names.  The reference's Saver paths carry no extension
        ('.../model_%d_%.2f_%.2f_%.2f_%.2f', train_script.py:181-182) and np.savez would silently append one: both save()
        and load() therefore use  path  if it ends in '.npz', else  path + '.npz'."""
        path = os.fspath(path)
        return path if path.endswith(".npz") else path + ".npz"

    def save(self, path, with_adam=True, prefix=""):
        """Checkpoint keyed by the TF variable names (the Saver's, train_script.py:181).  Returns the file written."""
        tree = {prefix + k: v for k, v in self.get_params().items()}
        if with_adam:
            m, v, step = self.get_adam_state()
            tree["__adam_m__"], tree["__adam_v__"], tree["__adam_step__"] = m, v, np.int64(step)
        fn = self.checkpoint_file(path)
        with open(fn, "wb") as f:                           # an open handle: numpy does not rename it
            np.savez(f, **tree)
        return fn

    def load(self, path, prefix=""):
        """saver.restore (base.py:144-145): accepts names with or without the 'contextmodel/' scope."""
        fn = self.checkpoint_file(path)
        if not os.path.exists(fn) and os.path.exists(os.fspath(path)):
            fn = os.fspath(path)                            # an .npz stored under an extension-less name
        with np.load(fn) as z:
            keys = set(z.files)
            first = self.param_info()[0][0]
            if prefix + first not in keys and "contextmodel/" + first in keys:
                prefix = "contextmodel/"
            self.set_params({k: z[k] for k in keys if not k.startswith("__")}, prefix=prefix)
            if "__adam_m__" in keys:
                self.set_adam_state(z["__adam_m__"], z["__adam_v__"], int(z["__adam_step__"]))

    # ------------------------------------------------------------------ inference (reward hook)
    def translate(self, obs_src, obs_tgt0):
        """obs_src uint8 [B,H,W,3]; obs_tgt0 uint8 [H,W,3] (first frame of the target context,
        broadcast like `[context] * batch_size`) or [B,H,W,3].  Returns (pred_frame f32 [B,H,W,3],
        feat f32 [B,featsize]) = (model.out, model.translated_z)."""
        src = _u8(obs_src)
        if src.ndim != 4 or src.shape[1:] != (self.H, self.W, 3):
            raise ValueError(f"obs_src must be [B,{self.H},{self.W},3], got {src.shape}")
        B = src.shape[0]
        ctx0 = _u8(obs_tgt0)
        batched = ctx0.ndim == 4
        if tuple(ctx0.shape) != ((B, self.H, self.W, 3) if batched else (self.H, self.W, 3)):
            raise ValueError(f"obs_tgt0 has shape {ctx0.shape}")
        pred = self._pool.get((B, self.H, self.W, 3))
        feat = self._pool.get((B, self.featsize))
        self._ck(self._lib.ctx_translate(self._h, _up(src), _up(ctx0), int(batched), B, _fp(pred), _fp(feat)))
        return pred, feat

    def translate_f32(self, src, ctx0):
        """translate() on float inputs [B,H,W,C]: frames in [-1,1] or, for variant "inception2", feature maps."""
        src = _f32(src)
        B = src.shape[0]
        src = _f32(src, (B, self.H, self.W, self.C))
        ctx0 = _f32(ctx0)
        batched = ctx0.ndim == 4
        ctx0 = _f32(ctx0, (B, self.H, self.W, self.C) if batched else (self.H, self.W, self.C))
        pred = np.empty(src.shape, np.float32)
        feat = np.empty((B, self.featsize), np.float32)
        self._ck(self._lib.ctx_translate_f32(self._h, _fp(src), _fp(ctx0), int(batched), B, _fp(pred), _fp(feat)))
        return pred, feat

    def translate_dev(self, d_src, d_ctx0, B, ctx_batched=False):
        """translate_f32 on DEVICE inputs (integer addresses of f32 [B,H,W,C] and [H,W,C] / [B,H,W,C]); host results."""
        pred = np.empty((B, self.H, self.W, self.C), np.float32)
        feat = np.empty((B, self.featsize), np.float32)
        self._ck(self._lib.ctx_translate_dev(self._h, ctypes.c_void_p(d_src), ctypes.c_void_p(d_ctx0), int(ctx_batched), B, _fp(pred), _fp(feat)))
        return pred, feat

    def encode_dev(self, d_frames, B):
        """encode_f32 on a DEVICE input (integer address of f32 [B,H,W,C]); host result."""
        feat = np.empty((B, self.featsize), np.float32)
        self._ck(self._lib.ctx_encode_dev(self._h, ctypes.c_void_p(d_frames), B, _fp(feat)))
        return feat

    def encode_f32(self, frames):
        """input_z of float inputs [B,H,W,C] (the `conv` encoder)."""
        fr = _f32(frames)
        fr = _f32(fr, (fr.shape[0], self.H, self.W, self.C))
        feat = np.empty((fr.shape[0], self.featsize), np.float32)
        self._ck(self._lib.ctx_encode_f32(self._h, _fp(fr), fr.shape[0], _fp(feat)))
        return feat

    def encode(self, frames, return_frames=True, out=None):
        """frames uint8 [B,H,W,3] -> (input_z f32 [B,featsize], image_trans[0] f32 [B,H,W,3]).

        out = (feat, frames_f32): caller-owned result arrays to fill instead of the translator's.  (Large results come from a pool that
        recycles an array once the caller has let go of it -- _ResultPool: a fresh > 32 MB numpy array is a fresh mmap whose pages
        fault in while the copy lands, 4 ms of a 7.5 ms call at B = 1000.)"""
        fr = _u8(frames)
        if fr.ndim != 4 or fr.shape[1:] != (self.H, self.W, 3):
            raise ValueError(f"frames must be [B,{self.H},{self.W},3], got {fr.shape}")
        B = fr.shape[0]
        if out is not None:
            feat, f32 = out
            if feat.shape != (B, self.featsize) or feat.dtype != np.float32 or not feat.flags.c_contiguous:
                raise ValueError("out[0] must be a C-contiguous float32 [B, featsize] array")
            if return_frames and (f32 is None or f32.shape != fr.shape or f32.dtype != np.float32 or not f32.flags.c_contiguous):
                raise ValueError("out[1] must be a C-contiguous float32 array of the frames' shape")
        else:
            feat = self._pool.get((B, self.featsize))
            f32 = self._pool.get(fr.shape) if return_frames else None
        self._ck(self._lib.ctx_encode(self._h, _up(fr), B, _fp(feat), _fp(f32) if return_frames else None))
        return feat, f32

    # ------------------------------------------------------------------ reward hook on the device (base.py:232-249)
    ABLATIONS = {"None": 0, "nofeat": 1, "noimage": 2}

    def reward_set_cache(self, vp, means, imgs):
        """Keep the demo cache of viewpoint vp (means [bs, featsize], imgs [bs,H,W,3]; base.py:221-222) on the device."""
        means = _f32(means)
        bs = means.shape[0]
        means, imgs = _f32(means, (bs, self.featsize)), _f32(imgs, (bs, self.H, self.W, 3))
        self._ck(self._lib.ctx_reward_set_cache(self._h, int(vp), _fp(means), _fp(imgs), bs))
        self._reward_bs = bs

    def reward_costs(self, vp, frames, scale, ablation_type="None"):
        """frames uint8 [npaths*bs,H,W,3] -> costs f32 [npaths, bs] of base.py:243-249, computed next to the encoder's output."""
        fr = _u8(frames)
        bs = self._reward_bs
        if fr.ndim != 4 or fr.shape[1:] != (self.H, self.W, 3) or fr.shape[0] % bs:
            raise ValueError(f"frames must be [npaths*{bs},{self.H},{self.W},3], got {fr.shape}")
        npaths = fr.shape[0] // bs
        costs = np.empty((npaths, bs), np.float32)
        self._ck(self._lib.ctx_reward_costs(self._h, int(vp), _up(fr), npaths, float(scale), self.ABLATIONS[ablation_type], _fp(costs)))
        return costs

    # ------------------------------------------------------------------ training
    def _triple(self, src, ctx, tgt):
        src = _f32(src)
        shp = (src.shape[0], self.H, self.W, self.C)
        return _f32(src, shp), _f32(ctx, shp), _f32(tgt, shp), shp[0]

    def train_step(self, src, ctx, tgt, lr=1e-4):
        """One Adam step on f32 frames in [-1,1]; returns dict(loss, simloss, recon1, recon2)."""
        src, ctx, tgt, B = self._triple(src, ctx, tgt)
        sc = np.empty(4, np.float32)
        self._ck(self._lib.ctx_train_step(self._h, _fp(src), _fp(ctx), _fp(tgt), B, float(lr), _fp(sc)))
        return dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))

    def set_dropout_seed(self, seed):
        """Seed of the dropout masks (variant "real" with keep_prob < 1); include/ctxtrans.h: ctx_set_dropout_seed."""
        self._ck(self._lib.ctx_set_dropout_seed(self._h, ctypes.c_uint64(int(seed))))

    def train_step_u8(self, src, ctx, tgt, lr=1e-4):
        src = _u8(src)
        shp = (src.shape[0], self.H, self.W, 3)
        src, ctx, tgt = _u8(src, shp), _u8(ctx, shp), _u8(tgt, shp)
        sc = np.empty(4, np.float32)
        self._ck(self._lib.ctx_train_step_u8(self._h, _up(src), _up(ctx), _up(tgt), shp[0], float(lr), _fp(sc)))
        return dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))

    def load_demos(self, vdata_u8):
        """Keep the demo tensor vdata[T,N,H,W,3] (uint8) resident on the device (train_script.py:59-96 builds it)."""
        v = _u8(vdata_u8)
        if v.ndim != 5 or v.shape[2:] != (self.H, self.W, 3):
            raise ValueError(f"vdata must be [T,N,{self.H},{self.W},3], got {v.shape}")
        self._ck(self._lib.ctx_demos_upload(self._h, _up(v), v.shape[0], v.shape[1]))
        self.demo_shape = v.shape[:2]

    def train_step_sampled(self, choicesrc, choicetgt, lr=1e-4):
        """One train step on the batch the reference samples from `traindata` (train_script.py:153-163):
        choicesrc / choicetgt = np.random.choice(ntrain, batch_size)."""
        cs = np.ascontiguousarray(choicesrc, dtype=np.int32)
        ct = np.ascontiguousarray(choicetgt, dtype=np.int32)
        if cs.shape != ct.shape or cs.ndim != 1:
            raise ValueError("choicesrc / choicetgt must be 1-D and equally long")
        sc = np.empty(4, np.float32)
        ip = ctypes.POINTER(ctypes.c_int32)
        self._ck(self._lib.ctx_train_step_sampled(self._h, cs.ctypes.data_as(ip), ct.ctypes.data_as(ip), cs.size, float(lr), _fp(sc)))
        return dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))

    def eval_sampled(self, choicesrc, choicetgt, outputs=True):
        """Forward + losses on the batch sampled from the resident demo tensor (the trainer's validation fetch,
        train_script.py:169-176); no update."""
        cs = np.ascontiguousarray(choicesrc, dtype=np.int32)
        ct = np.ascontiguousarray(choicetgt, dtype=np.int32)
        if cs.shape != ct.shape or cs.ndim != 1:
            raise ValueError("choicesrc / choicetgt must be 1-D and equally long")
        B = cs.size
        sc = np.empty(4, np.float32)
        out = np.empty((B, self.H, self.W, self.C), np.float32) if outputs else None
        out2 = np.empty((B, self.H, self.W, self.C), np.float32) if outputs else None
        ip = ctypes.POINTER(ctypes.c_int32)
        self._ck(self._lib.ctx_eval_sampled(self._h, cs.ctypes.data_as(ip), ct.ctypes.data_as(ip), B, _fp(sc),
                                            _fp(out) if outputs else None, _fp(out2) if outputs else None))
        res = dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))
        if outputs:
            res["out"], res["out2"] = out, out2
        return res

    def last_outputs(self, out=True, out2=False, tgt=False):
        """Host copies of (out, out2, tgt frames) of the last training-mode forward; None where not asked for."""
        B = ctypes.c_int()
        self._ck(self._lib.ctx_last_codes(self._h, None, None, ctypes.byref(B)))
        shp = (B.value, self.H, self.W, self.C)
        arrs = [np.empty(shp, np.float32) if w else None for w in (out, out2, tgt)]
        self._ck(self._lib.ctx_last_outputs(self._h, *[_fp(a) if a is not None else None for a in arrs]))
        return tuple(arrs)

    def evaluate(self, src, ctx, tgt, outputs=True):
        """Forward + losses (train_script.py:176,192-193)."""
        src, ctx, tgt, B = self._triple(src, ctx, tgt)
        sc = np.empty(4, np.float32)
        out = np.empty(src.shape, np.float32) if outputs else None
        out2 = np.empty(src.shape, np.float32) if outputs else None
        self._ck(self._lib.ctx_eval(self._h, _fp(src), _fp(ctx), _fp(tgt), B, _fp(sc),
                                    _fp(out) if outputs else None, _fp(out2) if outputs else None))
        res = dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))
        if outputs:
            res["out"], res["out2"] = out, out2
        return res

    # ------------------------------------------------------------------ device-resident phases
    def dev_forward_backward(self, d_src, d_ctx, d_tgt, B, sim_batch=0):
        """d_*: integer device addresses of f32 [B,H,W,3].  Asynchronous on the handle's stream."""
        self._ck(self._lib.ctx_dev_forward_backward(self._h, ctypes.c_void_p(d_src), ctypes.c_void_p(d_ctx),
                                                     ctypes.c_void_p(d_tgt), B, sim_batch))

    def dev_frames(self, B):
        """(d_src, d_ctx, d_tgt): integer device addresses of the handle's OWN frame slots for a batch of B.  A caller that writes its
        frames there and passes these addresses to dev_forward_backward / dev_train_step / dp_train_step saves the 3 B-frame copy."""
        ps = [ctypes.c_void_p() for _ in range(3)]
        self._ck(self._lib.ctx_dev_frames(self._h, B, *(ctypes.byref(p) for p in ps)))
        return tuple(int(p.value) for p in ps)

    def dev_train_step(self, d_src, d_ctx, d_tgt, B, lr=1e-4):
        """One whole training step (forward + backward + Adam) on device-resident frames, asynchronous on the handle's stream:
        bit-identical to dev_forward_backward + dev_adam, with Adam's slices enqueued beside the remaining backward."""
        self._ck(self._lib.ctx_dev_train_step(self._h, ctypes.c_void_p(d_src), ctypes.c_void_p(d_ctx), ctypes.c_void_p(d_tgt), B, float(lr)))

    def set_grad_bucket_callback(self, fn):
        """fn(first, count) is called from inside dev_forward_backward when gradients [first, first+count) of the gradient
        arena (translate/*, deconv/*) are final in stream order; None clears it (data-parallel overlap, dp.py)."""
        if fn is None:
            self._bucket_cb = None
            self._ck(self._lib.ctx_set_grad_bucket_callback(self._h, None, None))
            return
        self._bucket_cb = _lib.BUCKET_FN(lambda user, bucket, first, count: fn(int(first), int(count)))   # keep the thunk alive
        self._ck(self._lib.ctx_set_grad_bucket_callback(self._h, ctypes.cast(self._bucket_cb, ctypes.c_void_p), None))

    def dev_forward(self, d_src, d_ctx, d_tgt, B):
        self._ck(self._lib.ctx_dev_forward(self._h, ctypes.c_void_p(d_src), ctypes.c_void_p(d_ctx), ctypes.c_void_p(d_tgt), B))

    def dev_adam(self, lr=1e-4):
        self._ck(self._lib.ctx_dev_adam(self._h, float(lr)))

    def dev_scalars(self):
        sc = np.empty(4, np.float32)
        self._ck(self._lib.ctx_dev_scalars(self._h, _fp(sc)))
        return dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))

    def sync(self):
        self._ck(self._lib.ctx_sync(self._h))

    # ------------------------------------------------------------------ data parallel over RCCL, behind the C ABI
    @staticmethod
    def dp_unique_id():
        """The rendezvous blob rank 0 makes (an ncclUniqueId, 128 bytes); ship it to the other ranks, then dp_init everywhere."""
        lib = _lib.load()
        buf = (ctypes.c_uint8 * _lib.CTX_DP_UNIQUE_ID_BYTES)()
        rc = lib.ctx_dp_unique_id(buf)
        if rc != _lib.CTX_OK:
            msg = lib.ctx_last_error(None)
            raise CtxError(rc, msg.decode() if msg else "")
        return bytes(buf)

    def dp_init(self, unique_id, rank, world):
        """Collective: RCCL communicator on this handle's device; rank 0's parameters and Adam slots are broadcast."""
        if len(unique_id) != _lib.CTX_DP_UNIQUE_ID_BYTES:
            raise ValueError(f"unique_id must be {_lib.CTX_DP_UNIQUE_ID_BYTES} bytes")
        buf = (ctypes.c_uint8 * _lib.CTX_DP_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._ck(self._lib.ctx_dp_init(self._h, buf, int(rank), int(world)))

    def dp_allreduce_grads(self):
        """In-place SUM all-reduce of the gradient arena, stream-ordered after the backward pass (asynchronous)."""
        self._ck(self._lib.ctx_dp_allreduce_grads(self._h))

    def dp_train_step(self, d_src, d_ctx, d_tgt, B, lr=1e-4, scalars=False):
        """One data-parallel step on this rank's shard (device addresses of f32 [B,H,W,3]): forward, backward with the simloss
        mean over the global batch, bucketed all-reduce overlapped with the encoders' backward (translate + decoder, then each encoder's FC slice, the conv filters last), Adam.  scalars=True also
        returns the GLOBAL dict(loss, simloss, recon1, recon2) (one more tiny collective + a sync)."""
        sc = np.empty(4, np.float32) if scalars else None
        self._ck(self._lib.ctx_dp_train_step(self._h, ctypes.c_void_p(d_src), ctypes.c_void_p(d_ctx), ctypes.c_void_p(d_tgt), B,
                                             float(lr), _fp(sc) if scalars else None))
        if scalars:
            return dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))

    def dp_train_step_sampled(self, choicesrc, choicetgt, lr=1e-4, scalars=True):
        """The trainer's step on N GPUs (ctx_dp_train_step_sampled): every rank passes the SAME global index arrays
        (np.random.choice(ntrain, batch_size) twice, train_script.py:154-155) and gathers its own rows of the batch from its resident
        demo tensor (load_demos).  Returns the GLOBAL dict(loss, simloss, recon1, recon2) when scalars."""
        cs = np.ascontiguousarray(choicesrc, dtype=np.int32)
        ct = np.ascontiguousarray(choicetgt, dtype=np.int32)
        if cs.shape != ct.shape or cs.ndim != 1:
            raise ValueError("choicesrc / choicetgt must be 1-D and equally long")
        sc = np.empty(4, np.float32) if scalars else None
        ip = ctypes.POINTER(ctypes.c_int32)
        self._ck(self._lib.ctx_dp_train_step_sampled(self._h, cs.ctypes.data_as(ip), ct.ctypes.data_as(ip), cs.size, float(lr),
                                                     _fp(sc) if scalars else None))
        if scalars:
            return dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))

    def dp_eval_sampled(self, choicesrc, choicetgt, outputs=True):
        """The validation fetch sharded over the ranks (ctx_dp_eval_sampled; collective): GLOBAL scalars, and -- outputs=True -- out /
        out2 of THIS rank's rows [B_global / world, H, W, 3]."""
        cs = np.ascontiguousarray(choicesrc, dtype=np.int32)
        ct = np.ascontiguousarray(choicetgt, dtype=np.int32)
        if cs.shape != ct.shape or cs.ndim != 1:
            raise ValueError("choicesrc / choicetgt must be 1-D and equally long")
        world = max(1, self.dp_world()[1])
        Bl = cs.size // world
        sc = np.empty(4, np.float32)
        out = np.empty((Bl, self.H, self.W, self.C), np.float32) if outputs else None
        out2 = np.empty((Bl, self.H, self.W, self.C), np.float32) if outputs else None
        ip = ctypes.POINTER(ctypes.c_int32)
        self._ck(self._lib.ctx_dp_eval_sampled(self._h, cs.ctypes.data_as(ip), ct.ctypes.data_as(ip), cs.size, _fp(sc),
                                               _fp(out) if outputs else None, _fp(out2) if outputs else None))
        res = dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))
        if outputs:
            res["out"], res["out2"] = out, out2
        return res

    def dp_world(self):
        """(rank, world) of this handle's RCCL group; (0, 0) before dp_init (ctx_dp_world reports world 0 until a group exists)."""
        r, w = ctypes.c_int(0), ctypes.c_int(1)
        self._ck(self._lib.ctx_dp_world(self._h, ctypes.byref(r), ctypes.byref(w)))
        return int(r.value), int(w.value)

    def dp_allreduce_host(self, arr):
        """In-place SUM over the ranks of a contiguous float64 numpy array (synchronous): the reward hook's sharded demo cache."""
        if not (isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.flags.c_contiguous):
            raise ValueError("dp_allreduce_host wants a C-contiguous float64 array")
        self._ck(self._lib.ctx_dp_allreduce_host_f64(self._h, arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), arr.size))
        return arr

    def dp_scalars(self):
        sc = np.empty(4, np.float32)
        self._ck(self._lib.ctx_dp_scalars(self._h, _fp(sc)))
        return dict(loss=float(sc[0]), simloss=float(sc[1]), recon1=float(sc[2]), recon2=float(sc[3]))

    @property
    def stream_ptr(self):
        return self._lib.ctx_stream(self._h) or 0

    @property
    def grads_ptr(self):
        return self._lib.ctx_dev_grads(self._h)

    @property
    def scalars_ptr(self):
        return self._lib.ctx_dev_scalar_buf(self._h)

    def profile_step(self, d_src, d_ctx, d_tgt, B, lr=1e-4, iters=5):
        """Per-launch-group timing of a full train step (HIP events on the handle's stream).
        Returns [dict(name, kernel, flops, ms, useful_flops)]: flops counts every tap of a SAME-padded layer, useful_flops only the
        products that meet data (flops * ctx_prof_entry.useful_frac)."""
        ents = (_lib.CtxProfEntry * 256)()
        n = ctypes.c_int()
        self._ck(self._lib.ctx_profile_step(self._h, ctypes.c_void_p(d_src), ctypes.c_void_p(d_ctx), ctypes.c_void_p(d_tgt),
                                            B, float(lr), iters, ents, 256, ctypes.byref(n)))
        return [dict(name=e.name.decode(), kernel=e.kernel.decode(), flops=e.flops, ms=e.ms, useful_flops=e.flops * e.useful_frac)
                for e in ents[: n.value]]

    @staticmethod
    def kernel_table(entries):
        """Groups profile_step entries by kernel: {kernel: dict(ms, flops, useful_flops, launches)}, ms-descending."""
        tab = {}
        for e in entries:
            t = tab.setdefault(e["kernel"], dict(ms=0.0, flops=0.0, useful_flops=0.0, launches=0))
            t["ms"] += e["ms"]
            t["flops"] += e["flops"]
            t["useful_flops"] += e.get("useful_flops", e["flops"])
            t["launches"] += 1
        return dict(sorted(tab.items(), key=lambda kv: -kv[1]["ms"]))

    def last_codes(self):
        """(input_z, translated_z) [B, featsize] of the last training-mode forward (evaluate / train_step): the fetches
        `model.input_z` / `model.translated_z` next to the losses (arm_shaping.py:1298, :1312), row padding removed."""
        B = ctypes.c_int()
        self._ck(self._lib.ctx_last_codes(self._h, None, None, ctypes.byref(B)))
        iz = np.empty((B.value, self.featsize), np.float32)
        tz = np.empty((B.value, self.featsize), np.float32)
        self._ck(self._lib.ctx_last_codes(self._h, _fp(iz), _fp(tz), ctypes.byref(B)))
        return iz, tz

    def debug_read(self, name, n):
        out = np.empty(int(n), np.float32)
        self._ck(self._lib.ctx_debug_read(self._h, name.encode(), _fp(out), out.size))
        return out
