"""CPU oracle for the context-translation hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path (``imitation_from_observation_amd``) must never import,
call or fall back to anything in ``oracle/``.

PARITY UNPINNED.  The reference's arithmetic for this path lives in TensorFlow
1.x (``tf.nn.conv2d``, ``tf.nn.conv2d_transpose``, ``tf.matmul``,
``tf.nn.l2_loss``, ``tf.train.AdamOptimizer``), which is a third-party
dependency that is neither vendored in the reference tree nor installable here
(``environment.yml:25`` pins ``tensorflow=0.10.0rc0`` while the code uses the
TF>=1.0 API).  The reference holds no golden vectors, known-answer tests,
checkpoints or demo tensors for this path, so this
restatement cannot be pinned against reference outputs.  It is instead pinned
(tests/test_oracle_*.py) by
  * hand-computable known-answer tests of TF's published SAME-padding and
    conv2d_transpose index rules,
  * an independent torch-CPU autograd statement of the same graph,
  * finite-difference gradient checks,
  * the structural facts the reference records (parameter count, h3 shape
    ``(100, 3, 3, 512)`` at 48x48 from ``notebooks/reach.ipynb``).

Every function cites the reference ``file:line`` it restates (paths relative to
the reference root).  Layouts are the reference's: activations NHWC, conv
weights HWIO ``[5,5,in,out]``, transposed-conv weights ``[5,5,out,in]``, FC
``Matrix[in,out]``.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass

import numpy as np

LEAK = 0.2  # gym/envs/mujoco/arm_shaping.py:18
KS = 5      # k_h = k_w = 5, arm_shaping.py:21-23 / :62-64
STRIDE = 2  # d_h = d_w = 2


@dataclass(frozen=True)
class SkipNewConfig:
    """Constructor/`build` arguments of ContextSkipNew (arm_shaping.py:1260-1281)."""
    H: int = 64
    W: int = 64
    C: int = 3          # c_dim
    df_dim: int = 64    # encoder base width
    gf_dim: int = 64    # decoder base width
    featsize: int = 1024  # hard-coded at arm_shaping.py:1277; a knob here so tests can shrink it

    def __post_init__(self):
        # decoder sizes are int(s/16) and every skip concat needs matching dims
        # (arm_shaping.py:1314-1330): H and W must be multiples of 16.
        assert self.H % 16 == 0 and self.W % 16 == 0, "ContextSkipNew needs H, W divisible by 16"
        assert self.df_dim == self.gf_dim, "skip concat pairs encoder/decoder widths"


# --------------------------------------------------------------------------- #
# parameter inventory (TF variable names; SURVEY.md section 5 checkpoint row)
# --------------------------------------------------------------------------- #
def param_specs(cfg: SkipNewConfig):
    """Ordered (name, shape) list.  Names are the TF variable-scope names created by
    arm_shaping.py:1282-1343 via conv2d (:24-29), linear (:51-55), deconv2d (:66-79)."""
    d, g, F = cfg.df_dim, cfg.gf_dim, cfg.featsize
    h16, w16 = cfg.H // 16, cfg.W // 16
    specs = []

    def enc(scope):
        cin = cfg.C
        for k, cout in enumerate([d, 2 * d, 4 * d, 8 * d]):
            specs.append((f"{scope}/h{k}_conv/w", (KS, KS, cin, cout)))
            specs.append((f"{scope}/h{k}_conv/biases", (cout,)))
            cin = cout
        specs.append((f"{scope}/h4_lin/Matrix", (h16 * w16 * 8 * d, F)))
        specs.append((f"{scope}/h4_lin/bias", (F,)))
        specs.append((f"{scope}/hz_lin/Matrix", (F, F)))
        specs.append((f"{scope}/hz_lin/bias", (F,)))

    enc("conv_context")
    enc("conv")
    specs.append(("translate/trans_h0/Matrix", (2 * F, F)))
    specs.append(("translate/trans_h0/bias", (F,)))
    specs.append(("translate/trans_z/Matrix", (F, F)))
    specs.append(("translate/trans_z/bias", (F,)))
    specs.append(("deconv/d_h0_lin/Matrix", (F, g * 8 * h16 * w16)))
    specs.append(("deconv/d_h0_lin/bias", (g * 8 * h16 * w16,)))
    # deconv2d filter is [k, k, output_channels, in_channels] (arm_shaping.py:66-67);
    # in_channels = decoder stream + skip (arm_shaping.py:1323-1330)
    specs.append(("deconv/d_h1/w", (KS, KS, 4 * g, 8 * g + 8 * d)))
    specs.append(("deconv/d_h1/biases", (4 * g,)))
    specs.append(("deconv/d_h2/w", (KS, KS, 2 * g, 4 * g + 4 * d)))
    specs.append(("deconv/d_h2/biases", (2 * g,)))
    specs.append(("deconv/d_h3/w", (KS, KS, g, 2 * g + 2 * d)))
    specs.append(("deconv/d_h3/biases", (g,)))
    specs.append(("deconv/d_h4/w", (KS, KS, cfg.C, g + d)))
    specs.append(("deconv/d_h4/biases", (cfg.C,)))
    return specs


def param_count(cfg: SkipNewConfig) -> int:
    return int(sum(int(np.prod(s)) for _, s in param_specs(cfg)))


def init_params(cfg: SkipNewConfig, seed: int, dtype=np.float64, stddev=0.02):
    """Distribution-level restatement of the initialisers: conv w truncated-normal(0.02)
    (arm_shaping.py:25-26), deconv w / FC Matrix normal(0.02) (:67-68, :52-53), biases 0
    (:29, :54-55, :79).  TF truncates at 2 sigma by resampling."""
    rng = np.random.default_rng(seed)
    params = OrderedDict()
    for name, shape in param_specs(cfg):
        if name.endswith("biases") or name.endswith("bias"):
            params[name] = np.zeros(shape, dtype)
        else:
            w = rng.standard_normal(shape)
            if "_conv/w" in name:
                bad = np.abs(w) > 2.0
                while bad.any():
                    w[bad] = rng.standard_normal(int(bad.sum()))
                    bad = np.abs(w) > 2.0
            params[name] = (w * stddev).astype(dtype)
    return params


def flatten(tree, cfg: SkipNewConfig, dtype=None):
    """Concatenate a name->array dict in param_specs order (the C-ABI's flat arena order)."""
    return np.concatenate([np.asarray(tree[n]).reshape(-1) for n, _ in param_specs(cfg)]).astype(
        dtype or next(iter(tree.values())).dtype)


def unflatten(flat, cfg: SkipNewConfig):
    out, off = OrderedDict(), 0
    for name, shape in param_specs(cfg):
        n = int(np.prod(shape))
        out[name] = np.asarray(flat[off:off + n]).reshape(shape)
        off += n
    assert off == len(flat)
    return out


# --------------------------------------------------------------------------- #
# ops  (arm_shaping.py:18-85)
# --------------------------------------------------------------------------- #
def lrelu(x):
    """arm_shaping.py:18-19  tf.maximum(x, leak*x)."""
    return np.maximum(x, LEAK * x)


def lrelu_grad(y, dy):
    """Gradient of tf.maximum(x, 0.2x) expressed on the OUTPUT y (sign(y) == sign(x)).
    TF's MaximumGrad routes the gradient to the first argument where x >= 0.2x, i.e. x >= 0."""
    return np.where(y >= 0, dy, LEAK * dy)


def same_pad(n_in: int, k: int = KS, s: int = STRIDE):
    """TF 'SAME' rule: out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0);
    pad_before = pad_total // 2 (the odd unit goes after).  For k=5, s=2, even `in` this is
    (1, 2) -- asymmetric (SURVEY.md 2.1 rule 1)."""
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2, total - total // 2


def _im2col(x, s=STRIDE):
    N, H, W, C = x.shape
    Ho, pt, pb = same_pad(H, KS, s)
    Wo, pl, pr = same_pad(W, KS, s)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    sN, sH, sW, sC = xp.strides
    win = np.lib.stride_tricks.as_strided(
        xp, (N, Ho, Wo, KS, KS, C), (sN, s * sH, s * sW, sH, sW, sC), writeable=False)
    return win.reshape(N * Ho * Wo, KS * KS * C), (N, Ho, Wo), (pt, pl, xp.shape)


def conv2d(x, w, b, s=STRIDE):
    """arm_shaping.py:21-32: tf.nn.conv2d(x, w[5,5,in,out], strides s, 'SAME') + biases.
    Cross-correlation (no kernel flip), NHWC."""
    cols, (N, Ho, Wo), _ = _im2col(x, s)
    y = cols @ w.reshape(-1, w.shape[-1]) + b
    return y.reshape(N, Ho, Wo, w.shape[-1])


def conv2d_bwd(x, w, dy, s=STRIDE, need_dx=True):
    """Analytic gradients of conv2d: dw, db and (optionally) dx."""
    cols, (N, Ho, Wo), (pt, pl, pshape) = _im2col(x, s)
    Cout = w.shape[-1]
    dy2 = dy.reshape(-1, Cout)
    dw = (cols.T @ dy2).reshape(w.shape)
    db = dy2.sum(0)
    dx = None
    if need_dx:
        C = x.shape[-1]
        dcols = (dy2 @ w.reshape(-1, Cout).T).reshape(N, Ho, Wo, KS, KS, C)
        dxp = np.zeros(pshape, dy.dtype)
        for ky in range(KS):
            for kx in range(KS):
                dxp[:, ky:ky + s * Ho:s, kx:kx + s * Wo:s, :] += dcols[:, :, :, ky, kx, :]
        dx = dxp[:, pt:pt + x.shape[1], pl:pl + x.shape[2], :]
    return dx, dw, db


def deconv2d(x, w, b, out_hw, s=STRIDE):
    """arm_shaping.py:62-85: tf.nn.conv2d_transpose(x, w[5,5,out,in], output_shape, strides s)
    (padding defaults to 'SAME') + biases.  It is the input-gradient of the SAME stride-s
    forward conv whose input has spatial size `out_hw`:
        out[n, s*i + ky - pt, s*j + kx - pl, c] += x[n,i,j,k] * w[ky,kx,c,k]
    with (pt, pl) the SAME pad_before of that forward conv (SURVEY.md 2.1 rule 2)."""
    N, h, wd, Cin = x.shape
    Ho, Wo = out_hw
    ho, pt, pb = same_pad(Ho, KS, s)
    wo, pl, pr = same_pad(Wo, KS, s)
    assert (ho, wo) == (h, wd), "output_shape inconsistent with input under SAME/stride"
    Cout = w.shape[2]
    full = np.zeros((N, Ho + pt + pb, Wo + pl + pr, Cout), x.dtype)
    x2 = x.reshape(-1, Cin)
    for ky in range(KS):
        for kx in range(KS):
            full[:, ky:ky + s * h:s, kx:kx + s * wd:s, :] += (x2 @ w[ky, kx].T).reshape(N, h, wd, Cout)
    return full[:, pt:pt + Ho, pl:pl + Wo, :] + b


def deconv2d_bwd(x, w, dy, s=STRIDE):
    """Analytic gradients of deconv2d: dx, dw, db."""
    N, h, wd, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    _, pt, pb = same_pad(Ho, KS, s)
    _, pl, pr = same_pad(Wo, KS, s)
    dfull = np.pad(dy, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dx = np.zeros((N * h * wd, Cin), dy.dtype)
    dw = np.zeros_like(w, dtype=dy.dtype)
    x2 = x.reshape(-1, Cin)
    for ky in range(KS):
        for kx in range(KS):
            sl = dfull[:, ky:ky + s * h:s, kx:kx + s * wd:s, :].reshape(-1, Cout)
            dx += sl @ w[ky, kx]
            dw[ky, kx] = sl.T @ x2
    return dx.reshape(x.shape), dw, dy.reshape(-1, Cout).sum(0)


def linear(x, W, b):
    """arm_shaping.py:48-59: tf.matmul(x, Matrix) + bias."""
    return x @ W + b


def preprocess_u8(frames_u8):
    """rllab/sampler/base.py:116-119: tf.image.convert_image_dtype(uint8 -> float32) multiplies
    by float32(1/255); then subtract 0.5, multiply 2.0 (all in float32)."""
    x = frames_u8.astype(np.float32) * np.float32(1.0 / 255.0)
    return (x - np.float32(0.5)) * np.float32(2.0)


# --------------------------------------------------------------------------- #
# ContextSkipNew.build  (arm_shaping.py:1272-1354)
# --------------------------------------------------------------------------- #
def _encode(p, scope, img, z_lrelu):
    """arm_shaping.py:1282-1288 (conv_context: hz_lin is linear) and :1290-1307 (conv: lrelu on
    hz_lin).  Returns [h0,h1,h2,h3,h4,z]."""
    acts = []
    h = img
    for k in range(4):
        h = lrelu(conv2d(h, p[f"{scope}/h{k}_conv/w"], p[f"{scope}/h{k}_conv/biases"]))
        acts.append(h)
    flat = h.reshape(h.shape[0], -1)                       # NHWC flatten, :1287
    h4 = lrelu(linear(flat, p[f"{scope}/h4_lin/Matrix"], p[f"{scope}/h4_lin/bias"]))
    z = linear(h4, p[f"{scope}/hz_lin/Matrix"], p[f"{scope}/hz_lin/bias"])
    if z_lrelu:
        z = lrelu(z)
    return acts + [h4, z]


def _decode(p, cfg, z, skips):
    """arm_shaping.py:1314-1330 (and :1334-1343 for the truth pass): d_h0_lin -> reshape
    [-1, H/16, W/16, 8g] -> 4 x deconv2d(concat([decoder, ctx skip], 3)); lrelu on all but d_h4."""
    g = cfg.gf_dim
    H, W = cfg.H, cfg.W
    z_ = lrelu(linear(z, p["deconv/d_h0_lin/Matrix"], p["deconv/d_h0_lin/bias"]))
    h0 = z_.reshape(-1, H // 16, W // 16, 8 * g)
    cats, hs = [], [z_]
    h = h0
    sizes = [(H // 8, W // 8), (H // 4, W // 4), (H // 2, W // 2), (H, W)]
    for k in range(1, 5):
        cat = np.concatenate([h, skips[4 - k]], axis=3)   # [decoder, skip] order, :1323
        cats.append(cat)
        h = deconv2d(cat, p[f"deconv/d_h{k}/w"], p[f"deconv/d_h{k}/biases"], sizes[k - 1])
        if k < 4:
            h = lrelu(h)
        hs.append(h)
    return hs, cats


# ablations_code/ablations.py:175-182 (the same switch in every model class of that file): which terms make up `loss`
LOSS_ABLATIONS = {"None": ("recon1", "recon2", "simloss"), "L2": ("recon1", "recon2"), "L2L3": ("recon1",), "L1": ("recon2", "simloss")}


def forward(p, src, ctx, tgt, cfg: SkipNewConfig, ablation_type="None"):
    """Whole graph of ContextSkipNew.build.  Inputs are float arrays [B,H,W,C] in [-1,1]
    (image[0]=src :1278, image[2]=tgt :1279, image[1]=ctx :1280).  ablation_type: the loss switch of
    ablations_code/ablations.py:175-182 ("None" = the trainer's recon1 + recon2 + simloss)."""
    c = {"src": src, "ctx": ctx, "tgt": tgt}
    c["e_ctx"] = _encode(p, "conv_context", ctx, z_lrelu=False)      # :1282-1288
    c["e_src"] = _encode(p, "conv", src, z_lrelu=True)               # :1290-1298
    c["e_tgt"] = _encode(p, "conv", tgt, z_lrelu=True)               # :1302-1307
    src_z, ctx_z, tgt_z = c["e_src"][5], c["e_ctx"][5], c["e_tgt"][5]
    c["tcat"] = np.concatenate([src_z, ctx_z], axis=1)               # :1310
    c["trans_h0"] = lrelu(linear(c["tcat"], p["translate/trans_h0/Matrix"], p["translate/trans_h0/bias"]))
    c["trans_z"] = linear(c["trans_h0"], p["translate/trans_z/Matrix"], p["translate/trans_z/bias"])  # :1311
    skips = c["e_ctx"][:4]
    c["d1"], c["d1_cats"] = _decode(p, cfg, c["trans_z"], skips)     # :1321-1330
    c["d2"], c["d2_cats"] = _decode(p, cfg, tgt_z, skips)            # :1334-1343
    out, out2 = c["d1"][4], c["d2"][4]
    res = {
        "input_z": src_z,                                            # :1298
        "translated_z": c["trans_z"],                                # :1312
        "out": out, "out2": out2,                                    # :1350-1351
        "simloss": np.mean((c["trans_z"] - tgt_z) ** 2) * 1e3,       # :1345
        "recon1": 0.5 * np.sum((tgt - out) ** 2),                    # :1352  tf.nn.l2_loss
        "recon2": 0.5 * np.sum((tgt - out2) ** 2),                   # :1353
    }
    res["loss"] = sum(res[t] for t in LOSS_ABLATIONS[ablation_type])  # :1354 / ablations.py:175-182
    c["ablation_type"] = ablation_type
    return res, c


def backward(p, c, cfg: SkipNewConfig, sim_batch=None):
    """Analytic gradient of `loss` w.r.t. every parameter -- what TF autodiff computes for
    AdamOptimizer.minimize(test.loss) (scripts/train_script.py:128).  `sim_batch`: batch size in
    the simloss mean's denominator; defaults to the local batch.  A data-parallel shard passes
    the GLOBAL batch so that a SUM all-reduce of shard gradients equals the full-batch gradient
    (SURVEY.md 8e)."""
    g = OrderedDict((n, None) for n, _ in param_specs(cfg))
    tgt = c["tgt"]
    B = tgt.shape[0]
    F = cfg.featsize
    tgt_z = c["e_tgt"][5]
    terms = LOSS_ABLATIONS[c.get("ablation_type", "None")]
    w1, w2 = float("recon1" in terms), float("recon2" in terms)
    dsim = ("simloss" in terms) * (2e3 / ((sim_batch or B) * F)) * (c["trans_z"] - tgt_z)   # d simloss / d trans_z

    def acc(name, val):
        g[name] = val if g[name] is None else g[name] + val

    def decode_bwd(hs, cats, dout):
        dskips = [None] * 4
        dh = dout
        for k in range(4, 0, -1):
            if k < 4:
                dh = lrelu_grad(hs[k], dh)
            dcat, dw, db = deconv2d_bwd(cats[k - 1], p[f"deconv/d_h{k}/w"], dh)
            acc(f"deconv/d_h{k}/w", dw)
            acc(f"deconv/d_h{k}/biases", db)
            Cd = cats[k - 1].shape[3] - c["e_ctx"][4 - k].shape[3]
            dskips[4 - k] = dcat[..., Cd:]
            dh = dcat[..., :Cd]
        dz_ = lrelu_grad(hs[0], dh.reshape(B, -1))
        return dz_, dskips

    def lin_bwd(name, x, dy, bias="bias"):
        acc(f"{name}/Matrix", x.T @ dy)
        acc(f"{name}/{bias}", dy.sum(0))
        return dy @ p[f"{name}/Matrix"].T

    # decoder on translated z (recon1) and on tgt z (recon2)
    dz1_, dsk1 = decode_bwd(c["d1"], c["d1_cats"], w1 * (c["d1"][4] - tgt))
    dz2_, dsk2 = decode_bwd(c["d2"], c["d2_cats"], w2 * (c["d2"][4] - tgt))
    dtrans_z = lin_bwd("deconv/d_h0_lin", c["trans_z"], dz1_) + dsim
    dtgt_z = lin_bwd("deconv/d_h0_lin", tgt_z, dz2_) - dsim          # no stop-gradient on tgtimg_z
    # translate MLP
    dth0 = lrelu_grad(c["trans_h0"], lin_bwd("translate/trans_z", c["trans_h0"], dtrans_z))
    dtcat = lin_bwd("translate/trans_h0", c["tcat"], dth0)
    dsrc_z, dctx_z = dtcat[:, :F], dtcat[:, F:]

    def encode_bwd(scope, img, acts, dz, z_lrelu, dskips=None):
        if z_lrelu:
            dz = lrelu_grad(acts[5], dz)
        dh4 = lrelu_grad(acts[4], lin_bwd(f"{scope}/hz_lin", acts[4], dz))
        dh = lin_bwd(f"{scope}/h4_lin", acts[3].reshape(B, -1), dh4).reshape(acts[3].shape)
        for k in range(3, -1, -1):
            if dskips is not None:
                dh = dh + dskips[k]
            dh = lrelu_grad(acts[k], dh)
            x = acts[k - 1] if k > 0 else img
            dx, dw, db = conv2d_bwd(x, p[f"{scope}/h{k}_conv/w"], dh, need_dx=(k > 0))
            acc(f"{scope}/h{k}_conv/w", dw)
            acc(f"{scope}/h{k}_conv/biases", db)
            dh = dx

    encode_bwd("conv", c["src"], c["e_src"], dsrc_z, True)
    encode_bwd("conv", c["tgt"], c["e_tgt"], dtgt_z, True)
    encode_bwd("conv_context", c["ctx"], c["e_ctx"], dctx_z, False,
               dskips=[a + b for a, b in zip(dsk1, dsk2)])
    return g


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer(lr) defaults (scripts/train_script.py:124-128), TF formulation:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps)  (eps OUTSIDE the
    bias correction; SURVEY.md 2.1 rule 6).  `t` is the 1-based step.  In place."""
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    for n in p:
        m[n] = b1 * m[n] + (1 - b1) * g[n]
        v[n] = b2 * v[n] + (1 - b2) * g[n] * g[n]
        p[n] = p[n] - (lr_t * m[n] / (np.sqrt(v[n]) + eps)).astype(p[n].dtype)


def train_step(p, m, v, t, src, ctx, tgt, lr, cfg):
    """One `sess.run([optimizer, loss, simloss, recon1, recon2])` (train_script.py:163).  Scalars
    are those of the forward pass BEFORE the update."""
    res, c = forward(p, src, ctx, tgt, cfg)
    g = backward(p, c, cfg)
    adam_step(p, g, m, v, t, lr)
    return res, g


# --------------------------------------------------------------------------- #
# inference call sites  (rllab/sampler/base.py)
# --------------------------------------------------------------------------- #
def translate(p, src_u8, ctx0_u8, cfg):
    """base.py:216-218: sess.run([translated_z, out], {image: [src, [context]*B, [context]*B]})
    -> (pred_frame, feat).  ctx0 may be one frame [H,W,3] (broadcast) or [B,H,W,3]."""
    src = preprocess_u8(src_u8)
    ctx0 = preprocess_u8(ctx0_u8)
    if ctx0.ndim == 3:
        ctx0 = np.broadcast_to(ctx0, src.shape)
    res, _ = forward(p, src, ctx0, ctx0, cfg)
    return res["out"], res["translated_z"]


def encode(p, frames_u8, cfg):
    """base.py:234-235: sess.run([input_z, image_trans], {image: [curimgs, [curimgs[0]]*B,
    curimgs]}) -> (feat, preprocessed frames = image_trans[0]).  Only the `conv` encoder on
    slot 0 contributes to input_z."""
    x = preprocess_u8(frames_u8)
    return _encode(p, "conv", x, z_lrelu=True)[5], x


def reward_costs(feats, frames_f32, means, imgs, scale, ablation_type="None"):
    """base.py:243-249 for one viewpoint: cost_j = sum((means_j-feats_j)^2) +
    scale*sum((imgs_j-frames_j)^2).  ('nofeat' / 'noimage' restated with the intended [vp]
    indexing)"""
    cf = np.sum((means - feats) ** 2, axis=1)
    ci = scale * np.sum((imgs - frames_f32) ** 2, axis=(1, 2, 3))
    if ablation_type == "None":
        return cf + ci
    if ablation_type == "nofeat":
        return ci
    if ablation_type == "noimage":
        return cf
    raise ValueError(ablation_type)


def apply_costs(rewards, costs):
    """base.py:256-257: rewards[2j+1] -= costs[j] * j^2."""
    for j in range(len(costs)):
        rewards[2 * j + 1] -= costs[j] * (j ** 2)
    return rewards


# --------------------------------------------------------------------------- #
# workload constants
# --------------------------------------------------------------------------- #
def flops_forward(cfg: SkipNewConfig) -> int:
    """Multiply-add FLOPs (2/MAC) of one forward triple."""
    d, g, F, C = cfg.df_dim, cfg.gf_dim, cfg.featsize, cfg.C
    H, W = cfg.H, cfg.W
    enc, cin, h, w = 0, C, H, W
    for cout in [d, 2 * d, 4 * d, 8 * d]:
        h, w = h // 2, w // 2
        enc += h * w * 25 * cin * cout
        cin = cout
    enc += h * w * 8 * d * F + F * F
    trans = 2 * F * F + F * F
    dec = F * g * 8 * h * w
    for cin_, cout in [(16 * g, 4 * g), (8 * g, 2 * g), (4 * g, g), (2 * g, C)]:
        dec += h * w * 25 * cin_ * cout     # per INPUT pixel: every input pixel meets all 25 taps
        h, w = 2 * h, 2 * w
    return 2 * (3 * enc + trans + 2 * dec)
