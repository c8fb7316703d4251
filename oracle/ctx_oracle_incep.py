"""CPU oracle for ContextAEInception2  --  TEST INFRASTRUCTURE ONLY (same rules and the same "parity unpinned"
status as oracle/ctx_oracle.py: the reference cannot run here and holds no golden vectors for this path).

ContextAEInception2 (gym/envs/mujoco/arm_shaping.py:1786-1894) is the translator of mode 'oursinception': it runs on
Inception-v3 `Mixed_7c` FEATURE MAPS [3, B, h, w, 2048] (rllab/sampler/base.py:121-132, scripts/train_script.py:98-114),
built as ContextAEInception2(strides=[1,2,1,2], kernels=[3,3,3,3], filters=[1024,1024,512,512]) (base.py:126).
Differences from ContextSkipNew:
  * per-layer stride / kernel / filter count (:1801-1803); SAME padding for k = 3 is (1,1) at stride 1 and (0,1) at
    stride 2 on an even grid; a 1x1 grid under stride 2 stays 1x1 with pad (1,1) -- only the centre tap touches data;
  * lrelu on hz_lin in BOTH encoder scopes (:1812);
  * out = decode(.) + tgtctx, out2 likewise (:1890-1891): the decoder predicts a residual on the context features;
  * featsize 1024 hard-coded (:1797).
conv / conv-transpose here take the kernel size from the filter's shape and the SAME rule from ctx_oracle.same_pad.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass

import numpy as np

from .ctx_oracle import linear, lrelu, lrelu_grad, same_pad


@dataclass(frozen=True)
class Incep2Config:
    H: int = 2                      # Mixed_7c grid: 2x2 for 125x125 frames, 8x8 for 299x299
    W: int = 2
    C: int = 2048
    featsize: int = 1024            # :1797
    strides: tuple = (1, 2, 1, 2)   # base.py:126
    kernels: tuple = (3, 3, 3, 3)
    filters: tuple = (1024, 1024, 512, 512)

    @property
    def sizes(self):
        """grid after encoder layer k (tgtctx_h0..h3 shapes, :1833-1836)."""
        out, h, w = [], self.H, self.W
        for s in self.strides:
            h, w = -(-h // s), -(-w // s)
            out.append((h, w))
        return out


def param_specs(cfg: Incep2Config):
    F, (f1, f2, f3, f4), (k1, k2, k3, k4) = cfg.featsize, cfg.filters, cfg.kernels
    h3, w3 = cfg.sizes[3]
    specs = []
    for scope in ("conv_context", "conv"):                       # creation order :1816-1823
        cin = cfg.C
        for k, (f, ks) in enumerate(zip(cfg.filters, cfg.kernels)):
            specs += [(f"{scope}/h{k}_conv/w", (ks, ks, cin, f)), (f"{scope}/h{k}_conv/biases", (f,))]
            cin = f
        specs += [(f"{scope}/h4_lin/Matrix", (f4 * h3 * w3, F)), (f"{scope}/h4_lin/bias", (F,)),
                  (f"{scope}/hz_lin/Matrix", (F, F)), (f"{scope}/hz_lin/bias", (F,))]
    specs += [("translate/trans_h0/Matrix", (2 * F, F)), ("translate/trans_h0/bias", (F,)),
              ("translate/trans_z/Matrix", (F, F)), ("translate/trans_z/bias", (F,)),
              ("deconv/d_h0_lin/Matrix", (F, f4 * h3 * w3)), ("deconv/d_h0_lin/bias", (f4 * h3 * w3,)),
              ("deconv/d_h1/w", (k4, k4, f3, 2 * f4)), ("deconv/d_h1/biases", (f3,)),      # :1841-1843, filter [k,k,out,in]
              ("deconv/d_h2/w", (k3, k3, f2, 2 * f3)), ("deconv/d_h2/biases", (f2,)),
              ("deconv/d_h3/w", (k2, k2, f1, 2 * f2)), ("deconv/d_h3/biases", (f1,)),
              ("deconv/d_h4/w", (k1, k1, cfg.C, 2 * f1)), ("deconv/d_h4/biases", (cfg.C,))]
    return specs


def param_count(cfg):
    return int(sum(int(np.prod(s)) for _, s in param_specs(cfg)))


def init_params(cfg, seed, dtype=np.float64, stddev=0.02):
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shape in param_specs(cfg):
        if name.endswith("bias") or name.endswith("biases"):
            p[name] = np.zeros(shape, dtype)
        else:
            p[name] = (rng.standard_normal(shape) * stddev).astype(dtype)
    return p


def flatten(tree, cfg, dtype=None):
    return np.concatenate([np.asarray(tree[n]).reshape(-1) for n, _ in param_specs(cfg)]).astype(
        dtype or next(iter(tree.values())).dtype)


# ----------------------------------------------------------------------------- ops with the kernel size of the filter
def _windows(x, k, s):
    """x [N,H,W,C] -> SAME-padded k x k windows [N,Ho,Wo,k,k,C] (a view), pad_before (pt, pl), padded shape."""
    N, H, W, C = x.shape
    Ho, pt, pb = same_pad(H, k, s)
    Wo, pl, pr = same_pad(W, k, s)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    sN, sH, sW, sC = xp.strides
    win = np.lib.stride_tricks.as_strided(xp, (N, Ho, Wo, k, k, C), (sN, s * sH, s * sW, sH, sW, sC), writeable=False)
    return win, (pt, pl), xp.shape


def conv2d(x, w, b, s):
    """arm_shaping.py:21-32 with k_h = k_w = w.shape[0], d_h = d_w = s."""
    k = w.shape[0]
    win, _, _ = _windows(x, k, s)
    N, Ho, Wo = win.shape[:3]
    return (win.reshape(N * Ho * Wo, -1) @ w.reshape(-1, w.shape[-1]) + b).reshape(N, Ho, Wo, w.shape[-1])


def conv2d_bwd(x, w, dy, s, need_dx=True):
    k = w.shape[0]
    win, (pt, pl), pshape = _windows(x, k, s)
    N, Ho, Wo = win.shape[:3]
    dy2 = dy.reshape(-1, w.shape[-1])
    dw = (win.reshape(N * Ho * Wo, -1).T @ dy2).reshape(w.shape)
    dx = None
    if need_dx:
        dcols = (dy2 @ w.reshape(-1, w.shape[-1]).T).reshape(N, Ho, Wo, k, k, x.shape[-1])
        dxp = np.zeros(pshape, dy.dtype)
        for ky in range(k):
            for kx in range(k):
                dxp[:, ky:ky + s * Ho:s, kx:kx + s * Wo:s, :] += dcols[:, :, :, ky, kx, :]
        dx = dxp[:, pt:pt + x.shape[1], pl:pl + x.shape[2], :]
    return dx, dw, dy2.sum(0)


def deconv2d(x, w, b, out_hw, s):
    """arm_shaping.py:62-85: conv2d_transpose(x, w[k,k,out,in], output_shape, strides s), SAME: the input gradient of
    the stride-s SAME conv whose input grid is out_hw:  out[n, s*i+ky-pt, s*j+kx-pl, c] += x[n,i,j,:] . w[ky,kx,c,:]."""
    k = w.shape[0]
    N, h, wd, Cin = x.shape
    Ho, Wo = out_hw
    ho, pt, pb = same_pad(Ho, k, s)
    wo, pl, pr = same_pad(Wo, k, s)
    assert (ho, wo) == (h, wd), "output_shape inconsistent with input under SAME/stride"
    full = np.zeros((N, Ho + pt + pb, Wo + pl + pr, w.shape[2]), x.dtype)
    x2 = x.reshape(-1, Cin)
    for ky in range(k):
        for kx in range(k):
            full[:, ky:ky + s * h:s, kx:kx + s * wd:s, :] += (x2 @ w[ky, kx].T).reshape(N, h, wd, -1)
    return full[:, pt:pt + Ho, pl:pl + Wo, :] + b


def deconv2d_bwd(x, w, dy, s):
    k = w.shape[0]
    N, h, wd, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    _, pt, pb = same_pad(Ho, k, s)
    _, pl, pr = same_pad(Wo, k, s)
    dfull = np.pad(dy, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dx = np.zeros((N * h * wd, Cin), dy.dtype)
    dw = np.zeros(w.shape, dy.dtype)
    x2 = x.reshape(-1, Cin)
    for ky in range(k):
        for kx in range(k):
            sl = dfull[:, ky:ky + s * h:s, kx:kx + s * wd:s, :].reshape(-1, Cout)
            dx += sl @ w[ky, kx]
            dw[ky, kx] = sl.T @ x2
    return dx.reshape(x.shape), dw, dy.reshape(-1, Cout).sum(0)


# ----------------------------------------------------------------------------- the graph
def _encode(p, scope, img, cfg):
    """:1805-1813."""
    acts, h = [], img
    for k in range(4):
        h = lrelu(conv2d(h, p[f"{scope}/h{k}_conv/w"], p[f"{scope}/h{k}_conv/biases"], cfg.strides[k]))
        acts.append(h)
    h4 = lrelu(linear(h.reshape(h.shape[0], -1), p[f"{scope}/h4_lin/Matrix"], p[f"{scope}/h4_lin/bias"]))
    z = lrelu(linear(h4, p[f"{scope}/hz_lin/Matrix"], p[f"{scope}/hz_lin/bias"]))
    return acts + [h4, z]


def _decode(p, cfg, z, skips):
    """:1838-1855: deconvs use (stride, kernel) s4/k4, s3/k3, s2/k2, s1/k1 and produce the grids of h2, h1, h0, input."""
    h3, w3 = cfg.sizes[3]
    z_ = lrelu(linear(z, p["deconv/d_h0_lin/Matrix"], p["deconv/d_h0_lin/bias"]))
    h = z_.reshape(-1, h3, w3, cfg.filters[3])
    hs, cats = [z_], []
    out_sizes = [cfg.sizes[2], cfg.sizes[1], cfg.sizes[0], (cfg.H, cfg.W)]
    for k in range(1, 5):
        cat = np.concatenate([h, skips[4 - k]], axis=3)
        cats.append(cat)
        h = deconv2d(cat, p[f"deconv/d_h{k}/w"], p[f"deconv/d_h{k}/biases"], out_sizes[k - 1], cfg.strides[4 - k])
        if k < 4:
            h = lrelu(h)
        hs.append(h)
    return hs, cats


def forward(p, src, ctx, tgt, cfg: Incep2Config, ablation_type="None"):
    """src/ctx/tgt: feature maps [B,h,w,C] (image[0], image[1], image[2], :1798-1800).  ablation_type: the loss switch every model
    class of the ablation script carries (ablations_code/ablations.py:175-182): which terms make up `loss`."""
    c = {"src": src, "ctx": ctx, "tgt": tgt}
    c["e_ctx"] = _encode(p, "conv_context", ctx, cfg)
    c["e_src"] = _encode(p, "conv", src, cfg)
    c["e_tgt"] = _encode(p, "conv", tgt, cfg)
    src_z, ctx_z, tgt_z = c["e_src"][5], c["e_ctx"][5], c["e_tgt"][5]
    c["tcat"] = np.concatenate([src_z, ctx_z], axis=1)
    c["trans_h0"] = lrelu(linear(c["tcat"], p["translate/trans_h0/Matrix"], p["translate/trans_h0/bias"]))
    c["trans_z"] = linear(c["trans_h0"], p["translate/trans_z/Matrix"], p["translate/trans_z/bias"])
    skips = c["e_ctx"][:4]
    c["d1"], c["d1_cats"] = _decode(p, cfg, c["trans_z"], skips)
    c["d2"], c["d2_cats"] = _decode(p, cfg, tgt_z, skips)
    out, out2 = c["d1"][4] + ctx, c["d2"][4] + ctx                                    # :1890-1891
    res = {"input_z": src_z, "translated_z": c["trans_z"], "out": out, "out2": out2,
           "simloss": np.mean((c["trans_z"] - tgt_z) ** 2) * 1e3,                       # :1882
           "recon1": 0.5 * np.sum((tgt - out) ** 2), "recon2": 0.5 * np.sum((tgt - out2) ** 2)}
    from .ctx_oracle import LOSS_ABLATIONS
    res["loss"] = sum(res[t] for t in LOSS_ABLATIONS[ablation_type])                    # :1894 / ablations.py:175-182
    c["ablation_type"] = ablation_type
    return res, c


def backward(p, c, cfg: Incep2Config, sim_batch=None):
    """Gradients of loss w.r.t. the translator's parameters (the Inception front end is frozen, train_script.py:126-128,
    so nothing flows into src / ctx / tgt)."""
    g = OrderedDict((n, None) for n, _ in param_specs(cfg))
    tgt, ctx = c["tgt"], c["ctx"]
    B, F = tgt.shape[0], cfg.featsize
    tgt_z = c["e_tgt"][5]
    from .ctx_oracle import LOSS_ABLATIONS
    terms = LOSS_ABLATIONS[c.get("ablation_type", "None")]
    w1, w2 = float("recon1" in terms), float("recon2" in terms)
    dsim = ("simloss" in terms) * (2e3 / ((sim_batch or B) * F)) * (c["trans_z"] - tgt_z)

    def acc(name, val):
        g[name] = val if g[name] is None else g[name] + val

    def lin_bwd(name, x, dy):
        acc(f"{name}/Matrix", x.T @ dy)
        acc(f"{name}/bias", dy.sum(0))
        return dy @ p[f"{name}/Matrix"].T

    def decode_bwd(hs, cats, dout):
        dskips, dh = [None] * 4, dout
        for k in range(4, 0, -1):
            if k < 4:
                dh = lrelu_grad(hs[k], dh)
            dcat, dw, db = deconv2d_bwd(cats[k - 1], p[f"deconv/d_h{k}/w"], dh, cfg.strides[4 - k])
            acc(f"deconv/d_h{k}/w", dw)
            acc(f"deconv/d_h{k}/biases", db)
            Cd = cats[k - 1].shape[3] // 2
            dskips[4 - k], dh = dcat[..., Cd:], dcat[..., :Cd]
        return lrelu_grad(hs[0], dh.reshape(B, -1)), dskips

    dz1_, dsk1 = decode_bwd(c["d1"], c["d1_cats"], w1 * (c["d1"][4] + ctx - tgt))
    dz2_, dsk2 = decode_bwd(c["d2"], c["d2_cats"], w2 * (c["d2"][4] + ctx - tgt))
    dtrans_z = lin_bwd("deconv/d_h0_lin", c["trans_z"], dz1_) + dsim
    dtgt_z = lin_bwd("deconv/d_h0_lin", tgt_z, dz2_) - dsim
    dth0 = lrelu_grad(c["trans_h0"], lin_bwd("translate/trans_z", c["trans_h0"], dtrans_z))
    dtcat = lin_bwd("translate/trans_h0", c["tcat"], dth0)

    def encode_bwd(scope, img, acts, dz, dskips=None):
        dz = lrelu_grad(acts[5], dz)
        dh4 = lrelu_grad(acts[4], lin_bwd(f"{scope}/hz_lin", acts[4], dz))
        dh = lin_bwd(f"{scope}/h4_lin", acts[3].reshape(B, -1), dh4).reshape(acts[3].shape)
        for k in range(3, -1, -1):
            if dskips is not None:
                dh = dh + dskips[k]
            dh = lrelu_grad(acts[k], dh)
            x = acts[k - 1] if k > 0 else img
            dx, dw, db = conv2d_bwd(x, p[f"{scope}/h{k}_conv/w"], dh, cfg.strides[k], need_dx=(k > 0))
            acc(f"{scope}/h{k}_conv/w", dw)
            acc(f"{scope}/h{k}_conv/biases", db)
            dh = dx

    encode_bwd("conv", c["src"], c["e_src"], dtcat[:, :F])
    encode_bwd("conv", c["tgt"], c["e_tgt"], dtgt_z)
    encode_bwd("conv_context", c["ctx"], c["e_ctx"], dtcat[:, F:], dskips=[a + b for a, b in zip(dsk1, dsk2)])
    return g


def translate(p, src_feat, ctx0_feat, cfg):
    """base.py:216-218 in mode 'oursinception' after the Inception front end: [src, [ctx0]*B, [ctx0]*B] -> (out, translated_z)."""
    ctx = np.broadcast_to(ctx0_feat, src_feat.shape)
    res, _ = forward(p, src_feat, ctx, ctx, cfg)
    return res["out"], res["translated_z"]


def encode(p, feat, cfg):
    """base.py:234-235: input_z of the `conv` encoder."""
    return _encode(p, "conv", feat, cfg)[5]
