"""CPU oracle for ContextAEReal  --  TEST INFRASTRUCTURE ONLY (same rules and same "parity unpinned"
status as oracle/ctx_oracle.py, whose ops it reuses).

ContextAEReal (gym/envs/mujoco/arm_shaping.py:1599-1684) is the translator the sampler builds for
name in ('real', 'sweep') (rllab/sampler/base.py:134-135), on 36x64 frames
(sandbox/andrew/run_trpo_sweep_ours.py:64).  Differences from ContextSkipNew:
  * ONE encoder (scope "conv") shared by src, tgt and ctx (arm_shaping.py:1642-1647); lrelu on hz_lin for all;
  * filters nf = 32/16/16/8 with strides ns = 1/2/1/2 (:1622-1629), featsize 100 (:1616);
  * decoder deconvs mirror the strides: d_h1 s2, d_h2 s1, d_h3 s2, d_h4 s1 (:1664-1671);
  * tf.nn.dropout(., keep_prob) in front of every linear and on the reshaped d_h0 (:1637-1663) with the
    module-level keep_prob = 1.0 (:1476) -- the identity, and restated as such.  (ablations_code/ablations.py
    trains its copy with keep_prob 0.5; that path needs TF's RNG stream and is out of scope.)
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass

import numpy as np

from .ctx_oracle import KS, conv2d, conv2d_bwd, deconv2d, deconv2d_bwd, linear, lrelu, lrelu_grad

NF = (32, 16, 16, 8)        # nf0..nf3, arm_shaping.py:1622-1625
NS = (1, 2, 1, 2)           # ns0..ns3, :1626-1629


@dataclass(frozen=True)
class RealConfig:
    H: int = 36
    W: int = 64
    C: int = 3
    featsize: int = 100     # :1616

    def __post_init__(self):
        assert self.H % 4 == 0 and self.W % 4 == 0, "two stride-2 layers: H, W divisible by 4"

    @property
    def sizes(self):
        """spatial size after each encoder layer: s0..s3 (arm_shaping.py:1654-1657)."""
        out, h, w = [], self.H, self.W
        for s in NS:
            h, w = h // s, w // s
            out.append((h, w))
        return out


def param_specs(cfg: RealConfig):
    """TF variable names/shapes (scopes of arm_shaping.py:1642, 1648, 1674; conv2d :24-29, linear :51-55,
    deconv2d :66-79 with filter [k, k, out, in])."""
    F = cfg.featsize
    h3, w3 = cfg.sizes[3]
    specs, cin = [], cfg.C
    for k, cout in enumerate(NF):
        specs += [(f"conv/h{k}_conv/w", (KS, KS, cin, cout)), (f"conv/h{k}_conv/biases", (cout,))]
        cin = cout
    specs += [("conv/h4_lin/Matrix", (h3 * w3 * NF[3], F)), ("conv/h4_lin/bias", (F,)),
              ("conv/hz_lin/Matrix", (F, F)), ("conv/hz_lin/bias", (F,)),
              ("translate/trans_h0/Matrix", (2 * F, F)), ("translate/trans_h0/bias", (F,)),
              ("translate/trans_z/Matrix", (F, F)), ("translate/trans_z/bias", (F,)),
              ("deconv/d_h0_lin/Matrix", (F, NF[3] * h3 * w3)), ("deconv/d_h0_lin/bias", (NF[3] * h3 * w3,)),
              ("deconv/d_h1/w", (KS, KS, NF[2], 2 * NF[3])), ("deconv/d_h1/biases", (NF[2],)),
              ("deconv/d_h2/w", (KS, KS, NF[1], 2 * NF[2])), ("deconv/d_h2/biases", (NF[1],)),
              ("deconv/d_h3/w", (KS, KS, NF[0], 2 * NF[1])), ("deconv/d_h3/biases", (NF[0],)),
              ("deconv/d_h4/w", (KS, KS, cfg.C, 2 * NF[0])), ("deconv/d_h4/biases", (cfg.C,))]
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


# --------------------------------------------------------------------------------------------------------------------
# tf.nn.dropout(x, keep_prob) = x * floor(keep_prob + U) / keep_prob at six sites of the TRAINING graph (arm_shaping.py:1637-1661;
# keep_prob is a module-level placeholder_with_default(1.0), :1476; ablations_code/ablations.py:544 feeds 0.5 when it trains this
# class, 1.0 when it validates, :556).  TensorFlow's random stream cannot be reproduced, and need not be: parity for a random op
# means "given the same masks, the same numbers".  The HIP path draws its masks from a counter-based hash of
# (seed, step, site, element index) -- restated here so that the oracle can be run on exactly the masks the device used.
#   site 1  flatten(h3) before h4_lin     rows = the 3B encoder images in the library's order [tgt | src | ctx]
#   site 2  h4 before hz_lin              same rows
#   site 3  concat([src_z, ctx_z])        B rows, 2 F columns
#   site 4  trans_h0                      B rows
#   site 5  z before d_h0_lin             rows = the 2B decoder passes [translated | truth]
#   site 6  reshape(z_) before d_h1       same rows
# Element index = row * (real, unpadded) columns + column.
# --------------------------------------------------------------------------------------------------------------------
def drop_hash(seed, step, site, n):
    """u32 hash of element indices 0..n-1 (lowbias32 finaliser over a Weyl mix); numpy restatement of kernels.hip:drop_hash."""
    M = np.uint64(0xFFFFFFFF)
    x = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(site) * np.uint64(0x85EBCA77) +
         np.uint64(step & 0xFFFFFFFF) * np.uint64(0xC2B2AE3D) + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x27D4EB2F)) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def drop_masks(cfg, B, keep_prob, seed, step, dtype=np.float64):
    """{site: mask / keep_prob, shaped [rows, real columns]} for one training step of batch B."""
    h3, w3 = cfg.sizes[3]
    D0, F = h3 * w3 * NF[3], cfg.featsize
    thr = np.uint64(min(int(np.float32(keep_prob).astype(np.float64) * 4294967296.0), 4294967295))
    shapes = {1: (3 * B, D0), 2: (3 * B, F), 3: (B, 2 * F), 4: (B, F), 5: (2 * B, F), 6: (2 * B, D0)}
    inv = np.float32(1.0) / np.float32(keep_prob)           # the device multiplies by this f32 value
    return {site: (drop_hash(seed, step, site, r * c).astype(np.uint64) < thr).reshape(r, c).astype(dtype) * dtype(inv)
            for site, (r, c) in shapes.items()}


def _encode(p, img, m1=None, m2=None):
    """arm_shaping.py:1633-1640; m1 / m2: dropout factors (mask / keep_prob) of sites 1 and 2 for these images, or None."""
    acts, h = [], img
    for k in range(4):
        h = lrelu(conv2d(h, p[f"conv/h{k}_conv/w"], p[f"conv/h{k}_conv/biases"], s=NS[k]))
        acts.append(h)
    flat = h.reshape(h.shape[0], -1)
    x1 = flat if m1 is None else flat * m1
    h4 = lrelu(linear(x1, p["conv/h4_lin/Matrix"], p["conv/h4_lin/bias"]))
    x2 = h4 if m2 is None else h4 * m2
    z = lrelu(linear(x2, p["conv/hz_lin/Matrix"], p["conv/hz_lin/bias"]))
    return acts + [h4, z, x1, x2]


def _decode(p, cfg, z, skips, m5=None, m6=None):
    """arm_shaping.py:1659-1672: d_h0_lin -> [-1, s_h3, s_w3, nf3]; deconvs with strides ns3, ns2, ns1, ns0
    on concat([decoder, skip_h3 / h2 / h1 / h0], 3).  m5 / m6: dropout factors of sites 5 and 6 for this pass."""
    h3, w3 = cfg.sizes[3]
    zin = z if m5 is None else z * m5
    z_ = lrelu(linear(zin, p["deconv/d_h0_lin/Matrix"], p["deconv/d_h0_lin/bias"]))
    h = (z_ if m6 is None else z_ * m6).reshape(-1, h3, w3, NF[3])
    hs, cats = [z_], [zin]                      # cats[0] = what d_h0_lin was fed; cats[k] = the concat fed to d_hk
    out_sizes = [cfg.sizes[2], cfg.sizes[1], cfg.sizes[0], (cfg.H, cfg.W)]
    for k in range(1, 5):
        cat = np.concatenate([h, skips[4 - k]], axis=3)
        cats.append(cat)
        h = deconv2d(cat, p[f"deconv/d_h{k}/w"], p[f"deconv/d_h{k}/biases"], out_sizes[k - 1], s=NS[4 - k])
        if k < 4:
            h = lrelu(h)
        hs.append(h)
    return hs, cats


def forward(p, src, ctx, tgt, cfg: RealConfig, drop=None, ablation_type="None"):
    """drop: None (keep_prob = 1, the sampler's and the validation graph) or the dict of drop_masks().  ablation_type: the loss
    switch of the ablation script's copies of this class (ContextAEPushReal / ContextAESweep, ablations_code/ablations.py:175-182,
    :477-484): which terms make up `loss`."""
    B = src.shape[0]
    d = drop or {}
    rows = {"tgt": slice(0, B), "src": slice(B, 2 * B), "ctx": slice(2 * B, 3 * B)}            # the library's encoder batch order
    m = lambda site, sl: d[site][sl] if site in d else None
    c = {"src": src, "ctx": ctx, "tgt": tgt, "drop": d}
    c["e_src"] = _encode(p, src, m(1, rows["src"]), m(2, rows["src"]))                         # :1642-1647
    c["e_tgt"] = _encode(p, tgt, m(1, rows["tgt"]), m(2, rows["tgt"]))
    c["e_ctx"] = _encode(p, ctx, m(1, rows["ctx"]), m(2, rows["ctx"]))
    src_z, ctx_z, tgt_z = c["e_src"][5], c["e_ctx"][5], c["e_tgt"][5]
    c["tcat"] = np.concatenate([src_z, ctx_z], axis=1)
    c["tcat_d"] = c["tcat"] * d[3] if 3 in d else c["tcat"]
    c["trans_h0"] = lrelu(linear(c["tcat_d"], p["translate/trans_h0/Matrix"], p["translate/trans_h0/bias"]))
    c["trans_h0_d"] = c["trans_h0"] * d[4] if 4 in d else c["trans_h0"]
    c["trans_z"] = linear(c["trans_h0_d"], p["translate/trans_z/Matrix"], p["translate/trans_z/bias"])
    skips = c["e_ctx"][:4]
    c["d1"], c["d1_cats"] = _decode(p, cfg, c["trans_z"], skips, m(5, slice(0, B)), m(6, slice(0, B)))
    c["d2"], c["d2_cats"] = _decode(p, cfg, tgt_z, skips, m(5, slice(B, 2 * B)), m(6, slice(B, 2 * B)))
    out, out2 = c["d1"][4], c["d2"][4]
    res = {"input_z": src_z, "translated_z": c["trans_z"], "out": out, "out2": out2,
           "simloss": np.mean((c["trans_z"] - tgt_z) ** 2) * 1e3,          # :1676
           "recon1": 0.5 * np.sum((tgt - out) ** 2), "recon2": 0.5 * np.sum((tgt - out2) ** 2)}
    from .ctx_oracle import LOSS_ABLATIONS
    res["loss"] = sum(res[t] for t in LOSS_ABLATIONS[ablation_type])       # :1684 / ablations.py:175-182, :477-484
    c["ablation_type"] = ablation_type
    return res, c


def backward(p, c, cfg: RealConfig, sim_batch=None):
    from .ctx_oracle import LOSS_ABLATIONS
    g = OrderedDict((n, None) for n, _ in param_specs(cfg))
    tgt = c["tgt"]
    B, F = tgt.shape[0], cfg.featsize
    tgt_z = c["e_tgt"][5]
    terms = LOSS_ABLATIONS[c.get("ablation_type", "None")]
    w1, w2 = float("recon1" in terms), float("recon2" in terms)
    dsim = ("simloss" in terms) * (2e3 / ((sim_batch or B) * F)) * (c["trans_z"] - tgt_z)

    def acc(name, val):
        g[name] = val if g[name] is None else g[name] + val

    def lin_bwd(name, x, dy):
        acc(f"{name}/Matrix", x.T @ dy)
        acc(f"{name}/bias", dy.sum(0))
        return dy @ p[f"{name}/Matrix"].T

    d = c.get("drop", {})
    mul = lambda x, site, sl: x * d[site][sl] if site in d else x          # the gradient of x * m is dy * m

    def decode_bwd(hs, cats, dout, sl):
        dskips, dh = [None] * 4, dout
        for k in range(4, 0, -1):
            if k < 4:
                dh = lrelu_grad(hs[k], dh)
            dcat, dw, db = deconv2d_bwd(cats[k], p[f"deconv/d_h{k}/w"], dh, s=NS[4 - k])
            acc(f"deconv/d_h{k}/w", dw)
            acc(f"deconv/d_h{k}/biases", db)
            Cd = cats[k].shape[3] // 2
            dskips[4 - k], dh = dcat[..., Cd:], dcat[..., :Cd]
        return lrelu_grad(hs[0], mul(dh.reshape(B, -1), 6, sl)), dskips

    p1, p2 = slice(0, B), slice(B, 2 * B)
    dz1_, dsk1 = decode_bwd(c["d1"], c["d1_cats"], w1 * (c["d1"][4] - tgt), p1)
    dz2_, dsk2 = decode_bwd(c["d2"], c["d2_cats"], w2 * (c["d2"][4] - tgt), p2)
    dtrans_z = mul(lin_bwd("deconv/d_h0_lin", c["d1_cats"][0], dz1_), 5, p1) + dsim      # simloss sees the un-dropped codes
    dtgt_z = mul(lin_bwd("deconv/d_h0_lin", c["d2_cats"][0], dz2_), 5, p2) - dsim
    dth0 = lrelu_grad(c["trans_h0"], mul(lin_bwd("translate/trans_z", c["trans_h0_d"], dtrans_z), 4, slice(0, B)))
    dtcat = mul(lin_bwd("translate/trans_h0", c["tcat_d"], dth0), 3, slice(0, B))
    rows = {"tgt": slice(0, B), "src": slice(B, 2 * B), "ctx": slice(2 * B, 3 * B)}

    def encode_bwd(img, acts, dz, dskips=None, sl=None):
        dz = lrelu_grad(acts[5], dz)
        dh4 = lrelu_grad(acts[4], mul(lin_bwd("conv/hz_lin", acts[7], dz), 2, sl))
        dh = mul(lin_bwd("conv/h4_lin", acts[6], dh4), 1, sl).reshape(acts[3].shape)
        for k in range(3, -1, -1):
            if dskips is not None:
                dh = dh + dskips[k]
            dh = lrelu_grad(acts[k], dh)
            x = acts[k - 1] if k > 0 else img
            dx, dw, db = conv2d_bwd(x, p[f"conv/h{k}_conv/w"], dh, s=NS[k], need_dx=(k > 0))
            acc(f"conv/h{k}_conv/w", dw)
            acc(f"conv/h{k}_conv/biases", db)
            dh = dx

    encode_bwd(c["src"], c["e_src"], dtcat[:, :F], sl=rows["src"])
    encode_bwd(c["tgt"], c["e_tgt"], dtgt_z, sl=rows["tgt"])
    encode_bwd(c["ctx"], c["e_ctx"], dtcat[:, F:], dskips=[a + b for a, b in zip(dsk1, dsk2)], sl=rows["ctx"])
    return g


def translate(p, src_u8, ctx0_u8, cfg: RealConfig):
    """rllab/sampler/base.py:216-218 on ContextAEReal: feed [src, [ctx0]*B, [ctx0]*B], fetch (out, translated_z)."""
    from .ctx_oracle import preprocess_u8
    src = preprocess_u8(src_u8)
    ctx = np.broadcast_to(preprocess_u8(ctx0_u8), src.shape)
    res, _ = forward(p, src, ctx, ctx, cfg)
    return res["out"], res["translated_z"]


def encode(p, frames_u8, cfg: RealConfig):
    """base.py:234-235: fetch (input_z, image_trans[0])."""
    from .ctx_oracle import preprocess_u8
    x = preprocess_u8(frames_u8)
    return _encode(p, x)[5], x
