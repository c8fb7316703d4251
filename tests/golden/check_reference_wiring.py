"""Runs the REFERENCE'S OWN build() Python of the three live model classes (gym/envs/mujoco/arm_shaping.py: ContextSkipNew
:1260-1354, ContextAEReal :1599-1684, ContextAEInception2 :1786-1894), loaded from REFERENCE_ROOT (default /root/reference) at run
time, on the eager stand-in of tests/golden/tf_standin.py, and compares with oracle/ in float64:

    out, out2, input_z, translated_z, loss, simloss, recon1, recon2  and  d loss / d parameter for EVERY parameter   (bar 1e-9)

plus the variable inventory (names, shapes, creation order, which get_variable calls were reuses).  ContextAEReal is also run with the
module-level keep_prob (arm_shaping.py:1476) set to 0.5 and the oracle's dropout masks handed to tf.nn.dropout in the reference's call
order, which checks WHERE the dropout sites sit (:1637-1661).

Build container only (the GPU box has no /root/reference); nothing of the reference is stored.  This pins the WIRING of the oracle to
the reference's code, not TensorFlow's op semantics, and does not lift "parity unpinned" (DESIGN.md section 2).

    python tests/golden/check_reference_wiring.py            # prints the table, exit code 1 on any deviation > 1e-9
"""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import tf_standin  # noqa: E402
from oracle import ctx_oracle as o  # noqa: E402
from oracle import ctx_oracle_incep as ci  # noqa: E402
from oracle import ctx_oracle_real as r  # noqa: E402

BAR = 1e-9
FETCHES = ("out", "out2", "input_z", "translated_z", "loss", "simloss", "recon1", "recon2")


def reference_root():
    return os.environ.get("REFERENCE_ROOT", "/root/reference")


def load_reference_module(root):
    """gym/envs/mujoco/arm_shaping.py as a stand-alone module (the `gym` package itself would pull mujoco_py); must be called inside
    tf_standin.install() -- the module binds `tf` at import (arm_shaping.py:3,10; nets/inception_v3.py:21,25)."""
    # the reference root goes on sys.path only while its module is executed (`from nets import inception_v3`, arm_shaping.py:8) and comes
    # off again: it holds top-level packages (`tests`, `scripts`, ...) that must not shadow this repository's in later imports -- or in
    # the spawn()ed children of later tests, which inherit sys.path
    for m in [k for k in sys.modules if k == "nets" or k.startswith("nets.")]:
        del sys.modules[m]                                          # they hold the `tf` of an earlier install
    spec = importlib.util.spec_from_file_location("ref_arm_shaping", os.path.join(root, "gym", "envs", "mujoco", "arm_shaping.py"))
    mod = importlib.util.module_from_spec(spec)
    import warnings
    saved = list(sys.path)
    sys.path.insert(0, root)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                         # scipy.misc deprecation
            spec.loader.exec_module(mod)
    finally:
        sys.path[:] = saved
        for m in [k for k in sys.modules if k == "nets" or k.startswith("nets.")]:
            del sys.modules[m]
    return mod


def rand_params(specs, seed, stddev):
    """Every parameter random (biases too: a zero bias would hide a bias wired to the wrong layer)."""
    rng = np.random.default_rng(seed)
    return {n: rng.standard_normal(s) * stddev for n, s in specs}


def run_reference(make_model, p, image, keep_prob=None, masks=None):
    """image: float64 [3, B, H, W, C].  Returns (fetches, gradients by variable name, get_variable log, dropout call count)."""
    import torch
    with tf_standin.install(p) as st:
        mod = load_reference_module(reference_root())
        if keep_prob is not None:
            mod.keep_prob = keep_prob                               # the module global of arm_shaping.py:1476
            st.dropout_masks = iter(masks)
        model = make_model(mod)
        with contextlib.redirect_stdout(io.StringIO()):             # build() prints shapes
            model.build(tf_standin.placeholder(image))
        loss = model.loss.t
        names = list(st.vars)
        grads = torch.autograd.grad(loss, [st.vars[n] for n in names], allow_unused=True)
        got = {k: np.asarray(getattr(model, k).numpy()) for k in FETCHES}
        g = {n: (None if x is None else x.detach().numpy()) for n, x in zip(names, grads)}
        return got, g, list(st.get_variable_calls), st.dropout_calls


def compare(tag, got, g, res, og, specs, log):
    rows, worst = [], 0.0
    for k in FETCHES:
        a, b = np.asarray(got[k], np.float64), np.asarray(res[k], np.float64)
        assert a.shape == b.shape, (tag, k, a.shape, b.shape)
        dev = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
        rows.append((k, dev))
        worst = max(worst, dev)
    created = [n for n, new in log if new]
    assert created == [n for n, _ in specs] or sorted(created) == sorted(n for n, _ in specs), \
        f"{tag}: variables created by the reference {created} != the oracle's inventory"
    for n, _ in specs:
        assert g[n] is not None, f"{tag}: the reference's loss does not depend on {n}"
        dev = float(np.abs(g[n] - og[n]).max() / (np.abs(og[n]).max() + 1e-300))
        rows.append(("d loss / d " + n, dev))
        worst = max(worst, dev)
    return rows, worst, created


def case_skipnew(H, W, d, B, seed):
    cfg = o.SkipNewConfig(H=H, W=W, df_dim=d, gf_dim=d, featsize=1024)          # featsize is hard-coded at arm_shaping.py:1277
    specs = o.param_specs(cfg)
    p = rand_params(specs, seed, 0.05)
    rng = np.random.default_rng(seed + 1)
    img = rng.uniform(-1, 1, (3, B, H, W, 3))
    got, g, log, _ = run_reference(lambda m: m.ContextSkipNew(gf_dim=d, df_dim=d), p, img)
    res, c = o.forward(p, img[0], img[1], img[2], cfg)                          # slots: 0 src, 1 ctx, 2 tgt (:1278-1280)
    return compare(f"ContextSkipNew {H}x{W} d={d} B={B}", got, g, res, o.backward(p, c, cfg), specs, log) + (log,)


def case_real(H, W, B, seed, keep_prob=None):
    cfg = r.RealConfig(H=H, W=W, C=3, featsize=100)
    specs = r.param_specs(cfg)
    p = rand_params(specs, seed, 0.05)
    rng = np.random.default_rng(seed + 1)
    img = rng.uniform(-1, 1, (3, B, H, W, 3))
    drop = masks = None
    if keep_prob is not None:
        drop = r.drop_masks(cfg, B, keep_prob, seed=77, step=1)
        h3, w3 = cfg.sizes[3]
        rows = {"tgt": slice(0, B), "src": slice(B, 2 * B), "ctx": slice(2 * B, 3 * B)}       # the oracle's encoder row order
        masks = []
        for who in ("src", "tgt", "ctx"):                                        # encode(srcimg), encode(tgtimg), encode(tgtctx): :1642-1647
            masks += [drop[1][rows[who]], drop[2][rows[who]]]                    # :1637 reshape(h3) ; :1638 h4
        masks += [drop[3], drop[4]]                                              # :1649 concat ; :1650 trans_h0
        for sl in (slice(0, B), slice(B, 2 * B)):                                # decode(trans_z), decode(tgtimg_z): :1674-1677
            masks += [drop[5][sl], drop[6][sl].reshape(B, h3, w3, r.NF[3])]      # :1660 z ; :1661 reshape(z_)
    got, g, log, ncalls = run_reference(lambda m: m.ContextAEReal(), p, img, keep_prob, masks)
    assert ncalls == 12, f"ContextAEReal: {ncalls} tf.nn.dropout calls, 12 expected (6 in the three encoders, 2 translate, 4 decoder)"
    res, c = r.forward(p, img[0], img[1], img[2], cfg, drop=drop)
    tag = f"ContextAEReal {H}x{W} B={B}" + (f" keep_prob={keep_prob}" if keep_prob else "")
    return compare(tag, got, g, res, r.backward(p, c, cfg), specs, log) + (log,)


def case_incep2(H, W, C, strides, kernels, filters, B, seed):
    cfg = ci.Incep2Config(H=H, W=W, C=C, featsize=1024, strides=strides, kernels=kernels, filters=filters)
    specs = ci.param_specs(cfg)
    p = rand_params(specs, seed, 0.05)
    rng = np.random.default_rng(seed + 1)
    img = rng.uniform(-1, 1, (3, B, H, W, C))
    got, g, log, _ = run_reference(lambda m: m.ContextAEInception2(list(strides), list(kernels), list(filters)), p, img)
    res, c = ci.forward(p, img[0], img[1], img[2], cfg)
    return compare(f"ContextAEInception2 {H}x{W}x{C} s={strides} k={kernels} f={filters} B={B}", got, g, res,
                   ci.backward(p, c, cfg), specs, log) + (log,)


CASES = {
    "skipnew_32x32_d8_b3": lambda: case_skipnew(32, 32, 8, 3, 11),
    "skipnew_16x48_d4_b2": lambda: case_skipnew(16, 48, 4, 2, 12),              # non-square: H and W are not swapped anywhere
    "skipnew_64x64_d16_b2": lambda: case_skipnew(64, 64, 16, 2, 13),            # the production image size
    "real_36x64_b3": lambda: case_real(36, 64, 3, 21),                           # the reference's ContextAEReal size
    "real_36x64_b2_keep0.5": lambda: case_real(36, 64, 2, 22, keep_prob=0.5),    # dropout sites
    "incep2_2x2x32_s1212_k3333_b3": lambda: case_incep2(2, 2, 32, (1, 2, 1, 2), (3, 3, 3, 3), (16, 16, 8, 8), 3, 31),   # config 4's strides/kernels
    "incep2_8x4x8_s2121_k5331_b2": lambda: case_incep2(8, 4, 8, (2, 1, 2, 1), (5, 3, 3, 1), (8, 8, 4, 4), 2, 32),
}


def main():
    if not os.path.isdir(reference_root()):
        print("no reference tree at", reference_root(), "- nothing checked")
        return 2
    bad = 0
    for name, fn in CASES.items():
        rows, worst, created, log = fn()
        reuses = sum(1 for _, new in log if not new)
        print(f"{name:34s} worst deviation {worst:.2e}   {len(created)} variables created, {reuses} get_variable reuses   "
              f"{'OK' if worst <= BAR else 'DIFFERS'}")
        for k, dev in rows:
            if dev > BAR or "-v" in sys.argv:
                print(f"    {k:44s} {dev:.2e}")
        bad += worst > BAR
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
