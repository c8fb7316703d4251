"""ContextAEInception2 (CTX_VARIANT_INCEPTION2) through the C ABI against oracle/ctx_oracle_incep.py, on synthetic
feature maps (the Inception-v3 front end that would produce them is not built; its weights are absent from the
reference tree anyway -- SURVEY.md 8a row a8)."""
import numpy as np
import pytest

from oracle import ctx_oracle_incep as oi

pytestmark = pytest.mark.gpu


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import Translator
    return Translator


def make(H, W, C, d, F, B, seed=0, stddev=0.05):
    cfg = oi.Incep2Config(H=H, W=W, C=C, featsize=F, filters=(16 * d, 16 * d, 8 * d, 8 * d))
    p = oi.init_params(cfg, 60 + seed, np.float64, stddev=stddev)
    brng = np.random.default_rng(seed + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * stddev
    rng = np.random.default_rng(seed)
    # Mixed_7c activations are post-ReLU: non-negative, sparse-ish
    feats = [np.maximum(rng.standard_normal((B, H, W, C)), 0).astype(np.float32) for _ in range(3)]
    return cfg, p, feats


def test_param_inventory(T):
    cfg = oi.Incep2Config(H=2, W=2, C=64, featsize=64, filters=(64, 64, 32, 32))
    with T(2, 2, df_dim=4, featsize=64, max_batch=1, variant="inception2", C=64) as tr:
        info = tr.param_info()
        assert tr.n_params == oi.param_count(cfg)
    assert [(n, s) for n, s, _ in info] == [(n, tuple(s)) for n, s in oi.param_specs(cfg)]
    # the production shape (base.py:126): 2048-channel maps, filters 1024/1024/512/512
    from imitation_from_observation_amd import Translator
    assert Translator.param_total(2, 2, 64, 1024, variant="inception2", C=2048) == oi.param_count(oi.Incep2Config())


@pytest.mark.parametrize("H,W,C,d,F,B", [(2, 2, 64, 4, 64, 3), (8, 8, 32, 4, 32, 2), (4, 4, 96, 4, 64, 5), (4, 8, 64, 4, 32, 2)])
def test_incep2_forward_backward_matches_oracle(T, H, W, C, d, F, B):
    cfg, p, (src, ctx, tgt) = make(H, W, C, d, F, B)
    res, c = oi.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    g = oi.backward(p, c, cfg)
    with T(H, W, df_dim=d, featsize=F, max_batch=B, variant="inception2", C=C) as tr:
        tr.set_params(p)
        np.testing.assert_array_equal(tr.get_params_flat(), oi.flatten(p, cfg, np.float32))
        ev = tr.evaluate(src, ctx, tgt)
        for k in ("loss", "simloss", "recon1", "recon2"):
            assert abs(ev[k] - res[k]) <= 1e-5 * abs(res[k]) + 1e-6, k
        assert relmax(ev["out"], res["out"]) < 1e-5 and relmax(ev["out2"], res["out2"]) < 1e-5
        sc = tr.train_step(src, ctx, tgt, lr=0.0)
        assert abs(sc["loss"] - res["loss"]) <= 1e-5 * abs(res["loss"])
        gg = tr.get_grads()
        for n in g:
            assert relmax(gg[n], g[n]) < 1e-4, n
        # the reward hook's two fetches on feature maps (base.py:216-218, 234-235 with image_trans = features)
        pred, feat = tr.translate_f32(src, ctx[0])
        opred, ofeat = oi.translate(p, src.astype(np.float64), ctx[0].astype(np.float64), cfg)
        assert relmax(pred, opred) < 1e-5 and relmax(feat, ofeat) < 1e-5
        assert relmax(tr.encode_f32(tgt), oi.encode(p, tgt.astype(np.float64), cfg)) < 1e-5
        # uint8 frames are refused with a message, not mis-read
        from imitation_from_observation_amd import CtxError
        with pytest.raises(CtxError, match="Inception"):
            tr.translate(np.zeros((B, H, W, 3), np.uint8), np.zeros((H, W, 3), np.uint8))


def test_incep2_adam_and_split_precision(T):
    H, W, C, d, F, B = 2, 2, 128, 8, 128, 4
    cfg, p, (src, ctx, tgt) = make(H, W, C, d, F, B, seed=2)
    res, _ = oi.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    with T(H, W, df_dim=d, featsize=F, max_batch=B, variant="inception2", C=C) as a, \
         T(H, W, df_dim=d, featsize=F, max_batch=B, variant="inception2", C=C, precision="bf16x3") as b:
        a.set_params(p)
        b.set_params(p)
        l0 = a.train_step(src, ctx, tgt, lr=1e-4)["loss"]
        for _ in range(3):
            l1 = a.train_step(src, ctx, tgt, lr=1e-4)["loss"]
        assert abs(l0 - res["loss"]) <= 1e-5 * res["loss"] and l1 < l0
        evb = b.evaluate(src, ctx, tgt)
        assert relmax(evb["out"], res["out"]) < 2e-4 and abs(evb["loss"] - res["loss"]) <= 2e-4 * res["loss"]


@pytest.mark.parametrize("H,W,C,d,F,B", [(2, 2, 64, 4, 64, 64), (4, 4, 32, 4, 32, 32)])
def test_incep2_position_major_batches(T, H, W, C, d, F, B):
    """>= 64 images per launch: the position-major / rectangle-ordered kernels (on a 2x2 grid 4 of the 9 taps of a 3x3
    kernel see data, on the 1x1 grids one of 9 -- the rest is never multiplied)."""
    cfg, p, (src, ctx, tgt) = make(H, W, C, d, F, B, seed=4)
    res, c = oi.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    g = oi.backward(p, c, cfg)
    with T(H, W, df_dim=d, featsize=F, max_batch=B, variant="inception2", C=C) as tr:
        tr.set_params(p)
        ev = tr.evaluate(src, ctx, tgt)
        assert relmax(ev["out"], res["out"]) < 1e-5 and abs(ev["loss"] - res["loss"]) <= 1e-5 * res["loss"]
        tr.train_step(src, ctx, tgt, lr=0.0)
        gg = tr.get_grads()
        for n in g:
            assert relmax(gg[n], g[n]) < 1e-3, n            # lrelu' flips at fp32 zero are not aligned here (B*h*w*C activations)
            assert np.linalg.norm(np.asarray(gg[n], np.float64) - g[n]) <= 2e-3 * np.linalg.norm(g[n]), n


def test_reference_model_interface(T):
    """The drop-in mirror of the reference's class (arm_shaping.py:1786-1894; constructed at rllab/sampler/base.py:126): same
    constructor arguments and fetch names, `sess.run(fetches, {image: [src, ctx, tgt]})` as `model.run(fetches, image)`."""
    from imitation_from_observation_amd.arm_shaping import ContextAEInception2
    H, W, C, d, F, B = 2, 2, 64, 4, 1024, 3
    cfg, p, (src, ctx, tgt) = make(H, W, C, d, F, B, seed=4)
    res, _ = oi.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    with pytest.raises(ValueError, match="strides 1.2"):
        ContextAEInception2(strides=[3, 2, 2, 2], kernels=[3, 3, 3, 3], filters=[64, 64, 32, 32])
    m = ContextAEInception2(strides=[1, 2, 1, 2], kernels=[3, 3, 3, 3], filters=[16 * d, 16 * d, 8 * d, 8 * d])
    m.build((3, B, H, W, C))
    try:
        m.translator.set_params(p)
        image = [src, ctx, tgt]
        tfeat, timg = m.run([m.translated_z, m.out], [src, ctx, ctx])            # base.py:216-218
        assert relmax(timg, res["out"]) < 1e-5
        opred, ofeat = oi.translate(p, src.astype(np.float64), ctx.astype(np.float64), cfg)
        assert relmax(tfeat, ofeat) < 1e-5 and relmax(timg, opred) < 1e-5
        feats, image_trans = m.run([m.input_z, m.image_trans], image)            # base.py:234-235
        assert relmax(feats, oi.encode(p, src.astype(np.float64), cfg)) < 1e-5
        np.testing.assert_array_equal(image_trans[0], src)                        # base.py:132: image_trans IS the fed tensor
        loss, sim, r1, r2, out, out2 = m.run([m.loss, m.simloss, m.recon1, m.recon2, m.out, m.out2], image)   # train_script.py:176
        for got, k in ((loss, "loss"), (sim, "simloss"), (r1, "recon1"), (r2, "recon2")):
            assert abs(got - res[k]) <= 1e-5 * abs(res[k]) + 1e-6, k
        assert relmax(out, res["out"]) < 1e-5 and relmax(out2, res["out2"]) < 1e-5
        _, l0 = m.run([m.optimizer, m.loss], image, learning_rate=1e-4)          # train_script.py:163
        assert abs(l0 - res["loss"]) <= 1e-5 * abs(res["loss"])
        assert m.run(m.loss, image) != l0                                          # the step moved the parameters
    finally:
        m.translator.close()


# ---- the class as the reference defines it: ContextAEInception2(strides, kernels, filters), arm_shaping.py:1786-1803 -------------
def _param_case(H, W, C, F, strides, kernels, filters, B, seed):
    cfg = oi.Incep2Config(H=H, W=W, C=C, featsize=F, strides=tuple(strides), kernels=tuple(kernels), filters=tuple(filters))
    p = oi.init_params(cfg, 80 + seed, np.float64, stddev=0.05)
    brng = np.random.default_rng(seed + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * 0.05
    rng = np.random.default_rng(seed)
    feats = [np.maximum(rng.standard_normal((B, H, W, C)), 0).astype(np.float32) for _ in range(3)]
    return cfg, p, feats


@pytest.mark.parametrize("H,W,C,F,strides,kernels,filters,B", [
    (8, 4, 32, 64, (2, 1, 2, 1), (5, 3, 3, 1), (32, 64, 32, 32), 3),        # other strides, mixed kernel sizes incl. 1x1 and 5x5
    (8, 8, 32, 32, (2, 2, 1, 1), (3, 5, 3, 3), (32, 32, 64, 32), 2),        # two stride-2 layers in a row
    (4, 4, 64, 64, (1, 1, 1, 1), (3, 3, 1, 5), (64, 32, 32, 32), 2),        # no down-sampling at all
    (4, 4, 32, 32, (2, 2, 2, 2), (4, 2, 3, 3), (32, 32, 32, 32), 2),        # even kernels (SAME pads more behind than in front); 1x1 grids under stride 2
    (4, 4, 32, 32, (1, 2, 1, 2), (3, 3, 3, 3), (32, 32, 32, 64), 64),       # position-major launches (>= 64 images)
])
def test_incep2_parametric_strides_kernels_filters(T, H, W, C, F, strides, kernels, filters, B):
    cfg, p, (src, ctx, tgt) = _param_case(H, W, C, F, strides, kernels, filters, B, seed=B)
    res, c = oi.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    g = oi.backward(p, c, cfg)
    with T(H, W, df_dim=4, featsize=F, max_batch=B, variant="inception2", C=C, strides=strides, kernels=kernels, filters=filters) as tr:
        assert [(n, s) for n, s, _ in tr.param_info()] == [(n, tuple(s)) for n, s in oi.param_specs(cfg)]
        tr.set_params(p)
        ev = tr.evaluate(src, ctx, tgt)
        for k in ("loss", "simloss", "recon1", "recon2"):
            assert abs(ev[k] - res[k]) <= 1e-5 * abs(res[k]) + 1e-6, k
        assert relmax(ev["out"], res["out"]) < 1e-5 and relmax(ev["out2"], res["out2"]) < 1e-5
        tr.train_step(src, ctx, tgt, lr=0.0)
        gg = tr.get_grads()
        tol = 1e-4 if B < 64 else 1e-3                                        # (large batches: lrelu' flips at fp32 zero are not aligned)
        for n in g:
            assert relmax(gg[n], g[n]) < tol, n
        pred, feat = tr.translate_f32(src, ctx[0])
        opred, ofeat = oi.translate(p, src.astype(np.float64), ctx[0].astype(np.float64), cfg)
        assert relmax(pred, opred) < 1e-5 and relmax(feat, ofeat) < 1e-5


@pytest.mark.parametrize("tag", ["incep2_4x4x64_f32_b2", "incep2_8x4x32_k5331_s2121_b2"])
def test_incep2_golden_fixtures(T, tag):
    """The committed fixtures the TF recipe (tests/golden/make_tf_fixtures.py) reproduces with the unmodified reference class
    (featsize 1024 hard-coded, :1797): outputs, scalars, gradient digests and the Adam trajectory through the C ABI."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", tag + ".npz"))
    H, W, C, F = (int(v) for v in z["cfg"])
    strides, kernels, filters = ([int(v) for v in z[k]] for k in ("strides", "kernels", "filters"))
    cfg = oi.Incep2Config(H=H, W=W, C=C, featsize=F, strides=tuple(strides), kernels=tuple(kernels), filters=tuple(filters))
    p = oi.init_params(cfg, int(z["pseed"]), np.float64, stddev=float(z["stddev"]))
    brng = np.random.default_rng(int(z["pseed"]) + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * float(z["stddev"])
    B = int(z["B"])
    src, ctx, tgt = z["src_f32"], z["ctx_f32"], z["tgt_f32"]
    with T(H, W, df_dim=4, featsize=F, max_batch=B, variant="inception2", C=C, strides=strides, kernels=kernels, filters=filters) as tr:
        tr.set_params(p)
        ev = tr.evaluate(src, ctx, tgt)
        np.testing.assert_allclose([ev["loss"], ev["simloss"], ev["recon1"], ev["recon2"]], z["scalars"], rtol=2e-5)
        assert relmax(ev["out"], z["out"]) < 1e-5 and relmax(ev["out2"], z["out2"]) < 1e-5
        iz, tz = tr.last_codes()
        assert relmax(iz, z["input_z"]) < 1e-5 and relmax(tz, z["translated_z"]) < 1e-5
        traj = [tr.train_step(src, ctx, tgt, lr=float(z["lr"])) for _ in range(int(z["steps"]))]
        gdig = z["grad_digest"]
        np.testing.assert_allclose([[t["loss"], t["simloss"], t["recon1"], t["recon2"]] for t in traj], z["train_scalars"], rtol=1e-4)
        tr2_grads = None
    with T(H, W, df_dim=4, featsize=F, max_batch=B, variant="inception2", C=C, strides=strides, kernels=kernels, filters=filters) as tr:
        tr.set_params(p)
        tr.train_step(src, ctx, tgt, lr=0.0)
        for i, (n, gv) in enumerate(tr.get_grads().items()):
            a = np.asarray(gv, np.float64).reshape(-1)
            dig = np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())])
            assert abs(dig[2] - gdig[i][2]) <= 1e-4 * gdig[i][2] + 1e-12, n
            assert abs(dig[0] - gdig[i][0]) <= 1e-4 * gdig[i][1] + 1e-12, n
            assert relmax(a[:64], z["grad_head"][i][:min(64, a.size)]) < 1e-3 or np.abs(z["grad_head"][i]).max() < 1e-12, n
