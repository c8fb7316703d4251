"""The committed golden vectors (tests/golden/*.npz) against the oracle: guards the fixtures against
RNG / oracle drift.  (The GPU tests compare the HIP path with the same files.)"""
import glob
import os

import numpy as np
import pytest

from oracle import ctx_oracle as o

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "skipnew_*.npz")))


def load_case(path):
    z = np.load(path)
    H, W, C, d, F = (int(v) for v in z["cfg"])
    cfg = o.SkipNewConfig(H=H, W=W, C=C, df_dim=d, gf_dim=d, featsize=F)
    p = o.init_params(cfg, int(z["pseed"]), np.float64, stddev=float(z["stddev"]))
    brng = np.random.default_rng(int(z["pseed"]) + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * float(z["stddev"])
    return z, cfg, p


def digest(a):
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())])


def test_fixtures_exist():
    assert len(GOLD) >= 3


@pytest.mark.parametrize("path", [g for g in GOLD if "d32" in g], ids=os.path.basename)
def test_oracle_reproduces_golden(path):
    z, cfg, p = load_case(path)
    np.testing.assert_allclose(digest(o.flatten(p, cfg)), z["param_digest"], rtol=1e-12)
    src, ctx, tgt = (o.preprocess_u8(z[k]).astype(np.float64) for k in ("src_u8", "ctx_u8", "tgt_u8"))
    res, c = o.forward(p, src, ctx, tgt, cfg)
    for k in ["input_z", "translated_z", "out", "out2"]:
        np.testing.assert_allclose(res[k], z[k], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([res["loss"], res["simloss"], res["recon1"], res["recon2"]], z["scalars"], rtol=1e-12)
    g = o.backward(p, c, cfg)
    for i, (n, _) in enumerate(o.param_specs(cfg)):
        np.testing.assert_allclose(digest(g[n]), z["grad_digest"][i], rtol=1e-9, atol=1e-12)


def test_oracle_f32_is_within_budget_of_golden_f64():
    """The float32 oracle (the CPU baseline) vs the float64 fixture: sets the scale of fp32 noise the
    1e-3 parity budget has to absorb (expect ~1e-6)."""
    z, cfg, p = load_case([g for g in GOLD if "32x32" in g][0])
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    src, ctx, tgt = (o.preprocess_u8(z[k]) for k in ("src_u8", "ctx_u8", "tgt_u8"))
    res, _ = o.forward(p32, src, ctx, tgt, cfg)
    assert np.abs(res["out"] - z["out"]).max() <= 1e-4 * np.abs(z["out"]).max()
    assert abs(res["loss"] - z["scalars"][0]) <= 1e-5 * z["scalars"][0]


# ---- the other model classes' fixtures (tests/golden/make_golden.py extra): guards against RNG / oracle drift; the TF recipe
# (tests/golden/make_tf_fixtures.py) and tests/test_tf_pin.py read the same files
@pytest.mark.parametrize("tag", ["incep2_4x4x64_f32_b2", "incep2_8x4x32_k5331_s2121_b2"])
def test_incep2_oracle_reproduces_golden(tag):
    from oracle import ctx_oracle_incep as ci
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", tag + ".npz"))
    H, W, C, F = (int(v) for v in z["cfg"])
    cfg = ci.Incep2Config(H=H, W=W, C=C, featsize=F, strides=tuple(int(v) for v in z["strides"]),
                          kernels=tuple(int(v) for v in z["kernels"]), filters=tuple(int(v) for v in z["filters"]))
    p = ci.init_params(cfg, int(z["pseed"]), np.float64, stddev=float(z["stddev"]))
    brng = np.random.default_rng(int(z["pseed"]) + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * float(z["stddev"])
    np.testing.assert_allclose(digest(ci.flatten(p, cfg)), z["param_digest"], rtol=1e-12)
    src, ctx, tgt = (z[k].astype(np.float64) for k in ("src_f32", "ctx_f32", "tgt_f32"))
    res, c = ci.forward(p, src, ctx, tgt, cfg)
    for k in ["input_z", "translated_z", "out", "out2"]:
        np.testing.assert_allclose(res[k], z[k], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([res["loss"], res["simloss"], res["recon1"], res["recon2"]], z["scalars"], rtol=1e-12)
    g = ci.backward(p, c, cfg)
    for i, (n, _) in enumerate(ci.param_specs(cfg)):
        np.testing.assert_allclose(digest(g[n]), z["grad_digest"][i], rtol=1e-9, atol=1e-12)


def test_inception_oracle_reproduces_golden():
    from oracle import inception_oracle as io
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "inception_v3_125x125_b2.npz"))
    ep = io.forward(io.init_params(int(z["pseed"])), o.preprocess_u8(z["frames_u8"]).astype(np.float64))
    assert list(ep) == [str(n) for n in z["endpoints"]]
    np.testing.assert_allclose(ep["Mixed_7c"], z["Mixed_7c"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.stack([digest(v) for v in ep.values()]), z["endpoint_digest"], rtol=1e-9)


def test_inception_oracle_reproduces_the_299x299_golden():
    """The front end at the reference's own frame size (nets/inception_v3_test.py:45-54): tests/golden/make_golden.py ref299."""
    from oracle import inception_oracle as io
    from tests.golden import make_golden as mg
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", mg.REF299_FRONT_TAG + ".npz"))
    frames = np.random.default_rng(int(z["fseed"])).integers(0, 256, (2, 299, 299, 3), dtype=np.uint8)
    np.testing.assert_array_equal(digest(frames), z["frames_digest"])
    p = {k: v.astype(np.float64) for k, v in io.init_params(int(z["pseed"]), np.float32).items()}
    ep = io.forward(p, o.preprocess_u8(frames).astype(np.float64))
    assert list(ep) == [str(n) for n in z["endpoints"]]
    assert ep["Mixed_7c"].shape == (2, 8, 8, 2048)
    np.testing.assert_allclose(ep["Mixed_7c"][0], z["Mixed_7c_0"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.stack([digest(v) for v in ep.values()]), z["endpoint_digest"], rtol=1e-9)


def test_incep2_oracle_reproduces_the_reference_size_golden():
    """ContextAEInception2 at 8x8x2048, batch 25, 153 M parameters (run_train_strike_inception.py:39-43): parameters and inputs are
    regenerated from the recorded seeds; forward outputs and scalars of the float64 oracle against the fixture (the gradient digests
    are guarded by the generator run; a backward pass here would double the test's 10 GB)."""
    from oracle import ctx_oracle_incep as ci
    from tests.golden import make_golden as mg
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", mg.REF299_TAG + ".npz"))
    cfg, p32, feats = mg.ref299_case()
    np.testing.assert_allclose(digest(ci.flatten(p32, cfg)), z["param_digest"], rtol=1e-12)
    assert ci.param_count(cfg) == 153_111_040
    p = {k: v.astype(np.float64) for k, v in p32.items()}
    del p32
    res, _ = ci.forward(p, *(x.astype(np.float64) for x in feats), cfg)
    np.testing.assert_allclose([res["loss"], res["simloss"], res["recon1"], res["recon2"]], z["scalars"], rtol=1e-12)
    keep = list(z["keep"])
    for k in ("out", "out2", "translated_z", "input_z"):
        np.testing.assert_allclose(res[k][keep], z[k + "_keep"], rtol=1e-5, atol=1e-6)
