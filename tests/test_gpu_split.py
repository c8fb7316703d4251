"""CTX_PREC_BF16X3 (split-bf16 products, f32 accumulation -- csrc/igemm_split.h) through the C ABI against the
float64 oracle.  Budget: north_star's 1e-3 relative; the asserted bounds are 2e-4 (measured: 1e-5 .. 2e-5).

Gradients are compared with the oracle's lrelu' branches aligned to the device's (tests/_align.py): the 1e-5 product
error moves a few more near-zero activations across the kink than exact f32 does, and the two subgradients there
differ by 0.8*dy -- a property of lrelu, not of the arithmetic."""
import glob
import os

import numpy as np
import pytest

from oracle import ctx_oracle as o
from oracle import ctx_oracle_real as r
from tests._align import align_skipnew_cache
from tests.test_gpu_parity import load_golden, make_case, rel_l2, relmax

pytestmark = pytest.mark.gpu
TOL = 2e-4
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "skipnew_*.npz")))


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import Translator
    return lambda *a, **k: Translator(*a, precision="bf16x3", **k)


@pytest.mark.parametrize("H,W,d,F,B", [(32, 32, 32, 128, 4), (16, 48, 32, 128, 3), (16, 16, 32, 32, 1), (32, 32, 64, 256, 5),
                                       (64, 64, 64, 1024, 3)])
def test_split_forward_backward_matches_oracle(T, H, W, d, F, B):
    cfg, p, fr = make_case(H, W, d, F, B, stddev=0.05 if d < 64 or H < 64 else 0.02)
    src, ctx, tgt = (o.preprocess_u8(x) for x in fr)
    res, c = o.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    with T(H, W, d, F, max_batch=B) as tr:
        tr.set_params(p)
        ev = tr.evaluate(src, ctx, tgt)
        for k in ("loss", "simloss", "recon1", "recon2"):
            assert abs(ev[k] - res[k]) <= TOL * abs(res[k]) + 1e-6, k
        assert relmax(ev["out"], res["out"]) < TOL and relmax(ev["out2"], res["out2"]) < TOL
        nflip, worst = align_skipnew_cache(tr, c, B)
        assert worst < 1e-4                                  # only activations within the product error of zero move
        g = o.backward(p, c, cfg)
        tr.train_step(src, ctx, tgt, lr=0.0)
        gg = tr.get_grads()
        for n in g:
            assert relmax(gg[n], g[n]) < TOL, (n, nflip)
        np.testing.assert_array_equal(tr.get_params_flat(), o.flatten(p, cfg, np.float32))


def test_split_mode_is_active_and_deterministic(T):
    from imitation_from_observation_amd import Translator
    H, W, d, F, B = 32, 32, 32, 128, 6
    cfg, p, fr = make_case(H, W, d, F, B, seed=2)
    src, ctx, tgt = (o.preprocess_u8(x) for x in fr)
    outs = []
    for mk in (T, T, lambda *a, **k: Translator(*a, precision="f32", **k)):
        with mk(H, W, d, F, max_batch=B) as tr:
            tr.set_params(p)
            tr.train_step(src, ctx, tgt, lr=1e-3)
            outs.append((tr.evaluate(src, ctx, tgt)["out"], tr.get_params_flat()))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])    # bit-reproducible (no atomics, fixed split-K order)
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert not np.array_equal(outs[0][0], outs[2][0])        # and it is not the f32 path
    assert relmax(outs[0][0], outs[2][0]) < 2 * TOL          # after an Adam step of lr 1e-3 on both: the product error, amplified once


def test_split_adam_trajectory(T):
    H, W, d, F, B = 32, 32, 32, 128, 4
    cfg, p, fr = make_case(H, W, d, F, B, seed=3)
    src, ctx, tgt = (o.preprocess_u8(x) for x in fr)
    q = {k: v.copy() for k, v in p.items()}
    m = {k: np.zeros_like(v) for k, v in q.items()}
    v = {k: np.zeros_like(v_) for k, v_ in q.items()}
    with T(H, W, d, F, max_batch=B) as tr:
        tr.set_params(p)
        for t in range(1, 4):
            # lr 1e-5: the steps of unresolved weights (below) must not feed back into the next gradients
            rr, _ = o.train_step(q, m, v, t, *(x.astype(np.float64) for x in (src, ctx, tgt)), 1e-5, cfg)
            sc = tr.train_step(src, ctx, tgt, lr=1e-5)
            assert abs(sc["loss"] - rr["loss"]) <= TOL * abs(rr["loss"]), t
        mm, vv, step = tr.get_adam_state()
        # (branches are not aligned here: a handful of lrelu' flips put ~1e-3 into the moment estimates)
        assert step == 3 and rel_l2(mm, o.flatten(m, cfg)) < 1e-2 and rel_l2(vv, o.flatten(v, cfg)) < 1e-2
        # Adam's update m/(sqrt(v)+eps) is scale-free: a weight whose gradient is below the 1e-5 product error gets a
        # full-size step of arbitrary sign, so the update is compared where the gradient is resolved (|m| >= 1e-2 of
        # the tensor's max) and only loosely overall.
        got, p0 = tr.get_params(), p
        num = den = 0.0
        for n, _ in o.param_specs(cfg):
            mref = np.abs(m[n])
            sel = mref >= 1e-2 * mref.max()
            dg, dr = (np.asarray(got[n], np.float64) - p0[n])[sel], (q[n] - p0[n])[sel]
            num, den = num + float(((dg - dr) ** 2).sum()), den + float((dr ** 2).sum())
        assert (num / den) ** 0.5 < 1e-2
        assert rel_l2(o.flatten(got, cfg, np.float64) - o.flatten(p0, cfg), o.flatten(q, cfg) - o.flatten(p0, cfg)) < 0.15


@pytest.mark.parametrize("path", GOLD, ids=os.path.basename)
def test_split_golden_vectors(T, path):
    z, cfg, p = load_golden(path)
    B = int(z["B"])
    with T(cfg.H, cfg.W, cfg.df_dim, cfg.featsize, max_batch=max(B, 25)) as tr:
        tr.set_params(p)
        pred, feat = tr.translate(z["src_u8"], z["ctx_u8"][0])
        assert relmax(pred, z["translate_pred"]) < TOL and relmax(feat, z["translate_feat"]) < TOL
        f, x = tr.encode(z["src_u8"])
        assert relmax(f, z["encode_feat"]) < TOL
        np.testing.assert_array_equal(x, o.preprocess_u8(z["src_u8"]))
        ev = tr.evaluate(*(o.preprocess_u8(z[k]) for k in ("src_u8", "ctx_u8", "tgt_u8")))
        assert relmax(ev["out"], z["out"]) < TOL and relmax(ev["out2"], z["out2"]) < TOL
        for i, k in enumerate(("loss", "simloss", "recon1", "recon2")):
            assert abs(ev[k] - z["scalars"][i]) <= TOL * abs(z["scalars"][i])


def test_split_context_ae_real(T):
    cfg = r.RealConfig()
    p = r.init_params(cfg, 44, np.float64, stddev=0.1)
    rng = np.random.default_rng(3)
    B = 3
    fr = [rng.integers(0, 256, (B, 36, 64, 3), dtype=np.uint8) for _ in range(3)]
    src, ctx, tgt = (o.preprocess_u8(x) for x in fr)
    res, _ = r.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    with T(36, 64, featsize=100, max_batch=B, variant="real") as tr:
        tr.set_params(p)
        ev = tr.evaluate(src, ctx, tgt)
        assert relmax(ev["out"], res["out"]) < TOL and relmax(ev["out2"], res["out2"]) < TOL
        assert abs(ev["loss"] - res["loss"]) <= TOL * res["loss"]
        pred, feat = tr.translate(fr[0], fr[1][0])
        opred, ofeat = r.translate(p, fr[0], fr[1][0], cfg)
        assert relmax(pred, opred) < TOL and relmax(feat, ofeat) < TOL


def test_split_full_size_properties(T):
    """BASELINE batch (256 x 64 x 64): finite, loss decreases over Adam steps, bit-reproducible across handles."""
    import torch
    B = 256
    g = torch.Generator(device="cuda").manual_seed(5)
    fr = [torch.randint(0, 256, (B, 64, 64, 3), device="cuda", generator=g, dtype=torch.uint8).float() / 127.5 - 1 for _ in range(3)]
    losses, params = [], []
    for _ in range(2):
        with T(max_batch=B) as tr:
            tr.init_params(1234)
            ls = []
            for _ in range(3):
                tr.dev_forward_backward(*(t.data_ptr() for t in fr), B)
                tr.dev_adam(1e-4)
                ls.append(tr.dev_scalars()["loss"])
            losses.append(ls)
            params.append(tr.get_params_flat())
    assert all(np.isfinite(losses[0])) and losses[0][2] < losses[0][0]
    assert losses[0] == losses[1]
    np.testing.assert_array_equal(params[0], params[1])
