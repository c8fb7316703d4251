"""The reference's OWN configurations that earlier rounds only touched at reduced size (VERDICT r3 "next round" 1a-1c):

  (a) mode 'oursinception' as the launchers run it: 299x299 frames (sandbox/andrew/run_trpo_strike.py:84,
      run_train_strike_inception.py:39-43: idims=(299, 299), batch_size=25) -> Inception-v3 Mixed_7c 8x8x2048
      (rllab/sampler/base.py:121-132) -> ContextAEInception2(strides [1,2,1,2], kernels [3,3,3,3], filters [1024,1024,512,512]):
      the front end's 18 end points at 299x299, the translator at 8x8x2048 with batch 25 (forward, every gradient, an Adam step,
      both reward fetches) against committed float64-oracle fixtures (tests/golden/make_golden.py ref299), and the two chained.
  (b) ContextSkipNew at the reach / push launchers' 48x48 frames (run_trpo_reach.py:85: imsize=(48, 48)) with the PRODUCTION widths
      (df_dim 64, featsize 1024): grids 24 / 12 / 6 / 3, none of which the LDS-resident transposed-conv kernel takes.
  (c) the ablation script's loss switch (ablations_code/ablations.py:175-182, :477-484) on the ContextAEReal family and on
      ContextAEInception2 (the switch is carried by every model class of that file).
"""
import os

import numpy as np
import pytest

from oracle import ctx_oracle as o

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def digest(a):
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())])


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import Translator
    return Translator


# ------------------------------------------------------------------------------------------------ (a) 299x299 -> 8x8x2048
def test_inception_front_end_at_299x299_matches_the_fixture_and_the_oracle(T):
    """nets/inception_v3.py:93-416 at the size its own test pins (inception_v3_test.py:45-54: Mixed_7c = [N, 8, 8, 2048]): every end
    point of two seeded frames against the committed float64 fixture (digests, heads, Mixed_7c of frame 0 whole)."""
    from imitation_from_observation_amd.inception_frontend import InceptionFrontend
    from tests.golden import make_golden as mg
    z = np.load(os.path.join(GOLD, mg.REF299_FRONT_TAG + ".npz"))
    frames = np.random.default_rng(int(z["fseed"])).integers(0, 256, (2, 299, 299, 3), dtype=np.uint8)
    np.testing.assert_allclose(digest(frames), z["frames_digest"], rtol=0)                       # RNG drift guard
    with InceptionFrontend(299, 299, max_images=2) as f:
        f.init_synthetic(int(z["pseed"]))
        feat = f.features(frames)
        assert feat.shape == (2, 8, 8, 2048)
        assert relmax(feat[0], z["Mixed_7c_0"]) < 1e-4
        for i, name in enumerate(str(n) for n in z["endpoints"]):
            got = f.endpoint(name, 2)
            assert tuple(got.shape) == tuple(int(v) for v in z["endpoint_shapes"][i]), name
            d = digest(got)
            # l1 and l2 norms of the whole end point, and its first 64 entries
            np.testing.assert_allclose(d[1:], z["endpoint_digest"][i][1:], rtol=1e-4, err_msg=name)
            head = got.reshape(-1)[:64]
            assert np.abs(head - z["endpoint_head"][i]).max() <= 1e-4 * max(np.abs(got).max(), 1e-30), name


def _ref299_fixture():
    from tests.golden import make_golden as mg
    z = np.load(os.path.join(GOLD, mg.REF299_TAG + ".npz"))
    cfg, p32, feats = mg.ref299_case()
    from oracle import ctx_oracle_incep as ci
    np.testing.assert_allclose(digest(ci.flatten(p32, cfg)), z["param_digest"], rtol=1e-12)      # RNG drift guard
    return mg, ci, z, cfg, p32, feats


def test_inception2_at_8x8x2048_batch25_against_the_float64_fixture(T):
    """ContextAEInception2 at the reference's size and batch (153 M parameters, 8x8x2048 maps, batch 25): forward, the four scalars,
    every parameter gradient (digest norm, 1024 sampled entries, 16 random-sign projections), the scalars of the step after one Adam
    update, and the reward hook's two fetches -- against tests/golden/incep2_8x8x2048_f1024_b25.npz (float64 oracle)."""
    mg, ci, z, cfg, p32, (src, ctx, tgt) = _ref299_fixture()
    B = int(z["B"])
    keep = list(z["keep"])
    names = [n for n, _ in ci.param_specs(cfg)]
    with T(cfg.H, cfg.W, df_dim=64, featsize=cfg.featsize, max_batch=B, variant="inception2", C=cfg.C) as tr:
        assert tr.n_params == ci.param_count(cfg)
        tr.set_params(p32)
        ev = tr.evaluate(src, ctx, tgt)
        np.testing.assert_allclose([ev[k] for k in ("loss", "simloss", "recon1", "recon2")], z["scalars"], rtol=1e-5)
        iz, tz = tr.last_codes()
        for got, k in ((ev["out"], "out"), (ev["out2"], "out2"), (tz, "translated_z"), (iz, "input_z")):
            assert relmax(got[keep], z[k + "_keep"]) < 1e-5, k
            flat = np.asarray(got, np.float64).reshape(B, -1)
            rows = np.stack([np.abs(flat).sum(1), np.sqrt((flat * flat).sum(1))], 1)
            np.testing.assert_allclose(rows, z[k + "_rows"][:, 1:], rtol=1e-5, err_msg=k)         # every triple, not only the kept one
        # the reward hook's fetches at its batch (base.py:216-218, :234-235; image_trans = feature maps)
        pred, feat = tr.translate_f32(src, ctx[0])
        assert relmax(pred[keep], z["translate_pred_keep"]) < 1e-5 and relmax(feat, z["translate_feat"]) < 1e-5
        assert relmax(tr.encode_f32(src), z["encode_feat"]) < 1e-5
        # Adam step 1 (train_script.py:163): scalars before the update, the gradient it used
        sc = tr.train_step(src, ctx, tgt, lr=float(z["lr"]))
        assert abs(sc["loss"] - z["train_scalars"][0][0]) <= 1e-5 * z["train_scalars"][0][0]
        # lrelu' branches.  An f32 pass puts a few of the ~3e6 activations that sit within rounding of zero on the other side of the kink
        # than float64 does, and each one moves 0.8 dy of one (image, position, channel) into every gradient upstream of it (first box
        # run: conv/h0..h2 2e-3 off the fixture with everything downstream of h2 at 1e-4).  So the gradients are compared twice:
        # (1) with the FIXTURE as it is -- whole-tensor bounds that hold with the flips in; (2) with the float64 oracle re-run here on
        # the same inputs, its saved activations given the device's sign at exactly the flipped elements (tests/_align.py) -- tight.
        gold = {str(n): (int(a), int(b), int(c)) for n, a, b, c in zip(z["act_names"], z["act_negative"], z["act_near_zero"], z["act_size"])}
        delta = {}
        for buf in ("a0", "a1", "a2", "a3", "a4", "th0", "dz", "e1", "e2", "e3"):
            neg, near, size = gold[buf]
            delta[buf] = (int((tr.debug_read(buf, size) < 0).sum()) - neg, near)
        zall = tr.debug_read("Z", 4 * B * cfg.featsize).reshape(4, B, -1)               # [trans_z | tgt_z | src_z | ctx_z]
        delta["z"] = (int((zall[1:] < 0).sum()) - gold["z"][0], gold["z"][1])
        print("lrelu' branch report (buffer: net sign changes vs float64, candidates within 1e-6 of zero):", delta)
        assert all(abs(d) <= max(8, near) for d, near in delta.values()), delta          # a wrong activation would move thousands
        gg = tr.get_grads()
        probes = mg.ref299_probes([(n, int(np.prod(gg[n].shape))) for n in names])
        report = {}
        for i, n in enumerate(names):
            a = np.asarray(gg[n], np.float64).reshape(-1)
            seed, idx = probes[n]
            ref_s = z["grad_samples"][i][: len(idx)]
            gnorm = z["grad_digest"][i][2]
            samp = float(np.linalg.norm(a[idx] - ref_s) / (np.linalg.norm(ref_s) + 1e-30))
            proj = float(np.sqrt(np.mean((mg.ref299_project(a, seed) - z["grad_proj"][i]) ** 2)) / gnorm)
            nrm = abs(float(np.sqrt((a * a).sum())) - gnorm) / gnorm
            report[n] = (samp, proj, nrm)
        print("gradient deviation per tensor vs the fixture (rel-L2 on 1024 samples, projected rel-L2 of the whole tensor, |norm| deviation):",
              {k: tuple(float(f"{x:.1e}") for x in v) for k, v in report.items()})
        for n, (samp, proj, nrm) in report.items():
            tight = n.startswith("deconv/d_h4")                                         # upstream of every lrelu' mask
            # un-aligned bars = measured + 30 % (round 6: 1.4e-4 / 1.1e-4 / 1.2e-5, the `conv` encoder's tensors; round 5's build had flips
            # worth 2.3e-3 here -- which activations sit within rounding of zero moves with every summation order, so a kernel change may
            # move these: the aligned comparison (2) below is the invariant one)
            assert samp <= (1e-5 if tight else 2e-4), (n, samp, proj, nrm)
            assert proj <= (1e-5 if tight else 1.5e-4), (n, samp, proj, nrm)
            assert nrm <= (1e-5 if tight else 2e-5), (n, samp, proj, nrm)
        # (2) the oracle re-run with the device's branches
        from tests._align import align_gen_cache
        p64 = {k: v.astype(np.float64) for k, v in p32.items()}
        res, c = ci.forward(p64, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
        np.testing.assert_allclose([res["loss"], res["simloss"], res["recon1"], res["recon2"]], z["scalars"], rtol=1e-12)
        nflip, worst, where = align_gen_cache(tr, c, B)
        print(f"aligned {nflip} activations (largest |x| / max|x| among them {worst:.1e}) in {where}")
        assert nflip <= 64 and worst <= 1e-5
        g = ci.backward(p64, c, cfg)
        del p64, c
        worst_t = {n: relmax(gg[n], g[n]) for n in names}
        print("gradients vs the branch-aligned oracle, max-norm relative:", {k: float(f"{v:.1e}") for k, v in worst_t.items()})
        assert max(worst_t.values()) <= 2e-4, worst_t
        del g, gg
        # the step after the update: the loss fell by what the oracle's float64 Adam step takes off (4.5e7 -> 1.8e7)
        sc2 = tr.train_step(src, ctx, tgt, lr=float(z["lr"]))
        want = z["train_scalars"][1]
        assert abs(sc2["loss"] - want[0]) <= 2e-4 * want[0], (sc2, want)
        assert abs(sc2["recon1"] - want[2]) <= 2e-4 * want[2] and abs(sc2["recon2"] - want[3]) <= 2e-4 * want[3]


def test_oursinception_end_to_end_at_299x299(T):
    """rllab/sampler/base.py:121-132, 216-218, 234-235 at imsize 299x299: uint8 frames -> Inception-v3 -> ContextAEInception2 at the
    sampler's lists and PRODUCTION widths on device-resident 8x8x2048 maps; the trainer's validation fetch, one train step and the
    hook's translate against the two oracles composed on the CPU."""
    from imitation_from_observation_amd.oursinception import InceptionTranslator
    from oracle import ctx_oracle_incep as oi
    from oracle import inception_oracle as io
    rng = np.random.default_rng(77)
    B, S = 2, 299
    frames = [rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8) for _ in range(3)]
    with InceptionTranslator((S, S), max_batch=B) as it:
        assert it.pred_shape == (8, 8, 2048)
        ip = {k: v.astype(np.float64) for k, v in it.front.init_synthetic(4).items()}
        cfg = oi.Incep2Config(H=8, W=8)
        tp = oi.init_params(cfg, 9, np.float32, stddev=0.01)
        it.tr.set_params(tp)
        f = [io.forward(ip, o.preprocess_u8(x).astype(np.float64))["Mixed_7c"] for x in frames]
        p64 = {k: v.astype(np.float64) for k, v in tp.items()}
        res, _ = oi.forward(p64, *f, cfg)
        ev = it.evaluate_u8(*frames)
        for k in ("loss", "simloss", "recon1", "recon2"):
            assert abs(ev[k] - res[k]) <= 1e-3 * abs(res[k]) + 1e-6, k
        assert relmax(ev["out"], res["out"]) <= 1e-3 and relmax(ev["tgt"], f[2]) <= 1e-3
        l0 = it.train_step_u8(*frames, lr=1e-4)["loss"]
        assert abs(l0 - res["loss"]) <= 1e-3 * abs(res["loss"])
        l1 = it.train_step_u8(*frames, lr=1e-4)["loss"]
        assert l1 < l0
    with InceptionTranslator((S, S), max_batch=B, train=False) as it:                    # the reward hook's sizing: 2 B front-end images
        it.front.init_synthetic(4)
        it.tr.set_params(tp)
        pred, feat = it.translate(frames[0], frames[1][0])
        opred, ofeat = oi.translate(p64, f[0], f[1][0], cfg)
        assert relmax(pred, opred) <= 1e-3 and relmax(feat, ofeat) <= 1e-3
        with pytest.raises(ValueError, match="train=False"):
            it.evaluate_u8(*frames)


# ------------------------------------------------------------------------------------------------ (b) 48x48, production widths
def _skipnew48(B, seed):
    cfg = o.SkipNewConfig(H=48, W=48)                                                   # df_dim 64, featsize 1024
    p = o.init_params(cfg, 300 + seed, np.float32, stddev=0.02)
    brng = np.random.default_rng(301 + seed)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = (brng.standard_normal(p[n].shape) * 0.02).astype(np.float32)
    rng = np.random.default_rng(302 + seed)
    fr = [rng.integers(0, 256, (B, 48, 48, 3), dtype=np.uint8) for _ in range(3)]
    return cfg, p, fr


def test_skipnew_48x48_at_production_widths_matches_oracle(T):
    """run_trpo_reach.py:85 / run_trpo_push.py: imsize (48, 48) with the class defaults df_dim = 64, featsize = 1024
    (arm_shaping.py:1261-1277): h3 = 3x3x512 as notebooks/reach.ipynb records.  Forward, all 38 gradients, two Adam steps, both
    reward fetches against the float64 oracle."""
    B = 4
    cfg, p32, fr = _skipnew48(B, 0)
    src, ctx, tgt = (o.preprocess_u8(x) for x in fr)
    p = {k: v.astype(np.float64) for k, v in p32.items()}
    res, c = o.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    assert c["e_src"][3].shape == (B, 3, 3, 512)
    g = o.backward(p, c, cfg)
    with T(48, 48, 64, 1024, max_batch=B) as tr:
        assert tr.n_params == o.param_count(cfg)
        tr.set_params(p32)
        ev = tr.evaluate(src, ctx, tgt)
        for k in ("loss", "simloss", "recon1", "recon2"):
            assert abs(ev[k] - res[k]) <= 1e-5 * abs(res[k]), k
        assert relmax(ev["out"], res["out"]) < 1e-5 and relmax(ev["out2"], res["out2"]) < 1e-5
        pred, feat = tr.translate(fr[0], fr[1][0])
        opred, ofeat = o.translate(p32, fr[0], fr[1][0], cfg)
        assert relmax(pred, opred) < 1e-4 and relmax(feat, ofeat) < 1e-4
        f, x = tr.encode(fr[2])
        np.testing.assert_array_equal(x, tgt)
        assert relmax(f, o.encode(p32, fr[2], cfg)[0]) < 1e-4
        sc = tr.train_step(src, ctx, tgt, lr=1e-4)
        assert abs(sc["loss"] - res["loss"]) <= 1e-5 * res["loss"]
        gg = tr.get_grads()
        for n in g:
            assert relmax(gg[n], g[n]) < 1e-3 and rel_l2(gg[n], g[n]) < 1e-3, n
        m = {k: np.zeros_like(v) for k, v in p.items()}
        v = {k: np.zeros_like(v_) for k, v_ in p.items()}
        o.adam_step(p, g, m, v, 1, 1e-4)
        r2, _ = o.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
        sc2 = tr.train_step(src, ctx, tgt, lr=1e-4)
        assert abs(sc2["loss"] - r2["loss"]) <= 1e-4 * r2["loss"]
        assert r2["loss"] < res["loss"]


def test_skipnew_48x48_batch64_equals_the_sum_of_its_shards(T):
    """At 64 triples the 48x48 net runs on the position-major / rectangle-ordered launches (24 / 12 / 6 / 3 grids, 128 and 64 images per
    launch): its gradient must be the sum of eight batch-8 shard gradients -- the image-major kernels the test above pins on the
    oracle -- and its outputs must not depend on the batch mates."""
    import torch
    B, S = 64, 8
    cfg, p32, fr = _skipnew48(B, 1)
    host = [o.preprocess_u8(x) for x in fr]
    dev = [torch.from_numpy(x).cuda() for x in host]
    torch.cuda.synchronize()
    with T(48, 48, 64, 1024, max_batch=B) as tr:
        tr.set_params(p32)
        big = tr.evaluate(*host)
        small = tr.evaluate(*(x[16:24] for x in host))
        assert relmax(small["out"], big["out"][16:24]) < 1e-5 and relmax(small["out2"], big["out2"][16:24]) < 1e-5
        r1 = 0.5 * np.sum((host[2].astype(np.float64) - big["out"]) ** 2)
        assert abs(big["recon1"] - r1) <= 1e-5 * r1
        tr.dev_forward_backward(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), B, sim_batch=B)
        tr.sync()
        full = tr.get_grads_flat().astype(np.float64)
        acc = np.zeros_like(full)
        for i in range(0, B, S):
            sl = [x[i:i + S].contiguous() for x in dev]
            torch.cuda.synchronize()
            tr.dev_forward_backward(sl[0].data_ptr(), sl[1].data_ptr(), sl[2].data_ptr(), S, sim_batch=B)
            tr.sync()
            acc += tr.get_grads_flat()
        worst = {}
        for name, shape, off in tr.param_info():
            n = int(np.prod(shape))
            worst[name] = rel_l2(full[off:off + n], acc[off:off + n])
        print("48x48 batch 64 vs sum of 8 shards, rel-L2 per tensor:", {k: float(f"{v:.1e}") for k, v in worst.items()})
        assert max(worst.values()) <= 8e-4, worst                                       # lrelu' flips between differently ordered f32 sums: measured 5.9e-4 + 30 %
        assert worst["deconv/d_h4/w"] <= 1e-6                                           # upstream of every mask (measured 9.2e-8)


# ------------------------------------------------------------------------------------------------ (c) loss switches, other variants
def _check_ablation(tr, res, g, src, ctx, tgt, tol):
    sc = tr.train_step(src, ctx, tgt, lr=0.0)
    for k in ("loss", "simloss", "recon1", "recon2"):
        assert abs(sc[k] - res[k]) <= 1e-5 * abs(res[k]) + 1e-6, k
    gg = tr.get_grads()
    for n in g:
        den = np.abs(g[n]).max()
        if den == 0:                                       # a term that is switched off leaves some tensors without gradient
            assert np.abs(gg[n]).max() == 0, n
        else:
            assert np.abs(gg[n] - g[n]).max() <= tol * den, n
    ev = tr.evaluate(src, ctx, tgt)
    assert abs(ev["loss"] - res["loss"]) <= 1e-5 * abs(res["loss"]) + 1e-6


@pytest.mark.parametrize("ablation", ["L2", "L2L3", "L1"])
@pytest.mark.parametrize("shape", [(36, 64, 3), (64, 64, 32)])
def test_loss_ablations_on_context_ae_real(T, ablation, shape):
    """ContextAEPushReal / ContextAESweep of the ablation script (ablations_code/ablations.py:390-484) are ContextAEReal with the
    `ablation_type` switch: narrow image-major path (B = 3, 36x64) and the position-major one (B = 32, 64x64)."""
    from oracle import ctx_oracle_real as r
    H, W, B = shape
    cfg = r.RealConfig(H=H, W=W)
    p = r.init_params(cfg, 50, np.float64, stddev=0.1)
    rng = np.random.default_rng(51)
    src, ctx, tgt = (o.preprocess_u8(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)) for _ in range(3))
    res, c = r.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg, ablation_type=ablation)
    g = r.backward(p, c, cfg)
    full, _ = r.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    assert res["loss"] != full["loss"]
    with T(H, W, featsize=100, max_batch=B, variant="real", ablation_type=ablation) as tr:
        tr.set_params(p)
        _check_ablation(tr, res, g, src, ctx, tgt, 1e-4 if B == 3 else 1e-3)


@pytest.mark.parametrize("ablation", ["L2", "L2L3", "L1"])
def test_loss_ablations_on_inception2(T, ablation):
    from oracle import ctx_oracle_incep as oi
    H, W, C, d, F, B = 4, 4, 64, 4, 64, 3
    cfg = oi.Incep2Config(H=H, W=W, C=C, featsize=F, filters=(16 * d, 16 * d, 8 * d, 8 * d))
    p = oi.init_params(cfg, 61, np.float64, stddev=0.05)
    rng = np.random.default_rng(62)
    src, ctx, tgt = (np.maximum(rng.standard_normal((B, H, W, C)), 0).astype(np.float32) for _ in range(3))
    res, c = oi.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg, ablation_type=ablation)
    g = oi.backward(p, c, cfg)
    with T(H, W, df_dim=d, featsize=F, max_batch=B, variant="inception2", C=C, ablation_type=ablation) as tr:
        tr.set_params(p)
        _check_ablation(tr, res, g, src, ctx, tgt, 1e-4)
