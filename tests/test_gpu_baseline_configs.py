"""BASELINE.json's configurations at the sizes it states (configs[1] was pinned only through properties, configs[2..4]
ran reduced).  What one GPU can show of each:

  configs[1]  ContextSkipNew 64x64x3, batch 256: against the committed float64-oracle fixture of the SAME batch
              (tests/golden/make_golden.py: make_b256) -- outputs, the four scalars, every parameter gradient.
  configs[2]  the data-parallel step at 256 per GPU: the default (one all-reduce after backward) path on a one-rank RCCL
              group must leave exactly the parameters of ctx_train_step; N > 1 arithmetic is tests/test_dp_gloo.py.
  configs[3]  ContextAEInception2 at production width (2x2x2048 Mixed_7c maps of 125x125 frames, filters 1024/1024/512/512):
              train step against oracle/ctx_oracle_incep.py at small batch; batch 64 per GPU through linearity.
  configs[4]  ContextAEReal at 64x64 (BASELINE's size; the reference's own is 36x64) against oracle/ctx_oracle_real.py.
"""
import os

import numpy as np
import pytest

from oracle import ctx_oracle as o

pytestmark = pytest.mark.gpu


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import Translator
    return Translator


# ------------------------------------------------------------------------------------------------ configs[1]
# device buffer -> (fixture names of its row blocks): the sign of these activations is what lrelu' reads
_ACT_BUFFERS = {"s0": ("s0_tgt", "s0_src"), "s1": ("s1_tgt", "s1_src"), "s2": ("s2_tgt", "s2_src"), "s3": ("s3_tgt", "s3_src"),
                "s4": ("s4_tgt", "s4_src"), "c0": ("c0",), "c1": ("c1",), "c2": ("c2",), "c3": ("c3",), "c4": ("c4",),
                "th0": ("th0",), "dz": ("d1_0", "d2_0"), "e1": ("d1_1", "d2_1"), "e2": ("d1_2", "d2_2"), "e3": ("d1_3", "d2_3")}


def test_config1_batch256_against_the_float64_oracle_fixture(T):
    """The production net at BASELINE configs[1]'s batch against the oracle's float64 pass over the same 256 triples.
    Forward: <= 1e-5.  Gradients: 4096 sampled entries per tensor + 16 random-sign projections that cover every entry.
    lrelu' branch flips (activations the f32 pass puts on the other side of zero than float64 does) are COUNTED and printed
    per buffer, not aligned away; the gradient bounds below are what holds with them in."""
    from tests.golden import make_golden as mg
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", mg.B256_TAG + ".npz"))
    cfg, p, frames = mg.b256_case()
    np.testing.assert_allclose(mg.digest(o.flatten(p, cfg))[0], z["param_digest"], rtol=1e-12)      # RNG drift guard
    B = int(z["B"])
    src, ctx, tgt = (o.preprocess_u8(f) for f in frames)
    names = [n for n, _ in o.param_specs(cfg)]
    keep = list(z["keep"])
    with T(cfg.H, cfg.W, cfg.df_dim, cfg.featsize, max_batch=B) as tr:
        tr.set_params(p)
        ev = tr.evaluate(src, ctx, tgt)
        np.testing.assert_allclose([ev[k] for k in ("loss", "simloss", "recon1", "recon2")], z["scalars"], rtol=1e-5)
        iz, tz = tr.last_codes()
        for got, k in ((ev["out"], "out"), (ev["out2"], "out2"), (tz, "translated_z"), (iz, "input_z")):
            assert relmax(got[keep], z[k + "_keep"]) < 1e-5, k
            flat = np.asarray(got, np.float64).reshape(B, -1)
            rows = np.stack([flat.sum(1), np.abs(flat).sum(1), np.sqrt((flat * flat).sum(1))], 1)
            np.testing.assert_allclose(rows[:, 1:], z[k + "_rows"][:, 1:], rtol=1e-5, err_msg=k)     # every image, not only the kept ones
        sc = tr.train_step(src, ctx, tgt, lr=0.0)
        assert abs(sc["loss"] - z["scalars"][0]) <= 1e-5 * z["scalars"][0]
        # lrelu' branch report: negative entries per buffer, device vs float64
        gold_neg = dict(zip((str(s) for s in z["act_names"]), z["act_negative"]))
        gold_near = dict(zip((str(s) for s in z["act_names"]), z["act_near_zero"]))
        flips = {}
        for buf, parts in _ACT_BUFFERS.items():
            sizes = {"s": 2, "c": 1, "t": 1, "d": 2, "e": 2}[buf[0]]
            per = {"s0": 32 * 32 * 64, "s1": 16 * 16 * 128, "s2": 8 * 8 * 256, "s3": 4 * 4 * 512, "s4": 1024,
                   "c0": 32 * 32 * 64, "c1": 16 * 16 * 128, "c2": 8 * 8 * 256, "c3": 4 * 4 * 512, "c4": 1024, "th0": 1024,
                   "dz": 8192, "e1": 8 * 8 * 256, "e2": 16 * 16 * 128, "e3": 32 * 32 * 64}[buf]
            a = tr.debug_read(buf, sizes * B * per)
            flips[buf] = (int((a < 0).sum()) - sum(int(gold_neg[q]) for q in parts), sum(int(gold_near[q]) for q in parts))
        zall = tr.debug_read("Z", 3 * B * cfg.featsize).reshape(3, B, -1)
        flips["z"] = (int((zall[1:] < 0).sum()) - int(gold_neg["z_tgt"]) - int(gold_neg["z_src"]), int(gold_near["z_tgt"]) + int(gold_near["z_src"]))
        print("lrelu' branch report (buffer: net sign changes vs float64, candidates within 1e-6 of zero):", flips)
        assert all(abs(d) <= max(8, near) for d, near in flips.values()), flips          # a wrong activation would move thousands
        gg = tr.get_grads()
        probes = mg.b256_probes([(n, int(np.prod(gg[n].shape))) for n in names])
        report = {}
        for i, n in enumerate(names):
            a = np.asarray(gg[n], np.float64).reshape(-1)
            seed, idx = probes[n]
            ref_s = z["grad_samples"][i][: len(idx)]
            gnorm = z["grad_digest"][i][2]
            samp = float(np.linalg.norm(a[idx] - ref_s) / (np.linalg.norm(ref_s) + 1e-30))
            # E[<d, r>^2] = |d|^2 for random +-1 vectors r: the 16 projections estimate the L2 norm of the WHOLE difference
            proj = float(np.sqrt(np.mean((mg.b256_project(a, seed) - z["grad_proj"][i]) ** 2)) / gnorm)
            nrm = abs(float(np.sqrt((a * a).sum())) - gnorm) / gnorm
            report[n] = (samp, proj, nrm)
        print("gradient deviation per tensor (rel-L2 on 4096 samples, projected rel-L2 of the whole tensor, |norm| deviation):",
              {k: tuple(float(f"{x:.1e}") for x in v) for k, v in report.items()})
        # d_h4 sits upstream of every lrelu' mask: no flip can reach it.  Everything else carries the flips reported above;
        # north_star's budget is 1e-3 relative.
        # Un-aligned bars = what this build measures + 30 % (round 6: worst sample rel-L2 1.3e-3 -- the `conv` encoder behind four flipped
        # activations --, projections 1.5e-3, norms 2.8e-4; profiles/round6_c_unaligned_gradient_deviation.txt).  That these are flips and
        # nothing else: the aligned comparison below / tests/test_gpu_parity.py::test_adam_trajectory_branch_aligned (<= 2e-6 / 1e-4).
        for n, (samp, proj, nrm) in report.items():
            tight = n.startswith("deconv/d_h4")
            assert samp <= (1e-5 if tight else 1.5e-3), (n, samp, proj, nrm)
            assert proj <= (1e-5 if tight else 2e-3), (n, samp, proj, nrm)               # 16 projections: +-35 % on the estimate
            assert nrm <= (1e-5 if tight else 4e-4), (n, samp, proj, nrm)


# ------------------------------------------------------------------------------------------------ configs[2]
def test_config2_default_dp_path_on_one_rank_rccl_at_256_per_gpu(T, monkeypatch):
    """BASELINE configs[2] is 256 triples per GPU with an RCCL all-reduce of the gradient arena.  One GPU can run the
    per-rank half of that: the default schedule (backward, ONE all-reduce, Adam) on a one-rank RCCL group with the
    collective forced on (CTX_DP_FORCE=1), at the full size -- and it must leave bit-for-bit the parameters and scalars of
    ctx_train_step (a one-rank sum is the identity)."""
    import torch
    import torch.distributed as dist
    from imitation_from_observation_amd.dp import DataParallelTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29537"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        monkeypatch.setenv("CTX_DP_FORCE", "1")
        monkeypatch.setenv("CTX_DP_OVERLAP", "0")
        B = 256
        rng = np.random.default_rng(31)
        host = [o.preprocess_u8(rng.integers(0, 256, (B, 64, 64, 3), dtype=np.uint8)) for _ in range(3)]
        dev = [torch.from_numpy(x).cuda() for x in host]
        dp = DataParallelTrainer(64, 64, 64, 1024, max_batch=B, device=0, seed=99)
        calls = []
        real = dist.all_reduce

        def spy(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)
        monkeypatch.setattr(dist, "all_reduce", spy)
        with T(64, 64, 64, 1024, max_batch=B) as ref:
            ref.set_params_flat(dp.translator.get_params_flat())
            for _ in range(2):
                dp.step(*dev, lr=1e-4)
                sref = ref.train_step(*host, lr=1e-4)
            assert dp.scalars() == sref
            torch.cuda.synchronize()
            np.testing.assert_array_equal(dp.translator.get_params_flat(), ref.get_params_flat())
        assert calls.count(dp.engine.grads.numel()) == 2          # the whole gradient arena, once per step
        dp.translator.close()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ configs[3]
def _incep_case(B, seed):
    from oracle import ctx_oracle_incep as oi
    cfg = oi.Incep2Config()                                         # 2x2x2048 maps, filters 1024/1024/512/512, featsize 1024
    p = oi.init_params(cfg, 70 + seed, np.float32, stddev=0.02)
    brng = np.random.default_rng(seed + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = (brng.standard_normal(p[n].shape) * 0.02).astype(np.float32)
    rng = np.random.default_rng(seed)
    feats = [np.maximum(rng.standard_normal((B, cfg.H, cfg.W, cfg.C)), 0).astype(np.float32) for _ in range(3)]   # post-ReLU Mixed_7c
    return oi, cfg, p, feats


def test_config3_inception2_train_step_at_production_width_matches_oracle(T):
    """ContextAEInception2(strides [1,2,1,2], kernels [3,3,3,3], filters [1024,1024,512,512]) on 2x2x2048 maps
    (rllab/sampler/base.py:126; 125x125 frames): forward, every gradient and two Adam steps against the float64 oracle."""
    B = 3
    oi, cfg, p32, (src, ctx, tgt) = _incep_case(B, seed=0)
    p = {k: v.astype(np.float64) for k, v in p32.items()}
    res, c = oi.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    g = oi.backward(p, c, cfg)
    with T(cfg.H, cfg.W, df_dim=64, featsize=1024, max_batch=B, variant="inception2", C=cfg.C) as tr:
        assert tr.n_params == oi.param_count(cfg)
        tr.set_params(p32)
        ev = tr.evaluate(src, ctx, tgt)
        for k in ("loss", "simloss", "recon1", "recon2"):
            assert abs(ev[k] - res[k]) <= 1e-5 * abs(res[k]) + 1e-6, k
        assert relmax(ev["out"], res["out"]) < 1e-5 and relmax(ev["out2"], res["out2"]) < 1e-5
        tr.train_step(src, ctx, tgt, lr=0.0)
        gg = tr.get_grads()
        for n in g:
            assert relmax(gg[n], g[n]) < 1e-4, n
        del gg, c
        m = {k: np.zeros_like(v) for k, v in p.items()}
        v = {k: np.zeros_like(v_) for k, v_ in p.items()}
        p0 = oi.flatten(p, cfg)
        # the lr = 0 step above was Adam step 1 on the device (it moved m and v, not the parameters): replay it in the oracle
        o.adam_step({k: v_.copy() for k, v_ in p.items()}, g, m, v, 1, 0.0)
        for t in (2, 3):
            rr, cc = oi.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
            o.adam_step(p, oi.backward(p, cc, cfg), m, v, t, 1e-4)
            sc = tr.train_step(src, ctx, tgt, lr=1e-4)
            assert abs(sc["loss"] - rr["loss"]) <= 2e-5 * rr["loss"], t
        # Adam slots are linear / quadratic in the gradients: compared whole.  The UPDATE lr_t * m / (sqrt(v) + eps) of an entry whose
        # gradient sits at the f32 noise floor is +-lr whatever its sign turns out to be (148 M parameters fed by sparse post-ReLU
        # maps have many of those), so it is compared where the oracle's first-step gradient is above 1e-3 of the tensor's largest.
        mm, vv, step = tr.get_adam_state()
        assert step == 3                                                                  # the lr = 0 step counts too
        assert rel_l2(mm, oi.flatten(m, cfg)) < 1e-4 and rel_l2(vv, oi.flatten(v, cfg)) < 1e-4
        d_ref = oi.flatten(p, cfg) - p0
        d_got = tr.get_params_flat().astype(np.float64) - p0
        big = np.concatenate([(np.abs(g[n]) > 1e-3 * np.abs(g[n]).max()).reshape(-1) for n, _ in oi.param_specs(cfg)])
        assert big.mean() > 0.05
        assert np.linalg.norm((d_got - d_ref)[big]) <= 2e-3 * np.linalg.norm(d_ref[big])


def test_config3_inception2_batch64_per_gpu_through_linearity(T):
    """BASELINE configs[3] runs 64 triples per GPU: the batch-64 step (position-major convs on the 2x2 / 1x1 grids, rectangle-
    ordered filter gradients, split-K everywhere) must produce the sum of eight batch-8 shard gradients -- the small-batch
    kernels that the test above pins on the oracle -- and outputs that do not depend on the batch mates."""
    import torch
    B, S = 64, 8
    oi, cfg, p32, feats = _incep_case(B, seed=5)
    dev = [torch.from_numpy(x).cuda() for x in feats]
    torch.cuda.synchronize()
    with T(cfg.H, cfg.W, df_dim=64, featsize=1024, max_batch=B, variant="inception2", C=cfg.C) as tr:
        tr.set_params(p32)
        big = tr.evaluate(*feats)
        small = tr.evaluate(*(x[16:24] for x in feats))
        assert relmax(small["out"], big["out"][16:24]) < 1e-5 and relmax(small["out2"], big["out2"][16:24]) < 1e-5
        again = tr.evaluate(*feats)
        np.testing.assert_array_equal(again["out"], big["out"])
        assert again["loss"] == big["loss"]
        r1 = 0.5 * np.sum((feats[2].astype(np.float64) - big["out"]) ** 2)
        assert abs(big["recon1"] - r1) <= 1e-5 * r1                                     # out = decode + tgtctx is compared with tgt
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
        print("batch 64 vs sum of 8 shards, rel-L2 per tensor:", {k: float(f"{v:.1e}") for k, v in worst.items()})
        # measured 8.0e-7 (no activation changes sides between the two summation orders on this fixture); one lrelu' flip would show as ~1e-3
        assert max(worst.values()) <= 2e-6, worst
        s = tr.train_step(*feats, lr=1e-4)
        assert np.isfinite(s["loss"]) and abs(s["loss"] - big["loss"]) <= 1e-6 * big["loss"]


# ------------------------------------------------------------------------------------------------ configs[4]
@pytest.mark.parametrize("B", [2, 32])
def test_config4_context_ae_real_at_64x64_matches_oracle(T, B):
    """ContextAEReal (name 'sweep', base.py:134-135) at BASELINE's 64x64: B = 2 on the image-major kernels, B = 32 (96 encoder /
    64 decoder images per launch) on the position-major / rectangle-ordered ones; the reward hook's two fetches as well."""
    from oracle import ctx_oracle_real as r
    cfg = r.RealConfig(H=64, W=64)
    p = r.init_params(cfg, 90, np.float64, stddev=0.1)
    brng = np.random.default_rng(91)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * 0.1
    rng = np.random.default_rng(92 + B)
    fr = [rng.integers(0, 256, (B, 64, 64, 3), dtype=np.uint8) for _ in range(3)]
    src, ctx, tgt = (o.preprocess_u8(x) for x in fr)
    res, c = r.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    g = r.backward(p, c, cfg)
    with T(64, 64, featsize=100, max_batch=B, variant="real") as tr:
        tr.set_params(p)
        ev = tr.evaluate(src, ctx, tgt)
        for k in ("loss", "simloss", "recon1", "recon2"):
            assert abs(ev[k] - res[k]) <= 1e-5 * abs(res[k]) + 1e-6, k
        assert relmax(ev["out"], res["out"]) < 1e-5 and relmax(ev["out2"], res["out2"]) < 1e-5
        iz, tz = tr.last_codes()
        assert relmax(tz, res["translated_z"]) < 1e-5 and relmax(iz, res["input_z"]) < 1e-5
        tr.train_step(src, ctx, tgt, lr=0.0)
        gg = tr.get_grads()
        print(f"ContextAEReal 64x64 B = {B}: worst un-aligned gradient deviation max-norm {max(relmax(gg[n], g[n]) for n in g):.2e}, rel-L2 {max(rel_l2(gg[n], g[n]) for n in g):.2e}")
        for n in g:
            # un-aligned; measured + 30 %: B = 2 9.6e-7 / 8.6e-7, B = 32 6.4e-4 / 8.7e-5 (a flipped activation in the position-major launches)
            assert relmax(gg[n], g[n]) < (2e-6 if B == 2 else 8.5e-4), n
            assert rel_l2(gg[n], g[n]) < (2e-6 if B == 2 else 1.2e-4), n
        if B == 2:
            pred, feat = tr.translate(fr[0], fr[1][0])
            c0 = np.broadcast_to(o.preprocess_u8(fr[1][0]), src.shape).astype(np.float64)
            tres, _ = r.forward(p, src.astype(np.float64), c0, c0, cfg)
            assert relmax(pred, tres["out"]) < 1e-5 and relmax(feat, tres["translated_z"]) < 1e-5
            f, x = tr.encode(fr[2])
            np.testing.assert_array_equal(x, tgt)
            assert relmax(f, r._encode(p, tgt.astype(np.float64))[5]) < 1e-5
