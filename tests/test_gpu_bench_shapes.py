"""The launch shapes bench.py's `secondary` numbers are quoted on, against committed float64-oracle fixtures of the SAME size
(VERDICT r4 next-1a/1b): the position-major / rectangle-ordered / `rchain` launches of ContextAEReal at 36x64 with the reference's own
training batch 100 (ablations_code/ablations.py:503,536-544) and with 256, at 64x64 with 256; one GPU's share of BASELINE configs[3]:
ContextAEInception2 on 2x2x2048 maps (rllab/sampler/base.py:126) with 64 triples.

Per case: forward (four scalars, whole outputs of the kept triples, l1 / l2 digests of EVERY triple), every parameter gradient
(norm, up to 4096 sampled entries, 16 random-sign projections that cover every entry), then ONE Adam step: the update of the sampled
entries and the scalars of the pass after it.  Fixtures: tests/golden/make_golden.py `big`."""
import os

import numpy as np
import pytest

from oracle import ctx_oracle as o

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


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


def _open(T, kind, cfg, B):
    if kind == "real":
        return T(cfg.H, cfg.W, featsize=cfg.featsize, max_batch=B, variant="real")
    return T(cfg.H, cfg.W, df_dim=64, featsize=cfg.featsize, max_batch=B, variant="inception2", C=cfg.C)


@pytest.mark.parametrize("tag", ["real_f100_36x64_b100", "real_f100_36x64_b256", "real_f100_64x64_b256", "incep2_2x2x2048_f1024_b64"])
def test_bench_launch_shape_against_the_float64_fixture(T, tag):
    from tests.golden import make_golden as mg
    z = np.load(os.path.join(GOLD, tag + ".npz"))
    kind = mg.BIG_CASES[tag][0]
    mod, cfg, p32, (src, ctx, tgt) = mg.big_case(tag)
    np.testing.assert_allclose(mg.digest(mod.flatten(p32, cfg))[0], z["param_digest"], rtol=1e-12)         # RNG drift guards
    np.testing.assert_allclose(np.stack([mg.digest(a)[0] for a in (src, ctx, tgt)]), z["input_digest"], rtol=1e-12)
    B, lr = int(z["B"]), float(z["lr"])
    keep = list(z["keep"])
    names = [n for n, _ in mod.param_specs(cfg)]
    with _open(T, kind, cfg, B) as tr:
        assert tr.n_params == mod.param_count(cfg)
        tr.set_params(p32)
        ev = tr.evaluate(src, ctx, tgt)
        np.testing.assert_allclose([ev[k] for k in ("loss", "simloss", "recon1", "recon2")], z["scalars"], rtol=1e-5)
        iz, tz = tr.last_codes()
        for got, k in ((ev["out"], "out"), (ev["out2"], "out2"), (tz, "translated_z"), (iz, "input_z")):
            assert relmax(got[keep], z[k + "_keep"]) < 1e-5, k
            flat = np.asarray(got, np.float64).reshape(B, -1)
            rows = np.stack([np.abs(flat).sum(1), np.sqrt((flat * flat).sum(1))], 1)
            np.testing.assert_allclose(rows, z[k + "_rows"][:, 1:], rtol=1e-5, err_msg=k)                 # every triple
        p_before = tr.get_params_flat().astype(np.float64)
        sc = tr.train_step(src, ctx, tgt, lr=lr)                                                        # Adam step 1
        assert abs(sc["loss"] - z["train_scalars"][0][0]) <= 1e-5 * z["train_scalars"][0][0]
        # lrelu' branch report (as tests/test_gpu_baseline_configs.py): counted, not aligned away
        gold = {str(n): (int(a), int(b), int(c)) for n, a, b, c in zip(z["act_names"], z["act_negative"], z["act_near_zero"], z["act_size"])}
        delta = {}
        for buf in ("a0", "a1", "a2", "a3", "a4", "th0", "dz", "e1", "e2", "e3"):
            neg, near, size = gold[buf]
            if buf in ("a4", "th0"):                                                                   # code-wide rows at a stride of featsize rounded up to 32
                F = cfg.featsize
                Fp = -(-F // 32) * 32
                got = tr.debug_read(buf, size // F * Fp).reshape(-1, Fp)[:, :F]
            else:
                got = tr.debug_read(buf, size)
            delta[buf] = (int((got < 0).sum()) - neg, near)
        print(f"{tag}: lrelu' branch report (buffer: net sign changes vs float64, candidates within 1e-6 of zero):", delta)
        assert all(abs(d) <= max(8, near) for d, near in delta.values()), delta
        gg = tr.get_grads()
        probes = mg.big_probes([(n, int(np.prod(gg[n].shape))) for n in names], tag)
        report = {}
        for i, n in enumerate(names):
            a = np.asarray(gg[n], np.float64).reshape(-1)
            seed, idx = probes[n]
            ref_s = z["grad_samples"][i][: len(idx)]
            gnorm = z["grad_digest"][i][2]
            samp = float(np.linalg.norm(a[idx] - ref_s) / (np.linalg.norm(ref_s) + 1e-30))
            proj = float(np.sqrt(np.mean((mg.big_project(a, seed) - z["grad_proj"][i]) ** 2)) / gnorm)
            nrm = abs(float(np.sqrt((a * a).sum())) - gnorm) / gnorm
            report[n] = (samp, proj, nrm)
        print(f"{tag}: gradient deviation per tensor vs the fixture (rel-L2 on the samples, projected rel-L2 of the whole tensor, |norm| deviation):",
              {k: tuple(float(f"{x:.1e}") for x in v) for k, v in report.items()})
        for n, (samp, proj, nrm) in report.items():
            tight = n.startswith("deconv/d_h4")                                                        # upstream of every lrelu' mask
            assert samp <= (1e-5 if tight else 1e-3), (n, samp, proj, nrm)                             # north_star's budget
            assert proj <= (1e-5 if tight else 1.5e-3), (n, samp, proj, nrm)                           # 16 projections: +-35 % on the estimate
            assert nrm <= (1e-5 if tight else 1e-3), (n, samp, proj, nrm)
        # the Adam update itself (train_script.py:128,163), where the first-step gradient is above the noise floor of its tensor
        d_got = tr.get_params_flat().astype(np.float64) - p_before
        off, nchk = 0, 0
        for i, (n, shape) in enumerate(mod.param_specs(cfg)):
            size = int(np.prod(shape))
            idx = probes[n][1]
            want = z["update_samples"][i][: len(idx)]
            ok = np.isfinite(want)
            got = d_got[off:off + size][idx]
            # p - lr*m_hat/(sqrt(v_hat)+eps) rounded to f32: the update is seen through the parameter's own ulp
            tol = 2e-3 * lr + 2.0 ** -23 * np.abs(p_before[off:off + size][idx])
            assert np.all(np.abs(got - want)[ok] <= tol[ok]), n
            nchk += int(ok.sum())
            off += size
        assert nchk > 1000
        sc2 = tr.train_step(src, ctx, tgt, lr=lr)                                                       # scalars of the pass after the update
        want = z["train_scalars"][1]
        for j, k in enumerate(("loss", "simloss", "recon1", "recon2")):
            assert abs(sc2[k] - want[j]) <= 2e-4 * abs(want[j]), (k, sc2, want)


def test_real_36x64_batch100_gradients_against_the_branch_aligned_oracle(T):
    """The same launch shapes at the reference's training batch, gradients compared entry by entry: the float64 oracle is re-run here on
    the fixture's inputs with its saved activations given the device's sign at the (counted, bounded) elements within f32 rounding of
    zero -- tests/_align.py -- so that both sides differentiate lrelu on the same branch."""
    from tests._align import align_gen_cache
    from tests.golden import make_golden as mg
    tag = "real_f100_36x64_b100"
    mod, cfg, p32, (src, ctx, tgt) = mg.big_case(tag)
    B = src.shape[0]
    p = {k: v.astype(np.float64) for k, v in p32.items()}
    res, c = mod.forward(p, *(x.astype(np.float64) for x in (src, ctx, tgt)), cfg)
    with _open(T, "real", cfg, B) as tr:
        tr.set_params(p32)
        sc = tr.train_step(src, ctx, tgt, lr=0.0)
        assert abs(sc["loss"] - res["loss"]) <= 1e-5 * res["loss"]
        nflip, worst, where = align_gen_cache(tr, c, B)
        print(f"aligned {nflip} activations (largest |x| / max|x| among them {worst:.1e}) in {where}")
        assert nflip <= 64 and worst <= 1e-5
        g = mod.backward(p, c, cfg)
        gg = tr.get_grads()
        worst_t = {n: relmax(gg[n], g[n]) for n in g}
        print("gradients vs the branch-aligned oracle, max-norm relative:", {k: float(f"{v:.1e}") for k, v in worst_t.items()})
        assert max(worst_t.values()) <= 1e-4, worst_t
