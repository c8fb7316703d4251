"""The C-ABI data-parallel path (ctx_dp_unique_id / ctx_dp_init / ctx_dp_allreduce_grads / ctx_dp_train_step / ctx_dp_scalars,
include/ctxtrans.h) executed by TWO, FOUR and EIGHT PROCESSES.  The gpurun boxes have one GPU and real RCCL will not put two ranks on one
device, so the collectives go through tests/fake_rccl (a shared-memory stand-in loaded via CTX_RCCL_LIB: asynchronous,
stream-ordered, fixed rank-order sums); everything else -- the two-bucket schedule, the second stream, the events, the global
simloss denominator, Adam behind the reduced gradients -- is the shipped code.  The three claims of tests/test_dp_gloo.py,
now for this path: summed shard gradients = full-batch gradient (float64 oracle, B = 2 x 8), replicas bit-identical after 3
steps, global scalars right.  Reference: none (the reference is single-device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ctx_oracle as o
from tests import _dp_rank_worker as wk

HERE = os.path.dirname(os.path.abspath(__file__))
FAKE = os.path.join(HERE, "fake_rccl", "libfakerccl.so")


def test_fake_rccl_exports_what_libctxtrans_binds():
    """CPU: the stand-in is built (by __graft_entry__.build()) and exports the eight symbols rccl_load() looks up."""
    if not os.path.exists(FAKE):
        subprocess.run(["make", "-C", os.path.dirname(FAKE)], check=True)
    # looked up in a CHILD process: the stand-in links the system HIP runtime, and loading that into the pytest process ahead of
    # torch's bundled one leaves a later GPU test of the same process without a device
    code = ("import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); "
            "missing = [s for s in sys.argv[2:] if not hasattr(lib, s)]; print(missing); sys.exit(1 if missing else 0)")
    syms = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclBroadcast", "ncclGroupStart",
            "ncclGroupEnd", "ncclGetErrorString"]
    r = subprocess.run([sys.executable, "-c", code, FAKE] + syms, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_n_processes_run_the_c_abi_data_parallel_step(tmp_path, world):
    """world = 2, 4 and 8 (BASELINE configs[2] runs on 4 GPUs, configs[3] / [4] on 8): the same claims for every rank count the driver's
    scaling run uses -- N processes on this one GPU, the demo cache and the path costs sharded rank::N (5 videos / 5 paths: at N = 8
    three ranks hold none and still meet the others in the collective)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert os.path.exists(FAKE), "tests/fake_rccl/libfakerccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ, CTX_RCCL_LIB=FAKE, FAKE_RCCL_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_rank_worker.py"), str(r), str(world), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for pr in procs:
            logs.append(pr.communicate(timeout=600)[0])
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]

    # ---- ctx_dp_init: every replica starts as rank 0 (parameters, Adam slots' effect, step counter)
    for r in range(world):
        np.testing.assert_array_equal(z[0]["params0"], z[r]["params0"])
        assert int(z[r]["adam_step0"]) == 0
        assert tuple(z[r]["dp_world"]) == (r, world)                                   # ctx_dp_world on every rank
        assert int(z[r]["ragged_refused"]) == 1                                        # B_global % world != 0: CTX_E_INVALID on every rank

    # ---- the oracle on the FULL batch with rank 0's parameters
    cfg = o.SkipNewConfig(H=wk.H, W=wk.W, df_dim=wk.D, gf_dim=wk.D, featsize=wk.F)
    p = {k: v.astype(np.float64) for k, v in o.init_params(cfg, wk.PSEED, np.float32, stddev=0.05).items()}
    np.testing.assert_array_equal(o.flatten(p, cfg).astype(np.float32), z[0]["params0"])
    src, ctx, tgt = (x.astype(np.float64) for x in wk.full_batch(world))
    res, c = o.forward(p, src, ctx, tgt, cfg)
    g = o.flatten(o.backward(p, c, cfg), cfg)
    want = np.array([res["loss"], res["simloss"], res["recon1"], res["recon2"]])

    # ---- claim 1: summed shard gradients = full-batch gradient; the bucketed step and the plain exchange agree bit for bit
    for r in range(world):
        np.testing.assert_array_equal(z[r]["grads_phases"], z[r]["grads1"])
    for r in range(1, world):
        np.testing.assert_array_equal(z[0]["grads1"], z[r]["grads1"])
    off = 0
    for name, shape in o.param_specs(cfg):
        n = int(np.prod(shape))
        a, b = z[0]["grads1"][off:off + n].astype(np.float64), g[off:off + n]
        assert np.abs(a - b).max() <= 1e-3 * np.abs(b).max() + 1e-12, (name, np.abs(a - b).max() / np.abs(b).max())
        off += n
    # ---- claim 3: global scalars (sum of recon sums, mean of simloss means)
    for key in ("scalars_phases", "scalars1"):
        np.testing.assert_allclose(z[0][key], want, rtol=2e-5)
        for r in range(1, world):
            np.testing.assert_array_equal(z[0][key], z[r][key])
    # ---- claim 2: replicas bit-identical after three steps, and on the oracle's Adam trajectory
    for r in range(1, world):
        np.testing.assert_array_equal(z[0]["params3"], z[r]["params3"])
        assert int(z[0]["adam_step3"]) == int(z[r]["adam_step3"]) == 3
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    traj = []
    for t in range(1, 4):
        r_, _ = o.train_step(p, m, v, t, src, ctx, tgt, wk.LR, cfg)
        traj.append([r_["loss"], r_["simloss"], r_["recon1"], r_["recon2"]])
    for k in range(3):
        got = z[0][f"scalars{k + 1}"]
        np.testing.assert_allclose(got[[0, 2, 3]], np.array(traj[k])[[0, 2, 3]], rtol=1e-4)
        np.testing.assert_allclose(got[1], traj[k][1], rtol=2e-3)      # simloss (~3 of 16.5 k): a mean of small code differences, the
                                                                       # most sensitive scalar to f32 rounding in the +-lr Adam updates
    assert traj[2][0] < traj[0][0]                                                  # the steps moved the loss
    delta = z[0]["params3"].astype(np.float64) - z[0]["params0"]
    want_delta = o.flatten(p, cfg) - z[0]["params0"]
    # Adam's first steps are ~ +-lr per entry: an entry whose f32 gradient differs in the last bits from the f64 one may move the other
    # way, so the update is judged as a whole (direction and size), not entry by entry
    assert np.corrcoef(delta, want_delta)[0, 1] > 0.999
    assert abs(np.linalg.norm(delta) / np.linalg.norm(want_delta) - 1.0) < 1e-2
    # ---- the reward hook's demo cache sharded over the handle's own group (ctx_dp_allreduce_host_f64, no torch.distributed):
    # both ranks hold the same cache, and it is the one a single rank builds from all videos (f64 sums in another order)
    for r in range(1, world):
        np.testing.assert_array_equal(z[0]["cache_means"], z[r]["cache_means"])
        np.testing.assert_array_equal(z[0]["cache_imgs"], z[r]["cache_imgs"])
    np.testing.assert_allclose(z[0]["cache_means"], z[0]["solo_means"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(z[0]["cache_imgs"], z[0]["solo_imgs"], rtol=1e-6, atol=1e-7)
    assert np.abs(z[0]["cache_means"]).max() > 0
    # ---- the loss switch "L1" (recon2 + simloss) on two ranks: the global `loss` holds those terms only (ctx_dp_scalars
    # used to rebuild simloss + recon1 + recon2 whatever the switch), and the reduced gradient is the oracle's for that loss
    p0 = {k: v.astype(np.float64) for k, v in o.init_params(cfg, wk.PSEED, np.float32, stddev=0.05).items()}
    ares, ac = o.forward(p0, src, ctx, tgt, cfg, ablation_type="L1")
    awant = np.array([ares["loss"], ares["simloss"], ares["recon1"], ares["recon2"]])
    assert abs(ares["loss"] - (ares["recon2"] + ares["simloss"])) <= 1e-12 * ares["loss"]
    for key in ("abl_scalars", "abl_scalars_again"):
        np.testing.assert_allclose(z[0][key], awant, rtol=2e-5)
        for r in range(1, world):
            np.testing.assert_array_equal(z[0][key], z[r][key])
    assert abs(z[0]["abl_scalars"][0] - (z[0]["abl_scalars"][3] + z[0]["abl_scalars"][1])) <= 1e-6 * z[0]["abl_scalars"][0]
    ag = o.flatten(o.backward(p0, ac, cfg), cfg)
    for r in range(1, world):
        np.testing.assert_array_equal(z[0]["abl_grads"], z[r]["abl_grads"])
    off = 0
    for name, shape in o.param_specs(cfg):
        n = int(np.prod(shape))
        a, b = z[0]["abl_grads"][off:off + n].astype(np.float64), ag[off:off + n]
        assert np.abs(a - b).max() <= 1e-3 * np.abs(b).max() + 1e-12, (name, np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
        off += n
    # ---- the trainer's sampled step on two ranks (ctx_dp_train_step_sampled: each rank gathers its rows of the global batch from its
    # resident demo tensor) against ONE handle's ctx_train_step_sampled on the same index arrays
    for r in range(1, world):
        np.testing.assert_array_equal(z[0]["samp_params3"], z[r]["samp_params3"])                # replicas bit-identical
        np.testing.assert_array_equal(z[0]["samp_scalars"], z[r]["samp_scalars"])
    np.testing.assert_allclose(z[0]["samp_scalars"][:, [0, 2, 3]], z[0]["solo_scalars"][:, [0, 2, 3]], rtol=2e-5)
    np.testing.assert_allclose(z[0]["samp_scalars"][:, 1], z[0]["solo_scalars"][:, 1], rtol=2e-3)  # simloss: see above
    # (1) the reduced gradient of the FIRST step (both sides hold the same parameters): two 8-triple sums added by the all-reduce against
    # one 16-triple sum -- f32 sums in another order, per tensor 1e-5 of its largest entry
    ga, gb = z[0]["samp_grads1"].astype(np.float64), z[0]["solo_grads1"].astype(np.float64)
    for r in range(1, world):
        np.testing.assert_array_equal(z[0]["samp_grads1"], z[r]["samp_grads1"])
    off = 0
    for name, shape in o.param_specs(cfg):
        n = int(np.prod(shape))
        e = np.abs(ga[off:off + n] - gb[off:off + n]).max() / (np.abs(gb[off:off + n]).max() + 1e-30)
        assert e <= 1e-5, (name, e)
        off += n
    # (2) the parameters after three steps.  Adam's first steps move every entry by ~lr whatever the size of its gradient, so an entry
    # whose gradient sits at the rounding floor of its sum (on these smooth demo frames: many) may move the other way on the two sides,
    # and the next steps see slightly different models: the agreement after three steps is a statement about that cascade, not about
    # the kernels, and it moves with the rounding of a single FMA (the two contractions of b1 m + (1 - b1) g in adam_kernel: 1.2e-6 /
    # 33 k entries beyond 1e-6 against 9.6e-6 / 729 k, both within 2e-7 / 82 entries of the float64 oracle on random frames).  Bounds:
    # parameters to 5e-5 relative (L2), no entry further apart than the 6e-4 of three opposite steps, the update itself to 1e-2.
    a, b = z[0]["samp_params3"].astype(np.float64), z[0]["solo_params3"].astype(np.float64)
    dev = np.abs(a - b)
    print("sampled DP vs one handle after 3 steps: max |dp| / max |p| =", dev.max() / np.abs(b).max(), " rel-L2 =", np.linalg.norm(a - b) / np.linalg.norm(b),
          " entries beyond 1e-6 of max |p|:", int((dev > 1e-6 * np.abs(b).max()).sum()), "of", a.size)
    upd_a, upd_b = a - z[0]["params0"].astype(np.float64), b - z[0]["params0"].astype(np.float64)
    print("   update agreement |d_dp - d_solo| / |d_solo| =", np.linalg.norm(upd_a - upd_b) / np.linalg.norm(upd_b))
    assert np.linalg.norm(a - b) <= 5e-5 * np.linalg.norm(b)
    assert dev.max() <= 6.0 * 1e-4
    assert np.linalg.norm(upd_a - upd_b) <= 1e-2 * np.linalg.norm(upd_b)
    # the sharded validation fetch: global scalars, and the two ranks' rows side by side = the single handle's outputs -- of two models
    # that are 1e-5 apart after the three steps above (the cascade), hence 1e-3 here
    for r in range(1, world):
        np.testing.assert_array_equal(z[0]["samp_eval"], z[r]["samp_eval"])
    both = np.concatenate([z[r]["samp_eval_out"] for r in range(world)])
    assert both.shape == z[0]["solo_eval_out"].shape
    assert np.abs(both - z[0]["solo_eval_out"]).max() <= 1e-3 * np.abs(z[0]["solo_eval_out"]).max()
    np.testing.assert_allclose(z[0]["samp_eval"][[0, 2, 3]], z[0]["solo_eval"][[0, 2, 3]], rtol=1e-3)
    # ---- the per-path costs of the reward hook sharded over the two ranks (TranslatorReward.paths_costs(distributed=True)): every
    # rank ends with the full [paths, 25] table, equal to what one rank computes for all paths
    for r in range(1, world):
        np.testing.assert_array_equal(z[0]["path_costs"], z[r]["path_costs"])
    assert z[0]["path_costs"].shape == (5, 25) and np.abs(z[0]["path_costs"]).max() > 0
    np.testing.assert_allclose(z[0]["path_costs"], z[0]["solo_path_costs"], rtol=1e-5, atol=1e-6)
