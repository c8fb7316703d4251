"""The N > 1 path on CPU: world_size-2 gloo groups drive imitation_from_observation_amd.dp.
DataParallelTrainer exactly as bench.py does on RCCL, with a stand-in engine built on the oracle
(the HIP engine needs a GPU).  Checks SURVEY.md 8e: summed shard gradients == full-batch gradient
(simloss divided by the GLOBAL batch), replicas stay identical, global scalars are right."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ctx_oracle as o

CFG = o.SkipNewConfig(H=16, W=16, df_dim=4, gf_dim=4, featsize=8)


class OracleEngine:
    """Same interface as dp.HipEngine (params / grads flat tensors, forward_backward, adam,
    scalars_tensor), arithmetic from the float64 oracle."""

    def __init__(self, seed, ablation="None"):
        import types
        self.ablation = ablation
        # what dp.DataParallelTrainer reads the loss switch from: Translator.cfg.loss_terms (CTX_LOSS_* bits; 0 = all)
        bits = {"None": 0, "L2": 3, "L2L3": 1, "L1": 6}[ablation]
        self.translator = types.SimpleNamespace(cfg=types.SimpleNamespace(loss_terms=bits))
        self.p = o.init_params(CFG, seed, np.float64, stddev=0.2)
        self.n_params = o.param_count(CFG)
        self.params = torch.from_numpy(o.flatten(self.p, CFG).copy())
        self.grads = torch.zeros_like(self.params)
        self.m = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.t = 0
        self._sc = torch.zeros(4, dtype=torch.float64)

    def forward_backward(self, src, ctx, tgt, sim_batch, bucket_cb=None):
        self.p = o.unflatten(self.params.numpy().copy(), CFG)
        res, c = o.forward(self.p, src.numpy(), ctx.numpy(), tgt.numpy(), CFG, ablation_type=self.ablation)
        g = o.backward(self.p, c, CFG, sim_batch=sim_batch)
        self.grads.copy_(torch.from_numpy(o.flatten(g, CFG)))
        if bucket_cb is not None:           # like the HIP engine: the tail of the arena (translate/*, deconv/*) is announced first
            first = sum(int(np.prod(sh)) for n, sh in o.param_specs(CFG) if n.startswith("conv"))
            bucket_cb(first, self.n_params - first)
        self._sc = torch.tensor([res["loss"], res["simloss"], res["recon1"], res["recon2"]], dtype=torch.float64)

    def adam(self, lr):
        self.t += 1
        g = o.unflatten(self.grads.numpy().copy(), CFG)
        o.adam_step(self.p, g, self.m, self.v, self.t, lr)
        self.params.copy_(torch.from_numpy(o.flatten(self.p, CFG)))

    def scalars_tensor(self):
        return self._sc


def _data(B):
    rng = np.random.default_rng(42)
    return [torch.from_numpy(rng.uniform(-1, 1, (B, 16, 16, 3))) for _ in range(3)]


def _worker(rank, world, port, q, overlap="0", ablation="None"):
    os.environ["CTX_DP_OVERLAP"] = overlap          # "1": two buckets, the tail announced from inside the backward pass
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imitation_from_observation_amd.dp import DataParallelTrainer
    B = 4
    src, ctx, tgt = _data(B)
    sh = slice(rank * B // world, (rank + 1) * B // world)
    # different seeds per rank: the constructor's broadcast must make replicas identical
    tr = DataParallelTrainer(engine=OracleEngine(seed=7 + rank, ablation=ablation))
    p0 = tr.engine.params.clone()
    tr.step(src[sh], ctx[sh], tgt[sh], lr=1e-3)
    g1 = tr.engine.grads.clone()
    sc = tr.scalars()
    tr.step(src[sh], ctx[sh], tgt[sh], lr=1e-3)
    q.put((rank, p0.numpy(), g1.numpy(), tr.engine.params.numpy().copy(), sc))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
@pytest.mark.parametrize("overlap,ablation", [("0", "None"), ("1", "None"), ("0", "L1"), ("0", "L2L3")])
def test_two_rank_data_parallel_equals_full_batch(overlap, ablation):
    """ablation != "None": the loss switch of ablations_code/ablations.py:175-182 -- the global `loss` is rebuilt from the reduced
    terms it keeps (dp.py: _scalars), the gradients are those of that loss."""
    world = 2
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q, overlap, ablation)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process full-batch reference with rank 0's initial parameters
    ref = OracleEngine(seed=7, ablation=ablation)
    src, ctx, tgt = _data(4)
    np.testing.assert_array_equal(got[0][1], got[1][1])                 # broadcast made replicas equal
    np.testing.assert_array_equal(got[0][1], ref.params.numpy())
    ref.forward_backward(src, ctx, tgt, sim_batch=4)
    for r in range(world):
        np.testing.assert_allclose(got[r][2], ref.grads.numpy(), rtol=1e-9, atol=1e-12)   # summed grads == full batch
    sc_ref = ref.scalars_tensor().numpy()
    for r in range(world):
        np.testing.assert_allclose([got[r][4][k] for k in ("loss", "simloss", "recon1", "recon2")], sc_ref, rtol=1e-10)
    ref.adam(1e-3)
    ref.forward_backward(src, ctx, tgt, sim_batch=4)
    ref.adam(1e-3)
    for r in range(world):
        np.testing.assert_allclose(got[r][3], ref.params.numpy(), rtol=1e-9, atol=1e-12)   # two steps later
    np.testing.assert_array_equal(got[0][3], got[1][3])                  # replicas still bit-identical


# ------------------------------------------------------------------------------------------------ reward path
def _reward_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imitation_from_observation_amd.reward import TranslatorReward
    from tests.test_reward import CFG as RCFG, OracleTranslator, make_world
    p, validdata, paths = make_world(nvp=2, nvid=7, npaths=2, seed=5)
    first = [img for img in paths[0]["env_infos"]["imgs"] if img is not None][0]
    hook = TranslatorReward(OracleTranslator(p, 50), nvp=2, scale=0.01).build_demo_cache(validdata, first, distributed=True)
    # the per-path costs sharded the same way (base.py:232-257): 5 paths on 2 ranks, one all-reduce, then the rewards
    _, _, paths5 = make_world(nvp=2, nvid=7, npaths=5, seed=5)
    costs = hook.paths_costs(paths5, distributed=True)
    hook.process_paths(paths5, distributed=True)
    q.put((rank, [m.copy() for m in hook.means], [i.copy() for i in hook.imgs], costs, [p_["rewards"].copy() for p_ in paths5]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_demo_cache_sharded_over_ranks_equals_single_process():
    """SURVEY.md 8e (inference): demo videos sharded over ranks, partial sums combined with one all-reduce."""
    world = 2
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_reward_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from imitation_from_observation_amd.reward import TranslatorReward
    from tests.test_reward import OracleTranslator, make_world
    pr, validdata, paths = make_world(nvp=2, nvid=7, npaths=2, seed=5)
    first = [img for img in paths[0]["env_infos"]["imgs"] if img is not None][0]
    ref = TranslatorReward(OracleTranslator(pr, 50), nvp=2, scale=0.01).build_demo_cache(validdata, first)
    for r in range(world):
        for vp in range(2):
            np.testing.assert_allclose(got[r][1][vp], ref.means[vp], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(got[r][2][vp], ref.imgs[vp], rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(got[0][1][0], got[1][1][0])      # every rank ends with the same cache
    # sharded per-path costs = the single-process hook's, on every rank; rewards updated alike
    _, _, paths5 = make_world(nvp=2, nvid=7, npaths=5, seed=5)
    want = ref.paths_costs(paths5)
    ref.process_paths(paths5)
    for r in range(world):
        np.testing.assert_allclose(got[r][3], want, rtol=1e-5, atol=1e-7)
        for a, b in zip(got[r][4], (p_["rewards"] for p_ in paths5)):
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(got[0][3], got[1][3])
