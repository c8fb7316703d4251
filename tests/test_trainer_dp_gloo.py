"""`ModelTrainer(rank, world)` on TWO processes (gloo, CPU): the multi-GPU form of the reference's training loop
(scripts/train_script.py:144-203 + SURVEY.md 8e / 8f-3) with a stand-in translator that implements the data-parallel surface the
trainer drives -- load_demos / dp_world / dp_allreduce_host / dp_train_step_sampled / dp_eval_sampled / last_outputs / save -- on the
float64 oracle and a gloo group.  The claim: two ranks log what the single-process reference loop logs on the same np.random
stream (global scalars, nn_err summed over the ranks' shares), leave the same parameters on every rank, and only rank 0 writes
files.  The HIP form of the same surface (ctx_dp_train_step_sampled behind the C ABI) is tests/test_gpu_dp_two_ranks.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ctx_oracle as o
from tests.test_trainer import B, CFG, H, NITR, NLEN, NTRAIN, NVID, SAVE, W, make_vdata, reference_loop


class OracleDPModel:
    """One rank's replica on the oracle: gathers ITS rows of the global batch (t = b % T on the global row b), simloss mean over the
    global batch, SUM all-reduce of the gradients, identical Adam on every rank."""

    def __init__(self, seed, rank, world):
        self.rank, self.world = rank, world
        self.p = o.init_params(CFG, seed, np.float64, stddev=0.05)
        self.m = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.t = 0
        self.max_batch = B // world
        self.saved = []

    def load_demos(self, u8):
        self.demos = u8.astype(np.float64) / 127.5 - 1.0

    def dp_world(self):
        return self.rank, self.world

    def dp_allreduce_host(self, x):
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def _shard(self, cs, ct):
        Bg = len(cs)
        Bl = Bg // self.world
        rows = np.arange(self.rank * Bl, (self.rank + 1) * Bl)
        T = self.demos.shape[0]
        return (self.demos[rows % T, np.asarray(cs)[rows]], self.demos[0, np.asarray(ct)[rows]], self.demos[rows % T, np.asarray(ct)[rows]]), Bg

    def _global(self, res):
        s = self.dp_allreduce_host(np.array([res["simloss"], res["recon1"], res["recon2"]]))
        sim = s[0] / self.world
        return dict(loss=float(sim + s[1] + s[2]), simloss=float(sim), recon1=float(s[1]), recon2=float(s[2]))

    def dp_train_step_sampled(self, cs, ct, lr):
        (src, ctx, tgt), Bg = self._shard(cs, ct)
        res, c = o.forward(self.p, src, ctx, tgt, CFG)
        g = o.flatten(o.backward(self.p, c, CFG, sim_batch=Bg), CFG)
        g = o.unflatten(self.dp_allreduce_host(g), CFG)
        self.t += 1
        o.adam_step(self.p, g, self.m, self.v, self.t, lr)
        self._last = (res["out"], tgt)
        return self._global(res)

    def dp_eval_sampled(self, cs, ct):
        (src, ctx, tgt), _ = self._shard(cs, ct)
        res, _ = o.forward(self.p, src, ctx, tgt, CFG)
        ev = self._global(res)
        ev["out"], ev["out2"] = res["out"], res["out2"]
        return ev

    def last_outputs(self, out=True, out2=False, tgt=False):
        return self._last[0], None, self._last[1]

    def save(self, path, prefix=""):
        self.saved.append(path)
        np.savez(path + ".npz", **{prefix + k: v for k, v in self.p.items()})


def _worker(rank, world, port, q, base):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imitation_from_observation_amd.trainer import ModelTrainer
    vdata = make_vdata()
    np.random.seed(7 if rank == 0 else 1234 + rank)                 # only rank 0's stream counts: the trainer hands it to the others
    lines = []
    model = OracleDPModel(3, rank, world)
    ModelTrainer((H, W), NVID, NTRAIN, B, "ContextSkipNew", NITR, SAVE, NLEN, 1, vdata=vdata, basedir=base, translator=model,
                 log=lines.append, rank=rank, world=world).train()
    q.put((rank, lines, o.flatten(model.p, CFG), model.saved, int(np.random.randint(1 << 30))))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(600)
def test_two_rank_trainer_equals_the_single_process_reference_loop(tmp_path):
    world = 2
    base = str(tmp_path / "dp") + "/"
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q, base)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the single-process reference loop on the oracle, same np.random stream as rank 0
    vdata = make_vdata()
    np.random.seed(7)
    ref, lines, rows, validloss = reference_loop(vdata, seed=3, basedir=base)
    after_ref = int(np.random.randint(1 << 30))
    (r0, lines0, p0, saved0, rng0), (r1, lines1, p1, saved1, rng1) = got
    assert lines1 == [] and saved1 == []                            # one log, one set of checkpoints: rank 0's
    assert rng0 == rng1 == after_ref                                # every rank drew rank 0's batches and is left where the reference is
    np.testing.assert_array_equal(p0, p1)                           # replicas identical
    np.testing.assert_allclose(p0, o.flatten(ref.p, CFG), rtol=1e-9, atol=1e-12)
    assert len(lines0) == len(lines) + 3
    for a, b in zip(lines0[3:], lines):
        fa, fb = a.split(), b.split()
        assert fa[0] == fb[0] and fa[5:] == fb[5:], (a, b)          # iteration, nn_err (integer: summed shares), the "E" tag
        np.testing.assert_allclose([float(x) for x in fa[1:5]], [float(x) for x in fb[1:5]], rtol=1e-9)
    assert [os.path.basename(s)[:9] for s in saved0] == [os.path.basename(s)[:9] for s in ref.saved]
    assert sorted(os.listdir(base)) == sorted(["20", "40", "progress.csv", "vdata_train.npy"])
    from tests.test_trainer import read_clip
    clip = read_clip(base + "20/__0trans.gif")
    assert clip.dtype == np.uint8 and clip.shape == (NLEN, H, W, 3)
    # the clip is rows 0..nlen-1 of the global batch: with B = 6 on two ranks they all live on rank 0 -- and at nlen = 3 = B / world
    # exactly; the gather through the all-reduce must reproduce what one process saves
    np.random.seed(7)
    one = str(tmp_path / "one") + "/"
    from imitation_from_observation_amd.trainer import ModelTrainer
    from tests.test_trainer import OracleModel
    ModelTrainer((H, W), NVID, NTRAIN, B, "ContextSkipNew", NITR, SAVE, NLEN, 1, vdata=vdata, basedir=one, translator=OracleModel(3),
                 log=lambda s: None).train()
    for kk in range(10):
        for tag in ("trans", "recon"):
            np.testing.assert_array_equal(read_clip(f"{base}40/__{kk}{tag}.gif"), read_clip(f"{one}40/__{kk}{tag}.gif"))


def _videos_worker(rank, world, port, q, base):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imitation_from_observation_amd.trainer import ModelTrainer
    rng = np.random.default_rng(9)
    videos = [rng.integers(1, 256, (51, 24, 24, 3), dtype=np.uint8) for _ in range(NVID + 3)]    # more videos than nvideos: a SUBSET is chosen
    np.random.seed(7 if rank == 0 else 1234 + rank)                 # unsynchronised streams, as separate processes have
    model = OracleDPModel(3, rank, world)
    lines = []
    ModelTrainer((H, W), NVID, NTRAIN, B, "ContextSkipNew", 6, 5, NLEN, 17, vdata=None, videos=videos, basedir=base, translator=model,
                 log=lines.append, rank=rank, world=world).train()
    q.put((rank, model.demos.copy(), o.flatten(model.p, CFG), lines))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_trainer_builds_the_same_demo_tensor_from_videos(tmp_path):
    """ModelTrainer(videos=..., world=2): build_vdata's np.random.shuffle(videos) (train_script.py:66) picks the video order, the subset
    and the train / valid split -- rank 0's np.random state must reach the other ranks BEFORE it, or the ranks train on different
    demo tensors with shared index arrays.  Both ranks must hold rank 0's tensor = the single-process one."""
    world = 2
    base = str(tmp_path / "dpv") + "/"
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_videos_worker, args=(r, world, port, q, base)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, d0, p0, lines0), (_, d1, p1, lines1) = got
    np.testing.assert_array_equal(d0, d1)                           # the same videos, in the same order, on every rank
    np.testing.assert_array_equal(p0, p1)
    assert lines1 == []
    # the single-process pipeline on rank 0's stream
    from imitation_from_observation_amd.demo_pipeline import build_vdata
    rng = np.random.default_rng(9)
    videos = [rng.integers(1, 256, (51, 24, 24, 3), dtype=np.uint8) for _ in range(NVID + 3)]
    np.random.seed(7)
    vdata = build_vdata(videos, (H, W), NVID, NLEN, 17, True, False)
    np.testing.assert_allclose(d0, np.asarray(vdata, np.float64)[:NLEN], atol=1e-6)
    saved = [f for f in os.listdir(base) if f.startswith("vdata_strike")]
    assert len(saved) == 1                                          # rank 0 alone writes the tensor


def test_trainer_refuses_bad_data_parallel_arguments(tmp_path):
    from imitation_from_observation_amd.trainer import ModelTrainer
    with pytest.raises(ValueError, match="multiple of world"):
        ModelTrainer((H, W), NVID, NTRAIN, 7, "ContextSkipNew", 5, 5, NLEN, 1, rank=0, world=2)
    with pytest.raises(ValueError, match="rank"):
        ModelTrainer((H, W), NVID, NTRAIN, 6, "ContextSkipNew", 5, 5, NLEN, 1, rank=2, world=2)
