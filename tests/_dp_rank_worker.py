"""One rank of tests/test_gpu_dp_two_ranks.py (world 2, 4 or 8): a separate PROCESS that drives the C-ABI data-parallel path (ctx_dp_*) on its
shard.  usage: python tests/_dp_rank_worker.py <rank> <world> <workdir>   (CTX_RCCL_LIB = the stand-in, set by the parent)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

H = W = 32
D, F, SHARD, LR, PSEED = 32, 128, 8, 1e-3, 321
NLEN, NDEMO = 4, 6          # (e): the trainer's sampler -- demo tensor [NLEN, NDEMO, H, W, 3], global batch SHARD * world


def exchange_uid(work, tag, rank):
    """rank 0 makes a rendezvous blob (ctx_dp_unique_id) and publishes it through the work directory"""
    from imitation_from_observation_amd import Translator
    idfile = os.path.join(work, tag)
    if rank == 0:
        uid = Translator.dp_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
        return uid
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60:
            raise SystemExit(f"rank 0 never published {tag}")
        time.sleep(0.01)
    return open(idfile, "rb").read()


def demo_tensor():
    return np.random.default_rng(29).integers(0, 256, (NLEN, NDEMO, H, W, 3), dtype=np.uint8)


def sampled_choices(world, steps=3):
    rng = np.random.default_rng(31)
    return [(rng.integers(0, NDEMO, SHARD * world), rng.integers(0, NDEMO, SHARD * world)) for _ in range(steps)]


def full_batch(world):
    rng = np.random.default_rng(17)
    return [rng.uniform(-1, 1, (SHARD * world, H, W, 3)).astype(np.float32) for _ in range(3)]     # src, ctx, tgt


def main():
    rank, world, work = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch   # owner of the device buffers the shard lives in (plumbing)
    from imitation_from_observation_amd import Translator
    from oracle import ctx_oracle as o   # parameter initialiser only (so that the parent's oracle holds the same parameters)
    cfg = o.SkipNewConfig(H=H, W=W, df_dim=D, gf_dim=D, featsize=F)
    # rank 0 holds the parameters the parent's oracle uses; the other ranks start from DIFFERENT ones and a different Adam
    # step counter: ctx_dp_init must make every replica rank 0's
    p = o.init_params(cfg, PSEED + rank, np.float32, stddev=0.05)
    tr = Translator(H, W, D, F, max_batch=SHARD)
    tr.set_params(p)
    if rank:
        n = tr.n_params
        tr.set_adam_state(np.full(n, 0.5, np.float32), np.full(n, 0.25, np.float32), 7)
    idfile = os.path.join(work, "uid.bin")
    if rank == 0:
        uid = Translator.dp_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise SystemExit("rank 0 never published the unique id")
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    tr.dp_init(uid, rank, world)
    out = {"params0": tr.get_params_flat(), "adam_step0": np.int64(tr.get_adam_state()[2]), "dp_world": np.array(tr.dp_world())}
    sl = slice(rank * SHARD, (rank + 1) * SHARD)
    dev = [torch.from_numpy(np.ascontiguousarray(x[sl])).cuda() for x in full_batch(world)]
    torch.cuda.synchronize()
    ptr = [t.data_ptr() for t in dev]
    # (a) the exchange step alone between the phases: forward/backward with the GLOBAL simloss denominator -> all-reduce
    tr.dev_forward_backward(ptr[0], ptr[1], ptr[2], SHARD, sim_batch=SHARD * world)
    tr.dp_allreduce_grads()
    tr.sync()
    out["grads_phases"] = tr.get_grads_flat()
    out["scalars_phases"] = np.array(list(tr.dp_scalars().values()), np.float64)
    # (b) the whole step: two buckets, second stream, Adam -- three times
    for k in range(3):
        sc = tr.dp_train_step(ptr[0], ptr[1], ptr[2], SHARD, lr=LR, scalars=True)
        out[f"scalars{k + 1}"] = np.array([sc["loss"], sc["simloss"], sc["recon1"], sc["recon2"]], np.float64)
        if k == 0:
            tr.sync()
            out["grads1"] = tr.get_grads_flat()
    tr.sync()
    out["params3"] = tr.get_params_flat()
    out["adam_step3"] = np.int64(tr.get_adam_state()[2])
    # (c) the reward hook's demo cache, sharded over the SAME group (ctx_dp_allreduce_host_f64): 5 demo videos of 25 frames, 2 viewpoints
    from imitation_from_observation_amd.reward import TranslatorReward
    drng = np.random.default_rng(23)
    validdata = drng.integers(0, 256, (25, 5, H, W, 3), dtype=np.uint8).astype(np.float32) / 127.5 - 1.0
    first = [drng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(2)]
    trw = Translator(H, W, D, F, max_batch=50)          # the sampler-side translator (its own handle), same group
    trw.set_params_flat(out["params3"])
    idfile2 = os.path.join(work, "uid2.bin")
    if rank == 0:
        uid2 = Translator.dp_unique_id()
        with open(idfile2 + ".tmp", "wb") as f:
            f.write(uid2)
        os.rename(idfile2 + ".tmp", idfile2)
    else:
        t0 = time.time()
        while not os.path.exists(idfile2):
            if time.time() - t0 > 60:
                raise SystemExit("rank 0 never published the second unique id")
            time.sleep(0.01)
        uid2 = open(idfile2, "rb").read()
    trw.dp_init(uid2, rank, world)
    hook = TranslatorReward(trw, nvp=2, scale=0.01).build_demo_cache(validdata, first, distributed=True)
    out["cache_means"] = np.stack(hook.means)
    out["cache_imgs"] = np.stack(hook.imgs)
    # ... and the rollout paths of an iteration sharded rank::world over the same group (base.py:232-257): 5 paths, 2 viewpoints
    prng = np.random.default_rng(41)
    paths = []
    for _ in range(5):
        imgs = []
        for t in range(50):
            imgs.append([prng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(2)] if t % 2 == 0 else None)
        paths.append({"env_infos": {"imgs": imgs}, "rewards": np.zeros(50)})
    out["path_costs"] = hook.paths_costs(paths, distributed=True)
    if rank == 0:                                       # the same cache built by one rank alone
        solo = Translator(H, W, D, F, max_batch=50)
        solo.set_params_flat(out["params3"])
        h1 = TranslatorReward(solo, nvp=2, scale=0.01).build_demo_cache(validdata, first)
        out["solo_means"] = np.stack(h1.means)
        out["solo_imgs"] = np.stack(h1.imgs)
        out["solo_path_costs"] = h1.paths_costs(paths)
        solo.close()
    trw.close()
    # (d) the ablation script's loss switch under data parallelism (ablations_code/ablations.py:175-182; ctx_config.loss_terms): "L1" =
    # recon2 + simloss.  The GLOBAL `loss` must be made of those terms only (ctx_dp_scalars), the reduced gradient theirs.
    tra = Translator(H, W, D, F, max_batch=SHARD, ablation_type="L1")
    tra.set_params(p)
    idfile3 = os.path.join(work, "uid3.bin")
    if rank == 0:
        uid3 = Translator.dp_unique_id()
        with open(idfile3 + ".tmp", "wb") as f:
            f.write(uid3)
        os.rename(idfile3 + ".tmp", idfile3)
    else:
        t0 = time.time()
        while not os.path.exists(idfile3):
            if time.time() - t0 > 60:
                raise SystemExit("rank 0 never published the third unique id")
            time.sleep(0.01)
        uid3 = open(idfile3, "rb").read()
    tra.dp_init(uid3, rank, world)
    sc = tra.dp_train_step(ptr[0], ptr[1], ptr[2], SHARD, lr=0.0, scalars=True)
    out["abl_scalars"] = np.array([sc["loss"], sc["simloss"], sc["recon1"], sc["recon2"]], np.float64)
    out["abl_scalars_again"] = np.array(list(tra.dp_scalars().values()), np.float64)
    tra.sync()
    out["abl_grads"] = tra.get_grads_flat()
    tra.close()
    # (e) the trainer's loop body on two ranks (ctx_dp_train_step_sampled / ctx_dp_eval_sampled): every rank holds the demo tensor and is
    # handed the SAME global index arrays; it gathers its own rows of the global batch (t = b % T on the global row).  Rank 0 also runs
    # the single-handle ctx_train_step_sampled on the same arrays: the parameters must agree up to f32 summation order.
    trs = Translator(H, W, D, F, max_batch=SHARD)
    trs.set_params(p)
    trs.dp_init(exchange_uid(work, "uid4.bin", rank), rank, world)
    trs.load_demos(demo_tensor())
    choices = sampled_choices(world)
    # a global batch the ranks cannot split evenly is refused by EVERY rank before anything is enqueued (nobody is left in a collective)
    from imitation_from_observation_amd import CtxError
    try:
        trs.dp_train_step_sampled(choices[0][0][:-1], choices[0][1][:-1], lr=1e-4)
        out["ragged_refused"] = np.array(0)
    except CtxError as e:
        out["ragged_refused"] = np.array(int("multiple" in str(e)))
    ssc = []
    for cs, ct in choices:
        sc = trs.dp_train_step_sampled(cs, ct, lr=1e-4)
        ssc.append([sc["loss"], sc["simloss"], sc["recon1"], sc["recon2"]])
        if len(ssc) == 1:
            out["samp_grads1"] = trs.get_grads_flat()              # the reduced gradient of the first step (same parameters on both sides)
    out["samp_scalars"] = np.array(ssc, np.float64)
    trs.sync()
    out["samp_params3"] = trs.get_params_flat()
    ev = trs.dp_eval_sampled(*choices[0])
    out["samp_eval"] = np.array([ev["loss"], ev["simloss"], ev["recon1"], ev["recon2"]], np.float64)
    out["samp_eval_out"] = ev["out"]
    trs.close()
    if rank == 0:
        solo = Translator(H, W, D, F, max_batch=SHARD * world)
        solo.set_params(p)
        solo.load_demos(demo_tensor())
        ssc = []
        for cs, ct in choices:
            sc = solo.train_step_sampled(cs, ct, lr=1e-4)
            ssc.append([sc["loss"], sc["simloss"], sc["recon1"], sc["recon2"]])
            if len(ssc) == 1:
                out["solo_grads1"] = solo.get_grads_flat()
        out["solo_scalars"] = np.array(ssc, np.float64)
        out["solo_params3"] = solo.get_params_flat()
        ev = solo.eval_sampled(*choices[0])
        out["solo_eval"] = np.array([ev["loss"], ev["simloss"], ev["recon1"], ev["recon2"]], np.float64)
        out["solo_eval_out"] = ev["out"]
        solo.close()
    np.savez(os.path.join(work, f"rank{rank}.npz"), **out)
    tr.close()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
