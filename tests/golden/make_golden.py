"""Generates the committed golden vectors under tests/golden/ from the float64 oracle.

The reference itself cannot run in the build container (TensorFlow absent), so these
are outputs of oracle/ctx_oracle.py -- which tests/test_oracle.py pins against an independent
torch-autograd statement -- NOT outputs of the reference.  Run:  python tests/golden/make_golden.py

Fixtures hold data only: seeds, uint8 input frames, outputs, gradient / parameter digests.
Parameters are regenerated in the tests from the recorded seed (a digest guards RNG drift).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ctx_oracle as o  # noqa: E402

N_HEAD = 64


def digest(a):
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())]), a[:N_HEAD].copy()


def synth_frames(seed, B, H, W):
    """SURVEY.md 8d synthetic inputs: iid uint8 frames, one seed per slot."""
    return np.random.default_rng(seed).integers(0, 256, (B, H, W, 3), dtype=np.uint8)


def blob_frames(seed, B, H, W):
    """Low-frequency 'rendered-looking' frames: sum of 4 random 2-D Gaussians per channel."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.zeros((B, H, W, 3))
    for b in range(B):
        for c in range(3):
            for _ in range(4):
                cy, cx, s, a = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(H / 8, H / 2), rng.uniform(0.2, 1)
                out[b, :, :, c] += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    out = out / out.max()
    return np.clip(np.round(out * 255), 0, 255).astype(np.uint8)


def make(tag, cfg, B, pseed, stddev, frames, steps, lr=1e-4):
    p = o.init_params(cfg, pseed, np.float64, stddev=stddev)
    # non-zero biases so every bias path is exercised
    brng = np.random.default_rng(pseed + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * stddev
    su8, cu8, tu8 = (frames(s, B, cfg.H, cfg.W) for s in (0, 1, 2))
    src, ctx, tgt = (o.preprocess_u8(x).astype(np.float64) for x in (su8, cu8, tu8))
    fx = dict(cfg=np.array([cfg.H, cfg.W, cfg.C, cfg.df_dim, cfg.featsize]), B=B, pseed=pseed, stddev=stddev,
              src_u8=su8, ctx_u8=cu8, tgt_u8=tu8, lr=lr, steps=steps)
    fx["param_digest"], _ = digest(o.flatten(p, cfg))
    res, c = o.forward(p, src, ctx, tgt, cfg)
    for k in ["input_z", "translated_z", "out", "out2"]:
        fx[k] = res[k].astype(np.float32)
    fx["scalars"] = np.array([res["loss"], res["simloss"], res["recon1"], res["recon2"]])
    g = o.backward(p, c, cfg)
    names = [n for n, _ in o.param_specs(cfg)]
    fx["grad_digest"] = np.stack([digest(g[n])[0] for n in names])
    fx["grad_head"] = np.stack([np.pad(digest(g[n])[1], (0, N_HEAD - min(N_HEAD, g[n].size))) for n in names])
    # inference call sites on the same frames
    pred, feat = o.translate({k: v.astype(np.float32) for k, v in p.items()}, su8, cu8[0], cfg)
    fx["translate_pred"], fx["translate_feat"] = pred.astype(np.float32), feat.astype(np.float32)
    ef, _ = o.encode({k: v.astype(np.float32) for k, v in p.items()}, su8, cfg)
    fx["encode_feat"] = ef.astype(np.float32)
    # Adam trajectory
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    traj = []
    p0 = o.flatten(p, cfg)
    for t in range(1, steps + 1):
        r, _ = o.train_step(p, m, v, t, src, ctx, tgt, lr, cfg)
        traj.append([r["loss"], r["simloss"], r["recon1"], r["recon2"]])
    fx["train_scalars"] = np.array(traj)
    delta = o.flatten(p, cfg) - p0
    fx["delta_digest"] = np.stack([digest(dd)[0] for dd in np.split(delta, np.cumsum(
        [int(np.prod(s)) for _, s in o.param_specs(cfg)])[:-1])])
    fx["delta_head"] = delta[:N_HEAD]
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(tag, os.path.getsize(path), "bytes; loss", res["loss"])


def make_real(tag, cfg, B, pseed, stddev, frames, steps, lr=1e-4):
    """Same contents for ContextAEReal (oracle/ctx_oracle_real.py)."""
    from oracle import ctx_oracle_real as r
    p = r.init_params(cfg, pseed, np.float64, stddev=stddev)
    brng = np.random.default_rng(pseed + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * stddev
    su8, cu8, tu8 = (frames(s, B, cfg.H, cfg.W) for s in (0, 1, 2))
    src, ctx, tgt = (o.preprocess_u8(x).astype(np.float64) for x in (su8, cu8, tu8))
    fx = dict(cfg=np.array([cfg.H, cfg.W, cfg.C, cfg.featsize]), B=B, pseed=pseed, stddev=stddev,
              src_u8=su8, ctx_u8=cu8, tgt_u8=tu8, lr=lr, steps=steps)
    fx["param_digest"], _ = digest(r.flatten(p, cfg))
    res, c = r.forward(p, src, ctx, tgt, cfg)
    for k in ["input_z", "translated_z", "out", "out2"]:
        fx[k] = res[k].astype(np.float32)
    fx["scalars"] = np.array([res["loss"], res["simloss"], res["recon1"], res["recon2"]])
    g = r.backward(p, c, cfg)
    names = [n for n, _ in r.param_specs(cfg)]
    fx["grad_digest"] = np.stack([digest(g[n])[0] for n in names])
    fx["grad_head"] = np.stack([np.pad(digest(g[n])[1], (0, N_HEAD - min(N_HEAD, g[n].size))) for n in names])
    pred, feat = r.translate(p, su8, cu8[0], cfg)
    fx["translate_pred"], fx["translate_feat"] = pred.astype(np.float32), feat.astype(np.float32)
    fx["encode_feat"] = r.encode(p, su8, cfg)[0].astype(np.float32)
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    traj, p0 = [], r.flatten(p, cfg)
    for t in range(1, steps + 1):
        rr, cc = r.forward(p, src, ctx, tgt, cfg)
        o.adam_step(p, r.backward(p, cc, cfg), m, v, t, lr)
        traj.append([rr["loss"], rr["simloss"], rr["recon1"], rr["recon2"]])
    fx["train_scalars"] = np.array(traj)
    delta = r.flatten(p, cfg) - p0
    fx["delta_digest"] = np.stack([digest(dd)[0] for dd in np.split(delta, np.cumsum(
        [int(np.prod(s)) for _, s in r.param_specs(cfg)])[:-1])])
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(tag, os.path.getsize(path), "bytes; loss", res["loss"])


def make_incep2(tag, cfg, B, pseed, stddev, fseed, steps, lr=1e-4):
    """Same contents for ContextAEInception2 (oracle/ctx_oracle_incep.py) on synthetic post-ReLU feature maps [B,h,w,C]:
    the class takes strides / kernels / filters (arm_shaping.py:1786-1803), so the fixture carries them."""
    from oracle import ctx_oracle_incep as ci
    p = ci.init_params(cfg, pseed, np.float64, stddev=stddev)
    brng = np.random.default_rng(pseed + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * stddev
    frng = np.random.default_rng(fseed)
    src, ctx, tgt = (np.maximum(frng.standard_normal((B, cfg.H, cfg.W, cfg.C)), 0.0).astype(np.float32).astype(np.float64) for _ in range(3))
    fx = dict(cfg=np.array([cfg.H, cfg.W, cfg.C, cfg.featsize]), strides=np.array(cfg.strides), kernels=np.array(cfg.kernels),
              filters=np.array(cfg.filters), B=B, pseed=pseed, stddev=stddev, fseed=fseed, lr=lr, steps=steps,
              src_f32=src.astype(np.float32), ctx_f32=ctx.astype(np.float32), tgt_f32=tgt.astype(np.float32))
    fx["param_digest"], _ = digest(ci.flatten(p, cfg))
    res, c = ci.forward(p, src, ctx, tgt, cfg)
    for k in ["input_z", "translated_z", "out", "out2"]:
        fx[k] = res[k].astype(np.float32)
    fx["scalars"] = np.array([res["loss"], res["simloss"], res["recon1"], res["recon2"]])
    g = ci.backward(p, c, cfg)
    names = [n for n, _ in ci.param_specs(cfg)]
    fx["grad_digest"] = np.stack([digest(g[n])[0] for n in names])
    fx["grad_head"] = np.stack([np.pad(digest(g[n])[1], (0, N_HEAD - min(N_HEAD, g[n].size))) for n in names])
    pred, feat = ci.translate(p, src, ctx[0], cfg)
    fx["translate_pred"], fx["translate_feat"] = pred.astype(np.float32), feat.astype(np.float32)
    fx["encode_feat"] = ci.encode(p, src, cfg).astype(np.float32)
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    traj = []
    for t in range(1, steps + 1):
        rr, cc = ci.forward(p, src, ctx, tgt, cfg)
        o.adam_step(p, ci.backward(p, cc, cfg), m, v, t, lr)
        traj.append([rr["loss"], rr["simloss"], rr["recon1"], rr["recon2"]])
    fx["train_scalars"] = np.array(traj)
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(tag, os.path.getsize(path), "bytes; loss", res["loss"])


def make_inception_v3(tag, pseed, fseed, B, S):
    """Inception-v3 to Mixed_7c (nets/inception_v3.py:93-416) on the oracle's synthetic variables (seeded; the checkpoint is not
    in the reference tree): the Mixed_7c feature maps whole, a digest + head of each of the 18 end points."""
    from oracle import inception_oracle as io
    p = io.init_params(pseed)
    frames = np.random.default_rng(fseed).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    ep = io.forward(p, o.preprocess_u8(frames).astype(np.float64))
    fx = dict(pseed=pseed, fseed=fseed, B=B, S=S, frames_u8=frames, endpoints=np.array(list(ep)),
              Mixed_7c=ep["Mixed_7c"].astype(np.float32),
              endpoint_digest=np.stack([digest(v)[0] for v in ep.values()]),
              endpoint_head=np.stack([digest(v)[1] for v in ep.values()]))
    fx["param_digest"], _ = digest(np.concatenate([np.asarray(v).reshape(-1) for v in p.values()]))
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(tag, os.path.getsize(path), "bytes")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "extra":
    from oracle import ctx_oracle_incep as ci
    # ContextAEInception2 as the unmodified reference class can be built: featsize 1024 (hard-coded, :1797), free strides /
    # kernels / filters.  One case with the sampler's [1,2,1,2] / [3,3,3,3], one with other strides and kernel sizes.
    make_incep2("incep2_4x4x64_f32_b2", ci.Incep2Config(H=4, W=4, C=64, filters=(32, 32, 32, 32)), 2, 555, 0.05, 9, steps=2)
    make_incep2("incep2_8x4x32_k5331_s2121_b2", ci.Incep2Config(H=8, W=4, C=32, strides=(2, 1, 2, 1), kernels=(5, 3, 3, 1),
                                                                 filters=(32, 64, 32, 32)), 2, 556, 0.05, 10, steps=2)
    make_inception_v3("inception_v3_125x125_b2", 0, 3, 2, 125)


if __name__ == "__main__" and len(sys.argv) == 1:
    from oracle import ctx_oracle_real as r
    # ContextAEReal at the reference's sweep imsize (run_trpo_sweep_ours.py:64)
    make_real("real_f100_36x64_b3", r.RealConfig(), 3, 4321, 0.05, blob_frames, steps=3)
    # reduced net the HIP kernels accept (channels multiples of 32), iid frames
    make("skipnew_d32_f128_32x32_b4", o.SkipNewConfig(H=32, W=32, df_dim=32, gf_dim=32, featsize=128), 4, 1234, 0.05,
         synth_frames, steps=3)
    # non-square, blob frames
    make("skipnew_d32_f128_16x48_b3", o.SkipNewConfig(H=16, W=48, df_dim=32, gf_dim=32, featsize=128), 3, 77, 0.05,
         blob_frames, steps=2)
    # the production net at the reference's init scale
    make("skipnew_d64_f1024_64x64_b2", o.SkipNewConfig(), 2, 1234, 0.02, synth_frames, steps=2)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[1] at its full size: the production net (47.6 M parameters), 64x64x3, batch 256.
# The float64 oracle runs the batch in chunks (triples are independent except through the parameter-gradient sums and
# the simloss mean, whose denominator is the global batch: `sim_batch`), so it needs ~10 GB and ~3 minutes here.
# The fixture holds what a 1-2 MB file can: seeds (frames and parameters are regenerated from them), the four scalars,
# a few whole output frames, per-image digests of out / out2 / translated_z / input_z, and for every parameter
# gradient its digest, 16 random-sign projections (they cover every entry), and 4096 sampled entries.
# For the lrelu' branch report: per activation buffer the number of negative entries and of entries within 1e-6 of zero
# (relative to the buffer's max) -- the candidates an f32 pass can put on the other side of the kink.
# ------------------------------------------------------------------------------------------------------------------
B256_TAG = "b256_skipnew_d64_f1024_64x64"   # (not "skipnew_*": the small-fixture tests glob that)
B256_SEEDS = dict(pseed=2024, fseed=(10, 11, 12), proj_seed=5, sample_seed=6)
B256_NPROJ, B256_NSAMP = 16, 4096
B256_KEEP = (0, 85, 170, 255)


def b256_case():
    """(cfg, params float64, uint8 frames) of the batch-256 fixture -- also used by the GPU test to rebuild the inputs."""
    cfg = o.SkipNewConfig()
    p = o.init_params(cfg, B256_SEEDS["pseed"], np.float64, stddev=0.02)
    brng = np.random.default_rng(B256_SEEDS["pseed"] + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * 0.02
    frames = [synth_frames(s, 256, cfg.H, cfg.W) for s in B256_SEEDS["fseed"]]
    return cfg, p, frames


def b256_probes(name_sizes):
    """Per gradient tensor: random-sign projection vectors are too big to keep, so they are regenerated from a seed per
    tensor; sampled entry indices likewise.  Returns {name: (seed_for_signs, sample_indices)}."""
    out = {}
    for i, (n, size) in enumerate(name_sizes):
        rs = np.random.default_rng(B256_SEEDS["sample_seed"] * 1000 + i)
        idx = np.sort(rs.choice(size, min(B256_NSAMP, size), replace=False))
        out[n] = (B256_SEEDS["proj_seed"] * 1000 + i, idx)
    return out


def b256_project(a, seed):
    """16 projections of the flat array on random +-1 vectors (generated 1 at a time: the tensors have up to 8.4 M entries)."""
    a = np.asarray(a, np.float64).reshape(-1)
    rng = np.random.default_rng(seed)
    return np.array([float(a @ (rng.integers(0, 2, a.size, dtype=np.int8) * 2.0 - 1.0)) for _ in range(B256_NPROJ)])


def make_b256(chunk=16):
    import time
    cfg, p, frames = b256_case()
    B = 256
    names = [n for n, _ in o.param_specs(cfg)]
    g = {n: np.zeros_like(p[n]) for n in names}
    scal = np.zeros(3)                                             # recon1, recon2, sum of squared code differences
    outs = {k: [] for k in ("out", "out2", "translated_z", "input_z")}
    act_stats = {}
    t0 = time.time()
    for i0 in range(0, B, chunk):
        sl = slice(i0, i0 + chunk)
        src, ctx, tgt = (o.preprocess_u8(f[sl]).astype(np.float64) for f in frames)
        res, c = o.forward(p, src, ctx, tgt, cfg)
        gi = o.backward(p, c, cfg, sim_batch=B)
        for n in names:
            g[n] += gi[n]
        scal += [res["recon1"], res["recon2"], float(np.sum((c["trans_z"] - c["e_tgt"][5]) ** 2))]
        for k in outs:
            outs[k].append(res[k])
        # lrelu'-relevant activations (the buffers whose sign the backward pass reads)
        bufs = {}
        for k in range(5):
            bufs[f"s{k}_tgt"], bufs[f"s{k}_src"], bufs[f"c{k}"] = c["e_tgt"][k], c["e_src"][k], c["e_ctx"][k]
        bufs["z_tgt"], bufs["z_src"], bufs["th0"] = c["e_tgt"][5], c["e_src"][5], c["trans_h0"]
        for k in range(4):
            bufs[f"d1_{k}"], bufs[f"d2_{k}"] = c["d1"][k], c["d2"][k]
        for k, a in bufs.items():
            st = act_stats.setdefault(k, [0, 0, 0.0])
            st[0] += int((a < 0).sum())
            st[2] = max(st[2], float(np.abs(a).max()))
        for k, a in bufs.items():                                  # second pass needs the running max: use this chunk's (close enough for a count)
            act_stats[k][1] += int((np.abs(a) <= 1e-6 * np.abs(a).max()).sum())
        print(f"chunk {i0 // chunk + 1}/{B // chunk}  {time.time() - t0:.0f}s", flush=True)
    sim = scal[2] / (B * cfg.featsize) * 1e3
    fx = dict(cfg=np.array([cfg.H, cfg.W, cfg.C, cfg.df_dim, cfg.featsize]), B=B, stddev=0.02,
              pseed=B256_SEEDS["pseed"], fseed=np.array(B256_SEEDS["fseed"]), proj_seed=B256_SEEDS["proj_seed"],
              sample_seed=B256_SEEDS["sample_seed"], keep=np.array(B256_KEEP))
    fx["param_digest"], _ = digest(o.flatten(p, cfg))
    fx["scalars"] = np.array([scal[0] + scal[1] + sim, sim, scal[0], scal[1]])
    for k in outs:
        full = np.concatenate(outs[k])
        fx[k + "_keep"] = full[list(B256_KEEP)].astype(np.float32)
        flat = full.reshape(B, -1)
        fx[k + "_rows"] = np.stack([flat.sum(1), np.abs(flat).sum(1), np.sqrt((flat * flat).sum(1))], 1)   # per-image digests
    probes = b256_probes([(n, g[n].size) for n in names])
    fx["grad_digest"] = np.stack([digest(g[n])[0] for n in names])
    fx["grad_proj"] = np.stack([b256_project(g[n], probes[n][0]) for n in names])
    fx["grad_samples"] = np.stack([np.pad(g[n].reshape(-1)[probes[n][1]], (0, B256_NSAMP - len(probes[n][1]))) for n in names])
    fx["act_names"] = np.array(sorted(act_stats))
    fx["act_negative"] = np.array([act_stats[k][0] for k in sorted(act_stats)], np.int64)
    fx["act_near_zero"] = np.array([act_stats[k][1] for k in sorted(act_stats)], np.int64)
    path = os.path.join(HERE, f"{B256_TAG}.npz")
    np.savez_compressed(path, **fx)
    print(B256_TAG, os.path.getsize(path), "bytes; loss", fx["scalars"][0], f"{time.time() - t0:.0f}s")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "b256":
    make_b256()


# ------------------------------------------------------------------------------------------------------------------
# Mode 'oursinception' at the REFERENCE's own size: 299x299 frames (sandbox/andrew/run_trpo_strike.py:84,
# run_train_strike_inception.py:39-43: idims=(299, 299), batch_size=25) -> Inception-v3 Mixed_7c 8x8x2048
# (rllab/sampler/base.py:121-132, nets/inception_v3_test.py:45-54) -> ContextAEInception2(strides [1,2,1,2], kernels [3,3,3,3],
# filters [1024,1024,512,512]), 153 M parameters, batch 25.
# Two files: the front end at 299x299 (2 frames: digests + heads of all 18 end points, Mixed_7c of frame 0 whole) and the translator at
# 8x8x2048, batch 25 (seeds, the four scalars, whole out / out2 maps of two triples, per-triple digests, and per gradient tensor a
# digest, 16 random-sign projections and 1024 sampled entries -- the b256 scheme; two Adam steps' scalars).
# Needs ~12 GB and a few minutes here:  python tests/golden/make_golden.py ref299
# ------------------------------------------------------------------------------------------------------------------
REF299_FRONT_TAG = "inception_v3_299x299_b2"
REF299_TAG = "incep2_8x8x2048_f1024_b25"
REF299_SEEDS = dict(pseed=3100, fseed=41, proj_seed=7, sample_seed=8)
REF299_NPROJ, REF299_NSAMP, REF299_B = 16, 1024, 25
REF299_KEEP = (24,)


def ref299_case():
    """(cfg, params float32-representable float64, [src, ctx, tgt] float32 post-ReLU maps) of the batch-25 fixture."""
    from oracle import ctx_oracle_incep as ci
    cfg = ci.Incep2Config(H=8, W=8)
    p = ci.init_params(cfg, REF299_SEEDS["pseed"], np.float32, stddev=0.02)
    brng = np.random.default_rng(REF299_SEEDS["pseed"] + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = (brng.standard_normal(p[n].shape) * 0.02).astype(np.float32)
    frng = np.random.default_rng(REF299_SEEDS["fseed"])
    feats = [np.maximum(frng.standard_normal((REF299_B, cfg.H, cfg.W, cfg.C)), 0.0).astype(np.float32) for _ in range(3)]
    return cfg, p, feats


def ref299_probes(name_sizes):
    out = {}
    for i, (n, size) in enumerate(name_sizes):
        rs = np.random.default_rng(REF299_SEEDS["sample_seed"] * 1000 + i)
        idx = np.sort(rs.choice(size, min(REF299_NSAMP, size), replace=False))
        out[n] = (REF299_SEEDS["proj_seed"] * 1000 + i, idx)
    return out


def ref299_project(a, seed):
    a = np.asarray(a, np.float64).reshape(-1)
    rng = np.random.default_rng(seed)
    return np.array([float(a @ (rng.integers(0, 2, a.size, dtype=np.int8) * 2.0 - 1.0)) for _ in range(REF299_NPROJ)])


def make_ref299():
    import time
    from oracle import ctx_oracle_incep as ci
    t0 = time.time()
    make_inception_v3_299()
    cfg, p32, feats = ref299_case()
    B = REF299_B
    p = {k: v.astype(np.float64) for k, v in p32.items()}
    src, ctx, tgt = (x.astype(np.float64) for x in feats)
    names = [n for n, _ in ci.param_specs(cfg)]
    fx = dict(cfg=np.array([cfg.H, cfg.W, cfg.C, cfg.featsize]), strides=np.array(cfg.strides), kernels=np.array(cfg.kernels),
              filters=np.array(cfg.filters), B=B, stddev=0.02, lr=1e-4, keep=np.array(REF299_KEEP),
              **{k: np.array(v) for k, v in REF299_SEEDS.items()})
    fx["param_digest"], _ = digest(ci.flatten(p, cfg))
    res, c = ci.forward(p, src, ctx, tgt, cfg)
    print(f"forward {time.time() - t0:.0f}s loss {res['loss']:.6g}", flush=True)
    fx["scalars"] = np.array([res["loss"], res["simloss"], res["recon1"], res["recon2"]])
    for k in ("out", "out2", "translated_z", "input_z"):
        fx[k + "_keep"] = res[k][list(REF299_KEEP)].astype(np.float32)
        flat = res[k].reshape(B, -1)
        fx[k + "_rows"] = np.stack([flat.sum(1), np.abs(flat).sum(1), np.sqrt((flat * flat).sum(1))], 1)
    # lrelu'-relevant activations in the device's buffer order (a_k = [tgt | src | ctx] over 3B images, e_k = [pass 1 | pass 2]): negative
    # entries and entries within 1e-6 of zero (relative to the buffer's max) -- the candidates an f32 pass may put on the other side
    bufs = {f"a{k}": np.concatenate([c["e_tgt"][k], c["e_src"][k], c["e_ctx"][k]]) for k in range(5)}
    bufs["z"] = np.concatenate([c["e_tgt"][5], c["e_src"][5], c["e_ctx"][5]])
    bufs["th0"] = c["trans_h0"]
    bufs["dz"] = np.concatenate([c["d1"][0], c["d2"][0]])
    for k in (1, 2, 3):
        bufs[f"e{k}"] = np.concatenate([c["d1"][k], c["d2"][k]])
    fx["act_names"] = np.array(sorted(bufs))
    fx["act_negative"] = np.array([int((bufs[k] < 0).sum()) for k in sorted(bufs)], np.int64)
    fx["act_near_zero"] = np.array([int((np.abs(bufs[k]) <= 1e-6 * np.abs(bufs[k]).max()).sum()) for k in sorted(bufs)], np.int64)
    fx["act_size"] = np.array([bufs[k].size for k in sorted(bufs)], np.int64)
    del bufs
    g = ci.backward(p, c, cfg)
    print(f"backward {time.time() - t0:.0f}s", flush=True)
    probes = ref299_probes([(n, g[n].size) for n in names])
    fx["grad_digest"] = np.stack([digest(g[n])[0] for n in names])
    fx["grad_proj"] = np.stack([ref299_project(g[n], probes[n][0]) for n in names])
    fx["grad_samples"] = np.stack([np.pad(g[n].reshape(-1)[probes[n][1]], (0, REF299_NSAMP - len(probes[n][1]))) for n in names])
    # the reward hook's fetches at its batch (base.py:216-218, :234-235 with image_trans = feature maps)
    pred, feat = ci.translate(p, src, ctx[0], cfg)
    fx["translate_pred_keep"], fx["translate_feat"] = pred[list(REF299_KEEP)].astype(np.float32), feat.astype(np.float32)
    fx["encode_feat"] = ci.encode(p, src, cfg).astype(np.float32)
    # two Adam steps (train_script.py:163): scalars before each update
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    o.adam_step(p, g, m, v, 1, 1e-4)
    del g, c
    rr, cc = ci.forward(p, src, ctx, tgt, cfg)
    fx["train_scalars"] = np.array([fx["scalars"], [rr["loss"], rr["simloss"], rr["recon1"], rr["recon2"]]])
    path = os.path.join(HERE, f"{REF299_TAG}.npz")
    np.savez_compressed(path, **fx)
    print(REF299_TAG, os.path.getsize(path), "bytes; loss", fx["scalars"][0], "->", rr["loss"], f"{time.time() - t0:.0f}s")


def make_inception_v3_299():
    """Inception-v3 to Mixed_7c at the reference's own 299x299 (nets/inception_v3_test.py:45-54: Mixed_7c = [N, 8, 8, 2048]): two
    seeded uint8 frames (regenerated by the tests), digest + head of every end point, Mixed_7c of frame 0 whole."""
    from oracle import inception_oracle as io
    p = {k: v.astype(np.float64) for k, v in io.init_params(0, np.float32).items()}      # float32-representable: what the device holds
    frames = np.random.default_rng(REF299_SEEDS["fseed"] + 1).integers(0, 256, (2, 299, 299, 3), dtype=np.uint8)
    ep = io.forward(p, o.preprocess_u8(frames).astype(np.float64))
    fx = dict(pseed=0, fseed=REF299_SEEDS["fseed"] + 1, B=2, S=299, endpoints=np.array(list(ep)),
              frames_digest=digest(frames)[0], Mixed_7c_0=ep["Mixed_7c"][0].astype(np.float32),
              endpoint_shapes=np.stack([np.array(v.shape) for v in ep.values()]),
              endpoint_digest=np.stack([digest(v)[0] for v in ep.values()]),
              endpoint_head=np.stack([digest(v)[1] for v in ep.values()]))
    fx["param_digest"], _ = digest(np.concatenate([np.asarray(v).reshape(-1) for v in p.values()]))
    path = os.path.join(HERE, f"{REF299_FRONT_TAG}.npz")
    np.savez_compressed(path, **fx)
    print(REF299_FRONT_TAG, os.path.getsize(path), "bytes")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ref299":
    make_ref299()


# ------------------------------------------------------------------------------------------------------------------
# The launch shapes bench.py's `secondary` numbers are quoted on: ContextAEReal at 36x64 with the
# reference's own training batch 100 (ablations_code/ablations.py:503,536-544) and with 256, at 64x64 with 256; one GPU's share
# of BASELINE configs[3] -- ContextAEInception2 on 2x2x2048 Mixed_7c maps (rllab/sampler/base.py:126), 64 triples.
# Same scheme as make_b256: the float64 oracle runs the batch in chunks (`sim_batch` = the whole batch), the file holds seeds,
# the four scalars, whole outputs of a few triples, per-triple digests of out / out2 / translated_z / input_z, and per gradient
# tensor a digest, 16 random-sign projections and up to 4096 sampled entries; then ONE Adam step (lr 1e-4) in float64 and the
# scalars of the pass after it, plus sampled entries of the parameter update where the gradient is well above the f32 noise floor.
#   python tests/golden/make_golden.py big            (all four; ~10 minutes and ~12 GB here)
# ------------------------------------------------------------------------------------------------------------------
BIG_NPROJ, BIG_NSAMP = 16, 4096
BIG_CASES = {
    # tag: (kind, H, W, B, pseed, fseed, stddev, chunk)
    "real_f100_36x64_b100": ("real", 36, 64, 100, 5100, 51, 0.1, 50),
    "real_f100_36x64_b256": ("real", 36, 64, 256, 5256, 52, 0.1, 64),
    "real_f100_64x64_b256": ("real", 64, 64, 256, 6256, 62, 0.1, 64),
    "incep2_2x2x2048_f1024_b64": ("incep2", 2, 2, 64, 7064, 71, 0.02, 64),
}


def big_case(tag):
    """(oracle module, cfg, params {name: float32-representable float64}, [src, ctx, tgt] float32 inputs) -- also what the GPU test
    rebuilds.  ContextAEReal: smooth blob frames for slot 0 / 2 and iid frames for the context slot (both SURVEY 8d distributions in one
    batch); ContextAEInception2: post-ReLU maps."""
    kind, H, W, B, pseed, fseed, stddev, _ = BIG_CASES[tag]
    if kind == "real":
        from oracle import ctx_oracle_real as mod
        cfg = mod.RealConfig(H=H, W=W)
    else:
        from oracle import ctx_oracle_incep as mod
        cfg = mod.Incep2Config(H=H, W=W)
    p = mod.init_params(cfg, pseed, np.float32, stddev=stddev)
    brng = np.random.default_rng(pseed + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = (brng.standard_normal(p[n].shape) * stddev).astype(np.float32)
    if kind == "real":
        u8 = [blob_frames(fseed, B, H, W), synth_frames(fseed + 1, B, H, W), blob_frames(fseed + 2, B, H, W)]
        x = [o.preprocess_u8(f) for f in u8]
    else:
        frng = np.random.default_rng(fseed)
        x = [np.maximum(frng.standard_normal((B, H, W, cfg.C)), 0.0).astype(np.float32) for _ in range(3)]
    return mod, cfg, p, x


def big_probes(name_sizes, tag):
    base = BIG_CASES[tag][4]
    out = {}
    for i, (n, size) in enumerate(name_sizes):
        idx = np.sort(np.random.default_rng(base * 1000 + i).choice(size, min(BIG_NSAMP, size), replace=False))
        out[n] = (base * 2000 + i, idx)
    return out


def big_project(a, seed):
    a = np.asarray(a, np.float64).reshape(-1)
    rng = np.random.default_rng(seed)
    return np.array([float(a @ (rng.integers(0, 2, a.size, dtype=np.int8) * 2.0 - 1.0)) for _ in range(BIG_NPROJ)])


def _big_pass(mod, cfg, p, x, B, chunk, want_grads=True):
    names = [n for n, _ in mod.param_specs(cfg)]
    g = {n: np.zeros_like(p[n]) for n in names} if want_grads else None
    scal = np.zeros(3)
    outs = {k: [] for k in ("out", "out2", "translated_z", "input_z")}
    neg, near, size = {}, {}, {}
    for i0 in range(0, B, chunk):
        sl = slice(i0, min(B, i0 + chunk))
        res, c = mod.forward(p, *(a[sl].astype(np.float64) for a in x), cfg)
        scal += [res["recon1"], res["recon2"], float(np.sum((c["trans_z"] - c["e_tgt"][5]) ** 2))]
        for k in outs:
            outs[k].append(res[k])
        if want_grads:
            gi = mod.backward(p, c, cfg, sim_batch=B)
            for n in names:
                g[n] += gi[n]
            # lrelu'-relevant activations under the device's buffer names (tests/_align.py: align_gen_cache)
            bufs = {f"a{k}": (c["e_tgt"][k], c["e_src"][k], c["e_ctx"][k]) for k in range(5)}
            bufs["z"] = (c["e_tgt"][5], c["e_src"][5])
            bufs["th0"] = (c["trans_h0"],)
            bufs["dz"] = (c["d1"][0], c["d2"][0])
            for k in (1, 2, 3):
                bufs[f"e{k}"] = (c["d1"][k], c["d2"][k])
            for k, parts in bufs.items():
                for a in parts:
                    neg[k] = neg.get(k, 0) + int((a < 0).sum())
                    near[k] = near.get(k, 0) + int((np.abs(a) <= 1e-6 * np.abs(a).max()).sum())
                    size[k] = size.get(k, 0) + a.size
    F = cfg.featsize
    sim = scal[2] / (B * F) * 1e3
    scalars = np.array([scal[0] + scal[1] + sim, sim, scal[0], scal[1]])
    return scalars, {k: np.concatenate(v) for k, v in outs.items()}, g, (neg, near, size)


def make_big(tag):
    import time
    t0 = time.time()
    kind, H, W, B, pseed, fseed, stddev, chunk = BIG_CASES[tag]
    mod, cfg, p32, x = big_case(tag)
    p = {k: v.astype(np.float64) for k, v in p32.items()}
    names = [n for n, _ in mod.param_specs(cfg)]
    keep = sorted({0, B // 3, B - 1})
    fx = dict(kind=kind, cfg=np.array([H, W, cfg.C, cfg.featsize]), B=B, pseed=pseed, fseed=fseed, stddev=stddev, lr=1e-4, keep=np.array(keep))
    fx["param_digest"], _ = digest(mod.flatten(p, cfg))
    fx["input_digest"] = np.stack([digest(a)[0] for a in x])
    scalars, outs, g, (neg, near, size) = _big_pass(mod, cfg, p, x, B, chunk)
    print(f"{tag}: pass 1 {time.time() - t0:.0f}s loss {scalars[0]:.8g}", flush=True)
    fx["scalars"] = scalars
    for k, full in outs.items():
        fx[k + "_keep"] = full[keep].astype(np.float32)
        flat = full.reshape(B, -1)
        fx[k + "_rows"] = np.stack([flat.sum(1), np.abs(flat).sum(1), np.sqrt((flat * flat).sum(1))], 1)
    probes = big_probes([(n, g[n].size) for n in names], tag)
    fx["grad_digest"] = np.stack([digest(g[n])[0] for n in names])
    fx["grad_proj"] = np.stack([big_project(g[n], probes[n][0]) for n in names])
    fx["grad_samples"] = np.stack([np.pad(g[n].reshape(-1)[probes[n][1]], (0, BIG_NSAMP - len(probes[n][1]))) for n in names])
    fx["act_names"] = np.array(sorted(neg))
    fx["act_negative"] = np.array([neg[k] for k in sorted(neg)], np.int64)
    fx["act_near_zero"] = np.array([near[k] for k in sorted(neg)], np.int64)
    fx["act_size"] = np.array([size[k] for k in sorted(neg)], np.int64)
    # one TF-Adam step (train_script.py:128,163) in float64, then the scalars of the next pass
    p0 = {n: p[n].copy() for n in names}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    o.adam_step(p, g, m, v, 1, 1e-4)
    # the update of an entry is -lr * sign(g) (first step): it is well-defined where |g| is above the f32 noise floor of its tensor
    upd = []
    for n in names:
        idx = probes[n][1]
        gs = g[n].reshape(-1)[idx]
        d = (p[n] - p0[n]).reshape(-1)[idx]
        d = np.where(np.abs(gs) > 1e-3 * np.abs(g[n]).max(), d, np.nan)
        upd.append(np.pad(d, (0, BIG_NSAMP - len(idx)), constant_values=np.nan))
    fx["update_samples"] = np.stack(upd)
    del g, p0, m, v
    scalars2, _, _, _ = _big_pass(mod, cfg, p, x, B, chunk, want_grads=False)
    fx["train_scalars"] = np.stack([scalars, scalars2])
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(tag, os.path.getsize(path), "bytes; loss", scalars[0], "->", scalars2[0], f"{time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "big":
    for _tag in (sys.argv[2:] or list(BIG_CASES)):
        make_big(_tag)
