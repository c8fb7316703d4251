#!/usr/bin/env python3
"""bench.py -- translator training throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either form works -- plain `python bench.py --gpus N` starts its N ranks itself by re-executing under
     torch.distributed.run on 127.0.0.1 and relays rank 0's JSON line as the LAST line of stdout;
     `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` is used as launched.)

One "step" = forward + backward + Adam of ContextSkipNew at 64x64x3 on a per-GPU batch of 256
synthetic (src, ctx, tgt) frame triples that are already resident in HBM (BASELINE.json
configs[1]; weak scaling: every rank has its own 256).  With N > 1 the gradient arena is
sum-all-reduced over RCCL between backward and Adam (the path has one exchange step).
Prints ONE JSON line on rank 0.  PyTorch is plumbing only: device buffers, the stream, RCCL.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 64
DF, FEAT = 64, 1024
FLOPS_FWD_BWD_PER_TRIPLE = 7_072_382_976
BYTES_PER_TRIPLE = 15_155_200
BYTES_PER_STEP_FIXED = 1_905_912_440
PEAK_F32_MFMA = 157.3e12                           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_HBM = 8.0e12
PEAK_BF16_MFMA = 2.5e15                            # dense; a split product costs 3 bf16 MFMA flops per algorithmic flop


def skipnew_flops_per_triple(H, W, d=DF, F=FEAT, useful=False):
    """fwd+bwd FLOPs of one ContextSkipNew triple (2 per multiply-add; arm_shaping.py:1272-1354: 3 encoder images, translate MLP,
    2 decoder passes; backward = input + filter gradient of every layer except the first convs' input gradient).
    useful=False: SURVEY.md 8d's convention -- every one of the 25 taps counted at every position (7,072,382,976 at 64x64).
    useful=True: only the (position, tap) pairs whose tap lies inside the image -- (5n-3)^2 of (5n)^2 on an n x n small grid -- i.e.
    the products that do not multiply SAME padding; what a kernel that skips those has to execute."""
    def frac(n):
        return (5 * n - 3) / (5.0 * n) if useful else 1.0
    h, w, cin = H, W, 3
    enc, first, dec = 0.0, 0.0, 0.0
    chans = [d, 2 * d, 4 * d, 8 * d]
    grids = []
    for k in range(4):
        h, w = h // 2, w // 2
        fl = 2.0 * h * w * 25 * cin * chans[k] * frac(h) * frac(w)
        enc += fl
        if k == 0:
            first = fl
        grids.append((h, w))
        cin = chans[k]
    d0 = grids[3][0] * grids[3][1] * 8 * d
    fc = 3 * 2.0 * (d0 * F + F * F) + 2.0 * (2 * F * F + F * F) + 2 * 2.0 * F * d0
    for k in range(1, 5):                                    # d_hk: input grid = encoder layer 4-k's output grid, channels [decoder | skip]
        hi, wi = grids[4 - k]
        c_in = 2 * chans[4 - k]
        c_out = chans[3 - k] if k < 4 else 3
        dec += 2.0 * hi * wi * 25 * c_in * c_out * frac(hi) * frac(wi)
    fwd = 3 * enc + fc + 2 * dec
    return 3.0 * fwd - 3 * first


def csrc_sha16():
    """Identity of the kernel sources this tree builds libctxtrans.so from: the PMC-derived traffic figures under profiles/
    carry the hash of the sources they were measured on (tools/hbm_aggregate.py) and are only quoted for the same sources."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "imitation_from_observation_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cpp")) +
                    glob.glob(os.path.join(d, "*.inc"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def real_flops_per_triple(H, W):
    """Algorithmic fwd+bwd FLOPs of one ContextAEReal triple (arm_shaping.py:1611-1684: k 5, strides 1/2/1/2, filters 32/16/16/8,
    featsize 100; 3 encoder images and 2 decoder passes per triple; backward = input gradient + filter gradient of every layer
    except the first conv's input gradient)."""
    nf, st, F = (32, 16, 16, 8), (1, 2, 1, 2), 100
    h, w, cin, fwd, first = H, W, 3, 0.0, 0.0
    grids = []
    for k in range(4):
        h, w = h // st[k], w // st[k]
        fl = 2.0 * h * w * 25 * cin * nf[k]
        fwd += 3 * fl
        if k == 0:
            first = 3 * fl
        grids.append((h, w))
        cin = nf[k]
    d0 = grids[3][0] * grids[3][1] * nf[3]
    fc = 3 * 2.0 * (d0 * F + F * F) + 2.0 * (2 * F * F + F * F) + 2 * 2.0 * F * d0
    dec = 0.0
    for k in range(1, 5):                                    # d_hk mirrors encoder layer 4-k: input grid = that layer's output grid
        gi = 4 - k
        hi, wi = grids[gi]
        cout = nf[gi - 1] if gi else 3
        dec += 2 * 2.0 * hi * wi * 25 * (2 * nf[gi]) * cout
    fwd += fc + dec
    return 3.0 * fwd - first


def secondary_legs(device, steps=40):
    """BASELINE configs[4] and [3] on one GPU, timed by this process (never `value`): (a) ContextAEReal 36x64, batch 256, fwd + bwd
    + Adam, frames resident; (b) one GPU's share of config 4: 64 triples of 125x125 frames = 192 images through the frozen
    Inception-v3 front end, then ContextAEInception2 fwd + bwd + Adam on the 2x2x2048 maps, one stream, everything resident."""
    import torch
    from imitation_from_observation_amd import Translator
    out = {}
    for Hh, Ww, Bb in ((36, 64, 256), (64, 64, 256)):          # the reference's sweep size (base.py:134-135) and BASELINE configs[4]'s 64x64
        key = f"context_ae_real_{Hh}x{Ww}_b{Bb}"
        try:
            tr = Translator(Hh, Ww, featsize=100, max_batch=Bb, variant="real", device=device)
            tr.init_params(0)
            g = torch.Generator(device="cuda").manual_seed(7)
            d = [(torch.randint(0, 256, (Bb, Hh, Ww, 3), device="cuda", generator=g, dtype=torch.uint8).float() / 127.5 - 1.0).contiguous()
                 for _ in range(3)]
            torch.cuda.synchronize()
            # warm up by TIME, not by count: the handle's creation leaves the GPU idle long enough for the clock to fall back, and five
            # 3 ms steps do not bring it up again (the same build read 2.6 and 3.0 ms on one box depending on what ran before)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                for _ in range(5):
                    tr.dev_train_step(*(t.data_ptr() for t in d), Bb, 1e-4)
                tr.sync()
            steps = max(steps, 100)
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.dev_train_step(*(t.data_ptr() for t in d), Bb, 1e-4)
            tr.sync()
            dt = (time.perf_counter() - t0) / steps
            fl = real_flops_per_triple(Hh, Ww) * Bb
            out[key] = {"ms_per_step": 1e3 * dt, "frames_per_s": Bb / dt, "steps": steps, "gflop_per_step": fl / 1e9, "tflops": fl / dt / 1e12,
                        "frac_f32_mfma_peak": fl / dt / PEAK_F32_MFMA, "loss_after": tr.dev_scalars()["loss"],
                        "roofline": {"bound": "mfma", "achieved": fl / dt / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                                     "frac": fl / dt / PEAK_F32_MFMA, "traffic": None,
                                     "note": "whole step on the layers' FLOPs (channels as the reference defines them: 32/16/16/8, no padding)"}}
            tr.close()
        except Exception as e:                                   # a secondary leg must never cost the bench line
            out[key] = {"error": repr(e)[:300]}
    try:
        from imitation_from_observation_amd.inception_frontend import InceptionFrontend
        S, Bc = 125, 64
        g = torch.Generator(device="cuda").manual_seed(8)
        frames = (torch.randint(0, 256, (3 * Bc, S, S, 3), device="cuda", generator=g, dtype=torch.uint8).float() / 127.5 - 1.0).contiguous()
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        front = InceptionFrontend(S, S, max_images=3 * Bc, precision="f32", stream=stream.cuda_stream, device=device)
        front.init_synthetic(0)
        h, w, c = front.out_shape
        tr = Translator(h, w, 64, 1024, max_batch=Bc, variant="inception2", C=c, precision="f32", stream=stream.cuda_stream, device=device)
        tr.init_params(1)
        lay = front.flops_per_image()
        per = h * w * c * 4

        def step():
            dd = front.features_dev(frames.data_ptr(), 3 * Bc)
            tr.dev_train_step(dd, dd + Bc * per, dd + 2 * Bc * per, Bc, 1e-4)

        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:                      # (warm-up by time, as above)
            for _ in range(3):
                step()
            tr.sync()
        n = max(10, steps // 2)
        t0 = time.perf_counter()
        for _ in range(n):
            front.features_dev(frames.data_ptr(), 3 * Bc)
        front.sync()
        tf_ = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        tr.sync()
        dt = (time.perf_counter() - t0) / n
        out["config4_share_64_triples_125x125"] = {
            "ms_per_step": 1e3 * dt, "triples_per_s": Bc / dt, "steps": n,
            "frontend_ms": 1e3 * tf_, "frontend_images_per_s": 3 * Bc / tf_, "frontend_gflop_per_image": lay / 1e9,
            "frontend_tflops": lay * 3 * Bc / tf_ / 1e12, "frontend_frac_f32_mfma_peak": lay * 3 * Bc / tf_ / PEAK_F32_MFMA,
            "translator_ms": 1e3 * (dt - tf_), "loss_after": tr.dev_scalars()["loss"],
            "roofline_frontend": {"bound": "mfma", "achieved": lay * 3 * Bc / tf_ / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                                  "frac": lay * 3 * Bc / tf_ / PEAK_F32_MFMA, "traffic": None},
            # the translator's share at 64 triples of 2x2 maps is parameter traffic: 10 touches of every parameter per step (SURVEY 8d:
            # forward read, backward read, gradient write, Adam's 4 reads and 3 writes)
            "roofline_translator": {"bound": "hbm", "achieved": tr.n_params * 40.0 / (dt - tf_) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                                    "frac": tr.n_params * 40.0 / (dt - tf_) / PEAK_HBM, "traffic": None, "parameters": tr.n_params},
            "note": "synthetic Inception-v3 variables (the checkpoint is not in the reference tree), f32"}
        tr.close()
        front.close()
    except Exception as e:
        out["config4_share_64_triples_125x125"] = {"error": repr(e)[:300]}
    return out


def _median_min(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2], ts[0]


def cpu_baseline(batch=32, steps=10, budget_s=40.0):
    """CPU statements of the same arithmetic on the host cores, on a bounded sample (the reference's TensorFlow path cannot
    run here, so kind = "port").  Two workloads at batch `batch`:
      * BASELINE configs[0]: forward + the three losses (no backward);
      * the metric's workload: forward + backward + TF-Adam.
    Two statements: torch-CPU (tests/_torch_ref.py: oneDNN convolutions + autograd) and the numpy oracle.  The thread count
    is swept (8/16/32/64/128, capped by the machine) with short runs, then `steps` timed steps (after a warm-up) run at
    the best count: median and min are reported, `value` is batch / median of the faster statement's train step."""
    import torch
    from oracle import ctx_oracle as o
    from tests import _torch_ref as tref
    ncpu = os.cpu_count() or 1
    cfg = o.SkipNewConfig(H=H, W=W, df_dim=DF, gf_dim=DF, featsize=FEAT)
    p = o.init_params(cfg, 0, np.float32)
    rng = np.random.default_rng(0)
    src, ctx, tgt = (rng.uniform(-1, 1, (batch, H, W, 3)).astype(np.float32) for _ in range(3))
    t_begin = time.perf_counter()

    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    ts, tc, tt = (torch.tensor(a) for a in (src, ctx, tgt))
    tm = {k: torch.zeros_like(v) for k, v in tp.items()}
    tv = {k: torch.zeros_like(v) for k, v in tp.items()}

    def t_fwd():
        with torch.no_grad():
            return float(tref.forward(tp, ts, tc, tt, H, W, DF)["loss"])

    def t_train(t):
        loss = tref.forward(tp, ts, tc, tt, H, W, DF)["loss"]
        grads = torch.autograd.grad(loss, list(tp.values()))
        with torch.no_grad():                              # TF-Adam (train_script.py:128 defaults)
            lr_t = 1e-4 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
            for (k, w), g in zip(tp.items(), grads):
                tm[k].mul_(0.9).add_(g, alpha=0.1)
                tv[k].mul_(0.999).addcmul_(g, g, value=0.001)
                w.sub_(lr_t * tm[k] / (tv[k].sqrt() + 1e-8))

    def timed_calls(fn, n):
        out = []
        for i in range(n):
            t0 = time.perf_counter()
            fn(i)
            out.append(time.perf_counter() - t0)
        return out

    # thread sweep on the train step (1 warm-up + 2 timed per count)
    counts = [c for c in (8, 16, 32, 64, 128) if c <= ncpu] or [ncpu]
    sweep, step_no = {}, [0]

    def train_once(_):
        step_no[0] += 1
        t_train(step_no[0])

    torch.set_num_threads(counts[0])
    train_once(0)                                            # global warm-up: allocator, oneDNN primitive caches
    train_once(0)
    for c in counts:
        torch.set_num_threads(c)
        train_once(0)
        sweep[c] = min(timed_calls(train_once, 2))
        if time.perf_counter() - t_begin > budget_s * 0.5:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    n_train = steps
    est = sweep[best] * steps
    left = budget_s - (time.perf_counter() - t_begin)
    if est > left * 0.6:                                    # keep the default bench run within minutes
        n_train = max(3, int(left * 0.6 / sweep[best]))
    train_ts = timed_calls(train_once, n_train)
    t_fwd()
    fwd_ts = timed_calls(lambda i: t_fwd(), max(n_train, 3))
    tr_med, tr_min = _median_min(train_ts)
    fw_med, fw_min = _median_min(fwd_ts)

    # the numpy oracle at the same thread count (its im2col / col2im parts are single-threaded)
    numpy_info = {}
    try:
        from threadpoolctl import threadpool_limits
        m = {k: np.zeros_like(v) for k, v in p.items()}
        v = {k: np.zeros_like(v_) for k, v_ in p.items()}
        with threadpool_limits(limits=best, user_api="blas"):
            o.train_step(p, m, v, 1, src, ctx, tgt, 1e-4, cfg)
            nts = timed_calls(lambda i: o.train_step(p, m, v, i + 2, src, ctx, tgt, 1e-4, cfg), 3)
            nfs = timed_calls(lambda i: o.forward(p, src, ctx, tgt, cfg), 3)
        nm, nmin = _median_min(nts)
        fm, fmin = _median_min(nfs)
        numpy_info = {"train_frames_per_s": batch / nm, "train_ms_median": 1e3 * nm, "train_ms_min": 1e3 * nmin,
                      "fwd_loss_frames_per_s": batch / fm, "fwd_loss_ms_median": 1e3 * fm, "blas_threads": best, "steps": 3}
    except Exception as e:                                   # the torch figure stands
        numpy_info = {"error": repr(e)[:200]}

    value, stmt = batch / tr_med, "torch-CPU (oneDNN, autograd)"
    if numpy_info.get("train_frames_per_s", 0) > value:
        value, stmt = numpy_info["train_frames_per_s"], "numpy oracle"
    return {
        "value": value, "unit": "frames/s", "cores": best, "kind": "port",
        "sample": f"{n_train} fwd+bwd+Adam steps at batch {batch}, f32, {stmt}, {best} threads of {ncpu} logical CPUs "
                  f"(best of a sweep over {sorted(sweep)}): median {1e3 * tr_med:.0f} ms, min {1e3 * tr_min:.0f} ms per step; "
                  f"the reference's TensorFlow path cannot run here",
        "statement": stmt,
        "train_ms_median": 1e3 * tr_med, "train_ms_min": 1e3 * tr_min, "train_steps": n_train,
        "thread_sweep_train_s_per_step": {str(k): round(v_, 4) for k, v_ in sorted(sweep.items())},
        # BASELINE configs[0]: batch 32, encoder-decoder forward + the L2 / feature losses on CPU
        "config1_fwd_loss": {"value": batch / fw_med, "unit": "frames/s", "cores": best, "ms_median": 1e3 * fw_med,
                             "ms_min": 1e3 * fw_min, "steps": len(fwd_ts), "statement": "torch-CPU (oneDNN)"},
        "numpy_oracle": numpy_info,
        "wall_s": time.perf_counter() - t_begin,
    }


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1), relay everything the ranks print to stderr and rank 0's JSON line -- alone -- as the last line of stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    rest = []
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            try:
                json.loads(ln)
                line = ln
                continue
            except ValueError:
                pass
        rest.append(ln)
    if rest:
        sys.stderr.write("\n".join(rest) + "\n")
        sys.stderr.flush()
    if line is None:
        sys.stderr.write(f"bench.py: the {n}-rank run produced no JSON line (exit code {proc.returncode})\n")
        return proc.returncode or 1
    print(line, flush=True)
    return proc.returncode


def spawn_selftest(args):
    """BENCH_SPAWN_SELFTEST=1: the launch / barrier / max-over-ranks / one-JSON-line plumbing on a gloo group with a sleep in
    place of the train step, so `python bench.py --gpus 2` can be exercised on a box without GPUs (tests/test_bench_spawn.py)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for _ in range(args.warmup):
        time.sleep(0.001)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    print(f"rank {rank} of {world} done", flush=True)        # noise the parent must keep off the last line
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "selftest", "value": args.steps * world / float(dt), "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "selftest": True}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-iters", type=int, default=5)
    ap.add_argument("--host-buffers", action="store_true",
                    help="also time ctx_train_step on host (numpy) frames: the PCIe-inclusive rate quoted in DESIGN.md")
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default=os.environ.get("CTX_PRECISION", "f32"),
                    help="arithmetic of the contractions: exact f32 MFMA (default) or split-bf16 products (DESIGN.md section 6b)")
    ap.add_argument("--no-split-leg", action="store_true",
                    help="skip the extra bf16x3 measurement that a default (f32) run appends as line['bf16x3']")
    ap.add_argument("--sustained-s", type=float, default=3.0,
                    help="after the timed run, keep stepping for at least this many seconds and report sustained_ms_per_step (0 = skip)")
    ap.add_argument("--no-sampled", action="store_true",
                    help="skip the extra leg that runs the trainer's own step (ctx_train_step_sampled / ctx_dp_train_step_sampled: the batch "
                         "gathered on the device from a resident uint8 demo tensor, global index arrays from the host) -- line['sampled']")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (ContextAEReal 36x64 B=256, config-4 share) a default 1-GPU run appends as line['secondary']")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # plain `python bench.py --gpus N`: start the ranks ourselves
        raise SystemExit(spawn_ranks(args.gpus))
    if os.environ.get("BENCH_SPAWN_SELFTEST") == "1":
        return spawn_selftest(args)

    # stdout carries ONE line, rank 0's JSON: file descriptor 1 is pointed at stderr for the run (librccl prints a version banner to
    # stdout at communicator creation, torch.distributed warns there too) and the saved descriptor gets the line at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from imitation_from_observation_amd.dp import DataParallelTrainer, RcclTrainer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    # BENCH_ONE_GPU=1 (tests/test_gpu_bench_ranks.py, tools/scale_check.sh): the N ranks of the REAL --gpus N code path share device 0 --
    # the torch group runs on gloo (two ranks cannot share a device under RCCL) and ctx_dp_* talks to whatever CTX_RCCL_LIB names (the
    # shared-memory stand-in of tests/fake_rccl).  It exercises spawn -> group -> ctx_dp_init -> timed steps -> comm block -> one JSON
    # line before an 8-GPU node sees that path first; the line is marked and its value is NOT a scaling figure.
    one_gpu = os.environ.get("BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} visible")
    red_dev = "cpu" if one_gpu else "cuda"                          # where the group's small reductions live
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":     # (the env switch: a one-rank RCCL group, to exercise the N > 1 code on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist.barrier()
        # RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would otherwise land AFTER the
        # JSON line at exit: flush it now so that the JSON line is the last line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)

    B = args.batch
    # N > 1: the data-parallel step behind the C ABI (ctx_dp_train_step: RCCL inside libctxtrans, gradient buckets sent from inside backward, the tail
    # bucket reduced on a second stream under the encoders' backward) -- torch.distributed only ships the 128-byte rendezvous blob
    # and does the contract's barrier / max-over-ranks.  BENCH_DP=torch selects the torch.distributed client of the same step
    # (dp.DataParallelTrainer: one all-reduce after backward, or the bucketed schedule with CTX_DP_OVERLAP=1).
    dp_client = os.environ.get("BENCH_DP", "cabi") if dist.is_initialized() else "single"
    if dp_client == "cabi":
        # every rank must take the same client: agree first that librccl loads everywhere behind the C ABI (else the torch client)
        try:
            from imitation_from_observation_amd import Translator as _T
            _T.dp_unique_id()
            ok = 1
        except Exception as e:                                  # noqa: BLE001  (a missing / unloadable librccl on this rank)
            sys.stderr.write(f"bench.py: rank {rank}: RCCL behind the C ABI unavailable ({e!r}); asking for the torch client\n")
            ok = 0
        flag = torch.tensor([ok], device=red_dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # no silent change of client: the number would be another schedule's (one all-reduce after backward, no overlap).  Ask for
            # it by name (BENCH_DP=torch) if that is what should be measured.
            raise SystemExit(f"bench.py: rank {rank}: the C-ABI RCCL client (ctx_dp_*) is unavailable on at least one rank; "
                             "set BENCH_DP=torch to measure the torch.distributed client instead")
    if dp_client == "cabi":
        trainer = RcclTrainer(H, W, DF, FEAT, max_batch=B, device=local_rank, seed=1234, rank=rank, world=world, precision=args.precision)
    else:
        trainer = DataParallelTrainer(H, W, DF, FEAT, max_batch=B, device=local_rank, seed=1234, precision=args.precision)

    def stream_of(tr_):
        """torch view of the stream the step's kernels are enqueued on (for the HIP events)"""
        eng = getattr(tr_, "engine", None)
        return eng.stream if eng is not None else torch.cuda.ExternalStream(tr_.translator.stream_ptr)
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    frames = [torch.randint(0, 256, (B, H, W, 3), device="cuda", generator=g, dtype=torch.uint8) for _ in range(3)]
    src, ctx, tgt = (f.float() / 127.5 - 1.0 for f in frames)      # synthetic frames, train_script.py:16-19 scaling

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def timed(tr_):
        for _ in range(args.warmup):
            tr_.step(src, ctx, tgt, lr=1e-4)
        # HIP events on the stream the kernels are launched on, one per step boundary: median / min step time beside the
        # wall-clock mean that `value` is computed from
        stream = stream_of(tr_)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        barrier()
        t0 = time.perf_counter()
        evs[0].record(stream)
        for i in range(args.steps):
            tr_.step(src, ctx, tgt, lr=1e-4)
            evs[i + 1].record(stream)
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
        ev_ = {"median": per[len(per) // 2], "min": per[0], "max": per[-1], "mean": sum(per) / len(per)}
        return dt_, tr_.scalars(), ev_

    dt, scal, step_events = timed(trainer)

    # Sustained leg (never `value`): the K timed steps are a fraction of a second; clocks settle to the power budget over seconds
    # (MI355X_MICROARCH.md, DVFS), so the same step is run on for >= --sustained-s seconds and reported beside the contract's figure.
    sustained = None
    if args.sustained_s > 0:
        n_sus = max(args.steps, int(args.sustained_s / (dt / args.steps)) + 1)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            trainer.step(src, ctx, tgt, lr=1e-4)
        barrier()
        dts = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dts], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dts = float(tmax.item())
        sustained = {"steps": n_sus, "seconds": dts, "ms_per_step": 1e3 * dts / n_sus, "frames_per_s": n_sus * B * world / dts}

    # The trainer's own step (never `value`): a uint8 demo tensor [25 frames, 64 videos] resident in HBM, per step the host hands over
    # the two GLOBAL index arrays (np.random.choice, train_script.py:154-155) and every rank gathers its rows on the device.  Includes the
    # host-side enqueue of each step and the 2 x 4 B x world bytes of indices.
    sampled = None
    if not args.no_sampled and dp_client in ("cabi", "single") and args.precision == "f32":
        try:
            trl = trainer.translator
            rng = np.random.default_rng(5)
            trl.load_demos(rng.integers(0, 256, (25, 64, H, W, 3), dtype=np.uint8))
            Bg = B * world
            draws = [(rng.integers(0, 64, Bg), rng.integers(0, 64, Bg)) for _ in range(8)]

            def sstep(i):
                cs, ct = draws[i % len(draws)]
                if dp_client == "cabi":
                    trl.dp_train_step_sampled(cs, ct, lr=1e-4, scalars=False)
                else:
                    trl.train_step_sampled(cs, ct, lr=1e-4)          # (synchronous: returns the four scalars, as the reference's sess.run does)

            for i in range(args.warmup):
                sstep(i)
            trl.sync()
            barrier()
            t0 = time.perf_counter()
            for i in range(args.steps):
                sstep(i)
            trl.sync()
            barrier()
            dts = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([dts], device=red_dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dts = float(tmax.item())
            sampled = {"ms_per_step": 1e3 * dts / args.steps, "frames_per_s": args.steps * Bg / dts, "steps": args.steps,
                       "demo_tensor": "uint8 [25, 64, 64, 64, 3] resident per rank", "global_batch": Bg,
                       "entry": "ctx_dp_train_step_sampled" if dp_client == "cabi" else "ctx_train_step_sampled (synchronous, scalars fetched every step)"}
        except Exception as e:                      # an extra leg must never cost the bench line
            sampled = {"error": repr(e)[:300]}

    ms = 1e3 * dt / args.steps
    value = args.steps * B * world / dt
    line = {
        "metric": "frames/sec fwd+bwd+Adam, 64x64x3, batch 256 per GPU",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "ContextSkipNew 64x64x3 fwd+bwd+Adam, batch 256/GPU (BASELINE configs[1])",
                   "precision": ("exact f32 MFMA" if args.precision == "f32" else
                                 "split-bf16: a*b = hi*hi + hi*lo + lo*hi on bf16 MFMA, f32 accumulate; f32 everywhere else"),
                   "per_gpu_batch": B, "global_batch": B * world, "params": trainer.n_params,
                   "parallelism": f"dp{world}" + (" + RCCL grad all-reduce" if world > 1 else ""),
                   "dp_client": {"cabi": "ctx_dp_train_step (RCCL behind the C ABI; buckets from inside backward: translate+decoder, each encoder's FC slice; conv filters last; second stream)",
                                 "torch": "torch.distributed all-reduce between ctx_dev_forward_backward and ctx_dev_adam",
                                 "single": "none (one rank)"}[dp_client]},
        "loss_after": scal["loss"],
        **({"one_gpu_stand_in": True, "note": "BENCH_ONE_GPU=1: the N ranks time-share ONE device (torch group on gloo, ctx_dp_* through CTX_RCCL_LIB): a code-path run, NOT a scaling figure"} if one_gpu else {}),
        "sustained_ms_per_step": sustained["ms_per_step"] if sustained else None, "sustained": sustained,
        "sampled": sampled,
        "step_ms_hip_events": step_events,          # rank 0's stream; `ms_per_step` / `value` are the wall-clock mean, max over ranks
        "step_rates": {
            "tflops_f32": FLOPS_FWD_BWD_PER_TRIPLE * B / (dt / args.steps) / 1e12,
            "frac_f32_mfma_peak": FLOPS_FWD_BWD_PER_TRIPLE * B / (dt / args.steps) / PEAK_F32_MFMA,
            # the same step priced on USEFUL flops: products with SAME-padding zeros left out (15.5 % of SURVEY 8d's count at 64x64) --
            # the figure a kernel cannot inflate by multiplying zeros, nor "exceed the peak" with by skipping them
            "flops_per_triple_all_taps": FLOPS_FWD_BWD_PER_TRIPLE,
            "flops_per_triple_useful": skipnew_flops_per_triple(H, W, useful=True),
            "tflops_useful": skipnew_flops_per_triple(H, W, useful=True) * B / (dt / args.steps) / 1e12,
            "useful_frac_f32_mfma_peak": skipnew_flops_per_triple(H, W, useful=True) * B / (dt / args.steps) / PEAK_F32_MFMA,
            "algorithmic_hbm_GBps": (BYTES_PER_TRIPLE * B + BYTES_PER_STEP_FIXED) / (dt / args.steps) / 1e9,
            "frac_hbm_peak": (BYTES_PER_TRIPLE * B + BYTES_PER_STEP_FIXED) / (dt / args.steps) / PEAK_HBM,
            # where the step sits on the roofline: its arithmetic intensity is far right of the ridge (peak flops / HBM
            # bandwidth = 19.7 FLOP/B), so the HBM ceiling does not bind and the attainable rate is the MFMA peak
            "arithmetic_intensity_flop_per_byte": FLOPS_FWD_BWD_PER_TRIPLE * B / (BYTES_PER_TRIPLE * B + BYTES_PER_STEP_FIXED),
            "ridge_flop_per_byte": PEAK_F32_MFMA / PEAK_HBM,
        },
    }
    if rank == 0 and world == 1 and not args.no_secondary:
        # "confirm on the box with a copy kernel and an FMA loop and report measured peaks too".  tools/libpeaks.so (built by
        # __graft_entry__.build(), not part of the product): an f32 matrix-core loop and a stream copy, outside the timed region.
        try:
            import ctypes
            pk = ctypes.CDLL(os.path.join(ROOT, "tools", "libpeaks.so"))
            pk.peak_mfma_f32_tflops.restype = ctypes.c_double
            pk.peak_hbm_copy_gbps.restype = ctypes.c_double
            pk.peak_hbm_copy_gbps.argtypes = [ctypes.c_int64]
            mf, cp = pk.peak_mfma_f32_tflops(), pk.peak_hbm_copy_gbps(1 << 30)
            line["peaks_measured"] = {
                "mfma_f32_tflops": mf, "hbm_copy_GBps": cp,
                "note": "v_mfma_f32_32x32x2_f32 loop on 8 waves per CU; 1 GiB device-to-device copy (read + write); nominal 157.3 TF/s, 8 TB/s",
                "step_frac_of_measured_mfma": (line["step_rates"]["tflops_f32"] / mf) if mf > 0 else None,
                "step_useful_frac_of_measured_mfma": (line["step_rates"]["tflops_useful"] / mf) if mf > 0 else None}
        except OSError:
            line["peaks_measured"] = None
    if dist.is_initialized():
        # Outside the timed region: what one step spends in compute and what the gradient all-reduce costs alone, so that the
        # per-N values can be read (exposed communication = ms_per_step - compute_ms; an overlapped schedule can hide at most
        # min(allreduce_ms, backward time)).  Every rank runs it; rank 0 reports its own clock.
        try:
            nrep = 5
            trl = trainer.translator
            ptr = (src.data_ptr(), ctx.data_ptr(), tgt.data_ptr())
            eng = getattr(trainer, "engine", None)
            est = stream_of(trainer)

            def compute_only():
                if eng is not None:
                    with torch.cuda.stream(est):
                        eng.forward_backward(src, ctx, tgt, sim_batch=B * world)
                        eng.adam(1e-4)
                else:
                    trl.dev_forward_backward(*ptr, B, sim_batch=B * world)
                    trl.dev_adam(1e-4)

            def allreduce_only():
                if eng is not None:
                    with torch.cuda.stream(est):
                        dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM)
                else:
                    trl.dp_allreduce_grads()

            def drain():
                if eng is None:
                    trl.sync()
                barrier()

            compute_only()
            drain()
            t0 = time.perf_counter()
            for _ in range(nrep):
                compute_only()
            drain()
            compute_ms = 1e3 * (time.perf_counter() - t0) / nrep
            allreduce_only()
            drain()
            t0 = time.perf_counter()
            for _ in range(nrep):
                allreduce_only()
            drain()
            ar_ms = 1e3 * (time.perf_counter() - t0) / nrep
            nbytes = trl.n_params * 4
            worlds = [None] * world
            dist.all_gather_object(worlds, int(trl.dp_world()[1]) if dp_client == "cabi" else world)
            line["comm"] = {"payload_MB": nbytes / 1e6, "allreduce_ms": ar_ms, "compute_ms_per_step": compute_ms, "ctx_dp_world_by_rank": worlds,
                            "busbw_GBps": (2.0 * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9) if world > 1 else None,
                            "client": dp_client, "overlap": dp_client == "cabi" or os.environ.get("CTX_DP_OVERLAP", "0") == "1"}
        except Exception as e:                      # diagnostics must never cost the bench line
            line["comm"] = {"error": repr(e)}
    if rank == 0:
        # dominant kernel: timed with HIP events on the handle's stream
        tr = trainer.translator
        ents = tr.profile_step(src.data_ptr(), ctx.data_ptr(), tgt.data_ptr(), B, lr=1e-4, iters=args.kernel_iters)
        tab = tr.kernel_table(ents)
        kname, k = next(iter(tab.items()))          # the kernel with the most time in a step
        per_launch_ms = k["ms"] / k["launches"]
        ach = k["flops"] / (k["ms"] * 1e-3)
        ach_useful = k["useful_flops"] / (k["ms"] * 1e-3)
        # f32: algorithmic flops against the f32-MFMA peak.  bf16x3: every algorithmic flop is 3 bf16 MFMA flops, so the
        # algorithmic rate is priced against (dense bf16 peak) / 3.
        peak = PEAK_F32_MFMA if args.precision == "f32" else PEAK_BF16_MFMA / 3
        line["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": ach / 1e12, "peak": peak / 1e12,
                            "unit": "TFLOP/s", "frac": ach / peak,
                            # `achieved` / `frac` count all 25 taps at every position; `*_useful` leave out the products
                            # with SAME-padding zeros (valid (position, tap) pairs only): the honest utilisation of the matrix pipe
                            "achieved_useful": ach_useful / 1e12, "useful_frac": ach_useful / peak,
                            "useful_share_of_flops": k["useful_flops"] / k["flops"] if k["flops"] else None,
                            "launches_per_step": k["launches"],
                            "avg_ms_per_launch": per_launch_ms, "flops_per_launch": k["flops"] / k["launches"],
                            "share_of_step_ms": k["ms"] / sum(t["ms"] for t in tab.values()), "traffic": None}
        # HBM bytes per launch of that kernel: PMC counters cannot be read from inside the process, so the
        # figure comes from the committed rocprofv3 --pmc passes of the same workload (profiles/*_hbm_traffic.json)
        import glob
        prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json" if args.precision == "f32" else f"*_hbm_traffic_{args.precision}.json")))
        if prof and B == 256:
            # newest file measured on THESE kernel sources (the file records the hash of csrc/ it was taken on); a file from
            # another build is named but not quoted
            sha = csrc_sha16()
            line["roofline"]["csrc_sha16"] = sha
            match = None
            for pf in reversed(prof):
                with open(pf) as f:
                    tr_json = json.load(f)
                if tr_json.get("csrc_sha16") == sha:
                    match = (pf, tr_json)
                    break
            if match:
                ent = match[1]["per_kernel"].get(kname)
                if ent:
                    line["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = os.path.basename(match[0])
            else:
                line["roofline"]["traffic_stale_source"] = os.path.basename(prof[-1])
        line["kernels"] = {n: {"ms": round(t["ms"], 4), "launches": t["launches"],
                               "tflops": round(t["flops"] / (t["ms"] * 1e-3) / 1e12, 2) if t["flops"] else None,
                               "tflops_useful": round(t["useful_flops"] / (t["ms"] * 1e-3) / 1e12, 2) if t["flops"] else None}
                           for n, t in tab.items()}
        # serialised launch groups of one step: all-taps and useful flops as the library counts them (cross-check of the closed forms)
        line["step_rates"]["profiled_gflop_per_step"] = sum(e["flops"] for e in ents) / 1e9
        line["step_rates"]["profiled_useful_gflop_per_step"] = sum(e["useful_flops"] for e in ents) / 1e9
        line["step_rates"]["profiled_serialised_ms"] = sum(e["ms"] for e in ents)
        if os.environ.get("BENCH_LAYER_TABLE"):
            with open(os.environ["BENCH_LAYER_TABLE"], "w") as f:
                for e in ents:
                    tf = e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] > 0 else 0
                    tu = e["useful_flops"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] > 0 else 0
                    f.write(f"{e['name']:34s} {e['kernel']:34s} {e['ms']:9.4f} ms {tf:8.2f} TF/s all taps {tu:8.2f} TF/s useful ({tu / (peak / 1e12):5.3f} of peak)\n")
        if args.host_buffers:
            hs, hc, ht = (x.cpu().numpy() for x in (src, ctx, tgt))
            tr.train_step(hs, hc, ht, lr=1e-4)
            t0 = time.perf_counter()
            for _ in range(5):
                tr.train_step(hs, hc, ht, lr=1e-4)
            line["host_buffer_frames_per_s"] = 5 * B / (time.perf_counter() - t0)
        if args.precision == "f32" and not args.no_split_leg and world == 1:
            # the same workload with split-bf16 products (CTX_PREC_BF16X3): reported beside, never as `value`
            del trainer
            t2 = DataParallelTrainer(H, W, DF, FEAT, max_batch=B, device=local_rank, seed=1234, precision="bf16x3")
            dt2, scal2, ev2 = timed(t2)
            line["bf16x3"] = {"value": args.steps * B / dt2, "unit": "frames/s", "ms_per_step": 1e3 * dt2 / args.steps,
                              "step_ms_hip_events": ev2,
                              "tflops_algorithmic": FLOPS_FWD_BWD_PER_TRIPLE * B / (dt2 / args.steps) / 1e12,
                              "loss_after": scal2["loss"], "loss_rel_diff_vs_f32": abs(scal2["loss"] - scal["loss"]) / abs(scal["loss"]),
                              "note": "products a*b evaluated as hi*hi + hi*lo + lo*hi on bf16 MFMA with f32 accumulation "
                                      "(~1e-5 relative, tests/test_gpu_split.py); everything else f32"}
            del t2
        if not args.no_secondary and world == 1 and args.precision == "f32":
            try:
                del trainer
            except NameError:
                pass
            line["secondary"] = secondary_legs(local_rank)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(batch=32, steps=10)
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
