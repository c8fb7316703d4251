#!/usr/bin/env python3
"""bench.py -- translator training throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

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


def cpu_baseline(batch, steps):
    """The numpy oracle (a port of the reference's arithmetic; the reference's TensorFlow path cannot
    run here) timed on the host cores on a bounded sample: `steps` train steps at batch `batch`."""
    from oracle import ctx_oracle as o
    cfg = o.SkipNewConfig(H=H, W=W, df_dim=DF, gf_dim=DF, featsize=FEAT)
    p = o.init_params(cfg, 0, np.float32)
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    rng = np.random.default_rng(0)
    src, ctx, tgt = (rng.uniform(-1, 1, (batch, H, W, 3)).astype(np.float32) for _ in range(3))
    o.train_step(p, m, v, 1, src, ctx, tgt, 1e-4, cfg)          # warm-up (page-in, BLAS threads)
    t0 = time.perf_counter()
    for t in range(2, 2 + steps):
        o.train_step(p, m, v, t, src, ctx, tgt, 1e-4, cfg)
    dt = time.perf_counter() - t0
    threads = os.cpu_count()
    try:                                     # the threads numpy's BLAS actually runs the matmuls on
        from threadpoolctl import threadpool_info
        blas = [t["num_threads"] for t in threadpool_info() if t.get("user_api") == "blas"]
        if blas:
            threads = max(blas)
    except Exception:
        pass
    numpy_rate = batch * steps / dt
    out = {"value": numpy_rate, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{steps} fwd+bwd+Adam steps of the numpy oracle at batch {batch}, f32, {dt:.1f}s, "
                     f"BLAS threads {threads} of {os.cpu_count()} logical CPUs (im2col/col2im parts are single-threaded)",
           "numpy_oracle_frames_per_s": numpy_rate}
    # second CPU statement of the same arithmetic: torch-CPU (oneDNN convs, autograd), all host threads
    try:
        import torch
        from tests import _torch_ref as tref
        nthreads = torch.get_num_threads()
        tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
        ts, tc, tt = (torch.tensor(a) for a in (src, ctx, tgt))
        tm = {k: torch.zeros_like(v) for k, v in tp.items()}
        tv = {k: torch.zeros_like(v) for k, v in tp.items()}

        def tstep(t):
            loss = tref.forward(tp, ts, tc, tt, H, W, DF)["loss"]
            grads = torch.autograd.grad(loss, list(tp.values()))
            with torch.no_grad():                          # TF-Adam (train_script.py:128 defaults)
                lr_t = 1e-4 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
                for (k, w), g in zip(tp.items(), grads):
                    tm[k].mul_(0.9).add_(g, alpha=0.1)
                    tv[k].mul_(0.999).addcmul_(g, g, value=0.001)
                    w.sub_(lr_t * tm[k] / (tv[k].sqrt() + 1e-8))

        tstep(1)
        t0 = time.perf_counter()
        for t in range(2, 2 + steps):
            tstep(t)
        dtt = time.perf_counter() - t0
        torch_rate = batch * steps / dtt
        out["torch_cpu_frames_per_s"] = torch_rate
        if torch_rate > numpy_rate:
            out.update(value=torch_rate, cores=nthreads,
                       sample=f"{steps} fwd+bwd+Adam steps at batch {batch}, f32: torch-CPU statement (oneDNN, autograd, {nthreads} threads) "
                              f"{dtt:.1f}s = {torch_rate:.1f} frames/s; numpy oracle ({threads} BLAS threads) {dt:.1f}s = {numpy_rate:.1f} frames/s; "
                              f"{os.cpu_count()} logical CPUs")
    except Exception as e:                                   # the numpy figure stands
        out["torch_cpu_error"] = repr(e)[:200]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-iters", type=int, default=5)
    ap.add_argument("--host-buffers", action="store_true",
                    help="also time ctx_train_step on host (numpy) frames: the PCIe-inclusive rate quoted in DESIGN.md")
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default=os.environ.get("CTX_PRECISION", "f32"),
                    help="arithmetic of the contractions: exact f32 MFMA (default) or split-bf16 products (DESIGN.md section 6b)")
    ap.add_argument("--no-split-leg", action="store_true",
                    help="skip the extra bf16x3 measurement that a default (f32) run appends as line['bf16x3']")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from imitation_from_observation_amd.dp import DataParallelTrainer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":     # (the env switch: a one-rank RCCL group, to exercise the N > 1 code on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist.barrier()
        # RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would otherwise land AFTER the
        # JSON line at exit: flush it now so that the JSON line is the last line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)

    B = args.batch
    trainer = DataParallelTrainer(H, W, DF, FEAT, max_batch=B, device=local_rank, seed=1234, precision=args.precision)
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
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr_.step(src, ctx, tgt, lr=1e-4)
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_], device="cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_, tr_.scalars()

    dt, scal = timed(trainer)

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
                   "parallelism": f"dp{world}" + (" + RCCL grad all-reduce" if world > 1 else "")},
        "loss_after": scal["loss"],
        "step_rates": {
            "tflops_f32": FLOPS_FWD_BWD_PER_TRIPLE * B / (dt / args.steps) / 1e12,
            "frac_f32_mfma_peak": FLOPS_FWD_BWD_PER_TRIPLE * B / (dt / args.steps) / PEAK_F32_MFMA,
            "algorithmic_hbm_GBps": (BYTES_PER_TRIPLE * B + BYTES_PER_STEP_FIXED) / (dt / args.steps) / 1e9,
            "frac_hbm_peak": (BYTES_PER_TRIPLE * B + BYTES_PER_STEP_FIXED) / (dt / args.steps) / PEAK_HBM,
            # where the step sits on the roofline: its arithmetic intensity is far right of the ridge (peak flops / HBM
            # bandwidth = 19.7 FLOP/B), so the HBM ceiling does not bind and the attainable rate is the MFMA peak
            "arithmetic_intensity_flop_per_byte": FLOPS_FWD_BWD_PER_TRIPLE * B / (BYTES_PER_TRIPLE * B + BYTES_PER_STEP_FIXED),
            "ridge_flop_per_byte": PEAK_F32_MFMA / PEAK_HBM,
        },
    }
    if dist.is_initialized():
        # Outside the timed region: what one step spends in compute and what the gradient all-reduce costs alone, so that the
        # per-N values can be read (exposed communication = ms_per_step - compute_ms; an overlapped schedule can hide at most
        # min(allreduce_ms, backward time)).  Every rank runs it; rank 0 reports its own clock.
        try:
            eng = trainer.engine
            nrep = 5
            with torch.cuda.stream(eng.stream):
                eng.forward_backward(src, ctx, tgt, sim_batch=B * world)
                eng.adam(1e-4)
            barrier()
            t0 = time.perf_counter()
            with torch.cuda.stream(eng.stream):
                for _ in range(nrep):
                    eng.forward_backward(src, ctx, tgt, sim_batch=B * world)
                    eng.adam(1e-4)
            barrier()
            compute_ms = 1e3 * (time.perf_counter() - t0) / nrep
            with torch.cuda.stream(eng.stream):
                dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM)
            barrier()
            t0 = time.perf_counter()
            with torch.cuda.stream(eng.stream):
                for _ in range(nrep):
                    dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM)
            barrier()
            ar_ms = 1e3 * (time.perf_counter() - t0) / nrep
            nbytes = eng.grads.numel() * 4
            line["comm"] = {"payload_MB": nbytes / 1e6, "allreduce_ms": ar_ms, "compute_ms_per_step": compute_ms,
                            "busbw_GBps": (2.0 * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9) if world > 1 else None,
                            "overlap": os.environ.get("CTX_DP_OVERLAP", "0") == "1"}
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
        # f32: algorithmic flops against the f32-MFMA peak.  bf16x3: every algorithmic flop is 3 bf16 MFMA flops, so the
        # algorithmic rate is priced against (dense bf16 peak) / 3.
        peak = PEAK_F32_MFMA if args.precision == "f32" else PEAK_BF16_MFMA / 3
        line["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": ach / 1e12, "peak": peak / 1e12,
                            "unit": "TFLOP/s", "frac": ach / peak, "launches_per_step": k["launches"],
                            "avg_ms_per_launch": per_launch_ms, "flops_per_launch": k["flops"] / k["launches"],
                            "share_of_step_ms": k["ms"] / sum(t["ms"] for t in tab.values()), "traffic": None}
        # HBM bytes per launch of that kernel: PMC counters cannot be read from inside the process, so the
        # figure comes from the committed rocprofv3 --pmc passes of the same workload (profiles/*_hbm_traffic.json)
        import glob
        prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json" if args.precision == "f32" else f"*_hbm_traffic_{args.precision}.json")))
        if prof and B == 256:
            with open(prof[-1]) as f:
                tr_json = json.load(f)
            ent = tr_json["per_kernel"].get(kname)
            if ent:
                line["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = os.path.basename(prof[-1])
        line["kernels"] = {n: {"ms": round(t["ms"], 4), "launches": t["launches"],
                               "tflops": round(t["flops"] / (t["ms"] * 1e-3) / 1e12, 2) if t["flops"] else None}
                           for n, t in tab.items()}
        if os.environ.get("BENCH_LAYER_TABLE"):
            with open(os.environ["BENCH_LAYER_TABLE"], "w") as f:
                for e in ents:
                    tf = e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] > 0 else 0
                    f.write(f"{e['name']:34s} {e['kernel']:34s} {e['ms']:9.4f} ms {tf:8.2f} TF/s\n")
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
            dt2, scal2 = timed(t2)
            line["bf16x3"] = {"value": args.steps * B / dt2, "unit": "frames/s", "ms_per_step": 1e3 * dt2 / args.steps,
                              "tflops_algorithmic": FLOPS_FWD_BWD_PER_TRIPLE * B / (dt2 / args.steps) / 1e12,
                              "loss_after": scal2["loss"], "loss_rel_diff_vs_f32": abs(scal2["loss"] - scal["loss"]) / abs(scal["loss"]),
                              "note": "products a*b evaluated as hi*hi + hi*lo + lo*hi on bf16 MFMA with f32 accumulation "
                                      "(~1e-5 relative, tests/test_gpu_split.py); everything else f32"}
            del t2
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(batch=32, steps=3)
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
