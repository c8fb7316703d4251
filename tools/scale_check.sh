#!/bin/bash
# N > 1 readiness check for an 8-GPU MI355X node (the build's boxes have one GPU; VERDICT r3 "next round" 10).
#   tools/scale_check.sh [N ...]        default: 1 2 4 8 (those that fit the node)
# Per N it runs `python bench.py --gpus N` (C-ABI RCCL client, ctx_dp_train_step: gradient buckets from inside backward) and prints
#   value (frames/s), ms/step, comm.compute_ms_per_step, comm.allreduce_ms, comm.busbw_GBps, exposed communication
#   (ms_per_step - compute_ms_per_step), weak-scaling efficiency vs N = 1, the sampled-step leg, and checks that every rank's handle
#   saw N ranks (ctx_dp_world) -- bench.py itself refuses to fall back to another client when librccl is missing on a rank.
# Output: one table on stdout, the JSON lines under gpurun_out/scale_check/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/scale_check
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
NS=${@:-1 2 4 8}
echo "GPUs visible: $NGPU"
# ONE_GPU=1: the same commands with every rank on device 0 (bench.py BENCH_ONE_GPU=1: torch group on gloo, ctx_dp_* through the shared-memory
# stand-in of tests/fake_rccl) and a small batch -- a dry run of the N > 1 code path on a 1-GPU box; the table it prints is NOT a scaling curve
EXTRA=""
if [ "${ONE_GPU:-0}" = "1" ]; then
  export BENCH_ONE_GPU=1 CTX_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so FAKE_RCCL_TIMEOUT_S=300
  NGPU=8; EXTRA="--batch 8 --sustained-s 0.2 --kernel-iters 1"; STEPS=${STEPS:-3}
  echo "ONE_GPU=1: N ranks time-share device 0 -- code-path run, not a scaling measurement"
fi
# every rank's handle must report the world it was initialised with (one process per GPU, RCCL behind the C ABI)
for N in $NS; do
  [ "$N" -gt "$NGPU" ] && { echo "N=$N: skipped ($NGPU GPUs visible)"; continue; }
  if [ "$N" -gt 1 ] && [ "${ONE_GPU:-0}" != "1" ]; then
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) tools/dp_world_check.py > $OUT/world_$N.txt 2>&1 \
      && echo "N=$N: ctx_dp_world ok on every rank: $(grep -c 'world ok' $OUT/world_$N.txt) ranks" || { echo "N=$N: ctx_dp_world check FAILED (see $OUT/world_$N.txt)"; tail -5 $OUT/world_$N.txt; }
  fi
  python bench.py --gpus $N --steps ${STEPS:-30} --warmup 10 --no-cpu-baseline --no-secondary --no-split-leg $EXTRA > $OUT/bench_$N.json 2> $OUT/bench_$N.err || { echo "N=$N: bench.py failed"; tail -5 $OUT/bench_$N.err; }
done
python - <<'PY'
import glob, json, os
rows = {}
for f in sorted(glob.glob("gpurun_out/scale_check/bench_*.json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rows[l["n_gpus"]] = l
if not rows:
    raise SystemExit("no bench lines")
base = rows.get(1, {}).get("value")
print(f"{'N':>2} {'frames/s':>10} {'ms/step':>8} {'compute':>8} {'allreduce':>9} {'busbw GB/s':>10} {'exposed':>8} {'eff':>6} {'sampled ms':>10}  client")
for n in sorted(rows):
    l = rows[n]
    c = l.get("comm") or {}
    comp, ar, bw = c.get("compute_ms_per_step"), c.get("allreduce_ms"), c.get("busbw_GBps")
    exp = l["ms_per_step"] - comp if comp else None
    eff = l["value"] / (n * base) if base else None
    smp = (l.get("sampled") or {}).get("ms_per_step")
    f = lambda x, w, p=2: (f"{x:{w}.{p}f}" if x is not None else " " * (w - 1) + "-")
    print(f"{n:>2} {l['value']:>10.0f} {l['ms_per_step']:>8.2f} {f(comp, 8)} {f(ar, 9)} {f(bw, 10, 1)} {f(exp, 8)} {f(eff, 6, 3)} {f(smp, 10)}  {l['config'].get('dp_client', '')[:40]}")
PY
