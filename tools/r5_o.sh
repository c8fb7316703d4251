# conv-gather locality: problem order of the position-major conv on grids of > 16 positions.
#   CTX_BALANCE=3 balanced runs (round 4), 1 row-major runs, 9 Z-order runs.   HBM bytes per launch + step time.
O=gpurun_out/r5o; mkdir -p $O
for B in 3 1 9 3 9; do
  CTX_BALANCE=$B python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary --no-sampled --sustained-s 0 > $O/bench_$B.json 2>/dev/null
  python - <<P
import json
d = json.loads(open("$O/bench_$B.json").read().strip().splitlines()[-1])
k = d["kernels"]["igemm<ConvGather,Plain>"]
print("balance $B  ms/step", round(d["ms_per_step"], 3), " conv gather", k)
P
done
for B in 3 1 9; do
  CTX_BALANCE=$B bash tools/hbm_traffic.sh r5o_$B f32 > /dev/null 2>&1
  python - <<P
import json
d = json.load(open("gpurun_out/r5o_$B/hbm_traffic.json"))
for k in ("igemm<ConvGather,Plain>", "igemm<WgradBig,WgradSmall>", "wconvt_kernel"):
    v = d["per_kernel"][k]; print("balance $B", k, "MB per launch", round(v["hbm_bytes_per_launch"] / 1e6, 1), "launches", v["launches_profiled"])
P
done
