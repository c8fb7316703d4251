#!/bin/bash
# Kernel trace of the reward hook's two fetches at the reference's batch of 25 (rllab/sampler/base.py:216-218, 234-235): per-call wall time
# and the kernels of one call in order (start offset, duration) -- where a 1.4 ms translate call goes.   tools/reward_trace.sh OUTDIR
OUT=${1:-gpurun_out/reward_trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/bench_reward.py > $GRAFT_REPO_ROOT/$OUT/reward_calls.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python $GRAFT_REPO_ROOT/tools/translate_small.py 64 64 25 skipnew > $GRAFT_REPO_ROOT/$OUT/translate_small.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: kernels after the last big gap (> 200 us)
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if st[i] - en[i - 1] > 200000: cut = i
last = rows[cut:]
t0 = int(last[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
span = int(last[-1]["End_Timestamp"]) - t0
with open("$OUT/last_call_kernels.txt", "w") as o:
    o.write(f"kernels {len(last)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us\n")
    for r in last:
        o.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  {r['Kernel_Name'][:90]}\n")
print(open("$OUT/last_call_kernels.txt").read()[:6000])
PY
