#!/bin/bash
# Kernel timeline of one headline training step with the stream lanes on (rocprofv3 --kernel-trace -> tools/timeline.py): which
# HIP stream ran on which hardware queue, what ran beside what, idle time.   tools/step_trace.sh OUTDIR ["ENV=.. ENV2=.."]
OUT=${1:-gpurun_out/step_trace}; ENVS=$2
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
env $ENVS rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/rp -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-sampled --no-split-leg --sustained-s 0 --kernel-iters 1 > $R/$OUT/run.log 2>&1
TIMELINE_CUT=${TIMELINE_CUT:-loss_final_kernel} python $R/tools/timeline.py $R/$OUT/rp 2 > $R/$OUT/timeline.txt 2>&1
rm -rf $R/$OUT/rp
head -1 $R/$OUT/timeline.txt; tail -n 1 $R/$OUT/timeline.txt
awk '/ us /{print $5}' $R/$OUT/timeline.txt | sort | uniq -c
