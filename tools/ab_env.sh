#!/bin/bash
# Whole-step A/B of environment switches on ONE box (decisions are made on whole steps, not launch times).
# usage: tools/ab_env.sh OUT "VAR=1 VAR2=0" "VAR3=x" ...   -- one quick bench.py run per quoted variant ("" = defaults), repeated REPS times
OUT=$1; shift
REPS=${REPS:-2}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for rep in $(seq 1 $REPS); do
  for v in "default" "$@"; do
    envs=""; [ "$v" != "default" ] && envs="$v"
    ms=$(env $envs python bench.py --steps ${STEPS:-30} --warmup 8 --no-cpu-baseline --no-secondary --no-split-leg --no-sampled --sustained-s 0 --kernel-iters 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print('%.3f %.3f' % (l['ms_per_step'], l['step_ms_hip_events']['median']))")
    echo "rep $rep  [$v]  ms_per_step / event median: $ms" | tee -a "$OUT"
  done
done
