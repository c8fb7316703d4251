#!/bin/bash
# Kernel timeline of one config-4 step (Inception-v3 front end + ContextAEInception2 + Adam on one stream) with the translator's
# stream lanes on and off: where the chip idles.   tools/config4_trace.sh OUTDIR
OUT=${1:-gpurun_out/c4trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ov in ${OVS:-1 0}; do
  CTX_OVERLAP=$ov rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/ov$ov -- python $GRAFT_REPO_ROOT/tools/bench_config4.py 125 64 f32only > $GRAFT_REPO_ROOT/$OUT/run_ov$ov.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/$OUT/ov$ov 2 > $GRAFT_REPO_ROOT/$OUT/timeline_ov$ov.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/$OUT/ov$ov
done
tail -3 $GRAFT_REPO_ROOT/$OUT/timeline_ov1.txt $GRAFT_REPO_ROOT/$OUT/timeline_ov0.txt
