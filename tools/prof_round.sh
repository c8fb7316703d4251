# Regenerates the per-round measurement set under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   gpurun -- 'bash tools/prof_round.sh d'
set -x
TAG=${1:-x}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
BENCH_LAYER_TABLE=$O/layer_table.txt python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/bench_reward.py > $O/reward.txt 2>&1
python tools/bench_real.py > $O/real.txt 2>&1
cd /tmp && export TMPDIR=/tmp
# kernels serialised on one stream (CTX_OVERLAP=0) so per-kernel durations are comparable with bench.py's event table
CTX_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rp.log 2>&1
# the same with the three stream lanes on (what a normal step runs)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_lanes -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rp_lanes.log 2>&1
cd $R
rm -f $O/rp/*kernel_trace.csv $O/rp_lanes/*kernel_trace.csv
ls -R $O | head -30
