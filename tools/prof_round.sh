# Regenerates the per-round measurement set under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   gpurun -- 'bash tools/prof_round.sh r3'
# Order matters for the last step: the HBM-traffic file records the hash of csrc/ it was measured on (tools/hbm_aggregate.py) and
# bench.py only quotes a file whose hash equals the tree's -- so it is regenerated LAST, on the build everything else used.
set -x
TAG=${1:-x}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
BENCH_LAYER_TABLE=$O/layer_table.txt python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-secondary --no-sampled --sustained-s 0 > /dev/null 2>&1
BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-secondary --no-sampled --sustained-s 0 > $O/bench_force_dist.json 2> /dev/null
python tools/bench_reward.py > $O/reward.txt 2>&1
python tools/reward_latency.py 300 > $O/reward_latency.txt 2>&1
python tools/bench_real.py > $O/real.txt 2>&1
python tools/real_layer_table.py > $O/real_layers.txt 2>&1
python tools/bench_config4.py 125 64 > $O/config4.txt 2>&1
python tools/frontend_layer_table.py > $O/frontend_layers.txt 2>&1
cd /tmp && export TMPDIR=/tmp
# kernels serialised on one stream (CTX_OVERLAP=0) so per-kernel durations are comparable with bench.py's event table
CTX_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-sampled --no-split-leg --sustained-s 0 > $O/rp.log 2>&1
# the same with the three stream lanes on (what a normal step runs)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_lanes -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-sampled --no-split-leg --sustained-s 0 > $O/rp_lanes.log 2>&1
cd $R
rm -f $O/rp/*kernel_trace.csv $O/rp_lanes/*kernel_trace.csv
bash tools/pmc_kernel.sh wconvt_kernel $TAG > /dev/null 2>&1
bash tools/hbm_traffic.sh $TAG f32 > $O/hbm.log 2>&1
ls -R $O | head -40
