# HBM bytes per launch per kernel from PMC counters (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes, kernel-trace only alongside; bytes = FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024.
#   gpurun -- 'bash tools/hbm_traffic.sh e [f32|bf16x3]'   ->  gpurun_out/<tag>/hbm_traffic[_bf16x3].json
set -x
TAG=${1:-x}; PREC=${2:-f32}
R=$PWD; O=$R/gpurun_out/$TAG/pmc_hbm_$PREC; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  CTX_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg --no-secondary --no-sampled --sustained-s 0 --kernel-iters 1 --precision $PREC > $O/$C.log 2>&1
done
cd $R
SUF=""; [ "$PREC" != "f32" ] && SUF="_$PREC"
python tools/hbm_aggregate.py $O $PREC $R/gpurun_out/$TAG/hbm_traffic$SUF.json
