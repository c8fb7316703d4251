# A/B of the XCD swizzle bits (1 conv, 2 transposed conv, 4 filter gradient): fetch traffic per launch and step time
R=$PWD
for V in ${@:-0 5 7}; do
  O=$R/gpurun_out/swz$V; mkdir -p $O/WRITE_SIZE
  (cd /tmp && TMPDIR=/tmp CTX_XCD_SWIZZLE=$V CTX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/FETCH_SIZE -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg --kernel-iters 1 > $O/log 2>&1)
  echo "swizzle=$V"; python tools/hbm_aggregate.py $O f32 $O/t.json 2>/dev/null | grep "ConvGather,Plain\|WgradBig,\|ConvTGather" | sed "s/'launches_profiled': 27, //; s/'fetch_bytes.*//"
  CTX_XCD_SWIZZLE=$V python bench.py --no-cpu-baseline --steps 20 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['bf16x3']['ms_per_step'],3), {k.replace('igemm',''):v['ms'] for k,v in list(d['kernels'].items())[:3]})"
done
