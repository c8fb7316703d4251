O=$PWD/gpurun_out/r5k; mkdir -p $O; R=$PWD
CTX_TRACE_LAUNCH=1 python tools/reward_kernels.py run translate 2> $O/shapes.txt
cd /tmp && export TMPDIR=/tmp
for K in encode translate; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$K -- python $R/tools/reward_kernels.py run $K > $O/run_$K.txt 2>&1
  python $R/tools/reward_kernels.py report $O/tr_$K $O/kernels_$K.txt > /dev/null
  rm -rf $O/tr_$K
done
cd $R; python tools/reward_latency.py 200 > $O/latency.txt 2>&1
cat $O/shapes.txt | grep igemm; cat $O/kernels_translate.txt; cat $O/kernels_encode.txt; cat $O/latency.txt
