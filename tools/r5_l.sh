O=$PWD/gpurun_out/r5l; mkdir -p $O; R=$PWD
python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_reference_sizes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E  " | head
cd /tmp && export TMPDIR=/tmp
for K in translate; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$K -- python $R/tools/reward_kernels.py run $K > $O/run_$K.txt 2>&1
  python $R/tools/reward_kernels.py report $O/tr_$K $O/kernels_$K.txt > /dev/null
  rm -rf $O/tr_$K
done
cd $R
tail -12 $O/kernels_translate.txt
python tools/reward_latency.py 200 2>&1 | head -3
CTX_WCONVT=15 python tools/reward_latency.py 200 2>&1 | head -3
