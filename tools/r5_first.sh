#!/bin/bash
# round 5, first GPU call: the new parity tests, the clock trace, counters of the dominant implicit-GEMM kernels, a bench line of this box
set -x
O=gpurun_out/r5a; mkdir -p $O
python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_dp_two_ranks.py -m gpu -q -s > $O/pytest_new.txt 2>&1
tail -5 $O/pytest_new.txt
tools/clock_trace.bin 3 > $O/clock_trace.txt 2>&1
python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
bash tools/pmc_kernel.sh "igemm_kernel" r5a > $O/pmc.log 2>&1
rm -rf $O/pmc_igemm_kernel/p*/  # keep the summary only
ls $O
