R=$PWD; O=$R/gpurun_out/idle; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/skip -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-leg --kernel-iters 0 > $O/skip.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/real -o r -- python $R/tools/real_step_loop.py > $O/real.log 2>&1
cd $R
python tools/gpu_idle.py $O/skip/r_kernel_trace.csv 0.5
python tools/gpu_idle.py $O/real/r_kernel_trace.csv 0.5
