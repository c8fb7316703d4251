# kernel timeline of one ContextAEReal step (36x64, B = 256, fused step) with the lanes on
R=$PWD; O=$R/gpurun_out/r5y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/rp -- python $R/tools/real_step_loop.py > $O/run.log 2>&1
TIMELINE_CUT=loss_final_kernel python $R/tools/timeline.py $O/rp 20 > $O/timeline.txt 2>&1
rm -rf $O/rp
head -3 $O/timeline.txt; tail -n 3 $O/timeline.txt
