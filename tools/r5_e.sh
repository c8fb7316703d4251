#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_real.py -q -x  2>&1 | tail -5
DCB_AUTO_ONLY=1 CTX_DCONV=1 timeout 60 tools/dconv_bench_pipe.bin 2>&1 | grep -v " dw " > $O/old.txt
DCB_AUTO_ONLY=1 timeout 60 tools/dconv_bench_pipe.bin 2>&1 | grep -v " dw " > $O/new.txt
paste <(cut -c1-36 $O/old.txt) <(grep -o "[0-9.]* ms" $O/old.txt) <(grep -o "[0-9.]* ms" $O/new.txt) <(grep -o "dconv2.*slices [0-9]\|auto" $O/new.txt)
DCB_AUTO_ONLY=1 timeout 60 tools/dconv_bench_trace.bin 2>&1 | grep phases | cut -c1-300 > $O/trace.txt; cat $O/trace.txt
python tools/real_layer_table.py 2>&1 | head -3
python tools/bench_real.py 2>&1 | grep "train_step (frames"
