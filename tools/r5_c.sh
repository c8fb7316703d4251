#!/bin/bash
O=gpurun_out/r5d; mkdir -p $O
DCB_AUTO_ONLY=1 tools/dconv_bench_base.bin > $O/abl_base.txt 2>&1
DCB_AUTO_ONLY=1 tools/dconv_bench_pipe.bin > $O/abl_pipe.txt 2>&1
paste <(cut -c1-36 $O/abl_base.txt) <(grep -o "[0-9.]* ms" $O/abl_base.txt) <(grep -o "[0-9.]* ms" $O/abl_pipe.txt)
python -m pytest tests/test_gpu_real.py tests/test_gpu_bench_shapes.py -q -x 2>&1 | tail -3
python tools/real_layer_table.py 2>&1 | head -8
cd /tmp && export TMPDIR=/tmp
true
cd $GRAFT_REPO_ROOT

