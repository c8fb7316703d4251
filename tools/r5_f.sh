#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
(echo "== dconv_fwd_kernel (CTX_DCONV=1), phases per tile, cycles"; CTX_DCONV=1 DCB_AUTO_ONLY=1 tools/dconv_bench_trace.bin 2>&1 | grep -v " dw "; echo; echo "== default (dconv2 for the one-class layers), phases per slice of the compute waves, cycles"; DCB_AUTO_ONLY=1 tools/dconv_bench_trace.bin 2>&1 | grep -v " dw ") | cut -c1-330 > $O/dconv_phase_stamps.txt
python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; l=json.load(open('$O/bench.json')); print(l['ms_per_step'], {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in l.get('secondary',{}).items()})"
