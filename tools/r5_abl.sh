#!/bin/bash
# ablation builds of the narrow-channel direct kernels (tools/dconv_bench.hip with -DDC_ABL_*): wrong results, right amount of the other work
O=gpurun_out/r5b; mkdir -p $O
tools/clock_trace.bin 3 > $O/clock_trace.txt 2>&1
for v in base NOMFMA NOLDG NOLAND NOEPI NOLOOP; do
  DCB_AUTO_ONLY=1 timeout 120 tools/dconv_bench_$v.bin > $O/abl_$v.txt 2>&1
done
python tools/real_layer_table.py > $O/real_layers.txt 2>&1
paste -d'|' <(cut -c1-40 $O/abl_base.txt) <(for v in base NOMFMA NOLDG NOLAND NOEPI NOLOOP; do grep -o "[0-9.]* ms" $O/abl_$v.txt | tr '\n' ' ' > /tmp/$v.col; done; echo) | head -2
for v in base NOMFMA NOLDG NOLAND NOEPI NOLOOP; do echo "== $v"; grep -o "^.\{38\}\|[0-9.]* ms" $O/abl_$v.txt | paste - - | head -30; done
