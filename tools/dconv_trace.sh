# kernel-trace stats of one layer of tools/dconv_bench.bin (fwd: 0..13, wgrad: 100..107).  usage: tools/dconv_trace.sh <layer> [<layer> ...]
R=$PWD; cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  O=$R/gpurun_out/dconv_trace_$L; mkdir -p $O
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- $R/tools/dconv_bench.bin $L > $O/log.txt 2>&1
  echo "== layer $L"; cut -d, -f1-4 $O/r_kernel_stats.csv | cut -c1-150
done
