# Round-2 diagnostics: per-layer table of ContextAEReal, the encode cliff, ContextSkipNew's 3-channel layers on dconv.
#   gpurun -- 'bash tools/diag_r2.sh <tag>'
TAG=${1:-x}; O=gpurun_out/$TAG; mkdir -p $O
python tools/real_layer_table.py > $O/real_layers.txt 2>&1
python tools/encode_time.py > $O/encode_time.txt 2>&1
CTX_DCONV_C3=1 BENCH_LAYER_TABLE=$O/layer_table_dc3.txt python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dc3.json 2>&1
