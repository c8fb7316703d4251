mkdir -p gpurun_out/r6h
timeout 120 tools/convt3m_bench.bin > gpurun_out/r6h/convt3m_bench.txt 2>&1; grep -n "TF/s" gpurun_out/r6h/convt3m_bench.txt
python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_gpu_real.py -m gpu -x -q > gpurun_out/r6h/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r6h/pytest.log | tail -3
REPS=2 STEPS=30 tools/ab_env.sh gpurun_out/r6h/ab_convt3m.txt "CTX_DIRECT3=15"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-leg > gpurun_out/r6h/bench.json 2> gpurun_out/r6h/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6h/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['step_rates']['useful_frac_f32_mfma_peak'])
for k,v in d.get('kernels',{}).items(): print(k, v)
print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('secondary',{}).items()})
PY
