mkdir -p gpurun_out/r6j
python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_reference_sizes.py "tests/test_gpu_parity.py::test_adam_trajectory_matches_oracle" -m gpu -x -q -s > gpurun_out/r6j/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r6j/pytest.log | tail -3
grep -n "un-aligned\|gradient deviation\|rel-L2 per tensor\|ContextAEReal 64x64" gpurun_out/r6j/pytest.log | cut -c1-6000
