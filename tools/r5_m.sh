O=$PWD/gpurun_out/r5m; mkdir -p $O; R=$PWD
for c in HEAD 5613548; do
  d=$R/build_ab/$c
  cp -r $R/tests $R/oracle $d/ ; cp $R/pytest.ini $R/conftest.py $d/ 2>/dev/null
  (cd $d && python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "adam_trajectory" > $O/adam_$c.txt 2>&1)
  echo "== $c"; grep -E "^E  |passed|failed" $O/adam_$c.txt | head -6
done
