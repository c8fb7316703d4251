# whole-step time against the slab count of the bias-gradient column sums (they run on the side lane): bash tools/colsum_ab.sh
for i in 1 2 3; do for c in 64 128 256 512; do
  echo -n "splits=$c: "; CTX_COLSUM_SPLITS=$c python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-split-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['kernels']['colsum']['ms'])"
done; done
