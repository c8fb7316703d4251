for v in 4 8 16 4 8 16; do
  CTX_SPLIT_MINCH=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary --no-sampled --sustained-s 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('minch $v ms/step', round(d['ms_per_step'],3), ' FC:', [(n.split('<')[1][:-1], k[n]['ms']) for n in k if 'Plain' in n and 'Conv' not in n], 'reduce', [k[n]['ms'] for n in k if 'reduce' in n])"
done
