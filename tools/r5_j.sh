O=gpurun_out/r5j; mkdir -p $O
CTX_TRACE_LAUNCH=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split-leg --no-secondary --no-sampled --sustained-s 0 > $O/b.json 2> $O/trace.txt
grep igemm $O/trace.txt | sort | uniq -c | sort -k1nr | head -80
