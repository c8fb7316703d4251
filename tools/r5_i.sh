# bf16x3 side-line: the shipped split kernel against the same kernel with the f32 -> (hi, lo) conversion removed (timing ablation,
# wrong numbers): the upper bound of what operands pre-split by their producers could buy.   gpurun -- 'bash tools/r5_i.sh'
O=gpurun_out/r5i; mkdir -p $O
A="--steps 20 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary --no-sampled --sustained-s 0 --precision bf16x3"
python bench.py $A > $O/bf16x3_shipped.json 2> $O/bf16x3_shipped.err
cp imitation_from_observation_amd/libctxtrans.so /tmp/keep.so
cp tools/abl/libctxtrans_nocvt.so imitation_from_observation_amd/libctxtrans.so
python bench.py $A > $O/bf16x3_nocvt.json 2> $O/bf16x3_nocvt.err
cp /tmp/keep.so imitation_from_observation_amd/libctxtrans.so
python - <<'P'
import json
for t in ("shipped", "nocvt"):
    d = json.loads(open(f"gpurun_out/r5i/bf16x3_{t}.json").read().strip().splitlines()[-1])
    print(t, "ms_per_step", round(d["ms_per_step"], 3))
    for k, v in list(d.get("kernels", {}).items())[:12]:
        print("   ", k, v)
P
