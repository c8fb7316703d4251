# PMC passes over one bench step in the given precision (kernel-trace only alongside the counters)
set -x
PREC=${1:-bf16x3}
R=$PWD; O=$R/gpurun_out/pmc_$PREC; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --kernel-iters 1 --precision $PREC --no-split-leg --no-secondary --sustained-s 0 > $O/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
O="$O"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(O+"/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "igemm" not in k: continue
        k=("split " if "igemm_split_kernel" in k else "")+k.split("<",1)[1].split(">(")[0].replace("ctx::","")
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        cnt[(k,r["Counter_Name"])]+=1
with open(O+"/summary.txt","w") as out:
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))[:12]:
        out.write(k+"\n")
        for c,x in sorted(v.items()):
            out.write(f"    {c:28s} {x:16.0f}  (n={cnt[(k,c)]})\n")
print(open(O+"/summary.txt").read()[:6000])
PY
