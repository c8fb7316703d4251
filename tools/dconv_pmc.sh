# PMC passes over one layer of tools/dconv_bench.bin (kernel-trace only alongside the counters).  usage: tools/dconv_pmc.sh <layer index>
L=${1:-2}
R=$PWD; O=$R/gpurun_out/dconv_pmc_$L; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_FLAT" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o r -- $R/tools/dconv_bench.bin $L > $O/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
O="$O"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(O+"/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
with open(O+"/summary.txt","w") as out:
    for k,v in agg.items():
        out.write(k+"\n")
        for c,x in sorted(v.items()):
            out.write(f"    {c:28s} {x/max(1,cnt[(k,c)]):16.0f}  per launch (n={cnt[(k,c)]})\n")
print(open(O+"/summary.txt").read()[:5000])
PY
