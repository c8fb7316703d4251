# PMC passes (kernel-trace only alongside) over one bench step, summarised for the kernels whose name matches the regex $1.
#   gpurun -- 'bash tools/pmc_kernel.sh wconvt_kernel r3e'
#   PMC_CMD="python $PWD/tools/real_step_loop.py" bash tools/pmc_kernel.sh "dconv_fwd|dconv_wgrad|convt3|c3conv|c3wgrad" r4real     (another workload)
set -x
PAT=${1:-wconvt_kernel}; TAG=${2:-x}
R=$PWD; O=$R/gpurun_out/$TAG/pmc_$(echo $PAT | tr -c "A-Za-z0-9_\n" "_"); mkdir -p $O
CMD=${PMC_CMD:-python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --kernel-iters 1 --no-split-leg --no-secondary --no-sampled --sustained-s 0}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  CTX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o r -- $CMD > $O/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, re
O="$O"; PAT="$PAT"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
clk=collections.defaultdict(list)
for f in glob.glob(O+"/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not re.search(PAT, k): continue
        k=re.sub(r"^void |ctx::|\(anonymous namespace\)::", "", k).split("(")[0][:70]+" grid="+r.get("Grid_Size","?")
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):      # effective shader clock of the launch (MI355X_MICROARCH.md, DVFS give-back)
            dt=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
            if dt>0: clk[k].append((float(r["Counter_Value"])/dt, dt*1e-3))
with open(O+"/summary.txt","w") as out:
    for k,v in sorted(agg.items()):
        out.write(k+"\n")
        for c,x in sorted(v.items()):
            out.write(f"    {c:28s} {x/cnt[(k,c)]:16.0f}  per launch (n={cnt[(k,c)]})\n")
        if v.get("SQ_BUSY_CYCLES") and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            out.write(f"    mfma busy / (32 * busy)      {v['SQ_VALU_MFMA_BUSY_CYCLES']/cnt[(k,'SQ_VALU_MFMA_BUSY_CYCLES')]/(32*v['SQ_BUSY_CYCLES']/cnt[(k,'SQ_BUSY_CYCLES')]):.3f}\n")
        if clk[k]:
            out.write(f"    effective clock GRBM_GUI_ACTIVE / duration: {sum(c for c,_ in clk[k])/len(clk[k]):.3f} GHz over {sum(d for _,d in clk[k])/len(clk[k]):.1f} us launches (profiled pass: serialised, counters on)\n")
        a=lambda c: v[c]/cnt[(k,c)] if cnt[(k,c)] else 0.0
        if a("SQ_WAVE_CYCLES"):
            out.write(f"    of the wave cycles: waiting on any instruction {a('SQ_WAIT_INST_ANY')/a('SQ_WAVE_CYCLES'):.2f}, on LDS {a('SQ_WAIT_INST_LDS')/a('SQ_WAVE_CYCLES'):.2f}, issuing {a('SQ_ACTIVE_INST_ANY')/a('SQ_WAVE_CYCLES'):.2f}\n")
        if a("SQ_INSTS_MFMA"):
            out.write(f"    per MFMA: VALU {a('SQ_INSTS_VALU')/a('SQ_INSTS_MFMA')-1:.2f} (besides itself), SALU {a('SQ_INSTS_SALU')/a('SQ_INSTS_MFMA'):.2f}, LDS {a('SQ_INSTS_LDS')/a('SQ_INSTS_MFMA'):.2f}, VMEM {a('SQ_INSTS_VMEM')/a('SQ_INSTS_MFMA'):.2f};  LDS bank-conflict share {a('SQ_LDS_BANK_CONFLICT')/max(a('SQ_LDS_IDX_ACTIVE'),1):.2f}\n")
print(open(O+"/summary.txt").read()[:8000])
PY
