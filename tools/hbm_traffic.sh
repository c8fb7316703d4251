# HBM bytes per launch per kernel from PMC counters (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes, kernel-trace only alongside; bytes = FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024.
#   gpurun -- 'bash tools/hbm_traffic.sh e [f32|bf16x3]'   ->  gpurun_out/<tag>/hbm_traffic[_bf16x3].json
set -x
TAG=${1:-x}; PREC=${2:-f32}
R=$PWD; O=$R/gpurun_out/$TAG/pmc_hbm_$PREC; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  CTX_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg --kernel-iters 1 --precision $PREC > $O/$C.log 2>&1
done
cd $R
python - <<PY
import csv, glob, json, collections, re
O="$O"; prec="$PREC"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(f"{O}/{C}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != C: continue
            k=r["Kernel_Name"]
            m=re.match(r"void ctx::igemm(_split)?_kernel<ctx::(\w+), ctx::(\w+),", k)
            if m:
                a,b=m.group(2),m.group(3)
                a=a.replace("Km","").replace("Nm","") if a not in ("KmPlain","NmPlain") else a
                b=b.replace("Km","").replace("Nm","") if b not in ("KmPlain","NmPlain") else b
                a=a.rstrip("P") if a.startswith("Wgrad") else a
                b=re.sub(r"2?P?$","",b) if b.startswith("Wgrad") else b
                b="Plain" if (b=="NmPlain" and a=="ConvGather") else b
                k=f"igemm<{a},{b}>"
            elif "adam_kernel" in k: k="adam"
            acc[k][C]+=float(r["Counter_Value"]); 
            if C=="FETCH_SIZE": n[k]+=1
out={"source":"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --steps 1 --warmup 1 --kernel-iters 1, CTX_OVERLAP=0, B=256, precision "+prec+"; bytes = FETCH_SIZE*1024*2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE*1024",
     "per_kernel":{}}
for k,v in acc.items():
    if not n[k]: continue
    fb=v["FETCH_SIZE"]*1024*2/n[k]; wb=v["WRITE_SIZE"]*1024/n[k]
    out["per_kernel"][k]={"launches_profiled":n[k],"hbm_bytes_per_launch":fb+wb,"fetch_bytes_per_launch":fb,"write_bytes_per_launch":wb}
suffix = "" if prec=="f32" else "_"+prec
json.dump(out, open(f"$R/gpurun_out/$TAG/hbm_traffic{suffix}.json","w"), indent=1)
for k in sorted(out["per_kernel"], key=lambda k:-out["per_kernel"][k]["hbm_bytes_per_launch"]*out["per_kernel"][k]["launches_profiled"])[:12]:
    print(k, out["per_kernel"][k])
PY
