"""Aggregates rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace alongside) into HBM bytes per
launch per kernel GROUP, with the group names bench.py's event table uses.   python tools/hbm_aggregate.py <dir> <prec> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def group(k):
    m = re.match(r"void ctx::igemm(_split)?_kernel<ctx::(\w+), ctx::(\w+),", k)
    if not m:
        if "adam_kernel" in k:
            return "adam"
        d = re.match(r"void ctx::(?:\(anonymous namespace\)::)?(dconv_fwd_kernel|dconv_wgrad_kernel|convt3_kernel|wconvt_kernel|wconvt_row_kernel|c3conv_kernel|c3wgrad_kernel)<", k)     # the labels ctx_profile_step uses for the direct kernels
        if d and d.group(1) == "wconvt_row_kernel":          # the row-block instances of the same layer family: one label (ctx_profile_step's)
            return "wconvt_kernel"
        return d.group(1) if d else k
    a, b = m.group(2), m.group(3)
    if a.startswith("KmConvTGather"):
        return "igemm<ConvTGather,ConvTWeights>"
    if a.startswith("KmConvGather"):
        return "igemm<ConvGather,Plain>" if not b.startswith("KmConvTWeights") else "igemm<ConvGather,ConvTWeights>"
    if a.startswith("NmWgradBig"):
        return "igemm<WgradBig,WgradSmall>"
    if a.startswith("NmC3WgradBig"):
        return "igemm<C3WgradBig,WgradSmall>"
    if a.startswith("KmC3Gather"):
        return "igemm<C3Gather,C3Weights>"
    if a == "KmCat2":
        return "igemm<Cat2,KmPlain>"
    b = "NmPlain" if b == "NmPlain2" else b
    a = "NmPlain" if a == "NmPlain2" else a
    return f"igemm<{a},{b}>"


def main(d, prec, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{d}/{c}/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != c:
                    continue
                k = group(r["Kernel_Name"])
                acc[k][c] += float(r["Counter_Value"])
                if c == "FETCH_SIZE":
                    n[k] += 1
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res = {"csrc_sha16": bench.csrc_sha16(), "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --steps 1 "
                     "--warmup 1 --kernel-iters 1, CTX_OVERLAP=0, B=256, precision " + prec + "; bytes = FETCH_SIZE*1024*2 "
                     "(gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE*1024", "per_kernel": {}}
    for k, v in acc.items():
        if n[k]:
            fb, wb = v["FETCH_SIZE"] * 2048 / n[k], v["WRITE_SIZE"] * 1024 / n[k]
            res["per_kernel"][k] = {"launches_profiled": n[k], "hbm_bytes_per_launch": fb + wb, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb}
    json.dump(res, open(out, "w"), indent=1)
    for k in sorted(res["per_kernel"], key=lambda k: -res["per_kernel"][k]["hbm_bytes_per_launch"] * res["per_kernel"][k]["launches_profiled"])[:8]:
        print(k, res["per_kernel"][k])


if __name__ == "__main__":
    main(*sys.argv[1:4])
