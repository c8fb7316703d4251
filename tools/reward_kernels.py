"""The kernels of ONE reward-hook call at the reference's call shape (25 frames, rllab/sampler/base.py:216-218, 234-235), in order, with
start offset and duration, from a rocprofv3 kernel trace.
    driver:   python tools/reward_kernels.py run [encode|translate]        (what rocprofv3 wraps: 30 calls of one kind, handle max_batch = 25)
    report:   python tools/reward_kernels.py report <trace dir> <out.txt>"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import time
    import numpy as np
    from imitation_from_observation_amd import Translator
    kind = sys.argv[2]
    tr = Translator(64, 64, 64, 1024, max_batch=25)
    tr.init_params(0)
    x = np.random.default_rng(0).integers(0, 256, (25, 64, 64, 3), dtype=np.uint8)
    fn = (lambda: tr.encode(x)) if kind == "encode" else (lambda: tr.translate(x, x[0]))
    ts = []
    for i in range(30):
        time.sleep(0.002)                                  # a gap the report can cut at
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(kind, "median ms under the tracer", round(float(np.median(ts)), 3))
    tr.close()
else:
    f = sorted(glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
    cut = 0
    for i in range(1, len(rows)):
        if st[i] - max(en[:i][-8:]) > 1000000: cut = i       # the last gap of > 1 ms
    last = rows[cut:]
    t0 = st[cut]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
    with open(sys.argv[3], "w") as o:
        o.write(f"kernels {len(last)}  span {(max(en[cut:]) - t0) / 1e3:.1f} us  sum of durations {busy / 1e3:.1f} us\n")
        for r in last:
            o.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>5s}  {r['Kernel_Name'][:110]}\n")
    print(open(sys.argv[3]).read())
