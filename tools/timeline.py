"""Timeline of ONE training step with the stream lanes on, from a rocprofv3 --kernel-trace csv:
   python tools/timeline.py <dir with *_kernel_trace.csv> [step index from the end, default 2]
Steps are cut at the Adam kernels.  Prints, for the chosen step, every kernel with its start / end offset, the number of other
kernels running at its start, and at the end the time during which 0 / 1 / 2 / 3+ kernels were in flight (idle gaps = launch
latency the chip waits on)."""
import csv, glob, re, sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], str(r.get("Stream_Id", "?")) + "/" + str(r.get("Queue_Id", "?"))))
rows.sort()
import os
cut = os.environ.get("TIMELINE_CUT", "")          # e.g. loss_final_kernel: one per step (then a "step" runs from the backward of one
if cut:                                            # step through the forward of the next -- needed when Adam runs in slices)
    ends = [i for i, r in enumerate(rows) if cut in r[2]]
else:
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    # a step = (after the previous step's last adam launch, through this step's last adam launch)
    ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 20]
e1 = ends[-back]; e0 = ends[-back - 1]
step = rows[e0 + 1:e1 + 1]
t0 = step[0][0]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ctx::", "", n)
    return n.split("(")[0][:70]
ev = []
for s, e, n, q in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
print(f"step of {len(step)} kernels, {(step[-1][1] - t0) / 1e6:.3f} ms")
for s, e, n, q in step:
    conc = sum(1 for s2, e2, _, _ in step if s2 < s < e2)
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  stream/queue={q:>5} +{conc}  {short(n)}")
hist = {}
cur = 0; last = ev[0][0]
for t, dlt in ev:
    hist[min(cur, 3)] = hist.get(min(cur, 3), 0) + (t - last)
    cur += dlt; last = t
print({k: round(v / 1e6, 3) for k, v in sorted(hist.items())}, "ms with k kernels in flight")
