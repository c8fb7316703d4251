"""Busy / idle split of the GPU over the timed steps of a command, from a rocprofv3 kernel trace CSV: union of the kernels'
[start, end) intervals against the span from the first to the last kernel of the window.
   python tools/gpu_idle.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = iv[int(len(iv) * skip):]                      # the later part: steady-state steps
span = iv[-1][1] - iv[0][0]
busy = 0; cur_s, cur_e = iv[0][0], iv[0][1]; gaps = []
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps.sort(reverse=True)
print(f"kernels {len(iv)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {100*(span-busy)/span:.1f} %  gaps {len(gaps)}  median gap {gaps[len(gaps)//2]/1e3 if gaps else 0:.1f} us  sum of kernel durations {sum(e-s for s,e,_ in iv)/1e6:.3f} ms")
