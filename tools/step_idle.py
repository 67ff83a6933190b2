"""Dev aid: GPU idle time inside one training step (union of ALL kernel intervals vs the step span), and the idle gaps by
position.  usage: step_idle.py trace.csv [min_gap_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda r: r["Kernel_Name"].replace("void ", "").replace("plmc::", "").split("(")[0][:34]
asm = [i for i, r in enumerate(rows) if ("k_assemble<" in r["Kernel_Name"] or "k_assemble_small<" in r["Kernel_Name"])]
a, b = asm[-2], asm[-1]
step = rows[a:b]
t0, tend = int(step[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
cur_e, idle, gaps = int(step[0]["End_Timestamp"]), 0, []
last = step[0]
for r in step[1:] + [rows[b]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_e:
        idle += s - cur_e; gaps.append((cur_e, s, short(last), short(r)))
    if e > cur_e:
        cur_e, last = e, r
print("step span %.2f ms, GPU idle %.2f ms in %d gaps" % ((tend - t0) / 1e6, idle / 1e6, len(gaps)))
hist = {}
for s, e, pa, nx in gaps:
    k = (pa, nx); hist[k] = (hist.get(k, (0, 0))[0] + 1, hist.get(k, (0, 0))[1] + (e - s) / 1e3)
for k, v in sorted(hist.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%8.1f us in %3d gaps  after %-34s before %s" % (v[1], v[0], k[0], k[1]))
for s, e, pa, nx in gaps:
    if (e - s) / 1e3 >= ming:
        print("gap %7.1f us at %8.1f us  after %-34s before %s" % ((e - s) / 1e3, (s - t0) / 1e3, pa, nx))
