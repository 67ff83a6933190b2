"""Dev aid: what runs between the end of k_kinv_grad and the next k_assemble (the torch tail of a step)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_reduce_grad" in r["Kernel_Name"]]
starts = [i for i, r in enumerate(rows) if ("k_assemble<" in r["Kernel_Name"] or "k_assemble_small<" in r["Kernel_Name"]) or "k_assemble(" in r["Kernel_Name"]]
e = ends[-2]
s = min(i for i in starts if i > e)
t0 = int(rows[e]["End_Timestamp"])
print("tail span %.1f us, %d kernels" % ((int(rows[s]["Start_Timestamp"]) - t0) / 1e3, s - e - 1))
busy = 0.0
from collections import defaultdict
d = defaultdict(lambda: [0, 0.0])
for r in rows[e + 1:s]:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += dur
    k = r["Kernel_Name"].split("(")[0][-60:]
    d[k][0] += 1; d[k][1] += dur
print("kernel busy %.1f us" % busy)
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:18]:
    print("%7.1f us %3d  %s" % (v[1], v[0], k))
