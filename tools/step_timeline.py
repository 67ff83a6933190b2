"""Dev aid: every kernel of one whole training step (assemble of step k to assemble of step k+1) from a rocprofv3
kernel-trace CSV, in start order: offset, duration, queue, name.  Runs of sweep kernels are folded into one line."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
asm = [i for i, r in enumerate(rows) if "k_assemble" in r["Kernel_Name"]]
a, b = asm[-2], asm[-1]
t0 = int(rows[a]["Start_Timestamp"])
sweep = ("k_diag", "k_panel", "k_update", "k_gpanel", "k_vtrans", "k_zero_diag")
fold = None
def flush():
    global fold
    if fold:
        print("%9.1f %9.1f  q=%-3s [%d sweep kernels, last ends %.1f]" % (fold[0], fold[1] - fold[0], "*", fold[2], fold[1]))
    fold = None
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("plmc::", "")
    if any(k in nm for k in sweep):
        fold = [s, e, 1] if not fold else [fold[0], max(fold[1], e), fold[2] + 1]
        continue
    flush()
    print("%9.1f %9.1f  q=%-3s %s" % (s, e - s, r.get("Queue_Id", "?"), nm[:90]))
flush()
