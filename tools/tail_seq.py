"""Dev aid: ordered kernel list between the end of k_kinv_grad and the next k_assemble (one line per run of equal names)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
kg = [i for i, r in enumerate(rows) if "k_kinv_grad" in r["Kernel_Name"]]
asm = [i for i, r in enumerate(rows) if ("k_assemble<" in r["Kernel_Name"] or "k_assemble_small<" in r["Kernel_Name"])]
e = kg[-2]; s = min(i for i in asm if i > e)
t0 = int(rows[e]["Start_Timestamp"]); t1 = int(rows[e]["End_Timestamp"])
print("kinv_grad %.1f us; next assemble starts %.1f us after its end" % ((t1 - t0) / 1e3, (int(rows[s]["Start_Timestamp"]) - t1) / 1e3))
# kernels that START after kinv_grad started, up to the assemble
prev = None; cnt = 0; first = 0
for r in rows[e + 1:s] + [None]:
    nm = None if r is None else r["Kernel_Name"].replace("void ", "").split("(")[0][:70]
    if nm != prev:
        if prev is not None:
            print("%9.1f  x%-3d %s" % ((first - t1) / 1e3, cnt, prev))
        prev, cnt, first = nm, 0, (0 if r is None else int(r["Start_Timestamp"]))
    cnt += 1
