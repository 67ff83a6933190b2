"""Dev aid: print the kernel timeline (start offset, duration, gap to previous end) from a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = lambda r: r["Kernel_Name"].split("<")[0].replace("void plmc::", "")
rows = [r for r in rows if "plmc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last sweep = from the last k_zero_diag_out (its first kernel)
last = max(i for i, r in enumerate(rows) if "k_zero_diag_out" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["Start_Timestamp"])
lo, hi = int(sys.argv[2]), int(sys.argv[3])
prev_end = t0
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= i < hi:
        print("%4d %-16s q%-3s start %9.1f dur %7.1f gap_prev_end %7.1f grid %s" % (i, name(r)[:16], r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", "")))
    prev_end = max(prev_end, e)
print("total %.1f us over %d kernels" % ((prev_end - t0) / 1e3, len(rows)))
