"""Dev aid: per-group view of one sweep from a rocprofv3 kernel-trace CSV (heads/tails/chain spans, idle gaps)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "plmc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_zero_diag_out" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["Start_Timestamp"])
nm = lambda r: r["Kernel_Name"].split("<")[0].replace("void plmc::", "")
ev = [(nm(r), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // 256 if "Grid_Size_X" in r else 0) for r in rows]
big = [e for e in ev if e[0] == "k_update" and e[3] > 2000]
print("big updates:", len(big))
for e in big: print("  %-8s start %8.1f end %8.1f dur %7.1f wgs %6d" % ("upd", e[1], e[2], e[2] - e[1], e[3]))
# union busy of big updates
iv = sorted((e[1], e[2]) for e in big)
busy = 0.0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
sw_end = max(e[2] for e in ev if e[0] in ("k_update", "k_panel", "k_diag"))
sw_start = min(e[1] for e in ev if e[0] == "k_diag")
print("sweep %.1f us, big-update busy union %.1f us, not covered %.1f us" % (sw_end - sw_start, busy, sw_end - sw_start - busy))
