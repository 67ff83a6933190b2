"""Dev aid: per-queue busy time, union busy time and idle gaps of the last sweep in a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "plmc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_zero_diag_out" in r["Kernel_Name"])      # first kernel of a sweep
rows = rows[last:]
end = next((i for i, r in enumerate(rows) if "k_extract_col" in r["Kernel_Name"]), len(rows))   # first kernel behind it (k_logdet rides on the chain stream)
rows = rows[:end]
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void plmc::", "")
byq, byk = {}, {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    byq.setdefault(r.get("Queue_Id", "?"), []).append((s, e))
    k = byk.setdefault(name(r), [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e) / 1e3); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("sweep wall %.1f us, union busy %.1f us, idle %.1f us in %d gaps (max %.1f)" % ((t1 - t0) / 1e3, busy / 1e3, sum(gaps), len(gaps), max(gaps or [0])))
for q, l in byq.items():
    print("queue", q, "busy %.1f us over %d kernels" % (sum(e - s for s, e in l) / 1e3, len(l)))
for k, (n, t) in sorted(byk.items(), key=lambda kv: -kv[1][1]):
    print("  %-40s %4d launches %9.1f us  avg %7.1f" % (k[:40], n, t, t / n))
