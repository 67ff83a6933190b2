"""Dev aid: one training step from a rocprofv3 kernel-trace CSV -- where the heavy kernels (tail / head update, K^-1+gradient,
assembly) do NOT cover the time line, and what ran there.  usage: step_gaps.py trace.csv [min_gap_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda r: r["Kernel_Name"].replace("void plmc::", "").split("(")[0][:40]
asm = [i for i, r in enumerate(rows) if ("k_assemble<" in r["Kernel_Name"] or "k_assemble_small<" in r["Kernel_Name"])]
a, b = asm[-2], asm[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
print("step span %.2f ms, %d kernels" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e6, len(step)))
heavy = lambda r: any(k in r["Kernel_Name"] for k in ("k_update<float, 0", "k_update<float, 2", "k_kinv_grad", "k_assemble"))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step if heavy(r))
cover, gaps, cur_s, cur_e = 0, [], iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        cover += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cover += cur_e - cur_s
gaps.append((cur_e, int(rows[b]["Start_Timestamp"])))
print("heavy kernels cover %.2f ms; gaps total %.2f ms" % (cover / 1e6, sum(e - s for s, e in gaps) / 1e6))
for s, e in gaps:
    if (e - s) / 1e3 < ming: continue
    inside = [r for r in step if int(r["End_Timestamp"]) > s and int(r["Start_Timestamp"]) < e and not heavy(r)]
    names = {}
    for r in inside:
        names[short(r)] = names.get(short(r), 0) + 1
    print("gap at %8.1f us, %6.1f us: %s" % ((s - t0) / 1e3, (e - s) / 1e3, ", ".join("%s x%d" % kv for kv in names.items())))
# per-kernel-class totals
tot = {}
for r in step:
    k = short(r); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[k] = (tot.get(k, (0, 0))[0] + 1, tot.get(k, (0, 0))[1] + d)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%9.1f us %4d  %s" % (v[1], v[0], k))
