"""Dev aid: one whole training step (from one k_write_rhs to the next) out of a rocprofv3 kernel-trace CSV: what runs outside the
library's kernels (torch: projection, loss terms, optimiser), where, and how long the GPU idles between kernels.
    python tools/step_timeline.py <kernel_trace.csv> [verbose]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
asm = [i for i, r in enumerate(rows) if "k_write_rhs" in r["Kernel_Name"]]      # one per training step, in front of the factorisation
a, b = asm[-2], asm[-1]
win = rows[a:b]
t0 = int(win[0]["Start_Timestamp"])
T = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("plmc::", "")[:60]
is_lib = lambda r: "plmc" in r["Kernel_Name"]
lib = [r for r in win if is_lib(r)]
oth = [r for r in win if not is_lib(r)]
lib_end = max(int(r["End_Timestamp"]) for r in lib)
sweep_end = max(int(r["End_Timestamp"]) for r in lib if "k_logdet" in r["Kernel_Name"] or "k_update" in r["Kernel_Name"] or "k_gpanel" in r["Kernel_Name"])
print("step period %.1f us: %d library kernels, %d other kernels" % (T, len(lib), len(oth)))
print("  library kernels span 0 .. %.1f us (sweep ends %.1f)" % ((lib_end - t0) / 1e3, (sweep_end - t0) / 1e3))
# union busy / idle over the period
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
busy, cs, ce, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; gaps.append(((cs if False else ce) - t0, s - ce)); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("  GPU busy %.1f us, idle %.1f us in %d gaps" % (busy / 1e3, T - busy / 1e3, len(gaps) + 1))
big = sorted(gaps, key=lambda g: -g[1])[:8]
print("  largest gaps (at us: length us):", ", ".join("%.0f: %.0f" % (g[0] / 1e3, g[1] / 1e3) for g in sorted(big)))
after = [r for r in oth if int(r["Start_Timestamp"]) >= lib_end]
during = [r for r in oth if int(r["Start_Timestamp"]) < lib_end]
print("  other kernels while library kernels run: %d (%.1f us of kernel time); after the last library kernel: %d (%.1f us of kernel time, wall %.1f us)" % (
    len(during), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in during) / 1e3, len(after),
    sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in after) / 1e3, T - (lib_end - t0) / 1e3))
if len(sys.argv) > 2:
    for r in win:
        if not is_lib(r) or any(k in r["Kernel_Name"] for k in ("k_assemble", "k_logdet", "k_kinv", "k_wt_matvec", "k_extract", "k_reduce", "k_qr", "k_write_rhs")):
            print("%9.1f %7.1f q%-2s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), nm(r)))
