"""Dev aid: per-group phases of the LAST sweep in a rocprofv3 kernel-trace CSV (look-ahead schedule of potrf_impl):
chain (first k_diag .. k_vtrans), head panel, U1 on the chain stream; rest panel, head update on the helper stream; tail on
the caller's stream.  Times in us from the start of the sweep: start-end (duration)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "plmc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_zero_diag_out" in r["Kernel_Name"])      # first kernel of a sweep
rows = rows[last:]
end = next((i for i, r in enumerate(rows) if "k_extract_col" in r["Kernel_Name"]), len(rows))   # first kernel behind it (k_logdet rides on the chain stream)
rows = rows[:end]
t0 = int(rows[0]["Start_Timestamp"])
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void plmc::", "")
ev = [(nm(r), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?")) for r in rows]
groups = []
cur = None
for name, s, e, q in ev:
    if name.startswith("k_diag") or name.startswith("k_chain"):      # (k_chain: the resident chain, one launch per group)
        if cur is None or "vtrans" in cur:
            cur = {"chain": [s, e], "ndiag": 0}
            groups.append(cur)
        cur["chain"][1] = e
        cur["ndiag"] += 1
    elif cur is None:
        continue
    elif name.startswith("k_panel") or name.startswith("k_update<float, 1") or name.startswith("k_update<float, 4") or name.startswith("k_update<double, 1"):
        if "vtrans" not in cur:
            cur["chain"][1] = e
    elif name.startswith("k_vtrans"):
        cur["vtrans"] = [s, e]
def span(d, key, s, e):
    if key in d:
        d[key][0] = min(d[key][0], s); d[key][1] = max(d[key][1], e)
    else:
        d[key] = [s, e]
# bulk kernels: attribute to groups in order of appearance per class
chainq = next((q for name, s, e, q in ev if name.startswith("k_diag") or name.startswith("k_chain")), None)       # the chain's queue: the head panel runs there
cls_of = lambda n, q=None: ("gp_head" if n.startswith("k_gpanel_rows<float, 1") or n.startswith("k_gpanel_rows<double, 1") or (n.startswith("k_gpanel_bf3") and q == chainq) else
                    "gp_rest" if n.startswith("k_gpanel_rows") or n.startswith("k_gpanel_bf3") else
                    "U1" if n.startswith("k_update<float, 2") or n.startswith("k_update<double, 2") or (n.startswith("k_update_bf3<") and n.endswith(" 2>")) else
                    "head" if (n.startswith("k_update_bf3<") and n.endswith(" 3>")) or n.startswith("k_update<float, 3") or n.startswith("k_update<double, 3") else
                    "tail" if (n.startswith("k_update_bf3<") and n.endswith(" 0>")) or n.startswith("k_update<float, 0") or n.startswith("k_update<double, 0") else None)
cnt = {}
split_on = any("_bf3<" in name for name, *_ in ev)      # split engine: its launches define the phases; the fp32 launches of the
for name, s, e, q in ev:                                 # same class only carry the augmented columns
    c = cls_of(name, q)
    if split_on and c in ("gp_head", "gp_rest", "U1", "head", "tail") and "_bf3<" not in name:
        continue
    if c is None:
        continue
    i = cnt.get(c, 0); cnt[c] = i + 1
    if i < len(groups):
        groups[i][c] = [s, e]
f = lambda d, k: ("%7.0f-%-7.0f(%5.0f)" % (d[k][0], d[k][1], d[k][1] - d[k][0])) if k in d else " " * 22
print("sweep %.0f us" % (ev[-1][2]))
print("grp  %-22s %-22s %-22s %-22s | %-22s %-22s | %-22s" % ("chain", "vtrans", "gp_head", "U1", "gp_rest", "head", "tail"))
for i, g in enumerate(groups):
    print("%3d  %s %s %s %s | %s %s | %s" % (i, f(g, "chain"), f(g, "vtrans"), f(g, "gp_head"), f(g, "U1"), f(g, "gp_rest"), f(g, "head"), f(g, "tail")))
