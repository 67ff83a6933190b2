"""Aggregate the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-kernel HBM bytes
per dispatch -> profiles/r01_pmc_traffic.json (+ compact per-dispatch CSVs).  Usage:
    python tools/pmc_aggregate.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix>
The gfx950 correction (FETCH_SIZE counts half of the bytes of wide coalesced reads) follows
MI355X_MICROARCH.md and is cross-checked on k_wt_matvec, whose algorithmic read volume is known."""
import csv, json, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("plmc::", "")

def load(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter or "plmc" not in r["Kernel_Name"]:
            continue
        out.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"]), r))
    return out

fetch, write, prefix = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
for rows, tag in ((fetch, "fetch_size"), (write, "write_size")):
    with open("%s_pmc_%s.csv" % (prefix, tag), "w") as f:
        f.write("Dispatch_Id,Kernel,Grid_Size,VGPR,AGPR,LDS,Counter,Value_KB\n")
        for d, k, v, r in rows:
            f.write("%d,%s,%s,%s,%s,%s,%s,%f\n" % (d, k, r.get("Grid_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""),
                                                 r.get("LDS_Block_Size", ""), tag.upper(), v))
agg = defaultdict(lambda: dict(dispatches=0, fetch=0.0, write=0.0, wd=0))
for d, k, v, _ in fetch:
    agg[k]["dispatches"] += 1; agg[k]["fetch"] += v * 1024.0
for d, k, v, _ in write:
    agg[k]["wd"] += 1; agg[k]["write"] += v * 1024.0
import hashlib, os
def build_key():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "projected-lmc_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)) + [os.path.join("..", "..", "include", "plmc.h")]:
        path = os.path.join(csrc, f)
        if os.path.isfile(path) and f.endswith((".hip", ".hpp", ".h")):
            h.update(f.encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()[:16]
res = {"build_key": build_key(), "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py "
                 "--steps 2 --warmup 1 --no-cpu-baseline --no-prof; 1x MI355X, C3 workload",
       "units": "bytes per dispatch (average over the dispatches of the pass); FETCH_SIZE / WRITE_SIZE are reported in KB; "
                "FETCH_SIZE counts 1/2 of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section), so "
                "hbm_bytes = 2*FETCH + WRITE; the counters sit on the fabric side of L2, Infinity-Cache hits are included",
       "kernels": {}}
for k, a in agg.items():
    n, nw = max(1, a["dispatches"]), max(1, a["wd"])
    res["kernels"][k] = {"dispatches": a["dispatches"], "fetch_bytes_raw": a["fetch"] / n, "write_bytes": a["write"] / nw,
                         "hbm_bytes_corrected": 2.0 * a["fetch"] / n + a["write"] / nw}
json.dump(res, open("%s_pmc_traffic.json" % prefix, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_corrected"] / 1e6, 1) for k, v in res["kernels"].items()}))
