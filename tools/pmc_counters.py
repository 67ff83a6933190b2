"""Average per kernel of any rocprofv3 --pmc counter collection (one or more counters per pass):
    python tools/pmc_counters.py <counter_collection.csv> [more.csv ...] > out.json
Per library kernel: dispatches, and per counter the mean over its dispatches.  With TCC_HIT_sum and TCC_MISS_sum present the L2
hit rate hits / (hits + misses) is added (MI355X_MICROARCH.md, L2 section)."""
import csv, json, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("plmc::", "")

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "plmc" not in r["Kernel_Name"]:
            continue
        a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {}
for k, cs in acc.items():
    o = {"dispatches": max(v[0] for v in cs.values())}
    for c, (n, s) in cs.items():
        o[c] = s / n
    if "TCC_HIT_sum" in o and "TCC_MISS_sum" in o and o["TCC_HIT_sum"] + o["TCC_MISS_sum"] > 0:
        o["l2_hit_rate"] = o["TCC_HIT_sum"] / (o["TCC_HIT_sum"] + o["TCC_MISS_sum"])
    out[k] = o
import hashlib, os
def build_key():                                    # as bench.py / tools/pmc_aggregate.py: hash of the library sources
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "projected-lmc_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)) + [os.path.join("..", "..", "include", "plmc.h")]:
        path = os.path.join(csrc, f)
        if os.path.isfile(path) and f.endswith((".hip", ".hpp", ".h")):
            h.update(f.encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()[:16]
print(json.dumps({"build_key": build_key(), "source": "rocprofv3 --kernel-trace --pmc <counters> (one pass per counter group) -- python3 bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-prof --no-options; counter runs serialise the kernels (isolated figures); means over the dispatches",
                  "kernels": out}, indent=1))
