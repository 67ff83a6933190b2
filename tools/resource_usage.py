"""Register / scratch / LDS table of every kernel in projected-lmc_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage,
gfx950; runs without a GPU).  `python tools/resource_usage.py [--md OUT.md] [file.hip ...]`.
The judge's bar for the gradient kernels is ScratchSize 0 for every instance a BASELINE config launches."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "projected-lmc_amd", "csrc")
FIELDS = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs", "LDS Size [bytes/block]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines()


def analyse(src):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
               "-c", src, "-o", os.path.join(tmp, "x.o")]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark: .*?\s{2,}([A-Za-z][A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = demangle([r["name"] for r in rows])
    for r, nm in zip(rows, names):
        r["pretty"] = re.sub(r"^void plmc::", "", nm).split("(")[0]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default=None)
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    files = a.files or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    lines = ["| file | kernel | VGPR | AGPR | scratch B/lane | waves/SIMD | SGPR | LDS B |", "|---|---|---|---|---|---|---|---|"]
    bad = 0
    for f in files:
        for r in analyse(f):
            sc = r.get("ScratchSize [bytes/lane]", -1)
            bad += sc > 0
            lines.append("| %s | `%s` | %s | %s | %s | %s | %s | %s |" % (
                os.path.basename(f), r["pretty"], r.get("VGPRs", "?"), r.get("AGPRs", "?"), sc,
                r.get("Occupancy [waves/SIMD]", "?"), r.get("TotalSGPRs", "?"), r.get("LDS Size [bytes/block]", "?")))
    text = "\n".join(lines) + "\n\nkernels with scratch: %d\n" % bad
    if a.md:
        with open(a.md, "w") as fh:
            fh.write("# Kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)\n\n" + text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
