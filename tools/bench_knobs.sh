#!/bin/bash
# dev: sweep schedule knobs at several q; usage: bench_knobs.sh "ENV1=a ENV2=b" q...
mkdir -p gpurun_out/r2
ENVS="$1"; shift
for q in "$@"; do
  env $ENVS timeout -k 10 200 python bench.py --steps 12 --warmup 4 --latents $q --no-cpu-baseline > gpurun_out/r2/bk.json 2> gpurun_out/r2/bk.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2/bk.json"))
print("[$ENVS] q=$q", round(d["ms_per_step"],3), "sweep", round(d["cholesky_gemm"]["ms_per_step"],3), {k:(round(v["ms_per_step"],2), int(v["launches_per_step"]), v["tflops"] and round(v["tflops"],1)) for k,v in d["kernels"].items() if v["ms_per_step"]>0.3})
PY
done
