#!/bin/bash
# dev: bench at q = 8, 4, 2, 1 (per-rank shards of the 1/2/4/8-GPU runs), one line each
mkdir -p gpurun_out/r2
TAG=${1:-x}
for q in 8 4 2 1; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --latents $q --no-cpu-baseline > gpurun_out/r2/bench_${TAG}_q$q.json 2> gpurun_out/r2/bench_${TAG}_q$q.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2/bench_${TAG}_q$q.json"))
print($q, round(d["ms_per_step"],3), "sweep", round(d["cholesky_gemm"]["ms_per_step"],3), {k:(round(v["ms_per_step"],2), int(v["launches_per_step"]), v["tflops"] and round(v["tflops"],1)) for k,v in d["kernels"].items() if v["ms_per_step"]>0.15})
PY
done
