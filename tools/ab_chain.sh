#!/bin/bash
# dev: the resident chain kernel (PLMC_CHAIN=1, default) against the launch-per-step chain (PLMC_CHAIN=0), and helper counts, at
# q = 8, 4, 2, 1 local latents (per-rank shards): ms/step and sweep ms from bench.py
mkdir -p gpurun_out/ab
run() {  # tag, env...
  tag=$1; shift
  for q in ${QS:-8 1}; do
    env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --latents $q --no-cpu-baseline --no-options > gpurun_out/ab/${tag}_q$q.json 2> gpurun_out/ab/${tag}_q$q.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab/${tag}_q$q.json"))
    print("$tag q=$q", round(d["ms_per_step"],3), "ms/step; sweep", round(d["cholesky_gemm"]["ms_per_step"],3), {k:(round(v["ms_per_step"],2), int(v["launches_per_step"])) for k,v in d["kernels"].items() if k in ("k_diag","k_panel","k_trail_row","k_trail","k_trail_head","k_gpanel","k_kinv_grad")}, flush=True)
except Exception as e:
    print("$tag q=$q FAILED", e, open("gpurun_out/ab/${tag}_q$q.err").read()[-600:], flush=True)
PY
  done
}
"$@"
