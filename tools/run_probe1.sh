#!/bin/bash
# round-2 experiment 1: baseline tests + write-back race probes (dev)
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2/tests0.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2/tests0.log
for v in base x7 x1 x6; do
  lib=""; [ "$v" != base ] && lib="$PWD/tools/variants/libplmc_$v.so"
  PLMC_LIB=$lib timeout -k 10 300 python tools/wb_race_probe.py --tag $v --reps 8 > gpurun_out/r2/probe_$v.json 2> gpurun_out/r2/probe_$v.err || echo "probe $v failed"
  PLMC_LIB=$lib timeout -k 10 300 python tools/wb_race_probe.py --tag ${v}_q2 --q 2 --reps 6 > gpurun_out/r2/probe_${v}_q2.json 2>> gpurun_out/r2/probe_$v.err || echo "probe $v q2 failed"
  PLMC_LIB=$lib timeout -k 10 300 python tools/wb_race_probe.py --tag ${v}_f64 --n 4096 --q 4 --dtype f64 --reps 6 > gpurun_out/r2/probe_${v}_f64.json 2>> gpurun_out/r2/probe_$v.err || echo "probe $v f64 failed"
done
tail -3 gpurun_out/r2/tests0.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/probe_*.json')):
    try:
        d=json.load(open(f))
        print(d['tag'], 'serial_rep', d['serial_repeatable'], [ (r['live_tiles_differ'], r['logdet_equal']) for r in d['runs']])
    except Exception as e: print(f, 'ERR', e)
PY
