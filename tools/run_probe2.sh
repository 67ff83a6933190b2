#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
for v in x7d1 x7d2 x7d3 x7d5 x7; do
  lib="$PWD/tools/variants/libplmc_$v.so"
  PLMC_LIB=$lib timeout -k 10 300 python tools/wb_race_probe.py --tag $v --reps 4 > gpurun_out/r2/p2_$v.json 2> gpurun_out/r2/p2_$v.err || echo "probe $v failed"
  PLMC_LIB=$lib timeout -k 10 300 python tools/wb_race_probe.py --tag ${v}_f64 --n 4096 --q 4 --dtype f64 --reps 4 > gpurun_out/r2/p2_${v}_f64.json 2>> gpurun_out/r2/p2_$v.err || echo "probe $v f64 failed"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/p2_*.json')):
    try:
        d=json.load(open(f))
        print(d['tag'], 'serial_diff', d.get('serial_tiles_differ'), [ (r['live_tiles_differ'], r['logdet_equal']) for r in d['runs']])
    except Exception as e: print(f, 'ERR', e)
PY
