#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
for v in x6 x7 x7d5; do
  lib="$PWD/tools/variants/libplmc_$v.so"
  PLMC_LIB=$lib timeout -k 10 300 python tools/wb_race_probe.py --tag $v --reps 6 > gpurun_out/r2/p3_$v.json 2> gpurun_out/r2/p3_$v.err || echo "probe $v failed"
done
PLMC_LIB=$PWD/tools/variants/libplmc_x7.so PLMC_GRP=1 timeout -k 10 300 python tools/wb_race_probe.py --tag x7g1 --reps 4 > gpurun_out/r2/p3_x7g1.json 2> gpurun_out/r2/p3_x7g1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/p3_*.json')):
    try:
        d=json.load(open(f))
        print(d['tag'], 'serial_diff', d.get('serial_tiles_differ'))
        for r in d['runs']: print('   ', r['live_tiles_differ'], json.dumps(r['detail']))
    except Exception as e: print(f, 'ERR', e)
PY
