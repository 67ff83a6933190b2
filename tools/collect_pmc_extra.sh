#!/bin/bash
# Extra PMC passes behind the roofline discussion (run on the GPU box through gpurun, from the repo root): MFMA pipe utilisation
# and L2 hit rate per kernel of two steps of the bench workload.  One pass per counter group, --kernel-trace only.
R=${1:-r01}
OUT=$PWD/gpurun_out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $OUT/pmc_m -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-options > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-options > /dev/null 2>&1
python3 $ROOT/tools/pmc_counters.py $(find $OUT/pmc_m -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_l -name "*counter_collection.csv" | head -1) > $OUT/${R}_pmc_mfma_l2.json
rm -rf $OUT/pmc_m $OUT/pmc_l
echo done
