#!/bin/bash
# Round profile refresh (run on the GPU box through gpurun, from the repo root): the bench line (with the PLMC_SPLIT=0 / 3 runs
# beside the headline), rocprofv3 kernel stats of the same command with the default arithmetic only, and the two PMC passes
# behind roofline.traffic.  Outputs land in gpurun_out/; copy to profiles/.
set -o pipefail
R=${1:-r01}
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -- python3 $OUT/../bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-options > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -- python3 $OUT/../bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-options > /dev/null 2>&1
python3 $OUT/../tools/pmc_aggregate.py $(find $OUT/pmc_f -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_w -name "*counter_collection.csv" | head -1) $OUT/${R} || exit 1
rm -rf $OUT/pmc_f $OUT/pmc_w
# (the PMC aggregate is keyed on the build: copy it to profiles/ BEFORE the bench line is taken, so that the line quotes it)
cp $OUT/${R}_pmc_traffic.json $OUT/../profiles/${R}_pmc_traffic.json 2>/dev/null
# (the same for the MFMA-utilisation / L2-hit-rate passes)
(cd $OUT/.. && bash tools/collect_pmc_extra.sh $R > /dev/null 2>&1 && cp gpurun_out/${R}_pmc_mfma_l2.json profiles/${R}_pmc_mfma_l2.json)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $OUT/../bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --no-options > /dev/null 2>&1
cp $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv || exit 1
rm -rf $OUT/prof_stats
python3 $OUT/../bench.py --steps 20 --warmup 5 > $OUT/${R}_bench.json 2> $OUT/${R}_bench.log || exit 1
echo done
