#!/bin/bash
# dev: kernel timeline of the last sweep at a given q -> gpurun_out/r2/tl_q$q.txt
Q=${1:-1}
OUT=$PWD/gpurun_out/r3; mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_q$Q
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_q$Q -- python3 $ROOT/bench.py --steps 3 --warmup 2 --latents $Q --no-cpu-baseline --no-prof --no-options $EXTRA_BENCH > /dev/null 2>&1
f=$(find $OUT/trace_q$Q -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $f 0 400 > $OUT/tl_q$Q.txt; python3 $ROOT/tools/step_timeline.py $f v > $OUT/step_q$Q.txt; python3 $ROOT/tools/timeline_stats.py $f > $OUT/tls_q$Q.txt; cat $OUT/tls_q$Q.txt; python3 $ROOT/tools/sweep_phases.py $f > $OUT/phases_q$Q.txt; cat $OUT/phases_q$Q.txt
rm -rf $OUT/trace_q$Q
tail -1 $OUT/tl_q$Q.txt
