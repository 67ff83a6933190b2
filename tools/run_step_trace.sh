#!/bin/bash
# dev: whole-step kernel timeline at a given q -> gpurun_out/r4/step_q$q.txt
Q=${1:-1}
OUT=$PWD/gpurun_out/r4; mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/strace_q$Q
rocprofv3 --kernel-trace --output-format csv -d $OUT/strace_q$Q -- python3 $ROOT/bench.py --steps 3 --warmup 2 --latents $Q --no-cpu-baseline --no-prof --no-options > /dev/null 2>&1
f=$(find $OUT/strace_q$Q -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/step_timeline.py $f verbose > $OUT/step_q$Q.txt
rm -rf $OUT/strace_q$Q
