#!/bin/bash
# dev aid: SQ counters of the product tile engine on a plain 8192^3 GEMM (tools/mainloop_bench 1)
cd /tmp && export TMPDIR=/tmp
OUT=$1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pg$i -- $OUT/../tools/mainloop_bench 1 > /dev/null 2>&1
  f=$(find $OUT/pg$i -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv
from collections import defaultdict
d=defaultdict(list)
for r in csv.DictReader(open("$f")):
    if "k_gemm" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items(): print("%-28s mean per dispatch %.4g (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
  rm -rf $OUT/pg$i
done
