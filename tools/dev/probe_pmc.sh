#!/bin/bash
# dev: LDS / wait / TCP counters of the engine_rate_probe kernels (from-L2 loop vs the tail's real shape), one pass per group
OUT=$PWD/gpurun_out; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES" "MfmaUtil"; do
  i=$((i+1)); rm -rf $OUT/pp_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pp_$i -- $ROOT/tools/bin/engine_rate_probe k > /dev/null 2>&1
done
python3 $ROOT/tools/pmc_counters.py $(find $OUT/pp_* -name "*counter_collection.csv") > $OUT/probe_counters.json
rm -rf $OUT/pp_*
python3 - <<PY
import json
d = json.load(open("$OUT/probe_counters.json"))["kernels"]
for k, v in d.items():
    if "k_rate" in k or "k_tail" in k:
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        print("%-52s n=%3d mfma %5.1f%% ldsconf/idx %.3f wait_lds/wave %.3f wait_any/wave %.3f datafifo %.2e cmdfifo %.2e vmem_rd/busy %.3f" % (
            k[-52:], v["dispatches"], v.get("MfmaUtil", -1), v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
            v.get("SQ_WAIT_INST_LDS", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_LDS_DATA_FIFO_FULL", 0), v.get("SQ_LDS_CMD_FIFO_FULL", 0),
            v.get("SQ_INST_CYCLES_VMEM_RD", 0) / max(v.get("SQ_BUSY_CYCLES", 1), 1)))
PY
