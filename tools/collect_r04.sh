#!/bin/bash
# Round-4 profile refresh on the GPU box (gpurun, from the repo root).  Everything lands in gpurun_out/; copy to profiles/.
R=r04
OUT=$PWD/gpurun_out
ROOT=$PWD
mkdir -p $OUT
bash tools/collect_profiles.sh $R > $OUT/${R}_collect.log 2>&1
echo "collect_profiles rc=$?"
# A/B on this box: launch-per-step chain vs the resident chain kernel, per-rank shards q = 8, 4, 2, 1
( source tools/ab_chain.sh true; QS="8 4 2 1"; run launches PLMC_CHAIN=0; run chain PLMC_CHAIN=1 ) > $OUT/${R}_chain_ab.txt 2>&1
# what one rank of the N-GPU run executes
timeout -k 10 300 python tools/time_shard_rank.py 8 4 2 1 2>&1 | grep -v amdgpu.ids > $OUT/${R}_shard_rank_times.txt
PLMC_CHAIN=0 timeout -k 10 300 python tools/time_shard_rank.py 8 1 2>&1 | grep -v amdgpu.ids | sed 's/^/PLMC_CHAIN=0 /' >> $OUT/${R}_shard_rank_times.txt
PLMC_LATE_CHECK=0 timeout -k 10 300 python tools/time_shard_rank.py 8 1 2>&1 | grep -v amdgpu.ids | sed 's/^/PLMC_LATE_CHECK=0 /' >> $OUT/${R}_shard_rank_times.txt
# where the host waits inside one training step (settings.late_pivot_check on / off)
timeout -k 10 200 python tools/dev/host_phases.py 1 8 2>&1 | grep -v amdgpu.ids > $OUT/${R}_host_phases.txt
PLMC_LATE_CHECK=0 timeout -k 10 200 python tools/dev/host_phases.py 1 8 2>&1 | grep -v amdgpu.ids >> $OUT/${R}_host_phases.txt
# critical workgroup of the chain kernel, phase by phase (variant build with -DPLMC_CHAIN_TRACE)
PLMC_LIB=tools/variants/libplmc_trace.so timeout -k 10 120 python tools/chain_trace.py 1 2>&1 | grep -v amdgpu.ids > $OUT/${R}_chain_trace_q1.txt
# the other BASELINE configs
timeout -k 10 200 python tools/time_c2.py 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/${R}_c2_fp64.json
timeout -k 10 200 python tools/time_c4.py 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/${R}_c4_variational.json
timeout -k 10 400 python tools/time_c5.py 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/${R}_c5_share.json
timeout -k 10 300 python tools/time_predict.py 2>&1 | grep -v amdgpu.ids | tail -2 > $OUT/${R}_predict.txt
# sweep timelines
bash tools/run_trace.sh 8 > /dev/null 2>&1; cp $OUT/r3/phases_q8.txt $OUT/${R}_sweep_phases_q8.txt; cp $OUT/r3/tls_q8.txt $OUT/${R}_sweep_timeline_stats_q8.txt
bash tools/run_trace.sh 1 > /dev/null 2>&1; cp $OUT/r3/phases_q1.txt $OUT/${R}_sweep_phases_q1.txt
bash tools/run_step_trace.sh 1 > /dev/null 2>&1; cp $OUT/r4/step_q1.txt $OUT/${R}_step_timeline_q1.txt
echo done
