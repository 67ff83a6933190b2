#!/bin/bash
# dev: pool size of the resident chain (PLMC_CHAIN_NW) per local latent count, one session
for spec in "8:24 31 40" "4:24 36 48" "2:32 44 60" "1:60 80 100"; do
  N=${spec%%:*}
  for nw in ${spec#*:}; do
    PLMC_CHAIN_NW=$nw timeout -k 10 200 python tools/time_shard_rank.py $N 2>&1 | grep -v amdgpu.ids | sed "s/^/NW=$nw /"
  done
done
