#!/bin/bash
# wait_copy_out grid sweep on the one-rank proxy (T = 1, one-sided)
export HEPT_SHARD_MODES="plain,all_to_all/1/p2p,all_to_all/1/p2p+view"
for w in 32 64 128 256 512 1024; do
  echo "== HEPT_COPY_OUT_WGS=$w"
  HEPT_COPY_OUT_WGS=$w python3 tools/shard_overhead.py bf16 1 2>&1 | grep "us/step"
done
