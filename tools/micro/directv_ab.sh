#!/bin/bash
# f32 rows: value rows read from the caller's v (default) against the riders that build the v half (HEPT_NO_DIRECT_V=1)
export HEPT_SWEEP_ONLY=tracking-60k,pileup-8clouds,tracking-6k,example-4k
for i in 1 2; do
echo "== direct v"; python3 tools/config_sweep.py 2>&1 | grep "fp32"
echo "== riders (HEPT_NO_DIRECT_V=1)"; HEPT_NO_DIRECT_V=1 python3 tools/config_sweep.py 2>&1 | grep "fp32"
done
