#!/bin/bash
# A/B of an environment switch ($1) on a subset of the workloads ($2, comma separated): two alternating runs each
export HEPT_SWEEP_ONLY=${2:-tracking-60k,pileup-8clouds,tracking-6k}
for i in 1 2; do
echo "== default"; python3 tools/config_sweep.py 2>&1 | grep "us/forward" | grep -v mixed16
echo "== $1"; env $1 python3 tools/config_sweep.py 2>&1 | grep "us/forward" | grep -v mixed16
done
