#!/bin/bash
# small clouds: A/B of an environment switch ($1, e.g. HEPT_PREP_NO_EARLY=1) on example-4k / tracking-6k
export HEPT_SWEEP_ONLY=example-4k,tracking-6k
for i in 1 2; do
echo "== default"; python3 tools/config_sweep.py 2>&1 | grep "us/forward"
echo "== $1"; env $1 python3 tools/config_sweep.py 2>&1 | grep "us/forward"
done
