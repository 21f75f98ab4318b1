#!/bin/bash
# A/B of ONE build under environment switches, timed on the whole forward: gpurun -- bash tools/micro/env_ab.sh "<fwd_ab args>" "VAR=1" "VAR2=1" ...
# (the unswitched run goes first and saves the reference output; every arm runs twice, interleaved)
R=$GRAFT_REPO_ROOT
ARGS=$1; shift
read PREC WL BS NH <<< "$ARGS"
WL=${WL:-tracking-60k}
rm -f $R/gpurun_out/fwd_ab_base_*.pt
for rep in 1 2; do
  python3 $R/tools/micro/fwd_ab.py $PREC "[]" $WL $BS $NH 2>&1 | grep "us per"
  for sw in "$@"; do
    env $sw python3 $R/tools/micro/fwd_ab.py $PREC "[$sw]" $WL $BS $NH 2>&1 | grep "us per"
  done
done
