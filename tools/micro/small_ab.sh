#!/bin/bash
# round 5: one-workgroup-per-segment sort split over 4 workgroups (short clouds), A/B on one box
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/small_ab.txt
: > $OUT
for wl in tracking-6k example-4k; do
for prec in fp32 bf16; do
for rep in 1 2; do
  python3 tools/micro/fwd_ab.py $prec "[split 4]" $wl 2>&1 | grep "us per" >> $OUT
  HEPT_SMALL_NO_SPLIT=1 python3 tools/micro/fwd_ab.py $prec "[one workgroup]" $wl 2>&1 | grep "us per" >> $OUT
done
done
done
for rep in 1 2; do python3 tools/micro/fwd_ab.py bf16 "[sampled codes]" tracking-60k 2>&1 | grep "us per" >> $OUT; done
cat $OUT
