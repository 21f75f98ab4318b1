#!/bin/bash
# round 5 A/B on one box: sort layout, even / uneven row-builder grid, where the v rows are written
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r5_ab.txt
: > $OUT
for wl in tracking-60k pileup-8clouds; do
  python3 tools/micro/sort_time.py $wl region 2>&1 | grep "us per" >> $OUT
  HEPT_SORT_LINEAR=1 python3 tools/micro/sort_time.py $wl linear 2>&1 | grep "us per" >> $OUT
done
python3 tools/sort_stress.py 300 2>&1 | tail -3 >> $OUT
for prec in bf16 fp32; do
for rep in 1 2; do
  python3 tools/micro/fwd_ab.py $prec "[]" tracking-60k 2>&1 | grep "us per" >> $OUT
  HEPT_PREP_UNEVEN=1 python3 tools/micro/fwd_ab.py $prec "[uneven]" tracking-60k 2>&1 | grep "us per" >> $OUT
  HEPT_NO_ROW_RIDERS=1 python3 tools/micro/fwd_ab.py $prec "[v role]" tracking-60k 2>&1 | grep "us per" >> $OUT
  HEPT_NO_ROW_RIDERS=1 HEPT_PREP_UNEVEN=1 python3 tools/micro/fwd_ab.py $prec "[v role, uneven]" tracking-60k 2>&1 | grep "us per" >> $OUT
  HEPT_SORT_LINEAR=1 HEPT_PREP_UNEVEN=1 python3 tools/micro/fwd_ab.py $prec "[r04: linear, uneven]" tracking-60k 2>&1 | grep "us per" >> $OUT
done
done
cat $OUT
