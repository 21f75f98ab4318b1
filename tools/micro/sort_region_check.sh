#!/bin/bash
# round 5: the REGION sort layout (pipelined / one bucket per workgroup) against the round-3 layout (HEPT_SORT_LINEAR=1):
# exactness + timing on one box
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/sort_region.txt
mkdir -p $R/gpurun_out; : > $OUT
for wl in tracking-60k pileup-8clouds; do
  python3 tools/micro/sort_time.py $wl pipe 2>&1 | grep "us per" >> $OUT
  HEPT_BKT_NO_PIPE=1 python3 tools/micro/sort_time.py $wl region 2>&1 | grep "us per" >> $OUT
  HEPT_SORT_LINEAR=1 python3 tools/micro/sort_time.py $wl linear 2>&1 | grep "us per" >> $OUT
done
python3 tools/sort_stress.py 300 2>&1 | tail -5 >> $OUT
for wl in tracking-60k pileup-8clouds; do
  rm -f $R/gpurun_out/fwd_ab_base_*.pt
  for rep in 1 2; do
  HEPT_SORT_LINEAR=1 python3 tools/micro/fwd_ab.py bf16 "[linear]" $wl 2>&1 | grep "us per" >> $OUT
  python3 tools/micro/fwd_ab.py bf16 "[pipe]" $wl 2>&1 | grep "us per" >> $OUT
  HEPT_NO_ROW_RIDERS=1 python3 tools/micro/fwd_ab.py bf16 "[pipe, v role]" $wl 2>&1 | grep "us per" >> $OUT
  HEPT_BKT_NO_PIPE=1 python3 tools/micro/fwd_ab.py bf16 "[region]" $wl 2>&1 | grep "us per" >> $OUT
  done
done
cat $OUT
