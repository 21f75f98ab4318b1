#!/bin/bash
# round 5: where the v rows should be written now that the bucket pass is shorter (REGION layout + lean path)
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/riders_r5.txt
: > $OUT
for prec in bf16 fp32; do
for rep in 1 2 3; do
  python3 tools/micro/fwd_ab.py $prec "[riders 3]" tracking-60k 2>&1 | grep "us per" >> $OUT
  HEPT_NO_ROW_RIDERS=1 python3 tools/micro/fwd_ab.py $prec "[v role]" tracking-60k 2>&1 | grep "us per" >> $OUT
  HEPT_ROW_RIDERS=4 python3 tools/micro/fwd_ab.py $prec "[riders 4]" tracking-60k 2>&1 | grep "us per" >> $OUT
done
done
for rep in 1 2; do
HEPT_NO_ROW_RIDERS=1 python3 tools/micro/fwd_ab.py bf16 "[v role]" pileup-8clouds 2>&1 | grep "us per" >> $OUT
python3 tools/micro/fwd_ab.py bf16 "[riders 3]" pileup-8clouds 2>&1 | grep "us per" >> $OUT
HEPT_NO_ROW_RIDERS=1 python3 tools/micro/fwd_ab.py bf16 "[v role]" tracking-60k - 2 2>&1 | grep "us per" >> $OUT
python3 tools/micro/fwd_ab.py bf16 "[riders 3]" tracking-60k - 2 2>&1 | grep "us per" >> $OUT
done
cat $OUT
