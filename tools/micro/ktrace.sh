#!/bin/bash
# kernel-trace stats of any python script under environment switches: bash tools/micro/ktrace.sh <label> "<VAR=1 ...|->" script.py args...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LABEL=$1; shift
ENVS=$1; shift
SCRIPT=$R/$1; shift
OUT=$R/gpurun_out/kt_$LABEL
rm -rf $OUT; mkdir -p $OUT
if [ "$ENVS" != "-" ]; then export $ENVS; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python3 $SCRIPT "$@" > $OUT/ks.log 2>&1
echo "== $LABEL [$ENVS]"
python3 $R/tools/kstats.py "$OUT/**/ks_kernel_stats.csv" | head -8
