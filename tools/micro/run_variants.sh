#!/bin/bash
# On the GPU box: time every prebuilt variant library (tools/micro/mkvariants.sh) on the whole forward, base first and last.
# gpurun -- bash tools/micro/run_variants.sh "<precision workload [block]>;<...>" [name ...]   (default: every variant)
R=$GRAFT_REPO_ROOT
C=$R/hept_amd/csrc
ARGSL=$1; shift
NAMES="$@"
[ -z "$NAMES" ] && NAMES=$(cd $C/variants && ls *.so | sed 's/\.so$//')
OUT=$R/gpurun_out/variants.txt
mkdir -p $R/gpurun_out; : > $OUT
rm -f $R/gpurun_out/fwd_ab_base_*.pt
cp $C/libhept_hip.so /tmp/base.so
IFS=';' read -ra ARGSA <<< "$ARGSL"
# a name may carry environment switches: name@VAR=1@VAR2=3 (name "base" = the shipped library)
for full in base $NAMES base; do
  name=${full%%@*}
  envs=""; [ "$full" != "$name" ] && envs=$(echo "${full#*@}" | tr '@' ' ')
  if [ $name = base ]; then cp /tmp/base.so $C/libhept_hip.so; else cp $C/variants/$name.so $C/libhept_hip.so; fi
  for ARGS in "${ARGSA[@]}"; do
    read PREC WL BS NH <<< "$ARGS"
    env $envs python3 $R/tools/micro/fwd_ab.py $PREC "[$full]" $WL $BS $NH 2>&1 | grep -E "us per|Error|error" >> $OUT
  done
done
cp /tmp/base.so $C/libhept_hip.so
cat $OUT
