#!/bin/bash
# On the GPU box: run one command under every prebuilt variant library (tools/micro/mkvariants.sh), base first and last.
# gpurun -- bash tools/micro/run_variants_cmd.sh "<command>" "<grep pattern>" [name ...]
R=$GRAFT_REPO_ROOT
C=$R/hept_amd/csrc
CMD=$1; PAT=$2; shift; shift
NAMES="$@"
[ -z "$NAMES" ] && NAMES=$(cd $C/variants && ls *.so | sed 's/\.so$//')
OUT=$R/gpurun_out/variants_cmd.txt
mkdir -p $R/gpurun_out; : > $OUT
cp $C/libhept_hip.so /tmp/base.so
for full in base $NAMES base; do
  name=${full%%@*}
  envs=""; [ "$full" != "$name" ] && envs=$(echo "${full#*@}" | tr '@' ' ')
  if [ $name = base ]; then cp /tmp/base.so $C/libhept_hip.so; else cp $C/variants/$name.so $C/libhept_hip.so; fi
  echo "== $full" >> $OUT
  (cd $R && env $envs $CMD 2>&1 | grep -E "$PAT" >> $OUT)
done
cp /tmp/base.so $C/libhept_hip.so
cat $OUT
