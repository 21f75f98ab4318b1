#!/bin/bash
# A/B of builds of ONE source file x environment switches, timed on the whole forward (tools/micro/fwd_ab.py).
# gpurun -- bash tools/micro/build_env_ab.sh <file.hip> "<precision workload [block]>;<precision workload>..." "<env1> <env2>" "<flags 1>" "<flags 2>" ...
# The unflagged build runs too (first).  Results: gpurun_out/build_env_ab.txt
R=$GRAFT_REPO_ROOT
SRC=$1; shift
ARGSL=$1; shift
ENVS=$1; shift
cd $R/hept_amd/csrc
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OUT=$R/gpurun_out/build_env_ab.txt
mkdir -p $R/gpurun_out; : > $OUT
rm -f $R/gpurun_out/fwd_ab_base_*.pt
OBJ=${SRC%.hip}.o
IFS=';' read -ra ARGSA <<< "$ARGSL"
for flags in "" "$@"; do
  /opt/rocm/bin/hipcc $BASE $flags -c $SRC -o $OBJ 2>> $OUT || { echo "BUILD FAILED: $flags" >> $OUT; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhept_hip.so prep_hash.o sort_tables.o block_attn.o block_attn_bwd.o combine.o block_train.o prepare.o comm.o p2p.o capi.o -ldl
  for ARGS in "${ARGSA[@]}"; do
    read PREC WL BS <<< "$ARGS"
    python3 $R/tools/micro/fwd_ab.py $PREC "[$flags]" $WL $BS 2>&1 | grep "us per" >> $OUT
    for sw in $ENVS; do
      env $sw python3 $R/tools/micro/fwd_ab.py $PREC "[$flags $sw]" $WL $BS 2>&1 | grep "us per" >> $OUT
    done
  done
done
cat $OUT
