#!/bin/bash
# A/B builds of ONE source file on a GPU box, timed on the whole forward (tools/micro/fwd_ab.py).
# gpurun -- bash tools/micro/ab_sweep.sh <file.hip> "<fwd_ab args: precision [workload [block]]>" "<flags 1>" "<flags 2>" ...
# The unflagged build runs first (it saves the reference output).  Results: gpurun_out/ab_sweep.txt
R=$GRAFT_REPO_ROOT
SRC=$1; shift
ARGS=$1; shift
cd $R/hept_amd/csrc
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OUT=$R/gpurun_out/ab_sweep.txt
mkdir -p $R/gpurun_out; : > $OUT
read PREC WL BS <<< "$ARGS"
WL=${WL:-tracking-60k}
rm -f $R/gpurun_out/fwd_ab_base_*.pt
OBJ=${SRC%.hip}.o
for flags in "" "$@"; do
  /opt/rocm/bin/hipcc $BASE $flags -c $SRC -o $OBJ 2>> $OUT || { echo "BUILD FAILED: $flags" >> $OUT; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhept_hip.so prep_hash.o sort_tables.o block_attn.o block_attn_bwd.o combine.o block_train.o prepare.o comm.o p2p.o capi.o -ldl
  python3 $R/tools/micro/fwd_ab.py $PREC "[$flags]" $WL $BS 2>&1 | grep "us per" >> $OUT
done
cat $OUT
