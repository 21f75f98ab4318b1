#!/bin/bash
# Build-parameter sweep of combine_out on a GPU box (rebuilds combine.o with every flag set, relinks, times).
R=$GRAFT_REPO_ROOT
cd $R/hept_amd/csrc
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OUT=$R/gpurun_out/combine_sweep.txt
mkdir -p $R/gpurun_out; : > $OUT
for flags in "" "$@" ""; do
  /opt/rocm/bin/hipcc $BASE $flags -c combine.hip -o combine.o 2>> $OUT || { echo "BUILD FAILED: $flags" >> $OUT; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhept_hip.so prep_hash.o sort_tables.o block_attn.o block_attn_bwd.o combine.o block_train.o prepare.o comm.o p2p.o capi.o -ldl
  python3 $R/tools/micro/combine_time.py bf16 "[$flags]" 2>&1 | grep "us per" >> $OUT
  python3 $R/tools/micro/combine_time.py fp32 "[$flags]" 2>&1 | grep "us per" >> $OUT
done
cat $OUT
