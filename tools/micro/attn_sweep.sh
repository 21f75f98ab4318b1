#!/bin/bash
# Build-parameter sweep of block_attn on a GPU box (rebuilds block_attn.o with every flag set, relinks, times).
# gpurun -- bash tools/micro/attn_sweep.sh <block size> "<flags 1>" ...     (results: gpurun_out/attn_sweep.txt)
R=$GRAFT_REPO_ROOT
BS=$1; shift
cd $R/hept_amd/csrc
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OUT=$R/gpurun_out/attn_sweep.txt
mkdir -p $R/gpurun_out; : > $OUT
for flags in "" "$@"; do
  /opt/rocm/bin/hipcc $BASE $flags -c block_attn.hip -o block_attn.o 2>> $OUT || { echo "BUILD FAILED: $flags" >> $OUT; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhept_hip.so prep_hash.o sort_tables.o block_attn.o block_attn_bwd.o combine.o block_train.o prepare.o comm.o p2p.o capi.o -ldl
  python3 $R/tools/micro/attn_time.py $BS "[$flags]" 2>&1 | grep "us per" >> $OUT
done
cat $OUT
