#!/bin/bash
# Build-parameter sweep of the sort on a GPU box: rebuilds sort_tables.o with every flag set, relinks, times.
# gpurun -- bash tools/micro/sort_sweep.sh "<flags 1>" "<flags 2>" ...   (results: gpurun_out/sort_sweep.txt)
R=$GRAFT_REPO_ROOT
cd $R/hept_amd/csrc
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OUT=$R/gpurun_out/sort_sweep.txt
mkdir -p $R/gpurun_out; : > $OUT
for flags in "" "$@"; do
  /opt/rocm/bin/hipcc $BASE $flags -c sort_tables.hip -o sort_tables.o 2>> $OUT || { echo "BUILD FAILED: $flags" >> $OUT; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhept_hip.so prep_hash.o sort_tables.o block_attn.o block_attn_bwd.o combine.o block_train.o prepare.o comm.o p2p.o capi.o -ldl
  python3 $R/tools/micro/sort_time.py tracking-60k "[$flags]" >> $OUT 2>&1
  python3 $R/tools/micro/sort_time.py pileup-8clouds "[$flags]" >> $OUT 2>&1
done
cat $OUT
