#!/bin/bash
# Build-parameter sweep of prep_hash on a GPU box, judged on the WHOLE forward (bench.py, no extras), on the kernel's
# stage time, and on the tracking-6k forward.  gpurun -- bash tools/micro/prep_sweep.sh "<flags 1>" ...   (results: gpurun_out/prep_sweep.txt)
R=$GRAFT_REPO_ROOT
cd $R/hept_amd/csrc
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OUT=$R/gpurun_out/prep_sweep.txt
mkdir -p $R/gpurun_out; : > $OUT
for flags in "" "$@" ""; do
  /opt/rocm/bin/hipcc $BASE $flags -c prep_hash.hip -o prep_hash.o 2>> $OUT || { echo "BUILD FAILED: $flags" >> $OUT; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhept_hip.so prep_hash.o sort_tables.o block_attn.o block_attn_bwd.o combine.o block_train.o prepare.o comm.o p2p.o capi.o -ldl
  (cd $R; echo "[$flags] $(python3 bench.py --no-extra --no-cpu-baseline --stages 2>&1 | grep -E 'stage ms|ms_per_step' | sed 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/' | tr '\n' ' ')" ; python3 tools/host_overhead.py 2>&1 | grep "N_raw=6000") >> $OUT
done
cat $OUT
