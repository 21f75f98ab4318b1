#!/bin/bash
# Build-parameter sweep of the bucket-sort launch with its rider workgroups (the v rows): rebuilds sort_tables.o with
# every flag set, relinks, reports the forward's stage times.  Environment settings go before a '|':
# gpurun -- bash tools/micro/riders_sweep.sh "HEPT_ROW_RIDERS=8|-DHEPT_ROWS_UNROLL=2" ...   (gpurun_out/riders_sweep.txt)
R=$GRAFT_REPO_ROOT
cd $R/hept_amd/csrc
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
OUT=$R/gpurun_out/riders_sweep.txt
mkdir -p $R/gpurun_out; : > $OUT
for spec in "HEPT_NO_ROW_RIDERS=1|" "|" "$@"; do
  envs="${spec%%|*}"; flags="${spec#*|}"
  /opt/rocm/bin/hipcc $BASE $flags -c sort_tables.hip -o sort_tables.o 2>> $OUT || { echo "BUILD FAILED: $flags" >> $OUT; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhept_hip.so prep_hash.o sort_tables.o block_attn.o block_attn_bwd.o combine.o block_train.o prepare.o comm.o p2p.o capi.o -ldl
  echo "== [$envs] [$flags]" >> $OUT
  for prec in bf16 fp32; do
    env $envs python3 $R/bench.py --no-cpu-baseline --no-extra --precision $prec 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$prec %.1f M points/s  %.1f us' % (d['value']/1e6, d['ms_per_step']*1e3))" >> $OUT
    env $envs python3 $R/bench.py --stages --no-cpu-baseline --no-extra --precision $prec 2>&1 | grep "stage ms" >> $OUT
  done
done
cat $OUT
