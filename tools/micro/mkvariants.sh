#!/bin/bash
# Build A/B variants of libhept_hip.so HERE (no GPU minutes): each argument is "name|file.hip[,file2.hip]|flags".
# Result: hept_amd/csrc/variants/<name>.so (git-ignored, travels with gpurun).  The base objects must be up to date (make).
R=/root/repo
C=$R/hept_amd/csrc
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1"
ALL="prep_hash sort_tables block_attn block_attn_bwd combine block_train prepare comm p2p capi"
mkdir -p $C/variants
build_one() {
  IFS='|' read -r name files flags <<< "$1"
  tmp=/tmp/var_$name; mkdir -p $tmp
  objs=""
  for o in $ALL; do
    if [[ ",$files," == *",$o.hip,"* ]]; then
      /opt/rocm/bin/hipcc $BASE $flags -c $C/$o.hip -o $tmp/$o.o || { echo "BUILD FAILED $name"; return 1; }
      objs="$objs $tmp/$o.o"
    else
      objs="$objs $C/$o.o"
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/$name.so $objs -ldl && echo "built $name"
}
export -f build_one; export C BASE ALL
printf '%s\n' "$@" | xargs -P 8 -I{} bash -c 'build_one "$@"' _ {}
