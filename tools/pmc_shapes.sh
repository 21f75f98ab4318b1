#!/bin/bash
# PMC passes of the block-attention kernel for the shapes of bench.py's sub-records (one counter group per pass, as
# MI355X_MICROARCH.md prescribes; --kernel-trace only).  Run on the GPU box after tools/pmc.sh:
#   bash tools/pmc_shapes.sh          -> gpurun_out/pmcs_<key>_<precision>/p<i>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "c5 pileup-8clouds" "b100 tracking-60k 100" "c2x10 tracking-6k-x10"; do
  read KEY WL BS <<< "$spec"
  for prec in fp32 bf16; do
    OUT=$R/gpurun_out/pmcs_${KEY}_$prec
    rm -rf $OUT; mkdir -p $OUT
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/attn_shape_run.py $WL $prec $BS > $OUT/p$i.log 2>&1
    done
  done
done
ls -d $R/gpurun_out/pmcs_*
