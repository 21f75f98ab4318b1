#!/bin/bash
# Kernel-trace stats + two SQ counter passes of ANY python script (one build, quick look).  Run on the GPU box:
#   bash tools/pmc_any.sh <label> tools/micro/sort_only.py tracking-60k
# Results: gpurun_out/pa_<label>/{ks_kernel_stats.csv, pmc.txt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pa_$1
rm -rf $OUT; mkdir -p $OUT
shift
SCRIPT=$R/$1
shift
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $SCRIPT "$@" > $OUT/ks.log 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $SCRIPT "$@" > $OUT/p$i.log 2>&1
done
python3 $R/tools/kstats.py "$OUT/ks/**/ks_kernel_stats.csv" > $OUT/kstats.txt 2>&1
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc.txt 2>&1
cat $OUT/kstats.txt
