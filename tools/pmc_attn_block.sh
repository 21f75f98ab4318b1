#!/bin/bash
# SQ counter passes around tools/attn_block_bench.py (the fused Attn block: prep_fused_kernel, combine_out<FFN>). GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_attnblk
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/attn_block_bench.py > $OUT/p$i.log 2>&1
done
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_attnblk | grep -A16 "prep_fused\|combine_out_kernel<true, true"
