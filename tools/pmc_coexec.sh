#!/bin/bash
# MFMA / VALU co-execution counters for the training step kernels. Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_coexec
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/train_step_bench.py > $OUT/p1.log 2>&1
python3 $R/tools/pmc_summary.py $OUT | grep -A9 "block_attn_bwd_split\|block_attn_split"
