#!/bin/bash
# PMC passes of the block-attention kernel for EVERY record of bench.py's line, one directory per (record key, precision)
# (one counter group per pass, as MI355X_MICROARCH.md prescribes; --kernel-trace only), then profiles/attn_traffic.json
# keyed by (record key / precision, kernel template).  Run on the GPU box:
#   bash tools/pmc_all.sh [full]     -> gpurun_out/pmca_<key>_<precision>/p<i>/ , gpurun_out/attn_traffic.json
# The headline records (c3) also take the LDS / wait / cache-hit groups; "full" gives them to every record.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SPECS=""
for spec in "c3 tracking-60k" "c1 example-4k" "c2 tracking-6k" "c2x10 tracking-6k-x10" "c5 pileup-8clouds" "b100 tracking-60k 100"; do
  read KEY WL BS <<< "$spec"
  for prec in fp32 bf16; do
    OUT=$R/gpurun_out/pmca_${KEY}_$prec
    rm -rf $OUT; mkdir -p $OUT
    CGRPS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum")
    if [ "$KEY" = c3 ] || [ "$1" = full ]; then
      CGRPS+=("SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS")
    fi
    i=0
    for grp in "${CGRPS[@]}"; do
      i=$((i+1))
      rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/attn_shape_run.py $WL $prec $BS > $OUT/p$i.log 2>&1
    done
    SPECS="$SPECS $KEY/$prec=$OUT"
  done
done
cd $R
python3 tools/make_traffic.py gpurun_out/attn_traffic.json $SPECS > gpurun_out/attn_traffic_summary.txt 2>&1
for prec in bf16 fp32; do python3 tools/pmc_summary.py gpurun_out/pmca_c3_$prec > gpurun_out/pmc_c3_$prec.txt 2>&1; done
tail -60 gpurun_out/attn_traffic_summary.txt
