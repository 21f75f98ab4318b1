# Regenerates the files of profiles/ on a GPU box: gpurun -- bash tools/refresh_profiles.sh ; results under
# gpurun_out/refresh/ (copy the ones to keep into profiles/ with the round's prefix).  HEPT_GIT_HEAD=<sha> in the
# environment is recorded in attn_traffic.json (the box has no .git).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
# ---- PMC passes first: bench.py copies roofline.traffic / mfma_busy_frac from profiles/attn_traffic.json only when the
#      record carries the digest of THIS tree's block-attention sources and the kernel template this run launches
cd $R
bash tools/pmc_all.sh > $O/pmc_all.log 2>&1     # every record of bench.py's line, both precisions: gpurun_out/pmca_<key>_<precision>/
cp gpurun_out/attn_traffic.json $O/attn_traffic.json
cp gpurun_out/attn_traffic_summary.txt $O/attn_traffic_summary.txt
cp gpurun_out/pmc_c3_bf16.txt $O/pmc_bf16.txt
cp gpurun_out/pmc_c3_fp32.txt $O/pmc_fp32.txt
cp $O/attn_traffic.json profiles/attn_traffic.json
cd /tmp
for prec in fp32 bf16; do
  # kernel durations from the profiled run; the JSON line from an un-profiled run of the same command (the tracer
  # slows the host's launches, which shows up as gaps between kernels, not inside them)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$prec -o ks -- python3 $R/bench.py --precision $prec --no-cpu-baseline --no-extra > $O/bench_$prec.log 2>&1
done
# the driver's command (default precision; carries the fp32, mixed16 and c4 sub-records and the CPU baseline)
python3 $R/bench.py 2>/dev/null | grep '^{' | tail -1 > $O/bench.json
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_driver_cmd.json
python3 $R/bench.py --stages --no-cpu-baseline --no-extra 2>&1 | grep "stage ms" > $O/stage_ms.txt
python3 $R/bench.py --stages --no-cpu-baseline --no-extra --precision fp32 2>&1 | grep "stage ms" >> $O/stage_ms.txt
cd $R
python3 tools/config_sweep.py > $O/config_sweep.txt 2>&1
python3 tools/shard_overhead.py bf16 2>&1 | grep "us/step" > $O/shard_overhead.txt
python3 tools/shard_overhead.py bf16 1 2>&1 | grep "us/step" >> $O/shard_overhead.txt   # BASELINE config 4: one table per GPU
python3 tools/block_size_sweep.py 2>&1 | grep "B=" > $O/block_size_sweep.txt
python3 tools/micro/sort_time.py tracking-60k 2>&1 | grep "us per" > $O/sort_and_combine_micro.txt
python3 tools/micro/sort_time.py pileup-8clouds 2>&1 | grep "us per" >> $O/sort_and_combine_micro.txt
python3 tools/micro/combine_time.py bf16 2>&1 | grep "us per" >> $O/sort_and_combine_micro.txt
python3 tools/micro/combine_time.py fp32 2>&1 | grep "us per" >> $O/sort_and_combine_micro.txt
python3 tools/attn_block_bench.py > $O/attn_block.txt 2>&1
python3 tools/train_step_bench.py >> $O/attn_block.txt 2>&1
python3 tools/attn_block_train_bench.py >> $O/attn_block.txt 2>&1
python3 tools/host_overhead.py > $O/host_overhead.txt 2>&1
python3 tools/model_latency.py > $O/model_and_prepare.txt 2>&1
python3 tools/prepare_input_bench.py >> $O/model_and_prepare.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_train -o ks -- python3 $R/tools/train_step_bench.py > $O/train.log 2>&1
for how in plain p2p rccl; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$how -- python3 $R/tools/trace_step.py $how 2 30 > $O/tr_$how.log 2>&1
  python3 $R/tools/trace_summary.py $O/tr_$how > $O/timeline_$how.txt 2>&1
done
# BASELINE config 4 on one rank: one table, the direct-scatter exchange (one head group)
export HEPT_TRACE_TABLES=1
for how in plain p2p p2pview rccl; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr1_$how -- python3 $R/tools/trace_step.py $how 1 30 > $O/tr1_$how.log 2>&1
  python3 $R/tools/trace_summary.py $O/tr1_$how > $O/timeline_T1_$how.txt 2>&1
done
unset HEPT_TRACE_TABLES
ls $O
