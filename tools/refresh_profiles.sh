cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
for prec in fp32 bf16; do
  # kernel durations from the profiled run; the JSON line from an un-profiled run of the same command (the tracer
  # slows the host's launches, which shows up as gaps between kernels, not inside them)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$prec -o ks -- python3 $R/bench.py --precision $prec --no-cpu-baseline > $O/bench_$prec.log 2>&1
  python3 $R/bench.py --precision $prec 2>/dev/null | grep '^{' | tail -1 > $O/bench_$prec.json
done
cd $R
python3 tools/config_sweep.py > $O/config_sweep.txt 2>&1
python3 tools/attn_block_bench.py > $O/attn_block.txt 2>&1
python3 tools/train_step_bench.py >> $O/attn_block.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_train -o ks -- python3 $R/tools/train_step_bench.py > $O/train.log 2>&1
bash tools/pmc.sh fp32r --precision fp32 > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_fp32r > $O/pmc_fp32.txt 2>&1
ls $O
