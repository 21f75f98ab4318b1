cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O
for how in native; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$how -- python3 $R/tools/trace_step.py $how 2 30 > $O/tr_$how.log 2>&1
  python3 $R/tools/trace_summary.py $O/tr_$how > $O/tl_$how.txt 2>&1
done
cd $R
timeout 300 python -m pytest tests/test_gpu_module.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -30
cat $O/tl_native.txt
timeout 300 python tools/shard_overhead.py bf16 2>&1 | grep "us/step"
