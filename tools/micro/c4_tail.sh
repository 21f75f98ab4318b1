#!/bin/bash
# round 5, c4 exchange tail: the view mode and the staged combine on the receive region (one-rank proxy)
O=gpurun_out/c4tail; mkdir -p $O
python3 -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -5 > $O/tests.txt
python3 -m pytest tests/test_gpu_two_process.py tests/test_gpu_c_host.py -x -q 2>&1 | tail -5 >> $O/tests.txt
python3 tools/shard_overhead.py bf16 1 2>&1 | grep "us/step" > $O/shard_T1.txt
HEPT_NO_STAGED_COMBINE=1 python3 tools/shard_overhead.py bf16 1 2>&1 | grep "us/step" > $O/shard_T1_nostg.txt
python3 tools/shard_overhead.py bf16 3 2>&1 | grep "us/step" > $O/shard_T3.txt
R=$PWD; export HEPT_TRACE_TABLES=1
cd /tmp; export TMPDIR=/tmp
for how in p2p p2pview; do
  rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr1_$how -- python3 $R/tools/trace_step.py $how 1 30 > $R/$O/tr1_$how.log 2>&1
  python3 $R/tools/trace_summary.py $R/$O/tr1_$how > $R/$O/timeline_T1_$how.txt 2>&1
done
cd $R; rm -rf $O/tr1_p2p $O/tr1_p2pview
cat $O/tests.txt $O/shard_T1.txt $O/shard_T1_nostg.txt $O/timeline_T1_p2p.txt $O/timeline_T1_p2pview.txt
