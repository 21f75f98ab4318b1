#!/bin/bash
# kernel timelines of the small clouds (one step: kernel starts relative to the first, durations, gaps)
R=$PWD; O=$R/gpurun_out/small_tl; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for wl in example-4k tracking-6k; do for prec in bf16 fp32; do
  export HEPT_TRACE_WORKLOAD=$wl HEPT_TRACE_PREC=$prec
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_${wl}_$prec -- python3 $R/tools/trace_step.py plain 1 60 > $O/log_${wl}_$prec.txt 2>&1
  echo "== $wl $prec"; python3 $R/tools/trace_summary.py $O/tr_${wl}_$prec | grep -v amdgpu.ids
  rm -rf $O/tr_${wl}_$prec
done; done
