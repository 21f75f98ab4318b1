cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for prec in fp32; do
HEPT_TRACE_WORKLOAD=tracking-6k HEPT_TRACE_PREC=$prec rocprofv3 --kernel-trace --output-format csv -d /tmp/tr6 -- python3 $R/tools/trace_step.py plain 1 40 > /dev/null 2>&1
python3 $R/tools/trace_summary.py /tmp/tr6
done
cd $R; python tools/config_sweep.py 2>&1 | grep "tracking-6k\|example"
