cd $GRAFT_REPO_ROOT
(cd /tmp && TMPDIR=/tmp timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ssn -o ks -- python3 $GRAFT_REPO_ROOT/tools/micro/sort_only.py > /dev/null 2>&1)
python3 tools/kstats.py gpurun_out/ssn/ks_kernel_stats.csv | grep small_sort
