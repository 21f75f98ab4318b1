cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/train_step_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_train16 -o ks -- python3 $R/tools/train_step_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/ks_train16/ks_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"].split("::")[-1][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
