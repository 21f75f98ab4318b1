cd $GRAFT_REPO_ROOT
O=gpurun_out/riders.txt; : > $O
run() { echo "== $1" >> $O; env $1 python3 bench.py --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step']*1e3)" >> $O; env $1 python3 bench.py --stages --no-cpu-baseline --no-extra 2>&1 | grep "stage ms" >> $O; }
run HEPT_NO_ROW_RIDERS=1
run HEPT_ROW_RIDERS=4
run HEPT_ROW_RIDERS=2
run HEPT_ROW_RIDERS=8
run HEPT_ROW_RIDERS=16
echo "== fp32" >> $O
for e in HEPT_NO_ROW_RIDERS=1 HEPT_ROW_RIDERS=4 HEPT_ROW_RIDERS=8; do echo $e >> $O; env $e python3 bench.py --no-cpu-baseline --no-extra --precision fp32 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step']*1e3)" >> $O; done
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_module.py -x -q -m gpu 2>&1 | tail -5 >> $O
cat $O
