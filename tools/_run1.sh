cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepare.py -m gpu -x -q -k "sort or prepare or argsort or full_size" 2>&1 | tail -2
timeout 300 python tools/sort_stress.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trx -- python3 $GRAFT_REPO_ROOT/tools/trace_step.py plain 1 40 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/trx | grep "keygen\|scatter\|bucket\|step of"
cd $GRAFT_REPO_ROOT/hept_amd/csrc
rm -f sort_tables.o; make CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DHEPT_SCATTER_THREADS=1024" > /dev/null 2>&1
cd /tmp; rm -rf /tmp/trx; rocprofv3 --kernel-trace --output-format csv -d /tmp/trx -- python3 $GRAFT_REPO_ROOT/tools/trace_step.py plain 1 40 > /dev/null 2>&1
echo "== 1024 threads"; python3 $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/trx | grep "keygen\|scatter\|bucket\|step of"
