cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
mkdir -p $O
timeout 600 python tests/op_stress.py 45 > $O/stress.log 2>&1; tail -8 $O/stress.log
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_module.py tests/test_gpu_attn_block.py -m gpu -x -q 2>&1 | tail -15
