cd $GRAFT_REPO_ROOT
timeout 300 python tools/shard_overhead.py bf16 2>&1 | grep "us/step" | head -5
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_two_process.py -m gpu -x -q 2>&1 | grep "passed\|failed"
