cd $GRAFT_REPO_ROOT
# dev tool: the GPU tests that cover the dense / t-family paths, then the C3 timing
timeout 600 python -m pytest tests/test_gpu_fullrank.py tests/test_gpu_objectives.py tests/test_gpu_full_size.py tests/test_gpu_comm.py tests/test_gpu_two_ranks.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/c3_bench.py 2>&1 | grep "C3 shape" | cut -c1-110
