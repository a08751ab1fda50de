cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fullrank.py tests/test_gpu_objectives.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/c3_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
python tools/fr_bench.py 1024 4096 gauss_full 100 path_deriv 2>&1 | tail -2
python tools/fr_bench.py 512 4096 gauss_full 100 path_deriv 2>&1 | tail -1
