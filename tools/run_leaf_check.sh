cd $GRAFT_REPO_ROOT
# dev tool: the GPU tests that cover the dense / t-family paths, then timings of short shards
timeout 600 python -m pytest tests/test_gpu_fullrank.py tests/test_gpu_objectives.py tests/test_gpu_full_size.py tests/test_gpu_comm.py tests/test_gpu_two_ranks.py tests/test_gpu_fit.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for d in 1024 512; do for n in 2048 1024 512 256; do
a=$(python tools/fr_bench.py $d $n gauss_full 3000 2>&1 | grep "model=" | grep -o "[0-9.]* us/eval")
b=$(VB_FR_KPARTS=1 python tools/fr_bench.py $d $n gauss_full 3000 2>&1 | grep "model=" | grep -o "[0-9.]* us/eval")
c=$(VB_FR_KPARTS=2 python tools/fr_bench.py $d $n gauss_full 3000 2>&1 | grep "model=" | grep -o "[0-9.]* us/eval")
e=$(VB_FR_KPARTS=4 python tools/fr_bench.py $d $n gauss_full 3000 2>&1 | grep "model=" | grep -o "[0-9.]* us/eval")
echo "D=$d n=$n: auto $a | unsplit $b | 2 parts $c | 4 parts $e"
done; done
