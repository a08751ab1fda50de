#!/bin/bash
# usage: tools/c3_timeline.sh   (runs on the GPU box): kernel timeline (start order, gaps) of C3-shaped DIS calls under
# rocprofv3 --kernel-trace -> gpurun_out/c3_timeline.txt
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace -d $out/c3tl -o t -- python3 tools/c3_bench.py > $out/c3tl.log 2>&1 < /dev/null
python3 tools/rocpd_stats.py $out/c3tl/t_results.db --timeline ${1:-60} ${2:-70} > $out/c3_timeline.txt 2>&1
rm -rf $out/c3tl
