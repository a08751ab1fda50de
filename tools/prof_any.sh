#!/bin/bash
# usage: tools/prof_any.sh <outdir-name> <python script> [args...]   (runs on the GPU box): rocprofv3 kernel trace
name=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out -o t -- python3 "$@" > $out/run.log 2>&1
grep -v "rocprofv3\]\|^W2026\|^E2026" $out/run.log | tail -3
python3 tools/rocpd_stats.py $out/t_results.db > $out/kernel_stats.txt 2>&1; head -8 $out/kernel_stats.txt | cut -c1-175
