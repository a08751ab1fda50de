#!/bin/bash
# usage: tools_prof.sh <outdir-name> <bench args...>   (runs on the GPU box)
set -e
name=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out -o prof -- python3 bench.py "$@" --no-cpu-baseline > $out/bench.log 2>&1 || true
tail -2 $out/bench.log | cut -c1-600
find $out -name "*stats*" | head
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f"
