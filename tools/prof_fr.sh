#!/bin/bash
# usage: tools/prof_fr.sh <outdir-name> [D N model steps]   (runs on the GPU box): kernel trace of the full-rank pipeline
name=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out -o fr -- python3 tools/fr_bench.py "$@" > $out/fr_bench.log 2>&1
tail -2 $out/fr_bench.log
python3 tools/rocpd_stats.py $out/fr_results.db > $out/kernel_stats.txt 2>&1; head -12 $out/kernel_stats.txt
