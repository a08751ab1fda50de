#!/bin/bash
# usage: tools/pmc.sh <outdir-name> "<counter list>" <program> [args...]   (runs on the GPU box)
# one rocprofv3 --pmc pass per counter; prints the per-launch average of each counter per kernel
name=$1; shift
counters=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
for c in $counters; do
  rocprofv3 --pmc $c -d $out -o pmc_$c -- "$@" > $out/run_$c.log 2>&1
  echo "== $c"; python3 tools/rocpd_stats.py $out/pmc_${c}_results.db $c | grep -E "gemm|accum|kernel  " | cut -c1-40,65-200
done
