#!/bin/bash
# usage: tools/prof.sh <outdir-name> <bench args...>   (runs on the GPU box)
# kernel trace of the bench + (separate passes) PMC counters for HBM traffic
name=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 bench.py "$@" --no-cpu-baseline --no-fullrank > $out/bench_trace.log 2>&1
grep '"metric"' $out/bench_trace.log | cut -c1-400
python3 tools/rocpd_stats.py $out/trace_results.db > $out/kernel_stats.txt 2>&1; cat $out/kernel_stats.txt
if [ -n "$PMC" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out -o pmc_$c -- python3 bench.py "$@" --no-cpu-baseline --no-fullrank > $out/bench_pmc_$c.log 2>&1
    python3 tools/rocpd_stats.py $out/pmc_${c}_results.db $c >> $out/pmc_summary.txt 2>&1
  done
  cat $out/pmc_summary.txt
fi
