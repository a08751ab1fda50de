#!/bin/bash
# usage: tools/r6_c3_kernels_ab.sh "ENV=VAL ..." ...   (runs on the GPU box): one rocprofv3 kernel-trace pass of the C3 loop
# (tools/c3_bench_r6.py philox) per argument, each with that argument's environment settings exported ("-" = none)
# -> gpurun_out/c3_ab/<i>.txt (per-kernel averages), and the unprofiled per-call time of each setting (tools/r6_c3_ab.py's
# first line) -> gpurun_out/c3_ab/calls.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/c3_ab
mkdir -p $out
: > $out/calls.txt
i=0
for setting in "$@"; do
  i=$((i + 1))
  (
    if [ "$setting" != "-" ]; then export $setting; fi
    rocprofv3 --kernel-trace --stats -d $out/run$i -o t -- python3 tools/c3_bench_r6.py philox > $out/run$i.log 2>&1 < /dev/null
    { echo "# $setting"; python3 tools/rocpd_stats.py $out/run$i/t_results.db | cut -c1-150 | head -22; } > $out/$i.txt 2>&1
    rm -rf $out/run$i
    echo "# $setting" >> $out/calls.txt
    timeout 300 python3 tools/c3_call_time.py >> $out/calls.txt 2>&1
  )
done
cat $out/calls.txt
