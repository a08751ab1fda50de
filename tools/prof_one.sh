#!/bin/bash
# usage: tools/prof_one.sh <name> <python program and args>  (GPU box): one rocprofv3 kernel trace, summarised
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/prof_one
mkdir -p $out
name=$1; shift
rocprofv3 --kernel-trace --stats -d $out/$name -o t -- python3 "$@" > $out/$name.log 2>&1
python3 tools/rocpd_stats.py $out/$name/t_results.db > $out/${name}_kernel_stats.txt 2>&1
grep -v "rocprofv3\]\|^W20\|^E20\|^I20\|it/s\]" $out/$name.log | tail -4 | cut -c1-300
rm -rf $out/$name
cut -c1-150 $out/${name}_kernel_stats.txt | head -50
