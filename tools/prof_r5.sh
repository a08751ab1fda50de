#!/bin/bash
# usage: tools/prof_r5.sh name program args...   (runs on the GPU box): one rocprofv3 kernel trace, summarised into
# gpurun_out/prof_r5/<name>_kernel_stats.txt with the program's own last lines first.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r5
mkdir -p $out
name=$1; shift
timeout 600 rocprofv3 --kernel-trace --stats -d $out/$name -o t -- python3 "$@" > $out/$name.log 2>&1 < /dev/null
python3 tools/rocpd_stats.py $out/$name/t_results.db > $out/${name}_body.txt 2>&1
grep -v "rocprofv3\]\|^W20\|^E20\|^I20\|it/s\]" $out/$name.log | tail -8 | cut -c1-300 > $out/${name}_tail.txt
cat $out/${name}_tail.txt $out/${name}_body.txt > $out/${name}_kernel_stats.txt
rm -rf $out/$name $out/${name}_body.txt
cat $out/${name}_tail.txt; head -24 $out/${name}_kernel_stats.txt | cut -c1-170
