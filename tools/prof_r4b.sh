#!/bin/bash
# re-collection of the two round-4 traces whose kernel list changed late in the round (radix-4 jump ladder)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r4
mkdir -p $out
run() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/$name -o t -- python3 "$@" > $out/$name.log 2>&1 < /dev/null
  python3 tools/rocpd_stats.py $out/$name/t_results.db > $out/${name}_kernel_stats.txt 2>&1
  grep -v "rocprofv3\]\|^W20\|^E20\|^I20\|it/s\]" $out/$name.log | tail -4 | cut -c1-300 > $out/${name}_tail.txt
  rm -rf $out/$name
  cat $out/${name}_tail.txt; head -8 $out/${name}_kernel_stats.txt | cut -c1-150
}
run legacydev tools/legacy_dev_bench.py 4096 1024 20
run apicall tools/api_call_bench.py 1024 4096 60
run c3 tools/c3_bench.py
