#!/bin/bash
# usage: tools/prof_r2.sh   (runs on the GPU box): round-2 rocprofv3 kernel traces of the bench headline and the C2 / C3 / C4 shapes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r2
mkdir -p $out
run() {   # name, program args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/$name -o t -- python3 "$@" > $out/$name.log 2>&1
  python3 tools/rocpd_stats.py $out/$name/t_results.db > $out/${name}_kernel_stats.txt 2>&1
  grep -v "rocprofv3\]\|^W20\|^E20\|^I20" $out/$name.log | tail -4 | cut -c1-300
  head -12 $out/${name}_kernel_stats.txt | cut -c1-170
}
run headline bench.py --no-legs --no-cpu-baseline --no-profile
run fr512 tools/fr_bench.py 512 4096 gauss_full 300
run c3 tools/c3_bench.py
run c4 tools/c4_bench.py
