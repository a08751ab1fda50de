#!/bin/bash
# usage: tools/prof_r6_c3.sh   (GPU box): the C3 call's kernel stats, timeline and HBM counters after the round-6 changes
# -> gpurun_out/prof_r6/ (the c3 part of tools/prof_r6_all.sh + tools/r6_c3_pmc.sh + the PSIS variant)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/prof_r6
mkdir -p $out
one() {
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/$name -o t -- python3 "$@" > $out/$name.log 2>&1 < /dev/null
  { grep -v "rocprofv3\]\|^W2026\|^E2026\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $out/$name.log | tail -4 | cut -c1-400; python3 tools/rocpd_stats.py $out/$name/t_results.db; } > $out/${name}_kernel_stats.txt 2>&1
  rm -rf $out/$name
}
one c3 tools/c3_bench_r6.py philox
one c3_psis tools/c3_bench_r6.py philox psis
one c3_parity tools/c3_bench_r6.py numpy
bash tools/timeline.sh c3_r6 700 24 tools/c3_bench_r6.py philox; cp gpurun_out/timeline_c3_r6.txt $out/c3_timeline.txt
bash tools/timeline.sh c3psis_r6 800 28 tools/c3_bench_r6.py philox psis; cp gpurun_out/timeline_c3psis_r6.txt $out/c3_psis_timeline.txt
bash tools/r6_c3_pmc.sh > /dev/null 2>&1; cp gpurun_out/c3_pmc/c3_pmc.txt $out/c3_pmc_hbm.txt
python3 tools/c3_call_time.py > $out/c3_call_time.txt 2>&1
ls -la $out
