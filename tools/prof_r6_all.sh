#!/bin/bash
# usage: tools/prof_r6_all.sh   (runs on the GPU box): the round-6 rocprofv3 kernel traces / counter passes behind DESIGN.md's
# numbers, one summary per program under gpurun_out/prof_r6/ (as tools/prof_r5_all.sh did for round 5).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/prof_r6
mkdir -p $out
one() {   # name program args...: kernel trace + stats -> $out/<name>_kernel_stats.txt
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/$name -o t -- python3 "$@" > $out/$name.log 2>&1 < /dev/null
  { grep -v "rocprofv3\]\|^W2026\|^E2026\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $out/$name.log | tail -4 | cut -c1-400; python3 tools/rocpd_stats.py $out/$name/t_results.db; } > $out/${name}_kernel_stats.txt 2>&1
  rm -rf $out/$name
}
one fullrank_headline bench.py --no-legs --no-cpu-baseline --no-profile
one c3 tools/c3_bench_r6.py philox
one c3_parity tools/c3_bench_r6.py numpy
one api_numpy tools/r6_api_numpy_loop.py
bash tools/timeline.sh c3_r6 700 26 tools/c3_bench_r6.py philox; cp gpurun_out/timeline_c3_r6.txt $out/c3_timeline.txt
bash tools/timeline.sh c3p_r6 2400 60 tools/c3_bench_r6.py numpy; cp gpurun_out/timeline_c3p_r6.txt $out/c3_parity_timeline.txt
# counters: separate passes, no trace domains (MI355X_MICROARCH.md HBM section)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/pmc -o fr_$c -- python3 bench.py --no-legs --no-cpu-baseline --no-profile > $out/pmc_fr_$c.log 2>&1 < /dev/null
  { echo "== $c (headline, bench.py --no-legs)"; python3 tools/rocpd_stats.py $out/pmc/fr_${c}_results.db $c | grep -E "gemm|kernel  " | cut -c1-40,65-200; } >> $out/fullrank_gemm_pmc.txt
  rocprofv3 --pmc $c -d $out/pmc -o mf_$c -- python3 tools/mf_stream_bench.py > $out/pmc_mf_$c.log 2>&1 < /dev/null
  { echo "== $c (tools/mf_stream_bench.py)"; python3 tools/rocpd_stats.py $out/pmc/mf_${c}_results.db $c | grep -E "accum|kernel  " | cut -c1-40,65-200; } >> $out/meanfield_c1_pmc_hbm.txt
done
rm -rf $out/pmc
ls -la $out
