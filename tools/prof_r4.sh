#!/bin/bash
# usage: tools/prof_r4.sh [pmc]  (runs on the GPU box): round-4 rocprofv3 kernel traces of the bench headline and of the
# C2 / funnel / path-derivative / fit-loop / C3 / MultivariateT-ExclusiveKL / C4 / C1 / low-rank shapes; with `pmc` also the
# matrix-pipe and HBM counters of the three full-rank GEMMs and the HBM counters of the mean-field streaming kernel
# (one counter per pass: the guide's rule for this pool).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r4
rm -rf $out; mkdir -p $out
run() {   # name, program args...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/$name -o t -- python3 "$@" > $out/$name.log 2>&1 < /dev/null
  python3 tools/rocpd_stats.py $out/$name/t_results.db > $out/${name}_kernel_stats.txt 2>&1
  grep -v "rocprofv3\]\|^W20\|^E20\|^I20\|it/s\]" $out/$name.log | tail -4 | cut -c1-300 > $out/${name}_tail.txt
  rm -rf $out/$name
  cat $out/${name}_tail.txt; head -10 $out/${name}_kernel_stats.txt | cut -c1-170
}
run headline bench.py --no-legs --no-cpu-baseline --no-profile
run fr512 tools/fr_bench.py 512 4096 gauss_full 300
run funnel tools/fr_bench.py 1024 4096 funnel 200
run pathderiv tools/fr_bench.py 1024 4096 gauss_full 100 path_deriv
run frfit tools/fr_fit_bench.py
run c3 tools/c3_bench.py
run mvtekl tools/mvt_ekl_bench.py
run c4 tools/c4_bench.py
run fit tools/fit_bench.py
run lr8 tools/lr_bench.py 1024 4096 8
run alpha tools/alpha_bench.py
run legacydev tools/legacy_dev_bench.py 4096 1024 20
run apicall tools/api_call_bench.py 1024 4096 60
run lr32 tools/lr_bench.py 1024 4096 32 funnel 100
run lr64 tools/lr_bench.py 1024 4096 64 funnel 100
run psis tools/psis_bench.py
run disbisect tools/dis_bisect_bench.py
if [ "$1" = "pmc" ]; then
  for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/pmc_$c -o p -- python3 tools/fr_bench.py 1024 4096 gauss_full 100 > $out/pmc_$c.log 2>&1
    echo "== $c" >> $out/fr1024_pmc.txt
    python3 tools/rocpd_stats.py $out/pmc_$c/p_results.db $c | grep -E "gemm|reduce|kernel  " | cut -c1-44,65-200 >> $out/fr1024_pmc.txt
    rm -rf $out/pmc_$c
  done
  cat $out/fr1024_pmc.txt
  # mean-field streaming kernel (BASELINE configs[1]): HBM-side bytes of a 32-evaluation launch
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/pmc_mf_$c -o p -- python3 tools/mf_stream_bench.py > $out/pmc_mf_$c.log 2>&1
    echo "== $c" >> $out/meanfield_c1_hbm.txt
    python3 tools/rocpd_stats.py $out/pmc_mf_$c/p_results.db $c | grep -E "accum|kernel  " | cut -c1-44,65-200 >> $out/meanfield_c1_hbm.txt
    rm -rf $out/pmc_mf_$c
  done
  cat $out/meanfield_c1_hbm.txt
fi
