#!/bin/bash
# usage: tools/prof_r2.sh [pmc]  (runs on the GPU box): round-2 rocprofv3 kernel traces of the bench headline and the
# C2 / C3 / C4 shapes; with `pmc` also the matrix-pipe counters of the three full-rank GEMMs (separate passes).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r2
rm -rf $out; mkdir -p $out
run() {   # name, program args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/$name -o t -- python3 "$@" > $out/$name.log 2>&1
  python3 tools/rocpd_stats.py $out/$name/t_results.db > $out/${name}_kernel_stats.txt 2>&1
  grep -v "rocprofv3\]\|^W20\|^E20\|^I20" $out/$name.log | tail -4 | cut -c1-300 > $out/${name}_tail.txt
  rm -rf $out/$name
  cat $out/${name}_tail.txt; head -12 $out/${name}_kernel_stats.txt | cut -c1-170
}
run headline bench.py --no-legs --no-cpu-baseline --no-profile
run fr512 tools/fr_bench.py 512 4096 gauss_full 300
run c3 tools/c3_bench.py
run c4 tools/c4_bench.py
run fit tools/fit_bench.py
if [ "$1" = "pmc" ]; then
  for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
    rocprofv3 --pmc $c -d $out/pmc_$c -o p -- python3 tools/fr_bench.py 1024 4096 gauss_full 100 > $out/pmc_$c.log 2>&1
    echo "== $c" >> $out/fr1024_pmc.txt
    python3 tools/rocpd_stats.py $out/pmc_$c/p_results.db $c | grep -E "gemm|kernel  " | cut -c1-44,65-200 >> $out/fr1024_pmc.txt
    rm -rf $out/pmc_$c
  done
  cat $out/fr1024_pmc.txt
  # HBM traffic of the same kernels (separate passes): FETCH_SIZE / WRITE_SIZE are in KiB-sized units of the L2's
  # memory-side requests; on gfx950 FETCH_SIZE counts 128-B streaming requests at 64 B (MI355X_MICROARCH.md, HBM
  # section), so the fetched bytes are 2 x FETCH_SIZE x 1024
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/pmc_$c -o p -- python3 tools/fr_bench.py 1024 4096 gauss_full 100 > $out/pmc_$c.log 2>&1
    echo "== $c" >> $out/fr1024_hbm.txt
    python3 tools/rocpd_stats.py $out/pmc_$c/p_results.db $c | grep -E "gemm|reduce|kernel  " | cut -c1-44,65-200 >> $out/fr1024_hbm.txt
    rm -rf $out/pmc_$c
  done
  cat $out/fr1024_hbm.txt
fi
