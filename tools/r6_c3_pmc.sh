#!/bin/bash
# usage: tools/r6_c3_pmc.sh   (GPU box): HBM traffic per kernel of the C3 loop -- FETCH_SIZE and WRITE_SIZE in separate counter passes,
# no trace domains (MI355X_MICROARCH.md, HBM section; FETCH_SIZE is doubled for gfx950 by tools/rocpd_stats.py's reader of the note)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/c3_pmc
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $out/pmc -o c3_$c -- python3 tools/c3_bench_r6.py philox > $out/pmc_$c.log 2>&1 < /dev/null
  { echo "== $c (tools/c3_bench_r6.py philox)"; python3 tools/rocpd_stats.py $out/pmc/c3_${c}_results.db $c | cut -c1-60,65-200; } >> $out/c3_pmc.txt 2>&1
done
rm -rf $out/pmc
cat $out/c3_pmc.txt
