#!/bin/bash
# usage: tools/r6_c3_resample_prof.sh   (GPU box): kernel stats + timeline of the C3 call with multinomial resampling
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/c3_resample
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/run -o t -- python3 tools/c3_bench_r6.py philox resample > $out/run.log 2>&1 < /dev/null
python3 tools/rocpd_stats.py $out/run/t_results.db | cut -c1-150 | head -30 > $out/kernel_stats.txt
python3 tools/rocpd_stats.py $out/run/t_results.db --timeline 600 28 | cut -c1-150 > $out/timeline.txt
rm -rf $out/run
cat $out/kernel_stats.txt $out/timeline.txt
