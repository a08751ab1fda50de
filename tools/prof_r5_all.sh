#!/bin/bash
# usage: tools/prof_r5_all.sh   (runs on the GPU box): the round-5 rocprofv3 kernel traces behind DESIGN.md's numbers, one
# summary per program under gpurun_out/prof_r5/ (tools/prof_r5.sh does one), plus the untraced timings of the same programs.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r5
tools/prof_r5.sh headline bench.py --no-legs --no-cpu-baseline --no-profile > /dev/null 2>&1
tools/prof_r5.sh legacy_t tools/legacy_t_bench.py 4096 1024 20 > /dev/null 2>&1
tools/prof_r5.sh legacy_dev tools/legacy_dev_bench.py 4096 1024 20 > /dev/null 2>&1
tools/prof_r5.sh c3_parity tools/c3_parity_profile.py > /dev/null 2>&1
tools/prof_r5.sh c3 tools/c3_bench.py > /dev/null 2>&1
tools/prof_r5.sh one_launch tools/one_launch_bench.py > /dev/null 2>&1
python3 tools/one_launch_bench.py > gpurun_out/prof_r5/one_launch_untraced.txt 2>&1
python3 tools/legacy_t_bench.py 4096 1024 20 > gpurun_out/prof_r5/legacy_t_untraced.txt 2>&1
python3 tools/c3_bench.py 16384 gauss numpy > gpurun_out/prof_r5/c3_parity_untraced.txt 2>&1
ls -la gpurun_out/prof_r5/
