#!/bin/bash
# dev tool: A/B of the bisection look-ahead depth (library variants built with tools/build_variant.sh lookN "-DVB_LOOK=.. -DVB_LOOK_PARTS=..")
cd $GRAFT_REPO_ROOT
for v in "" look7 look8; do
  if [ -n "$v" ]; then export VIABEL_AMD_LIB=$PWD/tools/libviabel_hip_$v.so; else unset VIABEL_AMD_LIB; fi
  echo "== ${v:-base (6 levels x 4 parts)}"
  for i in 1 2; do timeout 120 python tools/c3_bench.py 2>&1 | grep "resampling=False"; done
  timeout 300 python -m pytest tests/test_gpu_objectives.py tests/test_gpu_full_size.py -m gpu -x -q -k "dis or DIS or c3" 2>&1 | tail -1
done
