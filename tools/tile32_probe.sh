#!/bin/bash
# dev tool (GPU box): 64 x 32 tiles (cfg 7: three LDS stages, cfg 8: two) against the 64 x 64 tiles the launcher picks
# (cfg 3 / 4) on the D <= 512 products -- triangular sampling product and the dense product of the same shape.
cd $GRAFT_REPO_ROOT
B=tools/gemm_bench.bin
timeout 120 $B 512 256 256 2>&1 | grep "^check"
for SHAPE in "4096 256 256" "16384 256 256" "4096 512 512" "4096 1024 1024"; do
  for MODE in t d; do
    for CFG in 3 4 7 8; do GEMM_REPS=2000 timeout 120 $B $SHAPE $CFG r $MODE 2>&1 | grep "^cfg"; done
  done
done
