#!/bin/bash
# dev tool (GPU box), VERDICT r4 item 8: what bounds the D = 512 / 256 triangular products (C2's sampling product
# Z = E L', 4096 x 512 x 512: 30 us, 0.45 of the fp64 MFMA roof)?  The whole triangle in every tile configuration, then
# with cfg 4 (64 x 64) only the heaviest column block / the heaviest half (the other tiles leave at once), then the dense
# product of the same shape for the launch's fixed cost.
cd $GRAFT_REPO_ROOT
B=tools/gemm_bench_clk.bin
run() { echo "== $*"; env "$@" GEMM_REPS=3000 timeout 120 $B $SHAPE $CFG r $MODE 2>&1 | grep -v "^empty"; }
for SHAPE in "4096 512 512" "4096 256 256" "16384 256 256"; do
  echo "#### M N K = $SHAPE"
  MODE=t
  for CFG in 1 2 3 4 5 6; do run GEMM_BN_MIN=0; done
  CFG=4 run GEMM_BN_MIN=7
  CFG=4 run GEMM_BN_MIN=4
  CFG=4 run GEMM_BN_MAX=3
  MODE=d
  for CFG in 2 4; do run GEMM_BN_MIN=0; done
done
