#!/bin/bash
# dev tool (GPU box): static wave priorities by k range in the triangular product Z = E L' (64 x 64 two-stage tiles): the light
# tiles first (they finish early instead of starving behind older heavy waves), or the heavy tiles first.
cd $GRAFT_REPO_ROOT
B=tools/gemm_bench_clk.bin
for p in 0 16 32 48 -32 -48; do
  echo "== GEMM_PRIO_SLABS=$p"; GEMM_PRIO_SLABS=$p GEMM_REPS=3000 timeout 120 $B 4096 1024 1024 4 r t 2>&1 | grep "cfg 4\|bn 15\|bn  8\|bn  7\|bn  4\|bn  0"
done
