#!/bin/bash
# dev tool (GPU box): what bounds the triangular sampling product Z = E L' (4096 x 1024 x 1024)?  The whole triangle with the
# product's tile (cfg 4: 64 x 64, two LDS stages), then only its heavy half / heavy quarter (the light tiles leave at
# once), then only the light half with 64 x 64 and with 128 x 64 tiles (cfg 2) -- VERDICT r4 item 4 "mixed tile shapes".
cd $GRAFT_REPO_ROOT
B=tools/gemm_bench_clk.bin
run() { echo "== $*"; env "$@" GEMM_REPS=3000 timeout 120 $B 4096 1024 1024 $CFG r t 2>&1 | grep -v "^empty"; }
CFG=4 run GEMM_BN_MIN=0
CFG=4 run GEMM_BN_MIN=8
CFG=4 run GEMM_BN_MIN=12
CFG=4 run GEMM_BN_MIN=14
CFG=4 run GEMM_BN_MAX=7
CFG=2 run GEMM_BN_MAX=7
CFG=2 run GEMM_BN_MIN=0
