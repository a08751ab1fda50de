#!/bin/bash
# dev tool: loop ablations (VB_ABL_*) of the LDS-DMA GEMM, one tile configuration per call: tools/run_abl.sh <cfg>
cd $GRAFT_REPO_ROOT
cfg=${1:-3}
for v in BASE NO_LGKM NO_READS NO_DMA NO_BARRIER; do
  B=tools/gb_abl_$v.bin
  [ -x $B ] || continue
  a=$(GEMM_REPS=2000 timeout 60 $B 4096 1024 1024 $cfg r d 2>&1 | grep "cfg $cfg" | sed 's/.*: //')
  b=$(GEMM_REPS=2000 timeout 60 $B 4096 512 512 $cfg r d 2>&1 | grep "cfg $cfg" | sed 's/.*: //')
  echo "cfg $cfg $v: dense 4096x1024x1024 $a | dense 4096x512x512 $b"
done
