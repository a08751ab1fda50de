#!/bin/bash
# Lists every kernel of the library whose code object uses scratch memory (register spills or stack).  The GEMM kernels
# must not: a spilling 128 x 128 epilogue once turned an 8 ms evaluation into 24 ms.  CPU-only (hipcc -S).
cd "$(dirname "$0")/../viabel_amd/csrc"
tmp=$(mktemp -d)
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I. --cuda-device-only -S -o $tmp/${f%.hip}.s $f 2>/dev/null &
done
wait
bad=0
for s in $tmp/*.s; do
  awk -v file=$(basename $s .s) '/^_Z[A-Za-z0-9_]*:/{name=$1} /; ScratchSize: [1-9]/{print file ": " substr(name, 1, 90) " " $0}' $s
done | tee $tmp/list.txt
if grep -q "gemm_f64" $tmp/list.txt; then echo "FAIL: a GEMM kernel spills"; bad=1; fi
rm -rf $tmp
exit $bad
