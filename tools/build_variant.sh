#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra compiler flags>": builds tools/libviabel_hip_<name>.so from the library
# sources with extra -D flags (A/B experiments; select at run time with VIABEL_AMD_LIB)
set -e
name=$1; extra=$2
cd "$(dirname "$0")/../viabel_amd/csrc"
mkdir -p /tmp/vbvar_$name
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra -c $f -o /tmp/vbvar_$name/${f%.hip}.o &
done
/opt/rocm/bin/hipcc -x c++ -O3 -std=c++17 -fPIC -ffp-contract=off -pthread -Wall -c vb_legacy_rng.cpp -o /tmp/vbvar_$name/vb_legacy_rng.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/vbvar_$name/*.o -shared -L/opt/rocm/lib -lrccl -ldl -pthread -Wl,-rpath,/opt/rocm/lib -o ../../tools/libviabel_hip_$name.so
ls -la ../../tools/libviabel_hip_$name.so
