#!/bin/bash
# Debug build of the library with the s_memrealtime stamps of mf_finalize_kernel compiled in (-DVB_FIN_CLOCK):
# tools/libviabel_hip_clk.so, selected at run time with VIABEL_AMD_LIB.  Also rebuilds the normal library.
set -e
cd "$(dirname "$0")/../viabel_amd/csrc"
make 2>&1 | grep -E "error|warning" || true
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DVB_FIN_CLOCK -c vb_meanfield.hip -o /tmp/vb_meanfield_clk.o
OBJS=$(ls *.o | grep -v vb_meanfield.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OBJS /tmp/vb_meanfield_clk.o -shared -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../../tools/libviabel_hip_clk.so
ls -la ../libviabel_hip.so ../../tools/libviabel_hip_clk.so
