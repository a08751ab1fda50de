// What does a scalar load from the kernel-argument segment cost when it is waited for, (a) the first time a line
// is touched, (b) the same line again, (c) another line?  Kernel with a 2 KB by-value argument; s_memrealtime stamps.
// build: hipcc -O3 --offload-arch=gfx950 tools/kernarg_probe.hip -o tools/kernarg_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big {
  long long x[256];
};

#define KLOAD(dst, off)                                                                        \
  asm volatile("s_load_dwordx2 %0, %1, " #off "\n s_waitcnt lgkmcnt(0)" : "=s"(dst) : "s"(kp) \
               : "memory")

__global__ void probe(Big big, long long* out) {
  typedef const long long __attribute__((address_space(4))) * KP;
  KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  long long t[8], v0, v1, v2, v3, v4;
  t[0] = wall_clock64();
  KLOAD(v0, 0x100);
  t[1] = wall_clock64();
  KLOAD(v1, 0x100);
  t[2] = wall_clock64();
  KLOAD(v2, 0x108);
  t[3] = wall_clock64();
  KLOAD(v3, 0x400);
  t[4] = wall_clock64();
  KLOAD(v4, 0x400);
  t[5] = wall_clock64();
  if (threadIdx.x == 0) {
    for (int i = 0; i < 5; ++i) out[blockIdx.x * 8 + i] = t[i + 1] - t[i];
    out[blockIdx.x * 8 + 7] = v0 + v1 + v2 + v3 + v4;
  }
}

int main() {
  long long* d;
  hipMalloc(&d, 4096);
  Big b;
  for (int i = 0; i < 256; ++i) b.x[i] = i;
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(probe, dim3(4), dim3(64), 0, 0, b, d);
    hipDeviceSynchronize();
    long long h[32];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("launch %d: first touch %lld | same address again %lld | same line +8 %lld | other line %lld | that line again %lld   (10 ns ticks)\n",
           rep, h[0], h[1], h[2], h[3], h[4]);
  }
  return 0;
}
