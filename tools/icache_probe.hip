// Is the instruction cache cold at the start of every kernel?  One wave runs a straight-line body of 1024 VALU
// instructions (8 KB of code) twice in a loop and stamps s_memrealtime (100 MHz) around each pass; the launch is
// repeated back to back.  If pass 0 of EVERY launch is much slower than pass 1, each dispatch starts with a cold
// I-cache and a latency-bound kernel pays (code bytes on its path / 64) x (fetch latency).
// build: hipcc -O3 --offload-arch=gfx950 tools/icache_probe.hip -o tools/icache_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(long long* out, int passes) {
  unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  for (int p = 0; p < passes; ++p) {
    long long t0 = wall_clock64();
    asm volatile(
        ".rept 128\n"
        "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n"
        "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0\n"
        ".endr\n"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x * 8 + p] = t1 - t0;
  }
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 0x12345) out[100] = 1;
}

__global__ void filler(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main() {
  long long* d;
  float* f;
  hipMalloc(&d, 4096);
  hipMalloc(&f, 64 << 20);
  hipMemset(d, 0, 4096);
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(filler, dim3(65536), dim3(256), 0, 0, f, 16 << 20);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 3);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("launch %d: 1024 instructions (4 KB of code)  pass 0 %lld  pass 1 %lld  pass 2 %lld   (10 ns ticks)\n", rep, h[0], h[1],
           h[2]);
  }
  // back to back without a filler in between
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 3);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("back-to-back: pass 0 %lld  pass 1 %lld  pass 2 %lld\n", h[0], h[1], h[2]);
  return 0;
}
