// What does a TAKEN branch over a large block of code cost the first time (cold instruction cache)?  One wave,
// s_memrealtime (100 MHz) stamps around `if (uniform_flag) { 16 KB of code }` with the flag false, four such skips in
// a row, repeated twice in a loop (second pass: targets already fetched).
// build: hipcc -O3 --offload-arch=gfx950 tools/branch_probe.hip -o tools/branch_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

// the branch is inside the asm so that the compiler cannot move the skipped block out of line
#define SKIP_BIG(x, bit)                                                                                         \
  asm volatile("s_bitcmp1_b32 %2, " #bit "\n s_cbranch_scc0 1f\n"                                               \
               ".rept 512\n v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, 1, %0\n" \
               ".endr\n 1:\n"                                                                                  \
               : "+v"(x)                                                                                       \
               : "v"(threadIdx.x), "s"(flag)                                                                   \
               : "scc")

__global__ void probe(long long* out, int flag, int passes) {
  unsigned x = threadIdx.x;
  for (int p = 0; p < passes; ++p) {
    long long t[6];
    t[0] = wall_clock64();
    SKIP_BIG(x, 0);
    t[1] = wall_clock64();
    SKIP_BIG(x, 1);
    t[2] = wall_clock64();
    SKIP_BIG(x, 2);
    t[3] = wall_clock64();
    SKIP_BIG(x, 3);
    t[4] = wall_clock64();
    if (threadIdx.x == 0)
      for (int i = 0; i < 4; ++i) out[p * 4 + i] = t[i + 1] - t[i];
  }
  if (x == 0x12345) out[100] = 1;
}

int main() {
  long long* d;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 0, 2);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("launch %d: skip over 16 KB x4: pass 0: %lld %lld %lld %lld   pass 1: %lld %lld %lld %lld  (10 ns ticks)\n", rep, h[0],
           h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  return 0;
}
