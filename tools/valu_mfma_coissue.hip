// Probe: do v_fma_f64 waves co-resident with v_mfma_f64_4x4x4 waves raise the chip's fp64
// throughput above the matrix pipe's 78.6 TFLOP/s?  (MI355X_MICROARCH.md: the MFMA and VALU pipes of a SIMD issue
// independently.)  One kernel, the role is a wave-uniform argument: wave w of a 512-thread workgroup (two waves per SIMD)
// runs the MFMA loop if bit w of `mask` is set, else the VALU loop.  Long runs (seconds) so that the clock has settled;
// every run reports wall time, the flops of each role and the shader clock it ran at (clock64 / wall_clock64).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_mfma_coissue.hip -o /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(512) mix(double* out, long long* clk, int iters_mfma, int iters_valu, unsigned mask) {
  const int wave = threadIdx.x >> 6;
  const long long c0 = clock64(), w0 = wall_clock64();
  double s = 0.0;
  if ((mask >> wave) & 1u) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    const double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i];
  } else {
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters_valu; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = fma(acc[i], 1.0000001, 1e-9);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    clk[2 * ((size_t)blockIdx.x * 8 + wave)] = clock64() - c0;
    clk[2 * ((size_t)blockIdx.x * 8 + wave) + 1] = wall_clock64() - w0;
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, grid = cus;      // one 8-wave workgroup per CU: two waves per SIMD
  double* out;
  long long* clk;
  hipMalloc(&out, (size_t)grid * 512 * sizeof(double));
  hipMalloc(&clk, (size_t)grid * 16 * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  struct Cfg { const char* name; unsigned mask; int im, iv; } cfgs[] = {
      {"MFMA on all 8 waves", 0xffu, 20000, 0},
      {"MFMA on 4 waves (one per SIMD), 4 idle", 0x0fu, 40000, 0},
      {"VALU fma on all 8 waves", 0x00u, 0, 40000},
      {"MFMA on 4 waves + VALU fma on 4 waves (one of each per SIMD)", 0x0fu, 40000, 40000},
      {"MFMA on 4 + VALU on 4, VALU at a quarter of the work", 0x0fu, 40000, 10000},
      {"MFMA on 6 waves + VALU on 2", 0x3fu, 27000, 40000},
  };
  for (const Cfg& c : cfgs) {
    const bool idle = c.iv == 0 && c.mask != 0xffu;
    for (int warm = 0; warm < 2; ++warm) {
      hipEventRecord(e0);
      for (int r = 0; r < (warm ? reps : 3); ++r)
        mix<<<grid, 512>>>(out, clk, c.im, idle ? 0 : c.iv, c.mask);
      hipEventRecord(e1);
      hipDeviceSynchronize();
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int n_m = __builtin_popcount(c.mask & 0xffu), n_v = 8 - n_m;
    if (idle) n_v = 0;
    const double f_m = 512.0 * 16 * c.im * n_m * grid * reps, f_v = 2.0 * 32 * 64 * (double)c.iv * n_v * grid * reps;
    long long h[16];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    printf("%-62s %8.2f ms  MFMA %6.2f + VALU %6.2f = %6.2f TFLOP/s  (wave 0: %.0f MHz, wave 7: %.0f MHz; lifetimes %.2f / %.2f ms)\n",
           c.name, ms, f_m / ms / 1e9, f_v / ms / 1e9, (f_m + f_v) / ms / 1e9, h[0] / (h[1] / 100.0), h[14] / (h[15] / 100.0),
           h[1] / 1e5, h[15] / 1e5);
  }
  return 0;
}
