// Microbenchmark: empirical fp64 ceilings of the device: VALU v_fma_f64 vs v_mfma_f64_16x16x4_f64 vs
// v_mfma_f64_4x4x4_4b_f64 (no memory traffic).  Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) valu_loop(double* out, int iters, double a, double b) {
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x * 1e-9 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) mfma16_loop(double* out, int iters) {
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) mfma4_loop(double* out, int iters) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  double* out; hipMalloc(&out, cus * 8 * 256 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int wg = 1; wg <= 4; wg *= 2) {
    int grid = cus * wg, iters = 4000;
    valu_loop<<<grid, 256>>>(out, 10, 1.0000001, 1e-9); hipDeviceSynchronize();
    hipEventRecord(e0); valu_loop<<<grid, 256>>>(out, iters, 1.0000001, 1e-9); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    printf("v_fma_f64       wg/cu=%d: %.3f ms  %.2f TFLOP/s\n", wg, ms, 2.0 * 32 * iters * 256.0 * grid / ms / 1e9);
    iters = 2000;
    mfma16_loop<<<grid, 256>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); mfma16_loop<<<grid, 256>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    printf("mfma_f64_16x16x4 wg/cu=%d: %.3f ms  %.2f TFLOP/s\n", wg, ms, 2048.0 * 16 * iters * 4.0 * grid / ms / 1e9);
    mfma4_loop<<<grid, 256>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); mfma4_loop<<<grid, 256>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    printf("mfma_f64_4x4x4_4b wg/cu=%d: %.3f ms  %.2f TFLOP/s\n", wg, ms, 512.0 * 16 * iters * 4.0 * grid / ms / 1e9);
  }
  return 0;
}
