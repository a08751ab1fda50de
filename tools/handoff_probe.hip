// Dev probe: cost and correctness of handing a tile from one workgroup to another INSIDE a kernel on gfx950, where
// every XCD has its own L2 and plain stores of one XCD are not visible to another until its L2 is written back.
//   producer: tile stores with agent-scope relaxed atomics (global_store ... sc1: written through the L2),
//             s_waitcnt vmcnt(0), workgroup barrier, one lane publishes a generation number (agent-scope store);
//   consumer: one lane spins on the generation (agent-scope loads), workgroup barrier, tile loads with agent-scope
//             relaxed atomics (sc1: not served from a stale line of the local L2), add, plain store.
// No buffer_wbl2 / buffer_inv (whole-L2 operations) anywhere.  Consumer j takes the tile of producer (j + shift) % P so
// that the pair sits on different XCDs (workgroup b runs on XCD b % 8).  Every iteration uses fresh data and a fresh
// generation; any stale read shows up as a mismatch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/handoff_probe.hip -o tools/handoff_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kTile = 128 * 64;   // doubles per tile (64 KB)

__device__ __forceinline__ double tile_value(int gen, int wg, int e) { return (double)((gen * 131 + wg * 17 + e) % 9973); }   // small integers: every sum is exact

template <int MODE>   // 0: everybody stores its own tile (baseline), 1: producer / consumer hand-off
__global__ void __launch_bounds__(256) handoff_kernel(double* __restrict__ P, double* __restrict__ C, int* flags, int gen,
                                                       int n_prod, int shift, int spin_us) {
  const int b = blockIdx.x, t = threadIdx.x;
  // stand-in for the main loop: stagger the workgroups a little
  if (spin_us > 0) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin_us * 100 + (b % 7) * 20) __builtin_amdgcn_s_sleep(8);
  }
  if (MODE == 0) {
    double* dst = C + (int64_t)b * kTile;
    for (int e = t; e < kTile; e += 256) dst[e] = tile_value(gen, b, e);
    return;
  }
  if (b < n_prod) {
    double* dst = P + (int64_t)b * kTile;
    for (int e = t; e < kTile; e += 256) __hip_atomic_store(dst + e, tile_value(gen, b, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);   // the write-through stores have been acknowledged
    __syncthreads();
    if (t == 0) __hip_atomic_store(flags + b, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const int src_wg = (b - n_prod + shift) % n_prod;
    if (t == 0) {
      while (__hip_atomic_load(flags + src_wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const double* src = P + (int64_t)src_wg * kTile;
    double* dst = C + (int64_t)b * kTile;
    double v[kTile / 256];
#pragma unroll
    for (int i = 0; i < kTile / 256; ++i) v[i] = __hip_atomic_load(src + t + 256 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < kTile / 256; ++i) dst[t + 256 * i] = v[i] + tile_value(gen, b, t + 256 * i);
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, spin_us = argc > 2 ? atoi(argv[2]) : 20;
  const int n_prod = 256, n_wg = 512;
  double *P, *C;
  int* flags;
  hipMalloc(&P, (size_t)n_prod * kTile * 8);
  hipMalloc(&C, (size_t)n_wg * kTile * 8);
  hipMalloc(&flags, n_prod * sizeof(int));
  hipMemset(flags, 0, n_prod * sizeof(int));
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<double> h((size_t)n_wg * kTile);
  for (int shift : {0, 1, 3}) {
    long bad = 0;
    int gen = 1;
    for (int check = 0; check < 20; ++check) {     // 20 verified iterations spread over the run
      for (int it = 0; it < iters / 20; ++it, ++gen)
        hipLaunchKernelGGL(handoff_kernel<1>, dim3(n_wg), dim3(256), 0, st, P, C, flags, gen, n_prod, shift, 2);
      hipStreamSynchronize(st);
      hipMemcpy(h.data(), C, h.size() * 8, hipMemcpyDeviceToHost);
      const int g = gen - 1;
      for (int b = n_prod; b < n_wg; ++b) {
        const int src = (b - n_prod + shift) % n_prod;
        for (int e = 0; e < kTile; ++e) {
          const double want = (double)((g * 131 + src * 17 + e) % 9973) + (double)((g * 131 + b * 17 + e) % 9973);
          if (h[(size_t)b * kTile + e] != want) ++bad;
        }
      }
    }
    float ms0, ms1;
    hipEventRecord(e0, st);
    for (int it = 0; it < 500; ++it) hipLaunchKernelGGL(handoff_kernel<0>, dim3(n_wg), dim3(256), 0, st, P, C, flags, gen, n_prod, shift, spin_us);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    hipEventElapsedTime(&ms0, e0, e1);
    hipEventRecord(e0, st);
    for (int it = 0; it < 500; ++it, ++gen)
      hipLaunchKernelGGL(handoff_kernel<1>, dim3(n_wg), dim3(256), 0, st, P, C, flags, gen, n_prod, shift, spin_us);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    hipEventElapsedTime(&ms1, e0, e1);
    printf("shift %d: %ld mismatches in 20 checked launches of %d; per launch: plain stores %.2f us, hand-off %.2f us (stand-in loop %d us)\n",
           shift, bad, iters, ms0 * 2.0, ms1 * 2.0, spin_us);
  }
  return 0;
}
