// Dev probe: what does the host pay to learn that a short kernel has finished?
//   (a) launch + hipStreamSynchronize          (b) launch + spin on a word the kernel stores into mapped pinned memory
//   (c) three dependent launches + sync (the mean-field call's chain)      hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_empty() {}
__global__ void k_flag(volatile unsigned long long* flag, unsigned long long v) {
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store((unsigned long long*)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc > 1) printf("hipSetDeviceFlags(%d) -> %d\n", atoi(argv[1]), (int)hipSetDeviceFlags((unsigned)atoi(argv[1])));
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  unsigned long long* host;
  hipHostMalloc((void**)&host, 64, hipHostMallocMapped);
  unsigned long long* dev;
  hipHostGetDevicePointer((void**)&dev, host, 0);
  *host = 0;
  const int reps = 2000;
  for (int i = 0; i < 100; ++i) {
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    hipStreamSynchronize(st);
  }
  double t0 = now();
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    hipStreamSynchronize(st);
  }
  printf("launch + hipStreamSynchronize:           %.2f us\n", 1e6 * (now() - t0) / reps);
  t0 = now();
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dev, (unsigned long long)(i + 1));
    while (__atomic_load_n(host, __ATOMIC_ACQUIRE) != (unsigned long long)(i + 1)) {
    }
  }
  printf("launch + spin on a mapped pinned word:   %.2f us\n", 1e6 * (now() - t0) / reps);
  hipStreamSynchronize(st);
  t0 = now();
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(k_empty, dim3(16), dim3(256), 0, st);
    hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, st);
    hipLaunchKernelGGL(k_empty, dim3(16), dim3(256), 0, st);
    hipStreamSynchronize(st);
  }
  printf("three launches + hipStreamSynchronize:   %.2f us\n", 1e6 * (now() - t0) / reps);
  t0 = now();
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(k_empty, dim3(16), dim3(256), 0, st);
    hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, st);
    hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dev, (unsigned long long)(reps + i + 1));
    while (__atomic_load_n(host, __ATOMIC_ACQUIRE) != (unsigned long long)(reps + i + 1)) {
    }
  }
  printf("three launches + spin:                   %.2f us\n", 1e6 * (now() - t0) / reps);
  hipStreamSynchronize(st);
  return 0;
}
