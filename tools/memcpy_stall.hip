// Dev tool: does a pageable hipMemcpyAsync + hipStreamSynchronize stall on this stack?  (hipcc tools/memcpy_stall.hip -o tools/memcpy_stall.bin)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void touch(double* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1.0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? atol(argv[1]) : 131072;
  const int reps = 400;
  double* dev; hipMalloc(&dev, bytes);
  hipMemset(dev, 0, bytes);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  double* pageable = (double*)malloc(bytes);
  double* pinned; hipHostMalloc((void**)&pinned, bytes, hipHostMallocMapped);
  for (int mode = 0; mode < 4; ++mode) {
    double* host = (mode & 1) ? pinned : pageable;
    const bool h2d = mode >= 2;
    std::vector<double> t(reps);
    for (int r = 0; r < reps; ++r) {
      const double t0 = now();
      hipLaunchKernelGGL(touch, dim3((unsigned)(bytes / 8 / 256)), dim3(256), 0, st, dev, (int)(bytes / 8));
      if (h2d) hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st);
      else hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
      t[r] = now() - t0;
    }
    double sum = 0, mx = 0; int slow = 0;
    for (int r = 10; r < reps; ++r) { sum += t[r]; if (t[r] > mx) mx = t[r]; if (t[r] > 2e-3) ++slow; }
    printf("%s %s %zu B: mean %.1f us, max %.2f ms, %d of %d calls > 2 ms\n", h2d ? "H2D" : "D2H", (mode & 1) ? "pinned  " : "pageable", bytes,
           1e6 * sum / (reps - 10), 1e3 * mx, slow, reps - 10);
  }
  return 0;
}
