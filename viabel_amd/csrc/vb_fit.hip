// Device-resident stochastic-gradient step: the descent directions of the reference's optimisers
// (optimization.py: StochasticGradientOptimizer :51-145, RMSProp :147-197, Adam :260-326, Adagrad :398-433)
// applied to the parameter where the objective kernels left (value, grad) -- so a whole fit is a chain of
// {Philox noise -> objective -> step} launches on one stream with no host round trip (vb_fit in vb_api.hip).
//
// The arithmetic is written operation by operation in numpy's order and compiled without floating-point
// contraction, so that a device fit reproduces the host loop (numpy update on the same gradients) bit for
// bit: IEEE fp64 multiply / add / divide / sqrt are correctly rounded on both sides.
#include "vb_common.h"
#include "vb_fit.h"

namespace vb {

namespace {

#pragma clang fp contract(off)

__global__ void __launch_bounds__(256) fit_step_kernel(FitStep a) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) a.values[a.k] = a.out[0];
  if (i >= a.p) return;
  fit_step_apply(a, i, a.out[1 + i]);
}

}  // namespace

int fit_step_enqueue(vb_ctx* ctx, const FitStep& a) {
  hipLaunchKernelGGL(fit_step_kernel, dim3((unsigned)((a.p + 255) / 256)), dim3(256), 0, ctx->stream, a);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

}  // namespace vb
