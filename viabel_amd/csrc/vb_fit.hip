// Device-resident stochastic-gradient step: the descent directions of the reference's optimisers
// (optimization.py: StochasticGradientOptimizer :51-145, RMSProp :147-197, Adam :260-326, Adagrad :398-433)
// applied to the parameter where the objective kernels left (value, grad) -- so a whole fit is a chain of
// {Philox noise -> objective -> step} launches on one stream with no host round trip (vb_fit in vb_api.hip).
//
// The arithmetic is written operation by operation in numpy's order and compiled without floating-point
// contraction, so that a device fit reproduces the host loop (numpy update on the same gradients) bit for
// bit: IEEE fp64 multiply / add / divide / sqrt are correctly rounded on both sides.
#include "vb_common.h"

namespace vb {

namespace {

#pragma clang fp contract(off)

__global__ void __launch_bounds__(256) fit_step_kernel(FitStep a) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) a.values[a.k] = a.out[0];
  if (i >= a.p) return;
  const double g = a.out[1 + i];
  double dir = g;
  if (a.kind == VB_OPT_RMSPROP) {
    // _avg_grad_sq starts as grad**2; then  *= beta;  += (1 - beta) * grad**2   (optimization.py:188-197)
    const double g2 = g * g;
    double v = a.first ? g2 : a.s1[i];
    v = v * a.beta1;
    v = v + a.one_minus_beta1 * g2;
    a.s1[i] = v;
    dir = g / sqrt(a.jitter + v);
  } else if (a.kind == VB_OPT_ADAGRAD) {
    const double v = (a.first ? 0.0 : a.s1[i]) + g * g;            // optimization.py:430-433
    a.s1[i] = v;
    dir = g / sqrt(a.jitter + v);
  } else if (a.kind == VB_OPT_ADAM) {
    double m, v;
    if (a.first) {
      // the reference aliases momentum = grad and scales it in place before the second moment is refreshed
      // (optimization.py:315-322): grad itself becomes beta1 grad before (1 - beta1) grad is added, and the
      // squared *momentum* enters v
      const double m1 = g * a.beta1;
      m = m1 + a.one_minus_beta1 * m1;
      v = (g * g) * a.beta2;
      v = v + a.one_minus_beta2 * (m * m);
    } else {
      m = a.s2[i] * a.beta1;
      m = m + a.one_minus_beta1 * g;
      v = a.s1[i] * a.beta2;
      v = v + a.one_minus_beta2 * (g * g);
    }
    a.s2[i] = m;
    a.s1[i] = v;
    dir = m / sqrt(a.jitter + v);
  }
  if (a.dirs) a.dirs[a.k * a.p + i] = dir;
  if (a.grads) a.grads[a.k * a.p + i] = g;
  const double t = a.theta[i] - a.lr * dir;                        // objective.update (objectives.py:57-59, optimization.py:97-98)
  a.theta[i] = t;
  if (a.hist && a.k >= a.hist_first) a.hist[(a.k - a.hist_first) * a.p + i] = t;
}

}  // namespace

int fit_step_enqueue(vb_ctx* ctx, const FitStep& a) {
  hipLaunchKernelGGL(fit_step_kernel, dim3((unsigned)((a.p + 255) / 256)), dim3(256), 0, ctx->stream, a);
  VB_HIP(ctx, hipGetLastError());
  return VB_OK;
}

}  // namespace vb
