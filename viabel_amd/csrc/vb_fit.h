// One optimiser step for parameter entry i with gradient g (device function shared by fit_step_kernel and the
// mean-field finalize kernel, which applies it to the columns it has just finished).  Written operation by
// operation in numpy's order and compiled without floating-point contraction: see vb_fit.hip.
#pragma once

#include "vb_common.h"

namespace vb {

__device__ __forceinline__ void fit_step_apply(const FitStep& a, int64_t i, double g) {
#pragma clang fp contract(off)
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

}  // namespace vb
