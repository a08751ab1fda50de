// One optimiser step for parameter entry i with gradient g (device function shared by fit_step_kernel and the
// mean-field finalize kernel, which applies it to the columns it has just finished).  Written operation by
// operation in numpy's order and compiled without floating-point contraction: see vb_fit.hip.
#pragma once

#include "vb_common.h"

namespace vb {

// The arithmetic of one step on values already in registers: (s1_old, s2_old) = the optimiser state of entry i,
// theta_old = the parameter entry.  Returns the new parameter entry.  fit_step_apply loads the three and calls this;
// the finalize kernel prefetches them at its top so that they do not queue behind its reduction.
__device__ __forceinline__ double fit_step_apply_vals(const FitStep& a, int64_t i, double g, double s1_old, double s2_old,
                                                      double theta_old) {
#pragma clang fp contract(off)
  double dir = g;
  if (a.kind == VB_OPT_RMSPROP) {
    // _avg_grad_sq starts as grad**2; then  *= beta;  += (1 - beta) * grad**2   (optimization.py:188-197)
    const double g2 = g * g;
    double v = a.first ? g2 : s1_old;
    v = v * a.beta1;
    v = v + a.one_minus_beta1 * g2;
    a.s1[i] = v;
    dir = g / sqrt(a.jitter + v);
  } else if (a.kind == VB_OPT_ADAGRAD) {
    const double v = (a.first ? 0.0 : s1_old) + g * g;             // optimization.py:430-433
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
      m = s2_old * a.beta1;
      m = m + a.one_minus_beta1 * g;
      v = s1_old * a.beta2;
      v = v + a.one_minus_beta2 * (g * g);
    }
    a.s2[i] = m;
    a.s1[i] = v;
    dir = m / sqrt(a.jitter + v);
  }
  if (a.dirs) a.dirs[a.k * a.p + i] = dir;
  if (a.grads) a.grads[a.k * a.p + i] = g;
  const double t = theta_old - a.lr * dir;                         // objective.update (objectives.py:57-59, optimization.py:97-98)
  a.theta[i] = t;
  if (a.hist && a.k >= a.hist_first) a.hist[(a.k - a.hist_first) * a.p + i] = t;
  return t;
}

// the state entries a step reads (nothing for the first call of an optimiser, whose state does not exist yet)
__device__ __forceinline__ void fit_step_load(const FitStep& a, int64_t i, double* s1_old, double* s2_old,
                                              double* theta_old) {
  *s1_old = (!a.first && (a.kind == VB_OPT_RMSPROP || a.kind == VB_OPT_ADAGRAD || a.kind == VB_OPT_ADAM)) ? a.s1[i] : 0.0;
  *s2_old = (!a.first && a.kind == VB_OPT_ADAM) ? a.s2[i] : 0.0;
  *theta_old = a.theta[i];
}

__device__ __forceinline__ void fit_step_apply(const FitStep& a, int64_t i, double g) {
  double s1_old, s2_old, theta_old;
  fit_step_load(a, i, &s1_old, &s2_old, &theta_old);
  fit_step_apply_vals(a, i, g, s1_old, s2_old, theta_old);
}

}  // namespace vb
