// Per-element arithmetic of the flat optimizers, shared by the stand-alone optimizer kernels
// (optim.hip) and by the update-log kernel, which applies the optimizer while it gathers the
// reduced gradient (csrc/comm/update_log.hip): the same operations in the same order, so a
// parameter updated by either path has the same bits.
#pragma once
#include "common.h"

namespace dtf {

// tf.train.AdamOptimizer with the bias correction folded into lr_t; g arrives unscaled.
__device__ __forceinline__ void adam_update4(floatx4& p, floatx4& m, floatx4& v, floatx4 g, float gs, float lr_t, float b1,
                                             float b2, float eps) {
  g = g * gs;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    m[e] = b1 * m[e] + (1.f - b1) * g[e];
    v[e] = b2 * v[e] + (1.f - b2) * g[e] * g[e];
    p[e] -= lr_t * m[e] / (sqrtf(v[e]) + eps);
  }
}

// momentum SGD with L2 weight decay
__device__ __forceinline__ void momentum_update4(floatx4& p, floatx4& m, floatx4 g, float gs, float lr, float mu, float wd) {
  g = g * gs + p * wd;
  m = m * mu + g;
  p -= m * lr;
}

}  // namespace dtf
