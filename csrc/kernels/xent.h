// Softmax cross-entropy of one row of logits, shared by the classifier-head kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace dtf {

// One thread computes the loss terms of one row of logits held in registers/LDS.
__device__ __forceinline__ void xent_row(const float* z, int K, int label, float inv_B, float* dz, float& loss,
                                         int& hit) {
  float mx = z[0];
  int arg = 0;
  for (int k = 1; k < K; ++k)
    if (z[k] > mx) { mx = z[k]; arg = k; }
  float se = 0.f;
  for (int k = 0; k < K; ++k) se += __expf(z[k] - mx);
  const float lse = mx + __logf(se);
  loss = lse - z[label];
  hit = (arg == label) ? 1 : 0;
  if (dz) {
    const float inv = 1.f / se;
    for (int k = 0; k < K; ++k) {
      const float p = __expf(z[k] - mx) * inv;
      dz[k] = (p - (k == label ? 1.f : 0.f)) * inv_B;
    }
  }
}

}  // namespace dtf
