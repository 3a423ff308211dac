// Philox-4x32-10 counter-based generator, shared by the dropout kernel and the kernels that
// regenerate its mask in place: elements 4*i .. 4*i+3 of a tensor take the four words drawn for
// counter (offset + i) under the key `seed`.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace dtf {

__device__ __forceinline__ void philox_round(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  const uint32_t c1 = ctr[1], c3 = ctr[3];
  ctr[0] = hi1 ^ c1 ^ k0;
  ctr[1] = lo1;
  ctr[2] = hi0 ^ c3 ^ k1;
  ctr[3] = lo0;
}

// The four 32-bit words of counter `cidx`.
__device__ __forceinline__ void philox4x32_10(unsigned long long cidx, unsigned long long seed, uint32_t (&out)[4]) {
  out[0] = (uint32_t)cidx;
  out[1] = (uint32_t)(cidx >> 32);
  out[2] = 0u;
  out[3] = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(out, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// Uniform in [0, 1) from the top 24 bits of a word; an element is kept when it is below keep_prob.
__device__ __forceinline__ float philox_uniform(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

}  // namespace dtf
