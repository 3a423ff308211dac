// Shared device/host helpers for the gfx950 (CDNA4) kernel suite.
//
// Conventions used by every kernel in this directory:
//   * activations are NHWC fp32, conv filters HWIO fp32 (TensorFlow-native layouts, so a
//     checkpoint tensor is the in-memory tensor);
//   * a wavefront is 64 lanes; blocks are 256 threads (4 waves, one per SIMD);
//   * GEMM-shaped work runs on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32:
//       A operand: lane l holds A[row = l & 15][k = l >> 4]
//       B operand: lane l holds B[k = l >> 4][col = l & 15]
//       C/D      : lane l, register r holds D[row = 4 * (l >> 4) + r][col = l & 15]
//   * per-channel BatchNorm statistics travel as kStatSlots partial blocks of 2*C doubles (sum,
//     then sum of squares), accumulated by the producing kernel's epilogue with fp64 atomics.
//     A workgroup adds into slot (blockIdx.x % kStatSlots): workgroups are dealt round-robin to
//     the 8 XCDs, so a slot's cache lines stay in one XCD's L2 and the atomics of different XCDs
//     never contend for a line (one shared block costs ~10 us per kernel in serialised
//     cross-die atomics). Consumers add the slots up.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define DTF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace dtf {

constexpr int kStatSlots = 8;
// Dynamic LDS a kernel may ask for: the CU's 160 KB less a line for the few statically allocated
// words (the abort guard's verdict).
constexpr size_t kLdsBudget = 160 * 1024 - 128;

// Total of statistic `idx` (0..2C-1) over the slots.
__device__ __forceinline__ double stat_total(const double* sums, int idx, int C) {
  double v[kStatSlots];
#pragma unroll
  for (int r = 0; r < kStatSlots; ++r) v[r] = sums[r * 2 * C + idx];
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}
// This workgroup's slot of a statistics block.
__device__ __forceinline__ double* stat_slot(double* sums, int C) {
  return sums + (size_t)(blockIdx.x & (kStatSlots - 1)) * 2 * C;
}

// Reference to one BatchNorm layer's batch statistics and affine parameters.
struct BnRef {
  const double* sums;   // [kStatSlots][2*C]: per-channel sum, then per-channel sum of squares
  const float* gamma;   // [C]
  const float* beta;    // [C]
  float inv_count;      // 1 / (B*H*W)
  float eps;
};

// scale/shift such that bn(x) = x*scale + shift, plus mean/rstd for the backward pass.
__device__ __forceinline__ void bn_coeffs(const BnRef& bn, int c, int C, float& scale, float& shift,
                                          float& mean_f, float& rstd_f) {
  double mean = stat_total(bn.sums, c, C) * (double)bn.inv_count;
  double var = stat_total(bn.sums, C + c, C) * (double)bn.inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  float rstd = rsqrtf((float)var + bn.eps);
  float g = bn.gamma[c];
  scale = g * rstd;
  shift = bn.beta[c] - (float)mean * scale;
  mean_f = (float)mean;
  rstd_f = rstd;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Statistics leave a workgroup as fp64 atomics, and fp64 addition is not associative: the last bits
// of a total would depend on the order in which the workgroups arrive. Each contribution is therefore
// rounded to a fixed grid first (multiples of 1 / scale): sums of grid points are exact in fp64 as
// long as they stay below 2^53 grid steps, so the total is the same whatever the order - and beyond
// that range the additions are merely rounded like any other, nothing overflows. Forward statistics
// (sums of activations and of their squares: up to ~1e9 over a batch) use steps of 2^-20, the
// backward sums (gradients: ~1e-3) steps of 2^-44.
constexpr double kStatGridFwd = 1048576.0;                  // 2^20
constexpr double kStatGridBwd = 17592186044416.0;           // 2^44
__device__ __forceinline__ double to_grid(double s, double scale) { return rint(s * scale) * (1.0 / scale); }

// Block-level flush of per-channel statistics held as per-lane partials in the swapped-operand
// accumulator layout (lane (q, l15), register r <-> channel 16*j + 4*q + r, pixel l15): reduce over
// the 16 pixel lanes, park one row per wave in `scratch` (WAVES * 2 * NT*16 floats that nobody
// reads any more), and let 2*NT*16 threads add the rows in wave order before the fp64 atomic into
// this workgroup's slot. Every step has a fixed order, so the block's contribution is bit-identical
// from run to run: with order-dependent float atomics here the batch statistics differed in their
// last bits between runs, enough to flip the ReLU mask of an activation within rounding of zero.
template <int NT, int WAVES>
__device__ __forceinline__ void block_stat_flush(float* scratch, const floatx4 (&st1)[NT], const floatx4 (&st2)[NT],
                                                 double* sums, int C, int c_first, double grid) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = lane >> 4, l15 = lane & 15;
  __syncthreads();   // scratch may alias tile buffers that other waves were still reading
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v1 = st1[j][r], v2 = st2[j][r];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        v1 += __shfl_xor(v1, off);
        v2 += __shfl_xor(v2, off);
      }
      if (l15 == 0) {
        scratch[(wave * 2 + 0) * NT * 16 + j * 16 + 4 * q + r] = v1;
        scratch[(wave * 2 + 1) * NT * 16 + j * 16 + 4 * q + r] = v2;
      }
    }
  }
  __syncthreads();
  if (tid < 2 * NT * 16) {
    const int which = tid / (NT * 16);
    const int c = tid - which * (NT * 16);
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += (double)scratch[(w * 2 + which) * NT * 16 + c];
    if (c_first + c < C) atomicAdd(&stat_slot(sums, C)[which * C + c_first + c], to_grid(s, grid));
  }
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace dtf

// Launch plumbing: every host-side launcher goes through dtf::launch so that a recording
// Program (csrc/runtime/program.h) can capture the launch instead of issuing it.
#include "../runtime/program.h"

namespace dtf {
// True when the replica's current update is already complete (see AbortRef). Both words live in
// memory that other GPUs write, so they are read past the caches. The verdict is taken ONCE per
// workgroup - thread 0 reads the two words, everybody branches on the copy in LDS - so a workgroup
// runs as a whole or not at all: were every thread to read for itself, a stop word rising while a
// workgroup starts could make some of its waves return and the others run on past barriers that no
// longer wait for the waves that left, over LDS they never filled. Between workgroups any mix of
// early and late readers is harmless: a tile is either computed or left at the zeros the step
// prologue wrote. Replicas recorded without an abort reference (null pointers, a kernel argument)
// skip the barrier as well.
__device__ __forceinline__ bool dtf_aborted(const AbortRef& r) {
  if (!r.stop) return false;
  __shared__ int s_aborted;
  if ((threadIdx.x | threadIdx.y | threadIdx.z) == 0) {
    const unsigned long long s = __hip_atomic_load(r.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long v = __hip_atomic_load(r.version, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_aborted = s > v;
  }
  __syncthreads();
  return s_aborted != 0;
}
}  // namespace dtf
#define DTF_ABORT_GUARD(args) do { if (dtf::dtf_aborted((args).abort)) return; } while (0)
