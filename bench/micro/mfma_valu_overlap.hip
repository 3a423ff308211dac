// Do vector-ALU instructions of one wave execute while another wave of the same SIMD keeps the
// matrix pipe busy? Two waves per SIMD (512-thread workgroups, one per CU): waves 0-3 run a pure
// MFMA loop, waves 4-7 a pure v_fma loop of adjustable length. If the pipes overlap, the combined
// time is max(t_mfma, t_valu); if a SIMD issues one or the other, it is their sum. Second part: the
// same question inside one wave (independent v_fma placed between consecutive MFMAs).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mixed(float* out, int mfma_iters, int valu_iters, float a0, float b0) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    floatx4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    floatx4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    if (s[0] == 12345.f) out[threadIdx.x] = s[1] + s[2] + s[3];
  } else {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], b0, a0);   // 16 independent v_fma_f32 chains
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.f) out[threadIdx.x] = s;
  }
}

// Same wave: K independent v_fma between consecutive (independent) MFMAs. One or two waves per SIMD.
template <int K>
__global__ __launch_bounds__(256) void interleaved(float* out, int iters, float a0, float b0) {
  floatx4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = __builtin_fmaf(v[k], b0, a0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  floatx4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += v[i];
  if (s[0] == 12345.f || t == 12345.f) out[threadIdx.x] = s[1] + s[2] + s[3] + t;
}

template <int K>
static void run_interleaved(float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      interleaved<K><<<256 * wps, 256>>>(out, 20000, 1.f, 1.0000001f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("same wave, %d v_fma after every MFMA, %d wave(s) per SIMD: %.3f ms\n", K, wps, best);
  }
}

static float run(float* out, int mi, int vi) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    mixed<<<256, 512>>>(out, mi, vi, 1.f, 1.0000001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float* out;
  hipMalloc(&out, 8192);
  const int mi = 20000;                 // 160000 MFMAs per wave = 5.12 M matrix cycles
  const float t_m = run(out, mi, 0);
  printf("MFMA waves alone: %.3f ms\n", t_m);
  for (int vi : {20000, 40000, 80000}) {   // 16 v_fma per iteration, 4 issue cycles each
    const float t_v = run(out, 0, vi);
    const float t_b = run(out, mi, vi);
    printf("v_fma x %d per wave: alone %.3f ms, with the MFMA waves %.3f ms (max %.3f, sum %.3f)\n", vi * 16, t_v, t_b,
           t_m > t_v ? t_m : t_v, t_m + t_v);
  }
  run_interleaved<0>(out);
  run_interleaved<2>(out);
  run_interleaved<4>(out);
  run_interleaved<6>(out);
  run_interleaved<8>(out);
  return 0;
}
