// Sustained throughput of v_mfma_f32_16x16x4_f32 with every CU busy: the real compute roof of the
// fp32 convolution kernels (the engine clock under matrix load is below the 2.4 GHz boost clock).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  floatx4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  floatx4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  if (s[0] == 12345.f) out[threadIdx.x] = s[1] + s[2] + s[3];
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd;
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = 20000;
      hipEventRecord(e0);
      mfma_loop<<<blocks, 256>>>(out, iters, 1.f, 2.f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 * iters * 8 * 2048.0;
      const double tf = flops / (ms * 1e-3) / 1e12;
      // one MFMA = 32 issue cycles per SIMD
      const double ghz = (double)waves_per_simd * iters * 8 * 32 / (ms * 1e-3) / 1e9;
      printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s, implied matrix clock %.2f GHz\n", waves_per_simd, ms, tf, ghz);
    }
  }
  return 0;
}
