// Gate measurements for the ResNet convolution kernels (one binary, all arms in one process):
//
//   A. exact-fp32 MFMA shapes under the convolution kernels' own operand traffic: 16x16x4 with a
//      2x1 and a 2x2 arrangement of tiles per wave, and 32x32x2 with one tile per wave, every operand
//      read from LDS with ds_read_b128 in the K-permuted order, double buffered as in
//      conv_fprop_band.hip; optionally with extra vector instructions per k-group.
//   B. do vector instructions of one wave overlap the matrix instructions of another wave on the
//      same SIMD? fp32 MFMA (16x16x4) and bf16 MFMA (16x16x32, 32x32x16) next to v_fma_f32 waves.
//   C. packed fp32 vector rate: v_pk_fma_f32 against v_fma_f32, alone and next to fp32 MFMA waves.
//   D. a grid-wide barrier inside one launch (flat counter and XCD-hierarchical) at one and two
//      workgroups per CU, against the boundary between two dependent launches.
//
// build: hipcc -O3 --offload-arch=gfx950 -o r3_gate r3_gate.hip ; run: ./r3_gate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <typename F>
static float time_ms(F&& f, int reps = 5) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0));
    f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return best;
}

// ---------------------------------------------------------------------------------------------------
// A. convolution-shaped operand traffic
// ---------------------------------------------------------------------------------------------------
// halo tile of an 8-row band: the image is 32 / 16 / 8 pixels wide at 16 / 32 / 64 channels
constexpr int tile_w(int c) { return c == 16 ? 32 : (c == 32 ? 16 : 8); }
constexpr int halo_pix(int c) { return 10 * (tile_w(c) + 2); }

// 16x16x4: MTW m-tiles x NT n-tiles per wave; per k-group (16 channels of one tap): MTW + NT
// ds_read_b128 feed 4 * MTW * NT MFMAs. EXTRA v_fma per k-group stand in for non-matrix work.
template <int C, int MTW, int NT, int EXTRA>
__global__ __launch_bounds__(256, 2) void loop16(const float* src, float* out, int iters) {
  constexpr int CPP = C + 4, CG = C / 16, KG = 9 * CG, TW = tile_w(C), LW = TW + 2, kPix = halo_pix(C), SEGS = 8 * TW / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + kPix * CPP;
  for (int i = threadIdx.x; i < kPix * CPP + KG * NT * 256; i += 256) smem[i] = src[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, q = lane >> 4, l15 = lane & 15;
  const int wave = threadIdx.x >> 6;
  floatx4 acc[MTW][NT];
  for (int i = 0; i < MTW; ++i)
    for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float v[EXTRA > 0 ? EXTRA : 1];
  for (int k = 0; k < (EXTRA > 0 ? EXTRA : 1); ++k) v[k] = src[k] + lane;
  const float c0 = src[1], c1 = src[2];
  for (int it = 0; it < iters; ++it) {
    const int mb = (((it + wave) * MTW) % SEGS);   // which 16-pixel segments of the band
    int aoff[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int p = ((mb + i) % SEGS) * 16 + l15;
      aoff[i] = ((p / TW) * LW + (p % TW)) * CPP + 4 * q;
    }
    floatx4 av[2][MTW], bv[2][NT];
    auto read_group = [&](int g, floatx4 (&ao)[MTW], floatx4 (&bo)[NT]) {
      const int tap = g / CG, cg = g % CG;
      const int toff = ((tap / 3) * LW + (tap % 3)) * CPP + cg * 16;
#pragma unroll
      for (int i = 0; i < MTW; ++i) ao[i] = *reinterpret_cast<const floatx4*>(s_in + aoff[i] + toff);
#pragma unroll
      for (int j = 0; j < NT; ++j) bo[j] = *reinterpret_cast<const floatx4*>(s_w + (g * NT + j) * 256 + lane * 4);
    };
    read_group(0, av[0], bv[0]);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      if (g + 1 < KG) read_group(g + 1, av[(g + 1) & 1], bv[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[g & 1][j][s], av[g & 1][i][s], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < EXTRA; ++k) v[k] = __builtin_fmaf(v[k], c0, c1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < MTW; ++i)
    for (int j = 0; j < NT; ++j) s += acc[i][j];
  float t = 0.f;
  for (int k = 0; k < (EXTRA > 0 ? EXTRA : 1); ++k) t += v[k];
  if (s[0] + t == 12345.f) out[threadIdx.x] = s[1] + s[2] + s[3];
}

// 32x32x2: one 32-pixel x 32-channel tile per wave; lane l holds pixel l & 31, channels
// 4 (l >> 5) .. +3 of an 8-channel k-group: one ds_read_b128 per operand feeds 4 MFMAs (k = {s, 4 + s}).
template <int C, int EXTRA>
__global__ __launch_bounds__(256, 2) void loop32(const float* src, float* out, int iters) {
  constexpr int CPP = C + 4, CG8 = C / 8, KG = 9 * CG8, TW = tile_w(C), LW = TW + 2, kPix = halo_pix(C), SEGS = 8 * TW / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + kPix * CPP;
  for (int i = threadIdx.x; i < kPix * CPP + KG * 256; i += 256) smem[i] = src[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
  const int wave = threadIdx.x >> 6;
  floatx16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float v[EXTRA > 0 ? EXTRA : 1];
  for (int k = 0; k < (EXTRA > 0 ? EXTRA : 1); ++k) v[k] = src[k] + lane;
  const float c0 = src[1], c1 = src[2];
  for (int it = 0; it < iters; ++it) {
    const int p = ((it + wave) % SEGS) * 32 + l31;
    const int aoff = ((p / TW) * LW + (p % TW)) * CPP + 4 * h;
    floatx4 av[2], bv[2];
    auto read_group = [&](int g, floatx4& ao, floatx4& bo) {
      const int tap = g / CG8, cg = g % CG8;
      ao = *reinterpret_cast<const floatx4*>(s_in + aoff + ((tap / 3) * LW + (tap % 3)) * CPP + cg * 8);
      bo = *reinterpret_cast<const floatx4*>(s_w + g * 256 + lane * 4);
    };
    read_group(0, av[0], bv[0]);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      if (g + 1 < KG) read_group(g + 1, av[(g + 1) & 1], bv[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[g & 1][s], av[g & 1][s], acc, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < EXTRA; ++k) v[k] = __builtin_fmaf(v[k], c0, c1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  float t = 0.f;
  for (int k = 0; k < (EXTRA > 0 ? EXTRA : 1); ++k) t += v[k];
  if (s + t == 12345.f) out[threadIdx.x] = s;
}

template <int C, int MTW, int NT, int EXTRA>
static void run16(const float* src, float* out, const char* label) {
  constexpr int KG = 9 * (C / 16);
  const size_t shmem = (size_t)(halo_pix(C) * (C + 4) + KG * NT * 256) * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop16<C, MTW, NT, EXTRA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const int iters = 4000;
  loop16<C, MTW, NT, EXTRA><<<512, 256, shmem>>>(src, out, 10);
  const float ms = time_ms([&] { loop16<C, MTW, NT, EXTRA><<<512, 256, shmem>>>(src, out, iters); });
  const double flop = 2.0 * 512 * 4 * (double)iters * KG * 4 * MTW * NT * 16 * 16 * 4;
  printf("A  %-44s %8.3f ms  %6.1f TFLOP/s  (%d reads, %d MFMA, %d v_fma per k-group)\n", label, ms, flop / ms * 1e-9,
         MTW + NT, 4 * MTW * NT, EXTRA);
}

template <int C, int EXTRA>
static void run32(const float* src, float* out, const char* label) {
  constexpr int KG = 9 * (C / 8);
  const size_t shmem = (size_t)(halo_pix(C) * (C + 4) + KG * 256) * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop32<C, EXTRA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const int iters = 4000;
  loop32<C, EXTRA><<<512, 256, shmem>>>(src, out, 10);
  const float ms = time_ms([&] { loop32<C, EXTRA><<<512, 256, shmem>>>(src, out, iters); });
  const double flop = 2.0 * 512 * 4 * (double)iters * KG * 4 * 32 * 32 * 2;
  printf("A  %-44s %8.3f ms  %6.1f TFLOP/s  (2 reads, 4 MFMA, %d v_fma per k-group)\n", label, ms, flop / ms * 1e-9, EXTRA);
}

// ---------------------------------------------------------------------------------------------------
// B / C. two waves per SIMD: waves 0-3 matrix only, waves 4-7 vector only
// ---------------------------------------------------------------------------------------------------
enum { M_F32_16 = 0, M_BF16_16 = 1, M_BF16_32 = 2 };
enum { V_FMA = 0, V_PK_FMA = 1 };

template <int MK, int VK>
__global__ __launch_bounds__(512) void mixed(float* out, int mfma_iters, int valu_iters, float a0, float b0) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (MK == M_F32_16) {
      floatx4 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      const float a = a0 + threadIdx.x, b = b0;
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
      floatx4 s = acc[0];
      for (int i = 1; i < 8; ++i) s += acc[i];
      if (s[0] == 12345.f) out[threadIdx.x] = s[1] + s[2] + s[3];
    } else if (MK == M_BF16_16) {
      floatx4 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + ((threadIdx.x * 7 + e) & 63)); b[e] = (short)(0xbf80 + ((threadIdx.x * 3 + e) & 63)); }
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
      floatx4 s = acc[0];
      for (int i = 1; i < 8; ++i) s += acc[i];
      if (s[0] == 12345.f) out[threadIdx.x] = s[1] + s[2] + s[3];
    } else {
      floatx16 acc[2];
      for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + ((threadIdx.x * 7 + e) & 63)); b[e] = (short)(0xbf80 + ((threadIdx.x * 3 + e) & 63)); }
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 1], 0, 0, 0);
      }
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
      if (s == 12345.f) out[threadIdx.x] = s;
    }
  } else {
    if (VK == V_FMA) {
      float v[16];
      for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x;
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], b0, a0);
      }
      float s = 0.f;
      for (int i = 0; i < 16; ++i) s += v[i];
      if (s == 12345.f) out[threadIdx.x] = s;
    } else {
      floatx2 v[16];
      for (int i = 0; i < 16; ++i) v[i] = floatx2{a0 + i + threadIdx.x, a0 - i};
      const floatx2 b2 = {b0, b0}, c2 = {a0, a0};
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b2), "v"(c2));
      }
      float s = 0.f;
      for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
      if (s == 12345.f) out[threadIdx.x] = s;
    }
  }
}

template <int MK, int VK>
static void run_mixed(float* out, const char* mname, const char* vname, int mi) {
  auto go = [&](int m, int v) { return time_ms([&] { mixed<MK, VK><<<256, 512>>>(out, m, v, 1.f, 1.0000001f); }, 3); };
  go(10, 10);
  const float t_m = go(mi, 0);
  for (int vi : {20000, 40000}) {
    const float t_v = go(0, vi);
    const float t_b = go(mi, vi);
    printf("BC %-26s alone %.3f ms | %-12s x %7d per wave alone %.3f ms | together %.3f ms (max %.3f, sum %.3f)\n", mname, t_m,
           vname, vi * 16, t_v, t_b, t_m > t_v ? t_m : t_v, t_m + t_v);
  }
}

// ---------------------------------------------------------------------------------------------------
// D. grid barrier inside a launch
// ---------------------------------------------------------------------------------------------------
struct BarrierState {
  unsigned flat[32];          // one counter, own 128-byte line
  unsigned xcd_cnt[8][32];    // per-group arrival counters
  unsigned top[32];
  unsigned gen[8][32];        // per-group release generation
  unsigned timeout[32];
};

__device__ __forceinline__ void spin_until(unsigned* word, unsigned target, unsigned* timeout) {
  int spins = 0;
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 22)) {   // never hang: report and go on
      __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

// every workgroup: release fence, arrive on one counter, poll it, acquire fence
__device__ __forceinline__ void barrier_flat(BarrierState* b, unsigned epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&b->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_until(&b->flat[0], epoch * gridDim.x, &b->timeout[0]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// groups of blockIdx % 8 (the blocks that usually share an XCD; correctness does not depend on it):
// arrive on the group's counter, the group's last arrival arrives on the top counter, the last of
// those raises every group's generation word; everybody polls its own group's word.
__device__ __forceinline__ void barrier_xcd(BarrierState* b, unsigned epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int g = blockIdx.x & 7;
    const unsigned per_group = gridDim.x >> 3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(&b->xcd_cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * per_group) {
      const unsigned t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1 == epoch * 8) {
        for (int x = 0; x < 8; ++x) __hip_atomic_store(&b->gen[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    spin_until(&b->gen[g][0], epoch, &b->timeout[0]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// KIND 0: no barrier (the work alone), 1: flat, 2: hierarchical. Between barriers every workgroup
// writes `work_floats` floats of its own slab and reads a neighbour's slab of the phase before, so
// the fences have something to write back / invalidate.
template <int KIND>
__global__ __launch_bounds__(256, 2) void barrier_loop(BarrierState* b, float* slab, int phases, int work_floats, unsigned epoch0) {
  float* mine = slab + (size_t)blockIdx.x * work_floats;
  const float* theirs = slab + (size_t)((blockIdx.x + 9) % gridDim.x) * work_floats;
  float acc = 0.f;
  for (int p = 0; p < phases; ++p) {
    for (int i = threadIdx.x; i < work_floats; i += 256) mine[i] = theirs[i] + 1.f + acc;
    if (KIND == 1) barrier_flat(b, epoch0 + p + 1);
    if (KIND == 2) barrier_xcd(b, epoch0 + p + 1);
    acc += 1e-9f;
  }
}

__global__ __launch_bounds__(256, 2) void phase_kernel(float* slab, int work_floats) {
  float* mine = slab + (size_t)blockIdx.x * work_floats;
  const float* theirs = slab + (size_t)((blockIdx.x + 9) % gridDim.x) * work_floats;
  for (int i = threadIdx.x; i < work_floats; i += 256) mine[i] = theirs[i] + 1.f;
}

static void run_barriers(int wgs, int work_floats) {
  BarrierState* b;
  float* slab;
  CHECK(hipMalloc(&b, sizeof(BarrierState)));
  CHECK(hipMalloc(&slab, (size_t)wgs * (work_floats > 0 ? work_floats : 1) * 4));
  CHECK(hipMemset(slab, 0, (size_t)wgs * (work_floats > 0 ? work_floats : 1) * 4));
  const int phases = 200;
  float t[3];
  for (int kind = 0; kind < 3; ++kind) {
    unsigned epoch = 0;
    CHECK(hipMemset(b, 0, sizeof(BarrierState)));
    auto go = [&] {
      if (kind == 0) barrier_loop<0><<<wgs, 256>>>(b, slab, phases, work_floats, epoch);
      if (kind == 1) barrier_loop<1><<<wgs, 256>>>(b, slab, phases, work_floats, epoch);
      if (kind == 2) barrier_loop<2><<<wgs, 256>>>(b, slab, phases, work_floats, epoch);
      epoch += phases;
    };
    go();
    CHECK(hipDeviceSynchronize());
    t[kind] = time_ms(go, 5);
  }
  BarrierState h;
  CHECK(hipMemcpy(&h, b, sizeof(h), hipMemcpyDeviceToHost));
  // the same phases as dependent launches on one stream, captured into a graph
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipGraph_t graph;
  hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int p = 0; p < phases; ++p) phase_kernel<<<wgs, 256, 0, s>>>(slab, work_floats);
  CHECK(hipStreamEndCapture(s, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CHECK(hipGraphLaunch(exec, s));
  CHECK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float t_graph = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0, s));
    CHECK(hipGraphLaunch(exec, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < t_graph) t_graph = ms;
  }
  printf("D  %4d workgroups, %6d B written per workgroup and phase: work alone %.2f us/phase | flat barrier +%.2f us | hierarchical barrier +%.2f us | launch per phase (graph) +%.2f us%s\n",
         wgs, work_floats * 4, t[0] * 1000 / phases, (t[1] - t[0]) * 1000 / phases, (t[2] - t[0]) * 1000 / phases,
         (t_graph - t[0]) * 1000 / phases, h.timeout[0] ? "  [A SPIN TIMED OUT]" : "");
  CHECK(hipGraphExecDestroy(exec));
  CHECK(hipGraphDestroy(graph));
  CHECK(hipStreamDestroy(s));
  CHECK(hipFree(b));
  CHECK(hipFree(slab));
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "ABCD";
  auto want = [&](char c) { for (const char* p = only; *p; ++p) if (*p == c) return true; return false; };
  float* out;
  float* src;
  CHECK(hipMalloc(&out, 1 << 16));
  CHECK(hipMalloc(&src, 4096 * 4));
  std::vector<float> h(4096);
  unsigned s = 12345u;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  CHECK(hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice));

  if (want('A')) {
    printf("A. fp32 MFMA shapes, operands from LDS (512 workgroups x 4 waves, 2 workgroups per CU)\n");
    run16<16, 2, 1, 0>(src, out, "16x16x4, C=16, 2x1 tiles per wave");
    run16<32, 2, 1, 0>(src, out, "16x16x4, C=32, 2x1 tiles per wave");
    run16<32, 2, 2, 0>(src, out, "16x16x4, C=32, 2x2 tiles per wave");
    run32<32, 0>(src, out, "32x32x2, C=32, one tile per wave");
    run16<64, 2, 1, 0>(src, out, "16x16x4, C=64, 2x1 tiles per wave");
    run16<64, 2, 2, 0>(src, out, "16x16x4, C=64, 2x2 tiles per wave");
    run32<64, 0>(src, out, "32x32x2, C=64, one tile per wave");
    run16<32, 2, 1, 4>(src, out, "16x16x4, C=32, 2x1 tiles, +4 v_fma");
    run16<32, 2, 2, 8>(src, out, "16x16x4, C=32, 2x2 tiles, +8 v_fma");
    run32<32, 4>(src, out, "32x32x2, C=32, one tile, +4 v_fma");
  }
  if (want('B')) {
    printf("B. matrix waves (0-3) next to v_fma_f32 waves (4-7) on the same SIMDs, 256 workgroups of 512\n");
    run_mixed<M_F32_16, V_FMA>(out, "v_mfma_f32_16x16x4_f32", "v_fma_f32", 20000);
    run_mixed<M_BF16_16, V_FMA>(out, "v_mfma_f32_16x16x32_bf16", "v_fma_f32", 40000);
    run_mixed<M_BF16_32, V_FMA>(out, "v_mfma_f32_32x32x16_bf16", "v_fma_f32", 20000);
  }
  if (want('C')) {
    printf("C. packed fp32: v_pk_fma_f32 (two lanes' worth per instruction) against v_fma_f32\n");
    run_mixed<M_F32_16, V_PK_FMA>(out, "v_mfma_f32_16x16x4_f32", "v_pk_fma_f32", 20000);
    run_mixed<M_BF16_16, V_PK_FMA>(out, "v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32", 40000);
  }
  if (want('D')) {
    printf("D. grid barrier inside a launch against a launch boundary (200 phases)\n");
    for (int wgs : {256, 512})
      for (int wf : {0, 1024, 16384}) run_barriers(wgs, wf);
  }
  return 0;
}
