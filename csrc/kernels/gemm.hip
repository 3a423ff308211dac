// fp32 GEMM on the exact-fp32 matrix cores (fully connected layers of LeNet).
//
// C[M,N] = op(A)[M,K] * op(B)[K,N] (+ bias[N]) (ReLU). Two kernels: a 128x128 block tile for
// large aligned problems (below) and the general one: 64x64 block tile, 32-deep k-step, four
// waves each owning a 32x32 quadrant as 2x2 MFMA 16x16x4 tiles. Both operand tiles are stored
// k-contiguous in LDS ([row][k] and [col][k], rows padded by 4 floats) so that one
// ds_read_b128 per operand feeds four MFMAs, using the same k-permutation as the conv kernels.
// The global loads of k-step i+1 are issued into registers before the MFMAs of k-step i (with
// batch-sized M a block walks few, latency-bound k-steps, so nothing else hides the latency).
// Small-M problems (batch-sized M) are split along K across blocks to fill the 256 CUs; the
// partial products are combined with fp32 atomics and bias/ReLU run as a second tiny pass.
// (reference ops: tf.matmul in DistributedLeNet/src/mnist.py:136,145)
#include "ops.h"

namespace dtf {
namespace {

constexpr int BM = 64, BN = 64, BK = 32, LDT = BK + 4;
constexpr int EPT = BM * BK / 256;   // tile elements staged per thread and operand

struct GemmKArgs {
  GemmArgs g;
  int k_per_split;
  int atomic_out;
  AbortRef abort;
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmKArgs ka) {
  DTF_ABORT_GUARD(ka);
  const GemmArgs& a = ka.g;
  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, l15 = lane & 15;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int k_begin = blockIdx.z * ka.k_per_split;
  int k_end = k_begin + ka.k_per_split;
  if (k_end > a.K) k_end = a.K;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

  floatx4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0, 0, 0, 0};

  float ra[EPT], rb[EPT];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
      const int e = tid + it * 256;
      {
        int r, kk;
        if (a.trans_a) { r = e & 63; kk = e >> 6; } else { kk = e & (BK - 1); r = e / BK; }
        const int m = m0 + r, k = k0 + kk;
        float v = 0.f;
        if (m < a.M && k < k_end) v = a.trans_a ? a.A[(size_t)k * a.lda + m] : a.A[(size_t)m * a.lda + k];
        ra[it] = v;
      }
      {
        int c, kk;
        if (a.trans_b) { kk = e & (BK - 1); c = e / BK; } else { c = e & 63; kk = e >> 6; }
        const int n = n0 + c, k = k0 + kk;
        float v = 0.f;
        if (n < a.N && k < k_end) v = a.trans_b ? a.B[(size_t)n * a.ldb + k] : a.B[(size_t)k * a.ldb + n];
        rb[it] = v;
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
      const int e = tid + it * 256;
      int r, kk, c, kk2;
      if (a.trans_a) { r = e & 63; kk = e >> 6; } else { kk = e & (BK - 1); r = e / BK; }
      if (a.trans_b) { kk2 = e & (BK - 1); c = e / BK; } else { c = e & 63; kk2 = e >> 6; }
      As[r * LDT + kk] = ra[it];
      Bs[c * LDT + kk2] = rb[it];
    }
  };

  if (k_begin < k_end) fetch(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    commit();
    __syncthreads();
    if (k0 + BK < k_end) fetch(k0 + BK);   // in flight during the MFMAs
#pragma unroll
    for (int kg = 0; kg < BK / 16; ++kg) {
      floatx4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const floatx4*>(&As[(wm + i * 16 + l15) * LDT + kg * 16 + 4 * q]);
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const floatx4*>(&Bs[(wn + j * 16 + l15) * LDT + kg * 16 + 4 * q]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = DTF_MFMA16(av[i][s], bv[j][s], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + 4 * q + r;
        const int n = n0 + wn + j * 16 + l15;
        if (m >= a.M || n >= a.N) continue;
        float v = acc[i][j][r];
        float* dst = a.C + (size_t)m * a.ldc + n;
        if (ka.atomic_out) {
          atomicAdd(dst, v);
        } else {
          if (a.bias) v += a.bias[n];
          if (a.accumulate) v += *dst;
          if (a.relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
}

// ---- 128x128 block tile for the large fully connected products (every dimension and leading
// dimension a multiple of 4). Four waves, each a 64x64 quadrant = 4x4 MFMA tiles (64 accumulator
// registers): 8 operand fragments feed 64 MFMAs per 16 k, twice the ratio of the 64x64 kernel.
// Global loads are 16 bytes wide along whichever dimension is contiguous in memory, and an
// operand is kept in LDS the way it is stored: k-contiguous operands ([row][k]) are read back
// with one ds_read_b128 per fragment, row-contiguous ones ([k][row], the transposed forms) with
// four conflict-free ds_read_b32 — no transposing scatter into LDS either way.
constexpr int TM = 128, TN = 128, TK = 32;
constexpr int LDK = TK + 4;    // [row][k] layout: row stride
constexpr int LDR = TM + 4;    // [k][row] layout: k stride

template <bool KC>   // KC: the operand is k-contiguous in memory (ld = row stride), else row-contiguous (ld = k stride)
struct Tile128 {
  static constexpr int FLOATS = KC ? TM * LDK : TK * LDR;
  // 1024 float4 per tile and k-step, 4 per thread
  __device__ static void fetch(const float* __restrict__ src, int ld, int row0, int rows, int k0, int k_end, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + it * 256;
      int row, k;
      if (KC) { k = (e & 7) * 4; row = e >> 3; } else { row = (e & 31) * 4; k = e >> 5; }
      const int gr = row0 + row, gk = k0 + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rows && gk < k_end) v = *reinterpret_cast<const float4*>(KC ? src + (size_t)gr * ld + gk : src + (size_t)gk * ld + gr);
      r[it] = v;
    }
  }
  __device__ static void commit(float* lds, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + it * 256;
      if (KC) *reinterpret_cast<float4*>(lds + (e >> 3) * LDK + (e & 7) * 4) = r[it];
      else *reinterpret_cast<float4*>(lds + (e >> 5) * LDR + (e & 31) * 4) = r[it];
    }
  }
  // values k = kg*16 + 4q + (0..3) of row `row`
  __device__ static floatx4 frag(const float* lds, int row, int kg, int q) {
    if (KC) return *reinterpret_cast<const floatx4*>(lds + row * LDK + kg * 16 + 4 * q);
    floatx4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = lds[(kg * 16 + 4 * q + s) * LDR + row];
    return v;
  }
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm128_kernel(GemmKArgs ka) {
  DTF_ABORT_GUARD(ka);
  const GemmArgs& a = ka.g;
  using TileA = Tile128<!TA>;   // A [M,K] row-major is k-contiguous; trans_a stores [K,M]
  using TileB = Tile128<TB>;    // B [K,N] row-major is n-contiguous; trans_b stores [N,K]
  __shared__ __attribute__((aligned(16))) float As[TileA::FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[TileB::FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, l15 = lane & 15;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int k_begin = blockIdx.z * ka.k_per_split;
  int k_end = k_begin + ka.k_per_split;
  if (k_end > a.K) k_end = a.K;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0, 0, 0, 0};

  float4 ra[4], rb[4];
  if (k_begin < k_end) {
    TileA::fetch(a.A, a.lda, m0, a.M, k_begin, k_end, tid, ra);
    TileB::fetch(a.B, a.ldb, n0, a.N, k_begin, k_end, tid, rb);
  }
  for (int k0 = k_begin; k0 < k_end; k0 += TK) {
    TileA::commit(As, tid, ra);
    TileB::commit(Bs, tid, rb);
    __syncthreads();
    if (k0 + TK < k_end) {   // in flight during the MFMAs
      TileA::fetch(a.A, a.lda, m0, a.M, k0 + TK, k_end, tid, ra);
      TileB::fetch(a.B, a.ldb, n0, a.N, k0 + TK, k_end, tid, rb);
    }
#pragma unroll
    for (int kg = 0; kg < TK / 16; ++kg) {
      floatx4 av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = TileA::frag(As, wm + i * 16 + l15, kg, q);
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = TileB::frag(Bs, wn + j * 16 + l15, kg, q);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = DTF_MFMA16(av[i][s], bv[j][s], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + 4 * q + r;
        const int n = n0 + wn + j * 16 + l15;
        if (m >= a.M || n >= a.N) continue;
        float v = acc[i][j][r];
        float* dst = a.C + (size_t)m * a.ldc + n;
        if (ka.atomic_out) {
          atomicAdd(dst, v);
        } else {
          if (a.bias) v += a.bias[n];
          if (a.accumulate) v += *dst;
          if (a.relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
}

struct BiasActArgs { float* C; const float* bias; int M, N, ldc, relu; AbortRef abort; };
__global__ __launch_bounds__(256) void bias_act_kernel(BiasActArgs a) {
  DTF_ABORT_GUARD(a);
  const long long total = (long long)a.M * a.N;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int n = (int)(i % a.N);
    const long long m = i / a.N;
    float v = a.C[m * a.ldc + n];
    if (a.bias) v += a.bias[n];
    if (a.relu) v = fmaxf(v, 0.f);
    a.C[m * a.ldc + n] = v;
  }
}

}  // namespace

void gemm_f32(const GemmArgs& a, hipStream_t stream) {
  // the large tile pays once there are enough of them to spread over the CUs with a modest k split
  // (measured on the LeNet products: batch 1024 gains 25-35 %, batch 256 loses to the split-k overhead)
  const int tiles128 = ceil_div(a.M, TM) * ceil_div(a.N, TN);
  const bool big = ((tiles128 >= 32 && a.K >= 1024) || (tiles128 >= 160 && a.K >= 4 * TK)) && ((a.M | a.N | a.K | a.lda | a.ldb) & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0;
  const int bm = big ? TM : BM, bn = big ? TN : BN;
  const int gx = ceil_div(a.N, bn), gy = ceil_div(a.M, bm);
  int splits = 1;
  const int k_steps = ceil_div(a.K, BK);
  while (gx * gy * splits < (big ? 200 : 256) && splits * 2 <= k_steps / 4 && splits < 32) splits *= 2;
  GemmKArgs ka;
  ka.g = a;
  ka.k_per_split = ceil_div(k_steps, splits) * BK;
  ka.atomic_out = splits > 1;
  auto kernel = !big ? gemm_kernel
                     : a.trans_a ? (a.trans_b ? gemm128_kernel<true, true> : gemm128_kernel<true, false>)
                                 : (a.trans_b ? gemm128_kernel<false, true> : gemm128_kernel<false, false>);
  if (splits > 1) {
    if (!a.accumulate) {
      if (a.ldc == a.N) {
        memset_async(a.C, 0, (size_t)a.M * a.N * sizeof(float), stream);
      } else {
        for (int m = 0; m < a.M; ++m) memset_async(a.C + (size_t)m * a.ldc, 0, (size_t)a.N * sizeof(float), stream);
      }
    }
    launch("gemm_f32_splitk", kernel, dim3(gx, gy, splits), dim3(256), 0, stream, ka);
    if (a.bias || a.relu) {
      BiasActArgs b{a.C, a.bias, a.M, a.N, a.ldc, a.relu};
      long long total = (long long)a.M * a.N;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      launch("bias_act", bias_act_kernel, dim3(blocks), dim3(256), 0, stream, b);
    }
  } else {
    launch("gemm_f32", kernel, dim3(gx, gy, 1), dim3(256), 0, stream, ka);
  }
}

}  // namespace dtf
