// fp32 GEMM on the exact-fp32 matrix cores (fully connected layers of LeNet).
//
// C[M,N] = op(A)[M,K] * op(B)[K,N] (+ bias[N]) (ReLU). Vector-load tile kernels for aligned
// problems (further down) and the general one: 64x64 block tile, 32-deep k-step, four
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
  int gx, gy, total;   // tile grid of the vector-load kernels (launched one-dimensional, see tile_of)
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

// ---- Vector-load tiles for aligned problems (every dimension and leading dimension a multiple
// of 4). Four waves in a 2x2 arrangement, each a (TM/2)x(TN/2) quadrant of 16x16 MFMA tiles.
// Global loads are 16 bytes wide along whichever dimension is contiguous in memory, and an
// operand is kept in LDS the way it is stored: k-contiguous operands ([row][k]) are read back
// with one ds_read_b128 per fragment, row-contiguous ones ([k][row], the transposed forms) with
// four conflict-free ds_read_b32 - no transposing scatter into LDS either way.
//
// Shapes in use (the others are kept for bench/gemm_sweep.py):
//   64x64, 64 deep     batch-sized products, split along k. A workgroup of such a product walks a
//                      few hundred k: with the general kernel's 32-deep step and scalar loads that
//                      is a dozen round trips to memory with little in flight each.
//   128x128, 32 deep   large products: 8 operand fragments feed 64 MFMAs per 16 k.
constexpr int TM = 128, TN = 128, TK = 32;

template <int ROWS, int DEPTH, bool KC, int THREADS = 256>   // KC: the operand is k-contiguous in memory (ld = row stride), else row-contiguous (ld = k stride)
struct OperandTile {
  static constexpr int LDK = DEPTH + 4;   // [row][k] layout: row stride
  static constexpr int LDR = ROWS + 4;    // [k][row] layout: k stride
  static constexpr int FLOATS = KC ? ROWS * LDK : DEPTH * LDR;
  static constexpr int NV = ROWS * DEPTH / 4 / THREADS;   // float4 per thread and k-step
  static_assert(ROWS * DEPTH % (4 * THREADS) == 0, "a k-step of an operand is a whole number of float4 per thread");
  __device__ static void fetch(const float* __restrict__ src, int ld, int row0, int rows, int k0, int k_end, int tid, float4 (&r)[NV]) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int e = tid + it * THREADS;
      int row, k;
      if (KC) { k = (e % (DEPTH / 4)) * 4; row = e / (DEPTH / 4); } else { row = (e % (ROWS / 4)) * 4; k = e / (ROWS / 4); }
      const int gr = row0 + row, gk = k0 + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rows && gk < k_end) v = *reinterpret_cast<const float4*>(KC ? src + (size_t)gr * ld + gk : src + (size_t)gk * ld + gr);
      r[it] = v;
    }
  }
  __device__ static void commit(float* lds, int tid, const float4 (&r)[NV]) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int e = tid + it * THREADS;
      if (KC) *reinterpret_cast<float4*>(lds + (e / (DEPTH / 4)) * LDK + (e % (DEPTH / 4)) * 4) = r[it];
      else *reinterpret_cast<float4*>(lds + (e / (ROWS / 4)) * LDR + (e % (ROWS / 4)) * 4) = r[it];
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

// WK = 2: eight waves. Waves 4-7 cover the same four quadrants as waves 0-3 and take every other
// 16-deep k group of a step; the two partial tiles are added through LDS at the end. Same bytes
// from L2 per workgroup, twice the waves to issue MFMAs and to keep loads in flight - for grids
// that put one workgroup on a CU.
template <int TM_, int TN_, int TK_, bool TA, bool TB, int WK = 1>
__global__ __launch_bounds__(256 * WK) void gemm_tile_kernel(GemmKArgs ka) {
  DTF_ABORT_GUARD(ka);
  const GemmArgs& a = ka.g;
  using TileA = OperandTile<TM_, TK_, !TA, 256 * WK>;   // A [M,K] row-major is k-contiguous; trans_a stores [K,M]
  using TileB = OperandTile<TN_, TK_, TB, 256 * WK>;    // B [K,N] row-major is n-contiguous; trans_b stores [N,K]
  constexpr int MI = TM_ / 32, NJ = TN_ / 32;  // MFMA tiles per wave
  constexpr int kOperandFloats = TileA::FLOATS + TileB::FLOATS;
  constexpr int kFoldFloats = WK > 1 ? MI * NJ * 4 * 256 : 0;
  __shared__ __attribute__((aligned(16))) float smem[kOperandFloats > kFoldFloats ? kOperandFloats : kFoldFloats];
  float* As = smem;
  float* Bs = smem + TileA::FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3, kh = wave8 >> 2;
  const int q = lane >> 4, l15 = lane & 15;
  // Workgroups go to the eight XCDs round-robin by their linear id, and every XCD has its own L2.
  // Work item t = (id % 8) * (items per XCD) + id / 8 gives each XCD one contiguous range of
  // (split, row tile, column tile) with the column tile running fastest: the workgroups an XCD runs
  // at a time share their A rows eight ways and their B columns across the row tiles, and over the
  // whole kernel an XCD's L2 fetches about an eighth of the operands instead of nearly all of them.
  const int per_xcd = (ka.total + 7) >> 3;
  const int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (t >= ka.total) return;
  const int tx = t % ka.gx, ty = (t / ka.gx) % ka.gy, tz = t / (ka.gx * ka.gy);
  const int m0 = ty * TM_, n0 = tx * TN_;
  const int k_begin = tz * ka.k_per_split;
  int k_end = k_begin + ka.k_per_split;
  if (k_end > a.K) k_end = a.K;
  const int wm = (wave >> 1) * (TM_ / 2), wn = (wave & 1) * (TN_ / 2);

  floatx4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = floatx4{0, 0, 0, 0};

  float4 ra[TileA::NV], rb[TileB::NV];
  if (k_begin < k_end) {
    TileA::fetch(a.A, a.lda, m0, a.M, k_begin, k_end, tid, ra);
    TileB::fetch(a.B, a.ldb, n0, a.N, k_begin, k_end, tid, rb);
  }
  for (int k0 = k_begin; k0 < k_end; k0 += TK_) {
    TileA::commit(As, tid, ra);
    TileB::commit(Bs, tid, rb);
    __syncthreads();
    if (k0 + TK_ < k_end) {   // in flight during the MFMAs
      TileA::fetch(a.A, a.lda, m0, a.M, k0 + TK_, k_end, tid, ra);
      TileB::fetch(a.B, a.ldb, n0, a.N, k0 + TK_, k_end, tid, rb);
    }
    const int k_left = k_end - k0;
#pragma unroll
    for (int kg = 0; kg < TK_ / 16; ++kg) {
      if (kg * 16 >= k_left) break;   // the last step of a split may be short: its empty groups hold zeros
      if (WK > 1 && (kg % WK) != kh) continue;
      floatx4 av[MI], bv[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) av[i] = TileA::frag(As, wm + i * 16 + l15, kg, q);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[j] = TileB::frag(Bs, wn + j * 16 + l15, kg, q);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = DTF_MFMA16(av[i][s], bv[j][s], acc[i][j]);
    }
    __syncthreads();
  }

  if (WK > 1) {
    // the operand tiles are dead (barrier above): the upper waves park their partial tile there
    floatx4* fold = reinterpret_cast<floatx4*>(smem);
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) fold[((wave * MI + i) * NJ + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] += fold[((wave * MI + i) * NJ + j) * 64 + lane];
  }

#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
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

// 128x128 tile, 32-deep steps, eight waves (2 x 4, each 64 x 32 of the tile as 4 x 2 MFMA tiles), two LDS
// stages: the LDS stores of step s + 1 and the global loads of step s + 2 are issued under the MFMAs of step
// s, one barrier per step. A 64x64 tile pulls 32 KB through L2 per 0.5 MFLOP - at batch 1024 the fully
// connected products move 205 MB in 47 us, 4.4 TB/s of L2 reads at 40 % MFMA busy; this tile needs half.
template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm_pipe_kernel(GemmKArgs ka) {
  DTF_ABORT_GUARD(ka);
  const GemmArgs& a = ka.g;
  constexpr int PM = 128, PN = 128, PK = 32;
  using TileA = OperandTile<PM, PK, !TA, 512>;
  using TileB = OperandTile<PN, PK, TB, 512>;
  constexpr int MI = 4, NJ = 2;
  constexpr int STAGE = TileA::FLOATS + TileB::FLOATS;
  extern __shared__ __attribute__((aligned(16))) float pipe_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, l15 = lane & 15;
  const int per_xcd = (ka.total + 7) >> 3;   // as in gemm_tile_kernel: an XCD takes one contiguous range of work items
  const int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (t >= ka.total) return;
  const int tx = t % ka.gx, ty = (t / ka.gx) % ka.gy, tz = t / (ka.gx * ka.gy);
  const int m0 = ty * PM, n0 = tx * PN;
  const int k_begin = tz * ka.k_per_split;
  int k_end = k_begin + ka.k_per_split;
  if (k_end > a.K) k_end = a.K;
  const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;

  floatx4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = floatx4{0, 0, 0, 0};

  const int nsteps = k_end > k_begin ? (k_end - k_begin + PK - 1) / PK : 0;
  float4 ra[TileA::NV], rb[TileB::NV];
  if (nsteps > 0) {
    TileA::fetch(a.A, a.lda, m0, a.M, k_begin, k_end, tid, ra);
    TileB::fetch(a.B, a.ldb, n0, a.N, k_begin, k_end, tid, rb);
    TileA::commit(pipe_smem, tid, ra);
    TileB::commit(pipe_smem + TileA::FLOATS, tid, rb);
    TileA::fetch(a.A, a.lda, m0, a.M, k_begin + PK, k_end, tid, ra);   // past k_end: zeros, never committed
    TileB::fetch(a.B, a.ldb, n0, a.N, k_begin + PK, k_end, tid, rb);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const float* As = pipe_smem + (s & 1) * STAGE;
    const float* Bs = As + TileA::FLOATS;
    if (s + 1 < nsteps) {
      // every wave left the other stage at the barrier that ended step s - 1
      float* An = pipe_smem + ((s + 1) & 1) * STAGE;
      TileA::commit(An, tid, ra);
      TileB::commit(An + TileA::FLOATS, tid, rb);
      TileA::fetch(a.A, a.lda, m0, a.M, k_begin + (s + 2) * PK, k_end, tid, ra);
      TileB::fetch(a.B, a.ldb, n0, a.N, k_begin + (s + 2) * PK, k_end, tid, rb);
    }
    const int k_left = k_end - (k_begin + s * PK);
#pragma unroll
    for (int kg = 0; kg < PK / 16; ++kg) {
      if (kg * 16 >= k_left) break;   // a short last step: its empty groups hold zeros
      floatx4 av[MI], bv[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) av[i] = TileA::frag(As, wm + i * 16 + l15, kg, q);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[j] = TileB::frag(Bs, wn + j * 16 + l15, kg, q);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = DTF_MFMA16(av[i][e], bv[j][e], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
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
constexpr size_t kPipeLds = 2 * (128 * 36 + 128 * 36) * sizeof(float);   // the larger ([row][k]) layout for both operands

using GemmKernel = void (*)(GemmKArgs);
GemmKernel pipe_kernel(bool ta, bool tb) {
  return ta ? (tb ? gemm_pipe_kernel<true, true> : gemm_pipe_kernel<true, false>)
            : (tb ? gemm_pipe_kernel<false, true> : gemm_pipe_kernel<false, false>);
}
template <int TM_, int TN_, int TK_, int WK = 1>
GemmKernel tile_kernel(bool ta, bool tb) {
  return ta ? (tb ? gemm_tile_kernel<TM_, TN_, TK_, true, true, WK> : gemm_tile_kernel<TM_, TN_, TK_, true, false, WK>)
            : (tb ? gemm_tile_kernel<TM_, TN_, TK_, false, true, WK> : gemm_tile_kernel<TM_, TN_, TK_, false, false, WK>);
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
  const bool aligned = ((a.M | a.N | a.K | a.lda | a.ldb) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.B) & 15) == 0;
  // variant (0 = choose here): 1 general 64x64x32, 2 128x128x32, 3 64x64x128, 4 64x128x64, 5 128x128x64, 6 64x64x64,
  // 7 64x64x64 with eight waves, 8 64x64x128 with eight waves, 9 128x128x32 two-stage pipeline with eight waves
  // Measured on the fully connected products of the MNIST network (bench/gemm_sweep.py, batch 128
  // to 2048): the 64x64 tile with a 64-deep step is the fastest or within 5 % of it on every one
  // of them (20-30 % ahead of the 32-deep general kernel and, up to batch 2048, of the 128x128
  // tile, which leaves most CUs without work at these sizes).
  int variant = a.variant;
  if (variant == 0) {
    // eight waves (7) where the tiles are few enough to leave a workgroup alone on its CU (the
    // batch <= 256 products: 3-8 % faster); with more tiles no different or slightly slower
    const int tiles64 = ceil_div(a.M, 64) * ceil_div(a.N, 64);
    const int tiles128 = ceil_div(a.M, TM) * ceil_div(a.N, TN);
    variant = tiles128 >= 512 ? 2 : tiles64 <= 200 ? 7 : 6;
    // about one 128x128 tile per CU and a short k: the two-stage pipeline without a split (the data gradient
    // of the first fully connected layer at batch 1024: 42.5 against 47.7 us; slower than the 64x64 tiles
    // wherever it needs a split to fill the CUs)
    if (tiles128 >= 192 && tiles128 < 512 && a.K <= 1024) variant = 9;
  }
  if (!aligned) variant = 1;
  int bm, bn, bk, threads = 256;
  size_t shmem = 0;
  GemmKernel kernel;
  switch (variant) {
    case 2: bm = 128; bn = 128; bk = 32; kernel = tile_kernel<128, 128, 32>(a.trans_a, a.trans_b); break;
    case 3: bm = 64; bn = 64; bk = 128; kernel = tile_kernel<64, 64, 128>(a.trans_a, a.trans_b); break;
    case 4: bm = 64; bn = 128; bk = 64; kernel = tile_kernel<64, 128, 64>(a.trans_a, a.trans_b); break;
    case 5: bm = 128; bn = 128; bk = 64; kernel = tile_kernel<128, 128, 64>(a.trans_a, a.trans_b); break;
    case 6: bm = 64; bn = 64; bk = 64; kernel = tile_kernel<64, 64, 64>(a.trans_a, a.trans_b); break;
    case 7: bm = 64; bn = 64; bk = 64; kernel = tile_kernel<64, 64, 64, 2>(a.trans_a, a.trans_b); threads = 512; break;
    case 8: bm = 64; bn = 64; bk = 128; kernel = tile_kernel<64, 64, 128, 2>(a.trans_a, a.trans_b); threads = 512; break;
    case 9: bm = 128; bn = 128; bk = 32; kernel = pipe_kernel(a.trans_a, a.trans_b); threads = 512; shmem = kPipeLds; break;
    default: variant = 1; bm = BM; bn = BN; bk = BK; kernel = gemm_kernel; break;
  }
  const int gx = ceil_div(a.N, bn), gy = ceil_div(a.M, bm);
  int splits = 1;
  if (a.splits > 0) {
    splits = a.splits;
  } else if (variant == 1) {
    const int k_steps = ceil_div(a.K, BK);
    while (gx * gy * splits < 256 && splits * 2 <= k_steps / 4 && splits < 32) splits *= 2;
  } else if (variant == 9 && a.variant == 0) {
    splits = 1;
  } else {
    // Split along k until about four workgroups per CU exist (they hide each other's memory round
    // trips), while a split keeps at least three 64-deep steps' worth of k (192): below that the
    // prologue, the epilogue and the atomics of a workgroup outweigh what the extra parallelism buys.
    // A product that does not accumulate also pays for clearing its output first, which only a
    // small output or a grid that would leave CUs idle is worth.
    const long long out = (long long)a.M * a.N;
    while (gx * gy * splits < 1024 && a.K / (splits * 2) >= 192 && out * splits * 2 <= (8ll << 20) &&
           (a.accumulate || out <= 1200000 || gx * gy * splits < 256) && splits < 32)
      splits *= 2;
  }
  if (shmem) {
    static const bool once = [] {
      for (int ta = 0; ta < 2; ++ta)
        for (int tb = 0; tb < 2; ++tb) ensure_dynamic_lds(reinterpret_cast<const void*>(pipe_kernel(ta, tb)), kPipeLds);
      return true;
    }();
    (void)once;
  }
  GemmKArgs ka;
  ka.g = a;
  // a split starts on a multiple of 16 (whole MFMA k groups, 16-byte aligned rows); its last k-step may be short
  ka.k_per_split = variant == 1 ? ceil_div(ceil_div(a.K, BK), splits) * BK : ceil_div(ceil_div(a.K, 16), splits) * 16;
  splits = ceil_div(a.K, ka.k_per_split);
  ka.atomic_out = splits > 1;
  ka.gx = gx; ka.gy = gy; ka.total = gx * gy * splits;
  (void)bk;
  const dim3 grid = variant == 1 ? dim3(gx, gy, splits) : dim3((ka.total + 7) & ~7);
  if (splits > 1) {
    if (!a.accumulate) {
      if (a.ldc == a.N) {
        memset_async(a.C, 0, (size_t)a.M * a.N * sizeof(float), stream);
      } else {
        for (int m = 0; m < a.M; ++m) memset_async(a.C + (size_t)m * a.ldc, 0, (size_t)a.N * sizeof(float), stream);
      }
    }
    launch("gemm_f32_splitk", kernel, grid, dim3(threads), shmem, stream, ka);
    if (a.bias || a.relu) {
      BiasActArgs b{a.C, a.bias, a.M, a.N, a.ldc, a.relu};
      long long total = (long long)a.M * a.N;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      launch("bias_act", bias_act_kernel, dim3(blocks), dim3(256), 0, stream, b);
    }
  } else {
    launch("gemm_f32", kernel, grid, dim3(threads), shmem, stream, ka);
  }
}

}  // namespace dtf
