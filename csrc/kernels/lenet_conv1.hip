// First LeNet layer fused with its pooling: conv 5x5x1 -> 32 SAME + bias + ReLU + 2x2/2 max pool
// over 28x28 MNIST images (reference: DistributedLeNet/src/mnist.py:107-118), forward and backward.
//
// With one input channel the layer is 25 multiply-adds per output and its full-resolution
// activation (B x 28 x 28 x 32 floats, 103 MB at batch 1024) is by far the largest tensor of the
// network, yet nothing downstream needs it: the forward pass continues from the pooled maximum and
// the backward pass needs only the pooled maximum (which doubles as the ReLU mask of its winner)
// and the 2-bit winner position. So
//   forward  computes the four convolution outputs of a pooling window in registers and writes the
//            pooled maximum + winner (25 MB + 6 MB instead of writing and re-reading 103 MB);
//   backward forms the filter and bias gradients straight from the pooled gradient: each pooled
//            cell contributes g * x[winner + tap] to the 25 taps of its channel, so the scattered
//            full-resolution gradient is never materialised either.
// Mapping: lane = output channel (32) x 2 pooling cells per wave, so the 25 filter taps of the
// lane's channel live in registers, the 6x6 input patch of a cell is a broadcast LDS read, and a
// cell's 32 channels are one 128-byte store. Plain VALU work: K = 25 is too small for MFMA tiles
// to pay, and the kernel is bound by its 60 MB of traffic.
#include "common.h"
#include "ops.h"

namespace dtf {
namespace {

constexpr int kIn = 28;              // image height / width
constexpr int kCellsW = 14;          // pooled width
constexpr int kCells = 196;          // pooled cells per image
constexpr int kCo = 32;              // output channels
constexpr int kRow = 32;             // padded LDS row: 2 halo + 28 + 2 halo
constexpr int kPlane = kRow * kRow;  // one padded image
constexpr int kRounds = (kCells + 7) / 8;

__device__ __forceinline__ void stage_image(float* plane, const float4& v, int tid) {
  // float4 number `tid` of the 784-float image: row tid / 7, columns 4 (tid % 7) ..
  float* d = plane + (tid / 7 + 2) * kRow + 2 + 4 * (tid % 7);
  *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
  *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
}

__device__ __forceinline__ void load_patch(const float* plane, int cell, float (&pt)[6][6]) {
  const int py = cell / kCellsW, px = cell % kCellsW;
  const float* o = plane + (2 * py) * kRow + 2 * px;   // rows 2py-2 .. 2py+3 of the image
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float2 v = *reinterpret_cast<const float2*>(o + r * kRow + 2 * c);
      pt[r][2 * c] = v.x;
      pt[r][2 * c + 1] = v.y;
    }
  }
}

__global__ __launch_bounds__(256) void conv1_pool_fwd_kernel(Conv1PoolFwdArgs a) {
  DTF_ABORT_GUARD(a);
  __shared__ __align__(16) float plane[2][kPlane];
  const int tid = threadIdx.x;
  const int co = tid & 31, sub = tid >> 5;
  for (int i = tid; i < 2 * kPlane; i += 256) (&plane[0][0])[i] = 0.f;   // halos stay zero
  float w[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) w[t] = a.w[t * kCo + co];
  const float bias = a.bias[co];
  float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
  int img = blockIdx.x;
  if (img < a.B && tid < kCells) pf = reinterpret_cast<const float4*>(a.x + (size_t)img * kIn * kIn)[tid];
  __syncthreads();
  for (int buf = 0; img < a.B; img += gridDim.x, buf ^= 1) {
    // The other plane may still be read by waves finishing the previous image; this one was last
    // read two images ago, before the previous barrier.
    if (tid < kCells) stage_image(plane[buf], pf, tid);
    __syncthreads();
    const int nxt = img + gridDim.x;
    if (nxt < a.B && tid < kCells) pf = reinterpret_cast<const float4*>(a.x + (size_t)nxt * kIn * kIn)[tid];
    // gridDim.y > 1 (fewer images than CUs): the workgroups of an image take every gridDim.y-th round
#pragma unroll 1
    for (int round = blockIdx.y; round < kRounds; round += gridDim.y) {
      const int cell = round * 8 + sub;
      if (cell < kCells) {
        float pt[6][6];
        load_patch(plane[buf], cell, pt);
        float acc[4] = {bias, bias, bias, bias};
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 5; ++kw) {
            const float wv = w[kh * 5 + kw];
            acc[0] = fmaf(pt[kh][kw], wv, acc[0]);
            acc[1] = fmaf(pt[kh][kw + 1], wv, acc[1]);
            acc[2] = fmaf(pt[kh + 1][kw], wv, acc[2]);
            acc[3] = fmaf(pt[kh + 1][kw + 1], wv, acc[3]);
          }
        }
        // max over the ReLU outputs, first maximum wins (window order (0,0) (0,1) (1,0) (1,1))
        float best = fmaxf(acc[0], 0.f);
        int arg = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float v = fmaxf(acc[k], 0.f);
          if (v > best) { best = v; arg = k; }
        }
        const size_t o = ((size_t)img * kCells + cell) * kCo + co;
        a.p[o] = best;
        a.arg[o] = (unsigned char)arg;
      }
    }
  }
}

__global__ __launch_bounds__(256) void conv1_pool_bwd_kernel(Conv1PoolBwdArgs a) {
  DTF_ABORT_GUARD(a);
  __shared__ __align__(16) float plane[2][kPlane];
  __shared__ float red[4][26][kCo];
  const int tid = threadIdx.x;
  const int co = tid & 31, sub = tid >> 5;
  for (int i = tid; i < 2 * kPlane; i += 256) (&plane[0][0])[i] = 0.f;
  float dw[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) dw[t] = 0.f;
  float db = 0.f;
  float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
  int img = blockIdx.x;
  if (img < a.B && tid < kCells) pf = reinterpret_cast<const float4*>(a.x + (size_t)img * kIn * kIn)[tid];
  // The pooled gradient, the pooled maximum and its argmax of a cell come from memory. A round
  // computes for ~0.2 us and a load takes ~1 us to arrive, and a CU holds one or two of these
  // workgroups, so the loads of five rounds are issued together, one batch ahead of their use.
  constexpr int kBatch = 5, kBatches = (kRounds + kBatch - 1) / kBatch;
  float g_c[kBatch], p_c[kBatch], g_n[kBatch], p_n[kBatch];
  int k_c[kBatch], k_n[kBatch];
  auto fetch = [&](int im, int batch, float (&g)[kBatch], float (&p)[kBatch], int (&k)[kBatch]) {
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int cell = (batch * kBatch + u) * 8 + sub;
      g[u] = 0.f; p[u] = 0.f; k[u] = 0;
      if (im < a.B && cell < kCells) {
        const size_t o = ((size_t)im * kCells + cell) * kCo + co;
        g[u] = a.dp[o];
        p[u] = a.p[o];
        k[u] = a.arg[o];
      }
    }
  };
  // gridDim.y > 1 (fewer images than CUs): the workgroups of an image take every gridDim.y-th batch
  const int batch0 = blockIdx.y, batch_step = gridDim.y;
  fetch(img, batch0, g_c, p_c, k_c);
  __syncthreads();
  for (int buf = 0; img < a.B; img += gridDim.x, buf ^= 1) {
    if (tid < kCells) stage_image(plane[buf], pf, tid);
    __syncthreads();
    const int nxt = img + gridDim.x;
    if (nxt < a.B && tid < kCells) pf = reinterpret_cast<const float4*>(a.x + (size_t)nxt * kIn * kIn)[tid];
#pragma unroll 1
    for (int batch = batch0; batch < kBatches; batch += batch_step) {
      const bool last = batch + batch_step >= kBatches;
      fetch(last ? nxt : img, last ? batch0 : batch + batch_step, g_n, p_n, k_n);
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int cell = (batch * kBatch + u) * 8 + sub;
        if (cell < kCells) {
          // a zero maximum means the winning pre-activation was clipped by the ReLU
          const float g = p_c[u] > 0.f ? g_c[u] : 0.f;
          const int k = k_c[u];
          const bool wy = (k >> 1) != 0, wx = (k & 1) != 0;
          float pt[6][6];
          load_patch(plane[buf], cell, pt);
          db += g;
#pragma unroll
          for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
              const float t0 = pt[kh][kw], t1 = pt[kh][kw + 1], b0 = pt[kh + 1][kw], b1 = pt[kh + 1][kw + 1];
              const float top = wx ? t1 : t0;
              const float bot = wx ? b1 : b0;
              dw[kh * 5 + kw] = fmaf(g, wy ? bot : top, dw[kh * 5 + kw]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) { g_c[u] = g_n[u]; p_c[u] = p_n[u]; k_c[u] = k_n[u]; }
    }
  }
  // the two cells of a wave, then the four waves, then one atomic per value into the copy of
  // the block's XCD
#pragma unroll
  for (int t = 0; t < 25; ++t) dw[t] += __shfl_xor(dw[t], 32, 64);
  db += __shfl_xor(db, 32, 64);
  const int wave = tid >> 6;
  if ((tid & 63) < 32) {
#pragma unroll
    for (int t = 0; t < 25; ++t) red[wave][t][co] = dw[t];
    red[wave][25][co] = db;
  }
  __syncthreads();
  const size_t slot = a.slot_stride ? (size_t)(blockIdx.x & (kStatSlots - 1)) * a.slot_stride : 0;
  for (int i = tid; i < 26 * kCo; i += 256) {
    const int t = i / kCo, c = i % kCo;
    const float s = (red[0][t][c] + red[1][t][c]) + (red[2][t][c] + red[3][t][c]);
    if (s != 0.f) atomicAdd((t < 25 ? a.dw + t * kCo + c : a.dbias + c) + slot, s);
  }
}

template <typename K>
int blocks_per_cu(K kernel) {
  int nb = 0;
  DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0));
  return nb < 1 ? 1 : nb;
}

}  // namespace

void conv1_pool_fwd(const Conv1PoolFwdArgs& a, hipStream_t stream) {
  if (a.B < 1) throw std::runtime_error("conv1_pool_fwd: empty batch");
  static const int bpc = blocks_per_cu(conv1_pool_fwd_kernel);
  const int cap = 256 * bpc;
  const int grid = ceil_div(a.B, ceil_div(a.B, cap));
  launch("conv1_pool_fwd", conv1_pool_fwd_kernel, dim3(grid, a.B < 512 ? 2 : 1), dim3(256), 0, stream, a);
}

void conv1_pool_bwd(const Conv1PoolBwdArgs& a, hipStream_t stream) {
  if (a.B < 1) throw std::runtime_error("conv1_pool_bwd: empty batch");
  static const int bpc = blocks_per_cu(conv1_pool_bwd_kernel);
  // fewer, longer blocks than the forward kernel: every block ends with 832 atomics
  const int cap = 256 * (bpc > 2 ? 2 : bpc);
  const int grid = ceil_div(a.B, ceil_div(a.B, cap));
  launch("conv1_pool_bwd", conv1_pool_bwd_kernel, dim3(grid, a.B < 512 ? 2 : 1), dim3(256), 0, stream, a);
}

}  // namespace dtf
