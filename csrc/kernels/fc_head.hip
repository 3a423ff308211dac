// The fully connected classifier of the MNIST network, forward and backward, in one launch.
//
// Input is the raw product of the first fully connected layer (batch x 512, before its bias).
// Per tile of eight samples a workgroup adds that bias, applies ReLU and the Philox dropout
// mask (the one dropout_kernel would draw for the same seed and offset), multiplies by the
// 512 x K weight of the second layer, adds its bias, evaluates softmax cross-entropy, and
// walks back: d(loss)/d(logits), the second layer's weight and bias gradients, the gradient
// through the weight, the dropout mask and the ReLU mask (both recovered from the kept
// activation: it is positive exactly where the mask kept a positive pre-activation), and the
// first layer's bias gradient. Only the logits and the gradient at the first layer's
// pre-activation are written; the hidden activations never leave registers.
//
// This replaces twelve launches of the step (bias + ReLU, dropout, three small GEMMs with
// their clears and bias passes, the loss kernel, two column sums, dropout replay, ReLU mask),
// each of which moved the batch x 512 tensor through HBM once more and cost a launch slot of
// 5-7 us for 1-2 us of work.
//
// Thread layout: 128 column groups of four hidden units x 2 row halves. A thread keeps the
// K weights of its four hidden units (4 x K registers) and the partial weight gradients
// (another 4 x K) across all tiles of its workgroup; these leave through one LDS fold of the
// two halves and one atomic per value per workgroup.
//
// reference: DistributedLeNet/src/mnist.py:118-164 (hidden layer, dropout, logits, loss, top-1)
#include "ops.h"
#include "philox.h"
#include "xent.h"

namespace dtf {
namespace {

constexpr int kH = 512;        // hidden units
constexpr int kCG = kH / 4;    // column groups
constexpr int kRows = 8;       // samples per tile
constexpr int kMaxBlocks = 128;

template <int K>
__global__ __launch_bounds__(256) void fc_head_kernel(FcHeadArgs a) {
  DTF_ABORT_GUARD(a);
  __shared__ float s_part[kRows][2][K];      // per-wave partial logits of a tile
  __shared__ float s_z[kRows][K];            // logits of a tile
  __shared__ float s_dz[kRows][K];           // d(loss)/d(logits) of a tile
  constexpr int kFoldDb1 = kH * K + kCG;     // dw2 (padded, see slot below), then db1
  __shared__ float s_fold[kFoldDb1 + kH];    // the workgroup's parameter gradients on their way out

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = tid & (kCG - 1), half = tid >> 7;
  const bool train = a.dh != nullptr;
  const bool drop = a.keep_prob < 1.f;
  const float inv_keep = drop ? 1.f / a.keep_prob : 1.f;
  const unsigned long long base = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);

  float w[4][K];
  {
    const floatx4* w4 = reinterpret_cast<const floatx4*>(a.w2 + (size_t)4 * cg * K);   // 4*K floats, 16-byte aligned for any K
    float flat[4 * K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const floatx4 v = w4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) flat[4 * i + e] = v[e];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < K; ++j) w[c][j] = flat[c * K + j];
  }
  const floatx4 b1 = reinterpret_cast<const floatx4*>(a.b1)[cg];

  float dw[4][K];
  float db1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < K; ++j) dw[c][j] = 0.f;
  // threads 0..kRows-1 own one row of every tile in the loss phase
  float loss_acc = 0.f, db2[K];
  int hit_acc = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) db2[j] = 0.f;

  const int tiles = (a.B + kRows - 1) / kRows;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int row0 = tile * kRows;
    float hd[kRows / 2][4];
    // ---- hidden activation, dropout, partial logits
#pragma unroll
    for (int i = 0; i < kRows / 2; ++i) {
      const int r = half + 2 * i, row = row0 + r;
      float pl[K];
#pragma unroll
      for (int j = 0; j < K; ++j) pl[j] = 0.f;
      if (row < a.B) {
        const floatx4 x = reinterpret_cast<const floatx4*>(a.h_raw + (size_t)row * kH)[cg];
        uint32_t rnd[4];
        if (drop) philox4x32_10(base + (unsigned long long)row * kCG + cg, a.seed, rnd);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = fmaxf(x[c] + b1[c], 0.f);
          if (drop) v = (philox_uniform(rnd[c]) < a.keep_prob) ? v * inv_keep : 0.f;
          hd[i][c] = v;
#pragma unroll
          for (int j = 0; j < K; ++j) pl[j] = fmaf(v, w[c][j], pl[j]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) hd[i][c] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float s = wave_sum(pl[j]);
        if (lane == 0) s_part[r][wave & 1][j] = s;
      }
    }
    __syncthreads();
    // ---- logits, loss, d(loss)/d(logits): one thread per row
    if (tid < kRows && row0 + tid < a.B) {
      const int row = row0 + tid;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float z = s_part[tid][0][j] + s_part[tid][1][j] + a.b2[j];
        s_z[tid][j] = z;
        a.logits[(size_t)row * K + j] = z;
      }
      float loss;
      int hit;
      xent_row(s_z[tid], K, a.labels[row], 1.f / (float)a.B, train ? s_dz[tid] : nullptr, loss, hit);
      loss_acc += loss;
      hit_acc += hit;
      if (train) {
#pragma unroll
        for (int j = 0; j < K; ++j) db2[j] += s_dz[tid][j];
      }
    }
    __syncthreads();   // s_part and s_z are free for the next tile, s_dz is complete
    if (!train) continue;
    // ---- back through the second layer, the dropout mask and the ReLU
#pragma unroll
    for (int i = 0; i < kRows / 2; ++i) {
      const int r = half + 2 * i, row = row0 + r;
      if (row >= a.B) continue;
      float dz[K];
#pragma unroll
      for (int j = 0; j < K; ++j) dz[j] = s_dz[r][j];
      floatx4 g;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          s = fmaf(dz[j], w[c][j], s);
          dw[c][j] = fmaf(hd[i][c], dz[j], dw[c][j]);
        }
        g[c] = hd[i][c] > 0.f ? s * inv_keep : 0.f;
        db1[c] += g[c];
      }
      reinterpret_cast<floatx4*>(a.dh + (size_t)row * kH)[cg] = g;
    }
  }

  // ---- leave: loss terms, then the parameter gradients
  if (tid < kRows) {
    if (loss_acc != 0.f) atomicAdd(a.loss, loss_acc / (float)a.B);
    if (a.correct && hit_acc) atomicAdd(a.correct, hit_acc);
    if (train) {
#pragma unroll
      for (int j = 0; j < K; ++j) atomicAdd(a.db2 + j, db2[j]);
    }
  }
  if (!train) return;
  // Fold the two halves in LDS, laid out as the gradients are in memory (dw2 row-major, then db1),
  // and add them to memory with consecutive lanes on consecutive addresses: a wave's atomic then
  // touches two cache lines instead of sixty-four, and it is the number of line operations that
  // bounds this step (every workgroup of the grid adds to the same 22 KB).
  auto slot = [](int n) { return n + n / (4 * K); };   // one pad word per column group: its 4*K floats start on an odd stride
  if (half == 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int j = 0; j < K; ++j) s_fold[slot((4 * cg + c) * K + j)] = dw[c][j];
      s_fold[kFoldDb1 + 4 * cg + c] = db1[c];
    }
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int j = 0; j < K; ++j) s_fold[slot((4 * cg + c) * K + j)] += dw[c][j];
      s_fold[kFoldDb1 + 4 * cg + c] += db1[c];
    }
  }
  __syncthreads();
  for (int n = tid; n < kH * K; n += 256) atomicAdd(a.dw2 + n, s_fold[slot(n)]);
  for (int n = tid; n < kH; n += 256) atomicAdd(a.db1 + n, s_fold[kFoldDb1 + n]);
}

}  // namespace

void fc_head(const FcHeadArgs& a, hipStream_t stream) {
  if (a.H != kH) throw std::runtime_error("fc_head: the hidden layer must have 512 units");
  if (a.K != 10) throw std::runtime_error("fc_head: built for 10 classes");
  if (a.B < 1) throw std::runtime_error("fc_head: empty batch");
  if (!(a.keep_prob > 0.f)) throw std::runtime_error("fc_head: keep_prob must be positive");
  if ((a.dh != nullptr) != (a.dw2 != nullptr) || (a.dh != nullptr) != (a.db2 != nullptr) || (a.dh != nullptr) != (a.db1 != nullptr))
    throw std::runtime_error("fc_head: give all gradient outputs or none");
  if ((reinterpret_cast<uintptr_t>(a.h_raw) | reinterpret_cast<uintptr_t>(a.w2) | reinterpret_cast<uintptr_t>(a.b1) |
       reinterpret_cast<uintptr_t>(a.dh)) & 15)
    throw std::runtime_error("fc_head: operands must be 16-byte aligned");
  const int tiles = ceil_div(a.B, kRows);
  const int cap = a.max_blocks > 0 ? a.max_blocks : kMaxBlocks;
  launch("fc_head", fc_head_kernel<10>, dim3(tiles < cap ? tiles : cap), dim3(256), 0, stream, a);
}

}  // namespace dtf
