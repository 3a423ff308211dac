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
// The logits of a tile are sums over the 128 column groups. A wave holds 4 rows x 16 (K padded)
// partial values per lane and reduces them with a transposing butterfly: at each of the six
// steps a lane hands half of its values to its partner and keeps the other half, so 63 lane
// exchanges leave lane L with the wave total of value L, where reducing the values one by one
// takes 6 exchanges each. The loss of a tile is then evaluated by sixteen lanes per row.
//
// Thread layout: 128 column groups of four hidden units x 2 row halves. A thread keeps the
// K weights of its four hidden units (4 x K registers) and the partial weight gradients
// (another 4 x K) across all tiles of its workgroup; these leave through one LDS fold of the
// two halves and one atomic per value per workgroup.
//
// reference: DistributedLeNet/src/mnist.py:118-164 (hidden layer, dropout, logits, loss, top-1)
#include "ops.h"
#include "philox.h"

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
  // loss phase: threads 0..127 = 8 rows x 16 lanes, lane j of a row owns class j (j < K)
  const int lr = tid >> 4, lj = tid & 15;
  const bool loss_lane = tid < 16 * kRows && lj < K;
  const float b2 = loss_lane ? a.b2[lj] : 0.f;
  float loss_acc = 0.f, hit_acc = 0.f, db2 = 0.f;

  const int tiles = (a.B + kRows - 1) / kRows;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int row0 = tile * kRows;
    // the label is needed after the first barrier only: its load rides behind the first phase
    const int lrow = row0 + lr;
    const int label = (tid < 16 * kRows && lrow < a.B) ? a.labels[lrow] : 0;
    float hd[kRows / 2][4];
    float v[64];   // partial logits: v[16 * i + j] for row half + 2 * i, class j; zero for j >= K
    // ---- hidden activation, dropout, partial logits
#pragma unroll
    for (int i = 0; i < kRows / 2; ++i) {
      const int row = row0 + half + 2 * i;
#pragma unroll
      for (int j = 0; j < 16; ++j) v[16 * i + j] = 0.f;
      if (row < a.B) {
        const floatx4 x = reinterpret_cast<const floatx4*>(a.h_raw + (size_t)row * kH)[cg];
        uint32_t rnd[4];
        if (drop) philox4x32_10(base + (unsigned long long)row * kCG + cg, a.seed, rnd);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float h = fmaxf(x[c] + b1[c], 0.f);
          if (drop) h = (philox_uniform(rnd[c]) < a.keep_prob) ? h * inv_keep : 0.f;
          hd[i][c] = h;
#pragma unroll
          for (int j = 0; j < K; ++j) v[16 * i + j] = fmaf(h, w[c][j], v[16 * i + j]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) hd[i][c] = 0.f;
      }
    }
    // transposing butterfly: lane L ends with the wave's total of v[L]
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const bool upper = (lane & off) != 0;
#pragma unroll
      for (int i = 0; i < off; ++i) {
        const float send = upper ? v[i] : v[i + off];
        const float keep = upper ? v[i + off] : v[i];
        v[i] = keep + __shfl_xor(send, off);
      }
    }
    if ((lane & 15) < K) s_part[half + 2 * (lane >> 4)][wave & 1][lane & 15] = v[0];
    __syncthreads();
    // ---- logits, loss, d(loss)/d(logits): sixteen lanes per row (waves 0 and 1)
    if (tid < 16 * kRows) {
      const bool live = loss_lane && lrow < a.B;
      const float z = loss_lane ? s_part[lr][0][lj] + s_part[lr][1][lj] + b2 : -INFINITY;
      float mx = z;
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      const float e = loss_lane ? __expf(z - mx) : 0.f;
      float se = e;
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) se += __shfl_xor(se, off);
      // the first class that attains the maximum is the prediction
      const unsigned long long at_max = __ballot(loss_lane && z == mx);
      const int arg = __ffsll((unsigned long long)((at_max >> (lane & 48)) & 0xffffull)) - 1;
      const float z_label = __shfl(z, (lane & 48) + label);
      if (live) {
        a.logits[(size_t)lrow * K + lj] = z;
        if (train) {
          const float dz = (e / se - (lj == label ? 1.f : 0.f)) / (float)a.B;
          s_dz[lr][lj] = dz;
          db2 += dz;
        }
        if (lj == 0) {
          loss_acc += mx + __logf(se) - z_label;
          hit_acc += (arg == label) ? 1.f : 0.f;
        }
      }
    }
    __syncthreads();   // s_part is free for the next tile, s_dz is complete
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

  // ---- leave: loss terms (summed over the four rows of a wave first), then the parameter gradients
  if (tid < 16 * kRows) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      loss_acc += __shfl_xor(loss_acc, off);
      hit_acc += __shfl_xor(hit_acc, off);
      db2 += __shfl_xor(db2, off);
    }
    if (lane == 0) {
      if (loss_acc != 0.f) atomicAdd(a.loss, loss_acc / (float)a.B);
      if (a.correct && hit_acc > 0.f) atomicAdd(a.correct, (int)(hit_acc + 0.5f));
    }
    if (train && lane < K) atomicAdd(a.db2 + lane, db2);
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
