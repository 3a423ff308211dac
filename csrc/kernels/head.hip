// Classifier head and loss kernels.
//
// head_fwd fuses BatchNorm + ReLU + global average pool + fully connected + softmax
// cross-entropy for one image per block: the 8x8x64 activation is read exactly once and only
// the pooled features, logits and d(loss)/d(logits) are written.
// head_bwd fuses the FC data gradient, the broadcast through the average pool, the ReLU mask
// and the two BatchNorm-backward reductions; the FC parameter gradients are reduced per block
// and leave with one atomic per element and block.
#include "ops.h"

namespace dtf {
namespace {

constexpr int kMaxK = 32;     // classes
constexpr int kMaxC = 256;    // pooled channels

// One thread computes the loss terms of one row of logits held in registers/LDS.
__device__ __forceinline__ void xent_row(const float* z, int K, int label, float inv_B, float* dz, float& loss,
                                         int& hit) {
  float mx = z[0];
  int arg = 0;
  for (int k = 1; k < K; ++k)
    if (z[k] > mx) { mx = z[k]; arg = k; }
  float se = 0.f;
  for (int k = 0; k < K; ++k) se += __expf(z[k] - mx);
  const float lse = mx + __logf(se);
  loss = lse - z[label];
  hit = (arg == label) ? 1 : 0;
  if (dz) {
    const float inv = 1.f / se;
    for (int k = 0; k < K; ++k) {
      const float p = __expf(z[k] - mx) * inv;
      dz[k] = (p - (k == label ? 1.f : 0.f)) * inv_B;
    }
  }
}

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadFwdArgs a) {
  __shared__ float s_scale[kMaxC], s_shift[kMaxC], s_feat[kMaxC], s_logit[kMaxK];
  __shared__ float s_w[kMaxC * kMaxK / 4], s_part[8][kMaxK + 1];
  const int n = blockIdx.x;
  const int C = a.C, C4 = C >> 2, K = a.K, tid = threadIdx.x;
  // everything the block will need is requested up front: the activation chunks, the FC weights
  // (to LDS, when they fit) and the BatchNorm statistics; the block's time is one memory round trip
  // plus the short dependent chain behind it
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x + (size_t)n * a.HW * C);
  const int n4 = a.HW * C4;
  constexpr int kPre = 4;
  floatx4 pre[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int i = tid + 256 * u;
    pre[u] = i < n4 ? x4[i] : floatx4{0.f, 0.f, 0.f, 0.f};
  }
  const bool w_in_lds = C * K <= kMaxC * kMaxK / 4;
  if (w_in_lds)
    for (int e = tid; e < C * K; e += 256) s_w[e] = a.fc_w[e];
  const float* fw = w_in_lds ? s_w : a.fc_w;
  for (int c = tid; c < C; c += 256) {
    float m, r;
    bn_coeffs(a.bn, c, C, s_scale[c], s_shift[c], m, r);
    s_feat[c] = 0.f;
  }
  __syncthreads();
  // 256 % C4 == 0 is guaranteed by the launcher, so each thread keeps one channel group.
  const int c = (tid % C4) * 4;
  floatx4 s = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    if (tid + 256 * u < n4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += fmaxf(pre[u][e] * s_scale[c + e] + s_shift[c + e], 0.f);
    }
  }
  for (int i = tid + 256 * kPre; i < n4; i += 256) {
    const floatx4 v = x4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += fmaxf(v[e] * s_scale[c + e] + s_shift[c + e], 0.f);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) atomicAdd(&s_feat[c + e], s[e]);
  __syncthreads();
  const float inv_hw = 1.f / (float)a.HW;
  for (int cc = tid; cc < C; cc += 256) {
    const float f = s_feat[cc] * inv_hw;
    s_feat[cc] = f;
    a.feat[(size_t)n * C + cc] = f;
  }
  __syncthreads();
  // logits: thread (part, k) sums every 8th channel, 8 partial sums per class
  {
    const int k = tid & 31, part = tid >> 5;   // 8 parts x 32 class slots
    float z = 0.f;
    if (k < K)
      for (int cc = part; cc < C; cc += 8) z += s_feat[cc] * fw[cc * K + k];
    if (k < K) s_part[part][k] = z;
  }
  __syncthreads();
  if (tid < K) {
    float z = a.fc_b[tid];
#pragma unroll
    for (int part = 0; part < 8; ++part) z += s_part[part][tid];
    s_logit[tid] = z;
    a.logits[(size_t)n * K + tid] = z;
  }
  __syncthreads();
  if (tid < 64) {
    // softmax cross-entropy of the row by one wave, lane = class (K <= 32)
    const int label = a.labels[n];
    const float z = tid < K ? s_logit[tid] : -3.0e38f;
    float mx = z;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float e = tid < K ? __expf(z - mx) : 0.f;
    const float se = wave_sum(e);
    const float zl = __shfl(z, label);
    if (tid < K) a.dlogits[(size_t)n * K + tid] = (e / se - (tid == label ? 1.f : 0.f)) / (float)a.B;
    const unsigned long long at_max = __ballot(tid < K && z == mx);
    if (tid == 0) {
      const int slot = n & (a.slots - 1);
      atomicAdd(a.loss + slot, (mx + __logf(se) - zl) / (float)a.B);
      if (a.correct && (__ffsll((long long)at_max) - 1) == label) atomicAdd(a.correct + slot, 1);   // first maximum wins, as argmax does
    }
  }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
  __shared__ float s_scale[kMaxC], s_shift[kMaxC], s_mean[kMaxC], s_rstd[kMaxC], s_df[kMaxC];
  __shared__ float s_red[2 * kMaxC];
  __shared__ float s_dz[kMaxK];
  const int n = blockIdx.x;
  const int C = a.C, C4 = C >> 2, K = a.K, tid = threadIdx.x;
  // the activation chunks are requested before the short dependent chain of coefficients
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x + (size_t)n * a.HW * C);
  const int n4 = a.HW * C4;
  constexpr int kPre = 4;
  floatx4 pre[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int i = tid + 256 * u;
    pre[u] = i < n4 ? x4[i] : floatx4{0.f, 0.f, 0.f, 0.f};
  }
  if (tid < K) s_dz[tid] = a.dlogits[(size_t)n * K + tid];
  for (int c = tid; c < C; c += 256) {
    bn_coeffs(a.bn, c, C, s_scale[c], s_shift[c], s_mean[c], s_rstd[c]);
    s_red[c] = 0.f;
    s_red[C + c] = 0.f;
  }
  __syncthreads();
  const float inv_hw = 1.f / (float)a.HW;
  for (int c = tid; c < C; c += 256) {
    float d = 0.f;
    for (int k = 0; k < K; ++k) d += s_dz[k] * a.fc_w[c * K + k];
    s_df[c] = d * inv_hw;
  }
  __syncthreads();
  floatx4* g4 = reinterpret_cast<floatx4*>(a.g + (size_t)n * a.HW * C);
  const int c = (tid % C4) * 4;
  floatx4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  auto one = [&](int i, const floatx4& x) {
    floatx4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[e] = (x[e] * s_scale[c + e] + s_shift[c + e] > 0.f) ? s_df[c + e] : 0.f;
      s1[e] += g[e];
      s2[e] += g[e] * ((x[e] - s_mean[c + e]) * s_rstd[c + e]);
    }
    g4[i] = g;
  };
#pragma unroll
  for (int u = 0; u < kPre; ++u)
    if (tid + 256 * u < n4) one(tid + 256 * u, pre[u]);
  for (int i = tid + 256 * kPre; i < n4; i += 256) one(i, x4[i]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    atomicAdd(&s_red[c + e], s1[e]);
    atomicAdd(&s_red[C + c + e], s2[e]);
  }
  __syncthreads();
  double* slot = stat_slot(a.gsums, C);
  for (int cc = tid; cc < 2 * C; cc += 256) atomicAdd(&slot[cc], (double)s_red[cc]);
}

// FC parameter gradients: one thread per (c, k) element walks the batch; dW[c][k] = sum_n
// feat[n][c] * dz[n][k], db[k] = sum_n dz[n][k]. The batch is split over gridDim.y blocks that
// combine with one atomic per element.
__global__ __launch_bounds__(256) void head_param_grad_kernel(HeadBwdArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int C = a.C, K = a.K;
  const int per = (a.B + gridDim.y - 1) / gridDim.y;
  const int n0 = blockIdx.y * per;
  const int n1 = (n0 + per < a.B) ? n0 + per : a.B;
  if (e < C * K) {
    const int c = e / K, k = e - c * K;
    float s = 0.f;
    for (int n = n0; n < n1; ++n) s += a.feat[(size_t)n * C + c] * a.dlogits[(size_t)n * K + k];
    atomicAdd(&a.d_fc_w[e], s);
  } else if (e < C * K + K) {
    const int k = e - C * K;
    float s = 0.f;
    for (int n = n0; n < n1; ++n) s += a.dlogits[(size_t)n * K + k];
    atomicAdd(&a.d_fc_b[k], s);
  }
}

__global__ __launch_bounds__(256) void softmax_xent_kernel(SoftmaxXentArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  float loss = 0.f;
  int hit = 0;
  if (n < a.B) {
    float z[kMaxK], dz[kMaxK];
    for (int k = 0; k < a.K; ++k) z[k] = a.logits[(size_t)n * a.K + k];
    xent_row(z, a.K, a.labels[n], 1.f / (float)a.B, a.dlogits ? dz : nullptr, loss, hit);
    if (a.dlogits)
      for (int k = 0; k < a.K; ++k) a.dlogits[(size_t)n * a.K + k] = dz[k];
    loss /= (float)a.B;
  }
  loss = wave_sum(loss);
  float fh = wave_sum((float)hit);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(a.loss, loss);
    if (a.correct) atomicAdd(a.correct, (int)(fh + 0.5f));
  }
}

__global__ __launch_bounds__(256) void top1_kernel(Top1Args a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  int hit = 0;
  if (n < a.B) {
    const float* z = a.logits + (size_t)n * a.K;
    float mx = z[0];
    int arg = 0;
    for (int k = 1; k < a.K; ++k)
      if (z[k] > mx) { mx = z[k]; arg = k; }
    hit = (arg == a.labels[n]);
  }
  const float fh = wave_sum((float)hit);
  if ((threadIdx.x & 63) == 0 && fh > 0.f) atomicAdd(a.correct, (int)(fh + 0.5f));
}

}  // namespace

void head_fwd(const HeadFwdArgs& a, hipStream_t stream) {
  if (a.K > kMaxK || a.C > kMaxC || (a.C & 3) || (256 % (a.C / 4)) != 0)
    throw std::runtime_error("head_fwd: unsupported C/K");
  if (a.slots < 1 || (a.slots & (a.slots - 1))) throw std::runtime_error("head_fwd: slots must be a power of two");
  launch("head_fwd", head_fwd_kernel, dim3(a.B), dim3(256), 0, stream, a);
}

void head_bwd(const HeadBwdArgs& a, hipStream_t stream) {
  if (a.K > kMaxK || a.C > kMaxC || (a.C & 3) || (256 % (a.C / 4)) != 0)
    throw std::runtime_error("head_bwd: unsupported C/K");
  launch("head_bwd", head_bwd_kernel, dim3(a.B), dim3(256), 0, stream, a);
  // ~16 samples per thread: the loop is a chain of dependent loads, so depth, not atomics, sets its time
  int splits = a.B / 16;
  if (splits < 1) splits = 1;
  if (splits > 128) splits = 128;
  launch("head_param_grad", head_param_grad_kernel, dim3(ceil_div(a.C * a.K + a.K, 256), splits), dim3(256), 0, stream, a);
}

void softmax_xent(const SoftmaxXentArgs& a, hipStream_t stream) {
  if (a.K > kMaxK) throw std::runtime_error("softmax_xent: too many classes");
  launch("softmax_xent", softmax_xent_kernel, dim3(ceil_div(a.B, 256)), dim3(256), 0, stream, a);
}

void top1_count(const Top1Args& a, hipStream_t stream) {
  launch("top1_count", top1_kernel, dim3(ceil_div(a.B, 256)), dim3(256), 0, stream, a);
}

}  // namespace dtf
