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
  const int n = blockIdx.x;
  const int C = a.C, C4 = C >> 2, tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    float m, r;
    bn_coeffs(a.bn, c, C, s_scale[c], s_shift[c], m, r);
    s_feat[c] = 0.f;
  }
  __syncthreads();
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x + (size_t)n * a.HW * C);
  const int n4 = a.HW * C4;
  // 256 % C4 == 0 is guaranteed by the launcher, so each thread keeps one channel group.
  const int c = (tid % C4) * 4;
  floatx4 s = {0, 0, 0, 0};
  for (int i = tid; i < n4; i += 256) {
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
  if (tid < a.K) {
    float z = a.fc_b[tid];
    for (int cc = 0; cc < C; ++cc) z += s_feat[cc] * a.fc_w[cc * a.K + tid];
    s_logit[tid] = z;
    a.logits[(size_t)n * a.K + tid] = z;
  }
  __syncthreads();
  if (tid == 0) {
    float dz[kMaxK];
    float loss;
    int hit;
    xent_row(s_logit, a.K, a.labels[n], 1.f / (float)a.B, dz, loss, hit);
    for (int k = 0; k < a.K; ++k) a.dlogits[(size_t)n * a.K + k] = dz[k];
    atomicAdd(a.loss, loss / (float)a.B);
    if (a.correct && hit) atomicAdd(a.correct, 1);
  }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
  __shared__ float s_scale[kMaxC], s_shift[kMaxC], s_mean[kMaxC], s_rstd[kMaxC], s_df[kMaxC];
  __shared__ float s_red[2 * kMaxC];
  __shared__ float s_dz[kMaxK];
  const int n = blockIdx.x;
  const int C = a.C, C4 = C >> 2, K = a.K, tid = threadIdx.x;
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
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x + (size_t)n * a.HW * C);
  floatx4* g4 = reinterpret_cast<floatx4*>(a.g + (size_t)n * a.HW * C);
  const int n4 = a.HW * C4;
  const int c = (tid % C4) * 4;
  floatx4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  for (int i = tid; i < n4; i += 256) {
    const floatx4 x = x4[i];
    floatx4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[e] = (x[e] * s_scale[c + e] + s_shift[c + e] > 0.f) ? s_df[c + e] : 0.f;
      s1[e] += g[e];
      s2[e] += g[e] * ((x[e] - s_mean[c + e]) * s_rstd[c + e]);
    }
    g4[i] = g;
  }
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
