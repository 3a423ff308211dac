// Classifier head and loss kernels.
//
// head_fwd fuses BatchNorm + ReLU + global average pool + fully connected + softmax
// cross-entropy for one image per wave: the 8x8x64 activation is read exactly once and only
// the pooled features, logits and d(loss)/d(logits) are written.
// head_bwd fuses the FC data gradient, the broadcast through the average pool, the ReLU mask
// and the two BatchNorm-backward reductions (combined per block of four images before they
// leave as fp64 atomics); the FC parameter gradients are a separate small kernel.
#include "ops.h"
#include "xent.h"

namespace dtf {
namespace {

constexpr int kMaxK = 32;     // classes
constexpr int kMaxC = 256;    // pooled channels

// One image per wave, four images per block: no block barrier after the shared coefficients are
// in LDS, and a quarter of the workgroups (1024 one-image blocks spent most of the kernel's time
// being dispatched). Lane l owns channel group l % C4 (64 % C4 == 0) of pixels l / C4, l / C4 + 64 / C4, ...
constexpr int kHeadImgs = 4;

__device__ __forceinline__ floatx4 group_sum(floatx4 v, int C4) {
  for (int off = C4; off < 64; off <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], off);
  }
  return v;
}

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadFwdArgs a) {
  DTF_ABORT_GUARD(a);
  __shared__ float s_scale[kMaxC], s_shift[kMaxC];
  __shared__ float s_w[kMaxC * kMaxK / 4];
  const int C = a.C, C4 = C >> 2, K = a.K, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x * kHeadImgs + wave;
  const bool live = n < a.B;
  const int n4 = a.HW * C4;
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x + (size_t)(live ? n : 0) * a.HW * C);
  // the first activation chunks are requested before the coefficient chain
  constexpr int kPre = 8;
  floatx4 pre[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int i = lane + 64 * u;
    pre[u] = (live && i < n4) ? x4[i] : floatx4{0.f, 0.f, 0.f, 0.f};
  }
  const bool w_in_lds = C * K <= kMaxC * kMaxK / 4;
  if (w_in_lds)
    for (int e = tid; e < C * K; e += 256) s_w[e] = a.fc_w[e];
  const float* fw = w_in_lds ? s_w : a.fc_w;
  for (int c = tid; c < C; c += 256) {
    float m, r;
    bn_coeffs(a.bn, c, C, s_scale[c], s_shift[c], m, r);
  }
  __syncthreads();
  if (!live) return;

  const int g = lane % C4, c = 4 * g;
  floatx4 sc, sh;
#pragma unroll
  for (int e = 0; e < 4; ++e) { sc[e] = s_scale[c + e]; sh[e] = s_shift[c + e]; }
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    if (lane + 64 * u < n4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += fmaxf(pre[u][e] * sc[e] + sh[e], 0.f);
    }
  }
  for (int i = lane + 64 * kPre; i < n4; i += 64) {
    const floatx4 v = x4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += fmaxf(v[e] * sc[e] + sh[e], 0.f);
  }
  s = group_sum(s, C4);   // every lane of group g now holds the group's four channel sums
  const float inv_hw = 1.f / (float)a.HW;
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] *= inv_hw;
  if (lane < C4) *reinterpret_cast<floatx4*>(a.feat + (size_t)n * C + c) = s;

  // logits: lane (k = lane % 32, half = lane / 32) sums the channel groups of its half
  const int k = lane & 31, half = lane >> 5;
  float z = 0.f;
  for (int gg = half; gg < C4; gg += 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = __shfl(s[e], gg);   // lane gg belongs to group gg
      if (k < K) z += f * fw[(4 * gg + e) * K + k];
    }
  }
  z += __shfl_xor(z, 32);
  if (k < K) z += a.fc_b[k];
  if (lane < K) a.logits[(size_t)n * K + lane] = z;

  // softmax cross-entropy of the row, lane = class (K <= 32; lanes 32.. mirror lanes 0..31)
  const int label = a.labels[n];
  const float zz = k < K ? z : -3.0e38f;
  float mx = zz;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  const float ex = (lane < K) ? __expf(zz - mx) : 0.f;
  const float se = wave_sum(ex);
  const float zl = __shfl(zz, label);
  if (lane < K) a.dlogits[(size_t)n * K + lane] = (ex / se - (lane == label ? 1.f : 0.f)) / (float)a.B;
  const unsigned long long at_max = __ballot(lane < K && zz == mx);
  if (lane == 0) {
    const int slot = n & (a.slots - 1);
    atomicAdd(a.loss + slot, (mx + __logf(se) - zl) / (float)a.B);
    if (a.correct && (__ffsll((long long)at_max) - 1) == label) atomicAdd(a.correct + slot, 1);   // first maximum wins, as argmax does
  }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
  DTF_ABORT_GUARD(a);
  __shared__ float s_scale[kMaxC], s_shift[kMaxC], s_mean[kMaxC], s_rstd[kMaxC];
  __shared__ float s_w[kMaxC * kMaxK / 4];
  __shared__ float s_red[kHeadImgs][2 * kMaxC];   // one row per wave, combined in wave order: no float atomics
  const int C = a.C, C4 = C >> 2, K = a.K, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x * kHeadImgs + wave;
  const bool live = n < a.B;
  const int n4 = a.HW * C4;
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x + (size_t)(live ? n : 0) * a.HW * C);
  constexpr int kPre = 8;
  floatx4 pre[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int i = lane + 64 * u;
    pre[u] = (live && i < n4) ? x4[i] : floatx4{0.f, 0.f, 0.f, 0.f};
  }
  const float dz_lane = (live && lane < K) ? a.dlogits[(size_t)n * K + lane] : 0.f;
  const bool w_in_lds = C * K <= kMaxC * kMaxK / 4;
  if (w_in_lds)
    for (int e = tid; e < C * K; e += 256) s_w[e] = a.fc_w[e];
  const float* fw = w_in_lds ? s_w : a.fc_w;
  for (int c = tid; c < C; c += 256) {
    bn_coeffs(a.bn, c, C, s_scale[c], s_shift[c], s_mean[c], s_rstd[c]);
#pragma unroll
    for (int w = 0; w < kHeadImgs; ++w) {
      s_red[w][c] = 0.f;
      s_red[w][C + c] = 0.f;
    }
  }
  __syncthreads();
  if (live) {
    const int g = lane % C4, c = 4 * g;
    floatx4 sc, sh, mean, rstd, df = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[e] = s_scale[c + e]; sh[e] = s_shift[c + e]; mean[e] = s_mean[c + e]; rstd[e] = s_rstd[c + e]; }
    // gradient of the pooled feature: d feat[c] = sum_k dz[k] w[c][k], spread over the HW pixels
    for (int k = 0; k < K; ++k) {
      const float dzk = __shfl(dz_lane, k);
#pragma unroll
      for (int e = 0; e < 4; ++e) df[e] += dzk * fw[(c + e) * K + k];
    }
    const float inv_hw = 1.f / (float)a.HW;
#pragma unroll
    for (int e = 0; e < 4; ++e) df[e] *= inv_hw;
    floatx4* g4 = reinterpret_cast<floatx4*>(a.g + (size_t)n * a.HW * C);
    floatx4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    auto one = [&](int i, const floatx4& x) {
      floatx4 gv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gv[e] = (x[e] * sc[e] + sh[e] > 0.f) ? df[e] : 0.f;
        s1[e] += gv[e];
        s2[e] += gv[e] * ((x[e] - mean[e]) * rstd[e]);
      }
      g4[i] = gv;
    };
#pragma unroll
    for (int u = 0; u < kPre; ++u)
      if (lane + 64 * u < n4) one(lane + 64 * u, pre[u]);
    for (int i = lane + 64 * kPre; i < n4; i += 64) one(i, x4[i]);
    s1 = group_sum(s1, C4);
    s2 = group_sum(s2, C4);
    if (lane < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s_red[wave][c + e] = s1[e];
        s_red[wave][C + c + e] = s2[e];
      }
    }
  }
  __syncthreads();
  double* slot = stat_slot(a.gsums, C);
  for (int cc = tid; cc < 2 * C; cc += 256) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kHeadImgs; ++w) s += (double)s_red[w][cc];
    atomicAdd(&slot[cc], to_grid(s, kStatGridBwd));
  }
}

// FC parameter gradients dW[c][k] = sum_n feat[n][c] * dz[n][k], db[k] = sum_n dz[n][k]: one wave per
// element; lane l walks samples l, l + 64, ... and the 64 partial sums are combined by a butterfly
// of lane exchanges - a fixed tree - then stored. No atomics: the same bits on every run.
constexpr int kParamElems = 4;   // waves (elements) per workgroup
__global__ __launch_bounds__(256) void head_param_grad_kernel(HeadBwdArgs a) {
  DTF_ABORT_GUARD(a);
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * kParamElems + (threadIdx.x >> 6);
  const int C = a.C, K = a.K;
  float s = 0.f;
  if (e < C * K) {
    const int c = e / K, k = e - c * K;
    for (int n = lane; n < a.B; n += 64) s += a.feat[(size_t)n * C + c] * a.dlogits[(size_t)n * K + k];
  } else if (e < C * K + K) {
    const int k = e - C * K;
    for (int n = lane; n < a.B; n += 64) s += a.dlogits[(size_t)n * K + k];
  }
  s = wave_sum(s);
  if (lane == 0) {
    if (e < C * K) a.d_fc_w[e] += s;
    else if (e < C * K + K) a.d_fc_b[e - C * K] += s;
  }
}

__global__ __launch_bounds__(256) void softmax_xent_kernel(SoftmaxXentArgs a) {
  DTF_ABORT_GUARD(a);
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
  if (a.K > kMaxK || a.C > kMaxC || (a.C & 3) || (64 % (a.C / 4)) != 0)
    throw std::runtime_error("head_fwd: unsupported C/K");
  if (a.slots < 1 || (a.slots & (a.slots - 1))) throw std::runtime_error("head_fwd: slots must be a power of two");
  launch("head_fwd", head_fwd_kernel, dim3(ceil_div(a.B, kHeadImgs)), dim3(256), 0, stream, a);
}

void head_bwd(const HeadBwdArgs& a, hipStream_t stream) {
  if (a.K > kMaxK || a.C > kMaxC || (a.C & 3) || (64 % (a.C / 4)) != 0)
    throw std::runtime_error("head_bwd: unsupported C/K");
  launch("head_bwd", head_bwd_kernel, dim3(ceil_div(a.B, kHeadImgs)), dim3(256), 0, stream, a);
  launch("head_param_grad", head_param_grad_kernel, dim3(ceil_div(a.C * a.K + a.K, kParamElems)), dim3(256), 0, stream, a);
}

void softmax_xent(const SoftmaxXentArgs& a, hipStream_t stream) {
  if (a.K > kMaxK) throw std::runtime_error("softmax_xent: too many classes");
  launch("softmax_xent", softmax_xent_kernel, dim3(ceil_div(a.B, 256)), dim3(256), 0, stream, a);
}

void top1_count(const Top1Args& a, hipStream_t stream) {
  launch("top1_count", top1_kernel, dim3(ceil_div(a.B, 256)), dim3(256), 0, stream, a);
}

}  // namespace dtf
