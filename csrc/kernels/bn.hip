// BatchNorm kernels (batch statistics, NHWC, fp32 data, fp64 cross-block accumulation).
//
// All kernels view the tensor as [n_rows, C] and vectorise 16 bytes per lane. The grid is
// sized so that every thread keeps a fixed group of four channels across its grid-stride loop
// (total threads is a multiple of C/4); per-thread partials are reduced per channel in LDS
// and leave the block as one fp64 atomic per channel and statistic.
#include "ops.h"

namespace dtf {
namespace {

constexpr int kThreads = 256;

// grid such that gridDim.x * 256 is a multiple of C/4 (C/4 divides 256 for C in {4..1024 pow2};
// otherwise fall back to per-element channel computation).
inline int stream_blocks(long long n4) {
  long long b = (n4 + kThreads - 1) / kThreads;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ __launch_bounds__(256) void bn_stats_kernel(BnStatsArgs a) {
  extern __shared__ float red[];  // [2][C]
  const int C = a.C, C4 = C >> 2;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  const long long n4 = a.n_rows * C4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x);
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool fixed = (stride % C4) == 0;
  if (fixed) {
    floatx4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    const int c4 = (int)(i % C4);
    for (; i < n4; i += stride) {
      const floatx4 v = x4[i];
      s1 += v;
      s2 += v * v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(&red[c4 * 4 + e], s1[e]);
      atomicAdd(&red[C + c4 * 4 + e], s2[e]);
    }
  } else {
    for (; i < n4; i += stride) {
      const floatx4 v = x4[i];
      const int c4 = (int)(i % C4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(&red[c4 * 4 + e], v[e]);
        atomicAdd(&red[C + c4 * 4 + e], v[e] * v[e]);
      }
    }
  }
  __syncthreads();
  double* slot = stat_slot(a.sums, C);
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) atomicAdd(&slot[c], to_grid((double)red[c], kStatGridFwd));
}

__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a) {
  extern __shared__ float coef[];  // scale[C], shift[C]
  const int C = a.C, C4 = C >> 2;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sc, sh, m, r;
    bn_coeffs(a.bn, c, C, sc, sh, m, r);
    coef[c] = sc;
    coef[C + c] = sh;
  }
  __syncthreads();
  const long long n4 = a.n_rows * C4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x);
  floatx4* y4 = reinterpret_cast<floatx4*>(a.y);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4) * 4;
    floatx4 v = x4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = v[e] * coef[c + e] + coef[C + c + e];
      if (a.relu) v[e] = fmaxf(v[e], 0.f);
    }
    y4[i] = v;
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(BnReluBwdReduceArgs a) {
  extern __shared__ float sh[];  // scale[C], shift[C], mean[C], rstd[C], red[2C]
  const int C = a.C, C4 = C >> 2;
  float* scale = sh;
  float* shift = sh + C;
  float* mean = sh + 2 * C;
  float* rstd = sh + 3 * C;
  float* red = sh + 4 * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    bn_coeffs(a.bn, c, C, scale[c], shift[c], mean[c], rstd[c]);
    red[c] = 0.f;
    red[C + c] = 0.f;
  }
  __syncthreads();
  const long long n4 = a.n_rows * C4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const floatx4* dy4 = reinterpret_cast<const floatx4*>(a.dy);
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x);
  floatx4* g4 = reinterpret_cast<floatx4*>(a.g);
  const bool fixed = (stride % C4) == 0;
  floatx4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c_fixed = (int)(i % C4) * 4;
  for (; i < n4; i += stride) {
    const int c = fixed ? c_fixed : (int)(i % C4) * 4;
    floatx4 g = dy4[i];
    const floatx4 x = x4[i];
    floatx4 gx;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (a.relu && !(x[e] * scale[c + e] + shift[c + e] > 0.f)) g[e] = 0.f;
      gx[e] = g[e] * ((x[e] - mean[c + e]) * rstd[c + e]);
    }
    g4[i] = g;
    if (fixed) {
      s1 += g;
      s2 += gx;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(&red[c + e], g[e]);
        atomicAdd(&red[C + c + e], gx[e]);
      }
    }
  }
  if (fixed) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(&red[c_fixed + e], s1[e]);
      atomicAdd(&red[C + c_fixed + e], s2[e]);
    }
  }
  __syncthreads();
  double* slot = stat_slot(a.gsums, C);
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) atomicAdd(&slot[c], to_grid((double)red[c], kStatGridBwd));
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdApplyArgs a) {
  DTF_ABORT_GUARD(a);
  extern __shared__ float sh[];  // k1[C] (=gamma*rstd), mean[C], rstd[C], mg[C], mgx[C]
  const int C = a.C, C4 = C >> 2;
  float* k1 = sh;
  float* mean = sh + C;
  float* rstd = sh + 2 * C;
  float* mg = sh + 3 * C;
  float* mgx = sh + 4 * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sc, shf;
    bn_coeffs(a.bn, c, C, sc, shf, mean[c], rstd[c]);
    k1[c] = sc;
    const double sg = stat_total(a.gsums, c, C), sgx = stat_total(a.gsums, C + c, C);
    mg[c] = (float)(sg * (double)a.bn.inv_count);
    mgx[c] = (float)(sgx * (double)a.bn.inv_count);
    if (blockIdx.x == 0) {
      if (a.dbeta) a.dbeta[c] = (float)sg;
      if (a.dgamma) a.dgamma[c] = (float)sgx;
    }
  }
  __syncthreads();
  const long long n4 = a.n_rows * C4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const floatx4* g4 = reinterpret_cast<const floatx4*>(a.g);
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x);
  floatx4* dx4 = reinterpret_cast<floatx4*>(a.dx);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4) * 4;
    const floatx4 g = g4[i];
    const floatx4 x = x4[i];
    floatx4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (x[e] - mean[c + e]) * rstd[c + e];
      d[e] = k1[c + e] * (g[e] - mg[c + e] - xh * mgx[c + e]);
    }
    if (a.res_mode == BWD_RES_IDENT) {
      d += reinterpret_cast<const floatx4*>(a.res)[i];
    } else if (a.res_mode == BWD_RES_SHORTCUT_A) {
      // x is [B,H,W,C]; the block output gradient is [B,H/2,W/2,2C]; this tensor fed its
      // channels [C/2, C/2 + C) through a 2x2 average pool.
      const long long row = i / C4;
      const int w = (int)(row % a.W);
      const int h = (int)((row / a.W) % a.H);
      const long long n = row / ((long long)a.W * a.H);
      const int H2 = a.H >> 1, W2 = a.W >> 1;
      if ((h >> 1) < H2 && (w >> 1) < W2) {
        const float* rp = a.res + (((n * H2 + (h >> 1)) * W2 + (w >> 1)) * (2 * C)) + (C >> 1) + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] += 0.25f * rp[e];
      }
    }
    dx4[i] = d;
  }
}

}  // namespace

void bn_stats(const BnStatsArgs& a, hipStream_t stream) {
  if (a.C & 3) throw std::runtime_error("bn_stats: C must be a multiple of 4");
  const long long n4 = a.n_rows * (a.C / 4);
  launch("bn_stats", bn_stats_kernel, dim3(stream_blocks(n4)), dim3(kThreads), 2 * a.C * sizeof(float), stream, a);
}

void bn_apply(const BnApplyArgs& a, hipStream_t stream) {
  if (a.C & 3) throw std::runtime_error("bn_apply: C must be a multiple of 4");
  const long long n4 = a.n_rows * (a.C / 4);
  launch("bn_apply", bn_apply_kernel, dim3(stream_blocks(n4)), dim3(kThreads), 2 * a.C * sizeof(float), stream, a);
}

void bn_relu_bwd_reduce(const BnReluBwdReduceArgs& a, hipStream_t stream) {
  if (a.C & 3) throw std::runtime_error("bn_relu_bwd_reduce: C must be a multiple of 4");
  const long long n4 = a.n_rows * (a.C / 4);
  launch("bn_relu_bwd_reduce", bn_relu_bwd_reduce_kernel, dim3(stream_blocks(n4)), dim3(kThreads),
         6 * a.C * sizeof(float), stream, a);
}

void bn_bwd_apply(const BnBwdApplyArgs& a, hipStream_t stream) {
  if (a.C & 3) throw std::runtime_error("bn_bwd_apply: C must be a multiple of 4");
  const long long n4 = a.n_rows * (a.C / 4);
  launch("bn_bwd_apply", bn_bwd_apply_kernel, dim3(stream_blocks(n4)), dim3(kThreads), 5 * a.C * sizeof(float),
         stream, a);
}

}  // namespace dtf
