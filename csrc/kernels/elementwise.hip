// Memory-bound kernels: elementwise, pooling, dropout, small reductions, augmentation, and
// the one-launch filter re-packer. All stream 16 bytes per lane where the layout allows.
#include "ops.h"
#include "split_pack.h"
#include "philox.h"

namespace dtf {
namespace {

inline int flat_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ __launch_bounds__(256) void add_relu_fwd_kernel(AddReluArgs a) {
  const long long n4 = a.n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const floatx4* a4 = reinterpret_cast<const floatx4*>(a.a);
  const floatx4* b4 = reinterpret_cast<const floatx4*>(a.b);
  floatx4* o4 = reinterpret_cast<floatx4*>(a.out);
  for (long long i = gi; i < n4; i += stride) {
    const floatx4 x = a4[i], y = b4[i];
    floatx4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a.relu_first ? fmaxf(x[e], 0.f) + y[e] : fmaxf(x[e] + y[e], 0.f);
    o4[i] = o;
  }
  const long long t0 = n4 << 2;
  if (gi < a.n - t0) {
    const float x = a.a[t0 + gi], y = a.b[t0 + gi];
    a.out[t0 + gi] = a.relu_first ? fmaxf(x, 0.f) + y : fmaxf(x + y, 0.f);
  }
}

__global__ __launch_bounds__(256) void add_relu_bwd_kernel(AddReluBwdArgs a) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const float d = a.dout[i];
    if (a.relu_first) {
      a.da[i] = a.a[i] > 0.f ? d : 0.f;
      a.db[i] = d;
    } else {
      const float m = (a.a[i] + a.b[i]) > 0.f ? d : 0.f;
      a.da[i] = m;
      a.db[i] = m;
    }
  }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(ReluMaskArgs a) {
  DTF_ABORT_GUARD(a);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride)
    a.dx[i] = a.y[i] > 0.f ? a.dy[i] : 0.f;
}

__global__ __launch_bounds__(256) void shortcut_a_fwd_kernel(ShortcutAArgs a) {
  const int H2 = a.H >> 1, W2 = a.W >> 1, C2 = 2 * a.C, cq = a.C >> 1;
  const long long total = (long long)a.B * H2 * W2 * C2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C2);
    long long r = i / C2;
    const int w = (int)(r % W2);
    r /= W2;
    const int h = (int)(r % H2);
    const long long n = r / H2;
    float v = 0.f;
    if (c >= cq && c < cq + a.C) {
      const float* p = a.x + (((n * a.H + 2 * h) * a.W + 2 * w) * a.C) + (c - cq);
      v = 0.25f * (p[0] + p[a.C] + p[(long long)a.W * a.C] + p[(long long)a.W * a.C + a.C]);
    }
    a.y[i] = v;
  }
}

__global__ __launch_bounds__(256) void shortcut_a_bwd_kernel(ShortcutABwdArgs a) {
  const int H2 = a.H >> 1, W2 = a.W >> 1, C2 = 2 * a.C, cq = a.C >> 1;
  const long long total = (long long)a.B * a.H * a.W * a.C;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % a.C);
    long long r = i / a.C;
    const int w = (int)(r % a.W);
    r /= a.W;
    const int h = (int)(r % a.H);
    const long long n = r / a.H;
    float v = 0.f;
    if ((h >> 1) < H2 && (w >> 1) < W2) v = 0.25f * a.dy[((n * H2 + (h >> 1)) * W2 + (w >> 1)) * C2 + cq + c];
    if (a.accumulate) a.dx[i] += v; else a.dx[i] = v;
  }
}

// 2x2/2 SAME max pool: output ceil(H/2) x ceil(W/2); windows hanging over the edge see only
// the valid elements (TensorFlow pads with -inf for max pooling).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(MaxPoolArgs a) {
  const int Ho = (a.H + 1) >> 1, Wo = (a.W + 1) >> 1;
  const long long total = (long long)a.B * Ho * Wo * a.C;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % a.C);
    long long r = i / a.C;
    const int w = (int)(r % Wo);
    r /= Wo;
    const int h = (int)(r % Ho);
    const long long n = r / Ho;
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 2 * h + (k >> 1), x = 2 * w + (k & 1);
      if (y < a.H && x < a.W) {
        const float v = a.x[((n * a.H + y) * a.W + x) * a.C + c];
        if (v > best) { best = v; arg = k; }
      }
    }
    a.y[i] = best;
    if (a.argmax) a.argmax[i] = (unsigned char)arg;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(MaxPoolBwdArgs a) {
  const int Ho = (a.H + 1) >> 1, Wo = (a.W + 1) >> 1;
  const long long total = (long long)a.B * a.H * a.W * a.C;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % a.C);
    long long r = i / a.C;
    const int x = (int)(r % a.W);
    r /= a.W;
    const int y = (int)(r % a.H);
    const long long n = r / a.H;
    const long long o = ((n * Ho + (y >> 1)) * Wo + (x >> 1)) * a.C + c;
    const int k = ((y & 1) << 1) | (x & 1);
    float v = (a.argmax[o] == k) ? a.dy[o] : 0.f;
    // relu_mask: the pooled tensor is the max of ReLU outputs; a zero maximum means the
    // winning pre-activation was clipped, so no gradient flows.
    if (a.relu_mask && !(a.y[o] > 0.f)) v = 0.f;
    a.dx[i] = v;
  }
}

__global__ __launch_bounds__(256) void dropout_kernel(DropoutArgs a) {
  DTF_ABORT_GUARD(a);
  const long long n4 = (a.n + 3) >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const float inv_keep = 1.f / a.keep_prob;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const unsigned long long cidx = a.offset + (a.offset_dev ? *a.offset_dev : 0ull) + (unsigned long long)i;
    uint32_t ctr[4];
    philox4x32_10(cidx, a.seed, ctr);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long j = i * 4 + e;
      if (j < a.n) {
        const float u = philox_uniform(ctr[e]);
        a.y[j] = (u < a.keep_prob) ? a.x[j] * inv_keep : 0.f;
      }
    }
  }
}

__global__ __launch_bounds__(256) void col_sum_kernel(ColSumArgs a) {
  DTF_ABORT_GUARD(a);
  extern __shared__ float red[];
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) red[c] = 0.f;
  __syncthreads();
  const long long total = a.n_rows * a.C;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = a.C >> 2;
  if ((a.C & 3) == 0 && (stride % C4) == 0) {
    // every thread keeps one group of four channels across its grid-stride loop: 16-byte loads,
    // register accumulation, one LDS atomic per channel at the end
    const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x);
    const long long n4 = total >> 2;
    const floatx4* m4 = reinterpret_cast<const floatx4*>(a.relu_of);
    floatx4* o4 = reinterpret_cast<floatx4*>(a.masked);
    floatx4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (long long i = gtid; i < n4; i += stride) {
      floatx4 v = x4[i];
      if (m4) {
        const floatx4 m = m4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
        if (o4) o4[i] = v;
      }
      s += v;
    }
    const int c = (int)(gtid % C4) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(&red[c + e], s[e]);
  } else {
    for (long long i = gtid; i < total; i += stride) {
      float v = a.x[i];
      if (a.relu_of) {
        v = a.relu_of[i] > 0.f ? v : 0.f;
        if (a.masked) a.masked[i] = v;
      }
      atomicAdd(&red[(int)(i % a.C)], v);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += blockDim.x)
    if (red[c] != 0.f) atomicAdd(&a.out[c], red[c]);
}

__global__ __launch_bounds__(256) void zero_count_kernel(ZeroFracArgs a) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned int cnt = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) cnt += (a.x[i] == 0.f);
  const float s = wave_sum((float)cnt);
  if ((threadIdx.x & 63) == 0 && s > 0.f) atomicAdd(a.zeros, (unsigned long long)(s + 0.5f));
}

// Privatised histogram: the limits and one count per bucket live in LDS (dynamic: nb doubles +
// nb counters), a value's bucket is found by binary search, block totals go out with one atomic
// per non-empty bucket.
__device__ __forceinline__ int float_order_key(float f) {
  const int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
__global__ __launch_bounds__(256) void tf_histogram_kernel(HistogramArgs a) {
  extern __shared__ __align__(8) unsigned char hist_smem[];
  double* lim = reinterpret_cast<double*>(hist_smem);
  unsigned int* cnt = reinterpret_cast<unsigned int*>(lim + a.nb);
  for (int i = threadIdx.x; i < a.nb; i += 256) {
    lim[i] = a.limits[i];
    cnt[i] = 0u;
  }
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  int kmin = 0x7fffffff, kmax = (int)0x80000000;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const float f = a.x[i];
    const double v = (double)f;
    int lo = 0, hi = a.nb - 1;   // first index with lim[idx] > v (the last limit is DBL_MAX)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (lim[mid] > v) hi = mid; else lo = mid + 1;
    }
    atomicAdd(&cnt[lo], 1u);
    s1 += v;
    s2 += v * v;
    const int k = float_order_key(f);
    kmin = min(kmin, k);
    kmax = max(kmax, k);
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
    kmin = min(kmin, __shfl_xor(kmin, off, 64));
    kmax = max(kmax, __shfl_xor(kmax, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&a.moments[0], s1);
    atomicAdd(&a.moments[1], s2);
    atomicMin(&a.minmax[0], kmin);
    atomicMax(&a.minmax[1], kmax);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.nb; i += 256)
    if (cnt[i]) atomicAdd(&a.counts[i], (unsigned long long)cnt[i]);
}

// One block per output image: two-pass mean / std over the cropped H*W*C values, then
// (x - mean) / max(std, 1/sqrt(H*W*C)).
__global__ __launch_bounds__(256) void crop_whiten_kernel(CropWhitenArgs a) {
  __shared__ float s_red[8];
  __shared__ float s_mean, s_inv;
  const int b = blockIdx.x;
  const int src_n = a.index ? a.index[b] : b;
  const int oy = a.offs ? a.offs[2 * b] : 0, ox = a.offs ? a.offs[2 * b + 1] : 0;
  const int per = a.H * a.W * a.C;
  const float* src = a.src + (size_t)src_n * a.HP * a.WP * a.C;
  float* dst = a.out + (size_t)b * per;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < per; i += blockDim.x) {
    const int c = i % a.C;
    const int x = (i / a.C) % a.W;
    const int y = i / (a.C * a.W);
    const float v = src[((size_t)(y + oy) * a.WP + (x + ox)) * a.C + c];
    dst[i] = v;
    s1 += v;
    s2 += v * v;
  }
  if (!a.whiten) return;
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[wv] = s1; s_red[4 + wv] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t1 = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const float t2 = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    const float mean = t1 / per;
    float var = t2 / per - mean * mean;
    if (var < 0.f) var = 0.f;
    const float sd = fmaxf(sqrtf(var), 1.0f / sqrtf((float)per));
    s_mean = mean;
    s_inv = 1.f / sd;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < per; i += blockDim.x) dst[i] = (dst[i] - s_mean) * s_inv;
}

__global__ __launch_bounds__(256) void zero_multi_kernel(ZeroMultiArgs a) {
  unsigned long long total = 0;
  for (int i = 0; i < a.count; ++i) total += a.chunks[i];
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long c = (unsigned long long)blockIdx.x * 256 + threadIdx.x; c < total; c += stride) {
    unsigned long long r = c;
    int seg = 0;
    while (r >= a.chunks[seg]) r -= a.chunks[seg++];   // total bounds c, so seg < count
    reinterpret_cast<floatx4*>(a.ptr[seg])[r] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
}

__global__ __launch_bounds__(256) void pack_multi_kernel(PackMultiArgs a) {
  // Find this block's table entry (entries are sorted by first_block): every lane tests one entry,
  // so the search costs one memory round trip instead of one per entry.
  int below = 0;
  for (int k0 = 0; k0 < a.n_entries; k0 += 64) {
    const int k = k0 + (int)(threadIdx.x & 63);
    const bool hit = k < a.n_entries && (int)blockIdx.x >= a.table[k].first_block;
    below += __popcll(__ballot(hit));
  }
  const PackEntry t = a.table[below > 0 ? below - 1 : 0];
  if (t.transpose_flip >= 2) {
    // operand of the split-operand kernel (split_pack.h): two bf16 values per float slot
    const int i2 = ((int)blockIdx.x - t.first_block) * 256 + threadIdx.x;
    if (i2 >= t.n_floats) return;
    const float* w2 = a.params + t.w_off;
    const unsigned lo = split_packed_element(w2, t.Cin, t.transpose_flip - 2, 2 * i2);
    const unsigned hi = split_packed_element(w2, t.Cin, t.transpose_flip - 2, 2 * i2 + 1);
    reinterpret_cast<unsigned*>(a.packed + t.out_off)[i2] = lo | (hi << 16);
    return;
  }
  const int ks = t.ks;
  const int cin_e = t.transpose_flip ? t.Cout : t.Cin;
  const int cout_e = t.transpose_flip ? t.Cin : t.Cout;
  const int cp = cin_e <= 4 ? 4 : (cin_e <= 16 ? 16 : (cin_e <= 32 ? 32 : 64));
  const int cg_n = cp >= 16 ? cp / 16 : 1;
  const int nt_n = (cout_e + 15) / 16;
  const float* w = a.params + t.w_off;
  float* out = a.packed + t.out_off;
  const int i = ((int)blockIdx.x - t.first_block) * 256 + threadIdx.x;
  if (i >= t.n_floats) return;
  const int s = i & 3;
  const int l = (i >> 2) & 63;
  const int j = (i >> 8) % nt_n;
  const int g = (i >> 8) / nt_n;
  const int q = l >> 4, nn = l & 15;
  int tap, ci;
  if (cp >= 16) {
    tap = g / cg_n;
    ci = (g % cg_n) * 16 + 4 * q + s;
  } else {
    tap = 4 * g + q;
    ci = s;
  }
  const int co = j * 16 + nn;
  float v = 0.f;
  if (tap < ks * ks && ci < cin_e && co < cout_e) {
    const int kh = tap / ks, kw = tap % ks;
    if (t.transpose_flip)
      v = w[(((ks - 1 - kh) * ks + (ks - 1 - kw)) * t.Cin + co) * t.Cout + ci];
    else
      v = w[((kh * ks + kw) * t.Cin + ci) * t.Cout + co];
  }
  out[i] = v;
}

}  // namespace

void add_relu_fwd(const AddReluArgs& a, hipStream_t stream) {
  launch("add_relu_fwd", add_relu_fwd_kernel, dim3(flat_blocks(a.n >> 2)), dim3(256), 0, stream, a);
}
void add_relu_bwd(const AddReluBwdArgs& a, hipStream_t stream) {
  launch("add_relu_bwd", add_relu_bwd_kernel, dim3(flat_blocks(a.n)), dim3(256), 0, stream, a);
}
void relu_bwd(const ReluMaskArgs& a, hipStream_t stream) {
  launch("relu_bwd", relu_bwd_kernel, dim3(flat_blocks(a.n)), dim3(256), 0, stream, a);
}
void shortcut_a_fwd(const ShortcutAArgs& a, hipStream_t stream) {
  const long long total = (long long)a.B * (a.H / 2) * (a.W / 2) * 2 * a.C;
  launch("shortcut_a_fwd", shortcut_a_fwd_kernel, dim3(flat_blocks(total)), dim3(256), 0, stream, a);
}
void shortcut_a_bwd(const ShortcutABwdArgs& a, hipStream_t stream) {
  const long long total = (long long)a.B * a.H * a.W * a.C;
  launch("shortcut_a_bwd", shortcut_a_bwd_kernel, dim3(flat_blocks(total)), dim3(256), 0, stream, a);
}
void maxpool2x2_fwd(const MaxPoolArgs& a, hipStream_t stream) {
  const long long total = (long long)a.B * ((a.H + 1) / 2) * ((a.W + 1) / 2) * a.C;
  launch("maxpool2x2_fwd", maxpool_fwd_kernel, dim3(flat_blocks(total)), dim3(256), 0, stream, a);
}
void maxpool2x2_bwd(const MaxPoolBwdArgs& a, hipStream_t stream) {
  const long long total = (long long)a.B * a.H * a.W * a.C;
  launch("maxpool2x2_bwd", maxpool_bwd_kernel, dim3(flat_blocks(total)), dim3(256), 0, stream, a);
}
void dropout_apply(const DropoutArgs& a, hipStream_t stream) {
  launch("dropout", dropout_kernel, dim3(flat_blocks((a.n + 3) >> 2)), dim3(256), 0, stream, a);
}
void col_sum(const ColSumArgs& a, hipStream_t stream) {
  long long total = a.n_rows * a.C;
  // Every block ends with one atomic per column on the same C addresses, and same-address atomics
  // serialise (~20 ns each: 1024 blocks cost 20 us in the tail alone), while a thread's loop is a
  // chain of memory round trips (~1 us each). 8 K floats per block = eight trips per thread;
  // between 32 and 512 blocks.
  long long nb = total >> 13;
  if (nb < 32) nb = 32;
  if (nb > 512) nb = 512;
  if (nb > flat_blocks(total >> 2)) nb = flat_blocks(total >> 2);
  const int blocks = (int)nb;
  launch("col_sum", col_sum_kernel, dim3(blocks), dim3(256), a.C * sizeof(float), stream, a);
}
void zero_count(const ZeroFracArgs& a, hipStream_t stream) {
  launch("zero_count", zero_count_kernel, dim3(flat_blocks(a.n)), dim3(256), 0, stream, a);
}
void tf_histogram(const HistogramArgs& a, hipStream_t stream) {
  if (a.nb < 1 || a.nb > 4096) throw std::runtime_error("tf_histogram: 1..4096 buckets");
  int blocks = flat_blocks(a.n);
  if (blocks > 1024) blocks = 1024;   // every block ends with up to nb atomics
  launch("tf_histogram", tf_histogram_kernel, dim3(blocks), dim3(256), (size_t)a.nb * 12, stream, a);
}
void crop_whiten(const CropWhitenArgs& a, hipStream_t stream) {
  launch("crop_whiten", crop_whiten_kernel, dim3(a.B), dim3(256), 0, stream, a);
}
void zero_multi(const ZeroMultiArgs& a, hipStream_t stream) {
  if (a.count < 1 || a.count > kZeroMultiMax) throw std::runtime_error("zero_multi: 1..8 buffers");
  unsigned long long total = 0;
  for (int i = 0; i < a.count; ++i) {
    if (!a.ptr[i] || (reinterpret_cast<uintptr_t>(a.ptr[i]) & 15)) throw std::runtime_error("zero_multi: buffers must be 16-byte aligned");
    total += a.chunks[i];
  }
  if (total == 0) return;
  const unsigned long long want = (total + 255) / 256;
  launch("zero_multi", zero_multi_kernel, dim3((unsigned)(want > 2048 ? 2048 : want)), dim3(256), 0, stream, a);
}
void pack_multi(const PackMultiArgs& a, int total_blocks, hipStream_t stream) {
  launch("pack_multi", pack_multi_kernel, dim3(total_blocks), dim3(256), 0, stream, a);
}

}  // namespace dtf
