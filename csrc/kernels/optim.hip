// Optimizer and flat-buffer kernels: one launch updates every parameter of the model.
//
// Parameters, gradients and optimizer slots each live in one contiguous fp32 arena, so the
// "multi-tensor apply" is a single vectorised stream over n floats. The step-dependent scalars
// (bias-corrected learning rate, gradient scale 1/sum(mask) that turns the all-reduced SUM into
// the mean of the aggregated replicas) are read from a small device buffer so that a captured
// hipGraph can be replayed with new values.
#include "ops.h"
#include "optim_math.h"

namespace dtf {
namespace {

inline int flat_blocks(long long n4) {
  long long b = (n4 + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ __launch_bounds__(256) void adam_kernel(FlatOptArgs a) {
  const float lr_t = a.hyper[0], b1 = a.hyper[1], b2 = a.hyper[2], eps = a.hyper[3], gs = a.hyper[4];
  const long long n4 = a.n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  floatx4* p4 = reinterpret_cast<floatx4*>(a.p);
  const floatx4* g4 = reinterpret_cast<const floatx4*>(a.g);
  floatx4* m4 = reinterpret_cast<floatx4*>(a.m);
  floatx4* v4 = reinterpret_cast<floatx4*>(a.v);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    floatx4 p = p4[i], m = m4[i], v = v4[i];
    adam_update4(p, m, v, g4[i], gs, lr_t, b1, b2, eps);
    p4[i] = p;
    m4[i] = m;
    v4[i] = v;
  }
  // tail (n not a multiple of 4)
  const long long tail0 = n4 << 2;
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi < a.n - tail0) {
    const long long i = tail0 + gi;
    const float g = a.g[i] * gs;
    const float m = b1 * a.m[i] + (1.f - b1) * g;
    const float v = b2 * a.v[i] + (1.f - b2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    a.p[i] -= lr_t * m / (sqrtf(v) + eps);
  }
}

__global__ __launch_bounds__(256) void momentum_kernel(FlatOptArgs a) {
  const float lr = a.hyper[0], gs = a.hyper[4], mu = a.hyper[5], wd = a.hyper[6];
  const long long n4 = a.n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  floatx4* p4 = reinterpret_cast<floatx4*>(a.p);
  const floatx4* g4 = reinterpret_cast<const floatx4*>(a.g);
  floatx4* m4 = reinterpret_cast<floatx4*>(a.m);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    floatx4 p = p4[i], m = m4[i];
    momentum_update4(p, m, g4[i], gs, lr, mu, wd);
    p4[i] = p;
    m4[i] = m;
  }
  const long long tail0 = n4 << 2;
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi < a.n - tail0) {
    const long long i = tail0 + gi;
    const float g = a.g[i] * gs + wd * a.p[i];
    const float m = mu * a.m[i] + g;
    a.m[i] = m;
    a.p[i] -= lr * m;
  }
}

// The values travel in the launch's own argument block, which the runtime copies when the launch
// is enqueued: a later call cannot change what an earlier, not yet executed one will write.
__global__ __launch_bounds__(64) void set_scalars_kernel(SetScalarsArgs a) {
  const int t = threadIdx.x;
  if (a.f_dst && t < a.nf) a.f_dst[t] = a.f[t];
  if (a.u_dst && t < a.nu) a.u_dst[t] = a.u[t];
}

__global__ __launch_bounds__(256) void scale_kernel(FlatScaleArgs a) {
  const float s = a.scale_dev ? *a.scale_dev : a.scale;
  const long long n4 = a.n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  floatx4* x4 = reinterpret_cast<floatx4*>(a.x);
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = gi; i < n4; i += stride) x4[i] = x4[i] * s;
  const long long tail0 = n4 << 2;
  if (gi < a.n - tail0) a.x[tail0 + gi] *= s;
}

__global__ __launch_bounds__(256) void axpy_kernel(AxpyArgs a) {
  const long long n4 = a.n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  floatx4* y4 = reinterpret_cast<floatx4*>(a.y);
  const floatx4* x4 = reinterpret_cast<const floatx4*>(a.x);
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = gi; i < n4; i += stride) y4[i] = y4[i] + x4[i] * a.alpha;
  const long long tail0 = n4 << 2;
  if (gi < a.n - tail0) a.y[tail0 + gi] += a.alpha * a.x[tail0 + gi];
}

// One work item = (tensor, chunk of 256 float4s, slot k): the copies k, k + 8, k + 16, ... of the chunk
// are added in that order - loads issued four at a time, additions strictly sequential.
__global__ __launch_bounds__(256) void part_reduce_kernel(PartReduceArgs a) {
  int e = 0;
  while (e + 1 < a.n_entries && (int)blockIdx.x >= a.table[e + 1].first_item) ++e;
  const PartEntry ent = a.table[e];
  const int local = (int)blockIdx.x - ent.first_item;
  const int k = local & (kStatSlots - 1);
  const int idx4 = (local >> 3) * 256 + (int)threadIdx.x;
  if (idx4 >= ent.numel4) return;
  const floatx4* p = reinterpret_cast<const floatx4*>(ent.part) + idx4;
  const size_t cs = (size_t)ent.numel4;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  int j = k;
  for (; j + 3 * kStatSlots < ent.copies; j += 4 * kStatSlots) {
    const floatx4 v0 = __builtin_nontemporal_load(p + (size_t)j * cs);
    const floatx4 v1 = __builtin_nontemporal_load(p + (size_t)(j + kStatSlots) * cs);
    const floatx4 v2 = __builtin_nontemporal_load(p + (size_t)(j + 2 * kStatSlots) * cs);
    const floatx4 v3 = __builtin_nontemporal_load(p + (size_t)(j + 3 * kStatSlots) * cs);
    acc += v0;
    acc += v1;
    acc += v2;
    acc += v3;
  }
  for (; j < ent.copies; j += kStatSlots) acc += __builtin_nontemporal_load(p + (size_t)j * cs);
  reinterpret_cast<floatx4*>(a.slots + (size_t)k * a.slot_stride + ent.dst_off)[idx4] = acc;
}

// y[i] += sum over the kStatSlots copies of x; the copies are cleared for the next step.
__global__ __launch_bounds__(256) void slot_reduce_kernel(SlotReduceArgs a) {
  const long long n4 = a.n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  floatx4* y4 = reinterpret_cast<floatx4*>(a.y);
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = gi; i < n4; i += stride) {
    floatx4 v[kStatSlots];
#pragma unroll
    for (int r = 0; r < kStatSlots; ++r) v[r] = *reinterpret_cast<const floatx4*>(a.slots + r * a.slot_stride + 4 * i);
    floatx4 s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) any = any || (s[e] != 0.f);
    if (any) {
      y4[i] = y4[i] + s;
#pragma unroll
      for (int r = 0; r < kStatSlots; ++r)
        *reinterpret_cast<floatx4*>(a.slots + r * a.slot_stride + 4 * i) = floatx4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const long long tail0 = n4 << 2;
  if (gi < a.n - tail0) {
    float s = 0.f;
    for (int r = 0; r < kStatSlots; ++r) {
      s += a.slots[r * a.slot_stride + tail0 + gi];
      a.slots[r * a.slot_stride + tail0 + gi] = 0.f;
    }
    a.y[tail0 + gi] += s;
  }
}

// One thread per 4 consecutive outputs and per group of kPartGroup partial copies (blockIdx.y), four
// loads in flight; a single group stores, several groups add atomically (y is an accumulator).
constexpr int kPartGroup = 32;
__global__ __launch_bounds__(256) void partial_sum_kernel(PartialSumArgs a) {
  DTF_ABORT_GUARD(a);
  const long long n4 = a.n >> 2;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  floatx4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = floatx4{0.f, 0.f, 0.f, 0.f};
  const float* src = a.ws + 4 * i;
  int p = blockIdx.y * kPartGroup;
  const int p_end = min(a.parts, p + kPartGroup);
  for (; p + 4 <= p_end; p += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += *reinterpret_cast<const floatx4*>(src + (p + u) * a.stride);
  }
  for (; p < p_end; ++p) s[0] += *reinterpret_cast<const floatx4*>(src + p * a.stride);
  const floatx4 t = (s[0] + s[1]) + (s[2] + s[3]);
  if (gridDim.y == 1) {
    floatx4* y4 = reinterpret_cast<floatx4*>(a.y);
    y4[i] = y4[i] + t;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(a.y + 4 * i + e, t[e]);
  }
}

}  // namespace

void partial_sum(const PartialSumArgs& a, hipStream_t stream) {
  if ((a.n & 3) || (a.stride & 3) || a.parts < 1) throw std::runtime_error("partial_sum: n and stride must be multiples of 4, parts >= 1");
  launch("partial_sum", partial_sum_kernel, dim3((unsigned)(((a.n >> 2) + 255) / 256), (unsigned)ceil_div(a.parts, kPartGroup)), dim3(256), 0,
         stream, a);
}

void part_reduce(const PartReduceArgs& a, int total_items, hipStream_t stream) {
  if (!a.table || a.n_entries < 1 || total_items < 1 || !a.slots || (a.slot_stride & 3)) throw std::runtime_error("part_reduce: bad arguments");
  launch("part_reduce", part_reduce_kernel, dim3(total_items), dim3(256), 0, stream, a);
}

void slot_reduce(const SlotReduceArgs& a, hipStream_t stream) {
  if ((a.slot_stride & 3) != 0) throw std::runtime_error("slot_reduce: slot stride must be a multiple of 4 floats");
  launch("slot_reduce", slot_reduce_kernel, dim3(flat_blocks(a.n >> 2)), dim3(256), 0, stream, a);
}

void fused_adam(const FlatOptArgs& a, hipStream_t stream) {
  launch("fused_adam", adam_kernel, dim3(flat_blocks(a.n >> 2)), dim3(256), 0, stream, a);
}
void fused_momentum_sgd(const FlatOptArgs& a, hipStream_t stream) {
  launch("fused_momentum_sgd", momentum_kernel, dim3(flat_blocks(a.n >> 2)), dim3(256), 0, stream, a);
}
void set_scalars(const SetScalarsArgs& a, hipStream_t stream) {
  if (a.nf < 0 || a.nf > kSetScalarsMax || a.nu < 0 || a.nu > kSetScalarsMax) throw std::runtime_error("set_scalars: at most 8 values of each kind");
  if ((a.nf > 0 && !a.f_dst) || (a.nu > 0 && !a.u_dst)) throw std::runtime_error("set_scalars: null destination");
  launch("set_scalars", set_scalars_kernel, dim3(1), dim3(64), 0, stream, a);
}
void flat_scale(const FlatScaleArgs& a, hipStream_t stream) {
  launch("flat_scale", scale_kernel, dim3(flat_blocks(a.n >> 2)), dim3(256), 0, stream, a);
}
void flat_axpy(const AxpyArgs& a, hipStream_t stream) {
  launch("flat_axpy", axpy_kernel, dim3(flat_blocks(a.n >> 2)), dim3(256), 0, stream, a);
}

}  // namespace dtf
