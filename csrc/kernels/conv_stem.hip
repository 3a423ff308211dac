// Stem convolution of the CIFAR ResNet: 3x3, 3 -> 16 channels, SAME, over 32x32 images (reference:
// DistributedResNet/src/resnet.py:183), forward (+ the statistics of the BatchNorm that follows)
// and filter gradient.
//
// With 3 input channels the whole reduction is K = 27 (seven MFMA k-steps, the 28th row is zero)
// and both passes are bound by the 16-channel tensor they write / read (67 MB at batch 1024), so
// the kernels are built around cheap staging instead of MFMA reuse: a tile is a band of 8 rows of
// one image, its 10 halo rows are contiguous 384-byte rows of the NHWC image and go to LDS as they
// are (one 16-byte chunk per thread, prefetched one band ahead), and element k = (kh, kw, ci) of a
// pixel's window is the float at  window origin + kh * row + (k mod 9): a per-lane constant offset,
// so operands are single ds_read_b32 with no index arithmetic in the loop. The 27 x 16 filter
// lives in 7 registers per lane.
//   forward:  M = output channel (A = filter), N = pixel, so a lane owns 4 consecutive channels of
//             a pixel (16-byte store) in the layout the statistics flush expects.
//   gradient: M = k (two tiles: 27 padded to 32), N = output channel, reduction over the pixels of
//             the band; the four waves split the band's 64 k-steps and combine through LDS.
#include "common.h"
#include "conv.h"

namespace dtf {
namespace {

constexpr int kW = 32, kTH = 8;                 // image width, rows per band
constexpr int kRowF = (kW + 2) * 3;             // floats of a padded LDS row (one halo pixel each side)
constexpr int kRow = 104;                       // LDS row stride
constexpr int kLH = kTH + 2;
constexpr int kPlane = kLH * kRow;
constexpr int kChunks = kLH * (kW * 3 / 4);     // 16-byte chunks of a band's halo rows (240)

__device__ __forceinline__ int k_offset(int k) {   // LDS offset of window element k = kh*9 + kw*3 + ci
  return k < 27 ? (k / 9) * kRow + (k % 9) : 0;
}

__device__ __forceinline__ void stem_stage(float* plane, const float4& v, int tid) {
  // chunk tid = (halo row tid / 24, floats 4 (tid % 24) ..) of the band; the row starts after the left halo pixel
  float* d = plane + (tid / 24) * kRow + 3 + 4 * (tid % 24);
  d[0] = v.x;
  d[1] = v.y;
  d[2] = v.z;
  d[3] = v.w;
}

__device__ __forceinline__ float4 stem_fetch(const float* x, int tile, int tid) {
  const int n = tile >> 2, y0 = (tile & 3) * kTH;
  const int y = y0 - 1 + tid / 24;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < kChunks && y >= 0 && y < kW) v = *reinterpret_cast<const float4*>(x + ((size_t)(n * kW + y) * kW) * 3 + 4 * (tid % 24));
  return v;
}

__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(ConvStemArgs a) {
  DTF_ABORT_GUARD(a);
  __shared__ __align__(16) float plane[2][kPlane];
  __shared__ float scratch[4 * 2 * 16];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  for (int i = tid; i < 2 * kPlane; i += 256) (&plane[0][0])[i] = 0.f;   // halo columns stay zero

  float wv[7];
  int offk[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + g;
    wv[s] = k < 27 ? a.w[k * 16 + n] : 0.f;   // A operand: row = output channel n, column k
    offk[s] = k_offset(k);
  }
  const int total_tiles = 4 * a.B;
  floatx4 st1[1] = {floatx4{0.f, 0.f, 0.f, 0.f}}, st2[1] = {floatx4{0.f, 0.f, 0.f, 0.f}};

  int tile = blockIdx.x;
  float4 pf = tile < total_tiles ? stem_fetch(a.x, tile, tid) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  for (int buf = 0; tile < total_tiles; tile += gridDim.x, buf ^= 1) {
    // the other plane may still be read by waves finishing the previous band
    if (tid < kChunks) stem_stage(plane[buf], pf, tid);
    __syncthreads();
    const int nxt = tile + gridDim.x;
    if (nxt < total_tiles) pf = stem_fetch(a.x, nxt, tid);
    const int img = tile >> 2, y0 = (tile & 3) * kTH;
    float* yb = a.y + ((size_t)(img * kW + y0) * kW) * 16 + 4 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int p = (wave * 4 + t) * 16 + n;                       // pixel of the band
      const float* win = plane[buf] + (p >> 5) * kRow + (p & 31) * 3;   // window origin in the padded rows
      float xv[7];
#pragma unroll
      for (int s = 0; s < 7; ++s) xv[s] = win[offk[s]];
      floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 7; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], xv[s], acc, 0, 0, 0);
      st1[0] += acc;
      st2[0] += acc * acc;
      *reinterpret_cast<floatx4*>(yb + (size_t)p * 16) = acc;
    }
  }
  if (a.sums) block_stat_flush<1, 4>(scratch, st1, st2, a.sums, 16, 0, kStatGridFwd);
}

// dw[k][co] += sum over pixels x[window element k of the pixel] * dy[pixel][co]
// DYBN: dy is staged as ca * dy + cb * y + cc per channel (BatchNorm backward of the layer after the stem)
template <bool DYBN>
__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(ConvStemArgs a) {
  DTF_ABORT_GUARD(a);
  constexpr int DS = 20;   // dy pixel stride in LDS (16 + 4: four pixels of a k-step start 16 banks apart)
  __shared__ __align__(16) float plane[2][kPlane];
  __shared__ __align__(16) float sdy[2][kTH * kW * DS];
  __shared__ float red[4][2][4][64];
  __shared__ __align__(16) float s_dco[3][16];
  const int tid = threadIdx.x;
  if (DYBN && tid < 16) {
    const double inv = (double)a.dy_bn.inv_count;
    const double ds1 = stat_total(a.dy_bn.sums, tid, 16), ds2 = stat_total(a.dy_bn.sums, 16 + tid, 16);
    const double dg1 = stat_total(a.dy_gsums, tid, 16), dg2 = stat_total(a.dy_gsums, 16 + tid, 16);
    const double mean = ds1 * inv;
    double var = ds2 * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = rsqrtf((float)var + a.dy_bn.eps);
    const float k1 = a.dy_bn.gamma[tid] * rstd;
    const float mg = (float)(dg1 * inv), mgx = (float)(dg2 * inv);
    s_dco[0][tid] = k1;
    s_dco[1][tid] = -k1 * mgx * rstd;
    s_dco[2][tid] = k1 * (mgx * (float)mean * rstd - mg);
    if (blockIdx.x == 0) {
      if (a.dy_dbeta) a.dy_dbeta[tid] = (float)dg1;
      if (a.dy_dgamma) a.dy_dgamma[tid] = (float)dg2;
    }
  }
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  for (int i = tid; i < 2 * kPlane; i += 256) (&plane[0][0])[i] = 0.f;
  // A operand rows: window elements k = n (tile 0) and 16 + n (tile 1)
  const int off0 = k_offset(n), off1 = k_offset(16 + n);
  const int total_tiles = 4 * a.B;
  floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

  // the band's 256 pixels x 16 channels are 1024 contiguous 16-byte chunks, four per thread. The
  // prefetch registers are named scalars on purpose: an array handed to a helper by reference is
  // demoted to LDS by the compiler, which turns the prefetch into a synchronous copy.
  const floatx4* dy4 = reinterpret_cast<const floatx4*>(a.dy) + tid;
  const floatx4* yx4 = reinterpret_cast<const floatx4*>(DYBN ? a.dy_x : a.dy) + tid;
#define STEM_FETCH_DY(t)                                        \
  do {                                                          \
    const size_t o_ = (size_t)(t) * (kTH * kW * 4);             \
    pd0 = dy4[o_];                                              \
    pd1 = dy4[o_ + 256];                                        \
    pd2 = dy4[o_ + 512];                                        \
    pd3 = dy4[o_ + 768];                                        \
    if (DYBN) {                                                 \
      pc0 = yx4[o_];                                            \
      pc1 = yx4[o_ + 256];                                      \
      pc2 = yx4[o_ + 512];                                      \
      pc3 = yx4[o_ + 768];                                      \
    }                                                           \
  } while (0)
  float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
  floatx4 pd0 = {0.f, 0.f, 0.f, 0.f}, pd1 = pd0, pd2 = pd0, pd3 = pd0, pc0 = pd0, pc1 = pd0, pc2 = pd0, pc3 = pd0;
  int tile = blockIdx.x;
  if (tile < total_tiles) {
    pf = stem_fetch(a.x, tile, tid);
    STEM_FETCH_DY(tile);
  }
  __syncthreads();
  // every chunk of a thread carries the same four channels 4 (tid % 4) ..
  floatx4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f}, cc = cb;
  if (DYBN) {
    ca = *reinterpret_cast<const floatx4*>(&s_dco[0][4 * (tid & 3)]);
    cb = *reinterpret_cast<const floatx4*>(&s_dco[1][4 * (tid & 3)]);
    cc = *reinterpret_cast<const floatx4*>(&s_dco[2][4 * (tid & 3)]);
  }
  for (int buf = 0; tile < total_tiles; tile += gridDim.x, buf ^= 1) {
    if (tid < kChunks) stem_stage(plane[buf], pf, tid);
    float* sdst = sdy[buf] + (tid >> 2) * DS + 4 * (tid & 3);   // chunk tid + 256 k: pixel tid / 4 + 64 k
    if (DYBN) {
      pd0 = ca * pd0 + (cb * pc0 + cc);
      pd1 = ca * pd1 + (cb * pc1 + cc);
      pd2 = ca * pd2 + (cb * pc2 + cc);
      pd3 = ca * pd3 + (cb * pc3 + cc);
    }
    *reinterpret_cast<floatx4*>(sdst) = pd0;
    *reinterpret_cast<floatx4*>(sdst + 64 * DS) = pd1;
    *reinterpret_cast<floatx4*>(sdst + 128 * DS) = pd2;
    *reinterpret_cast<floatx4*>(sdst + 192 * DS) = pd3;
    __syncthreads();
    const int nxt = tile + gridDim.x;
    if (nxt < total_tiles) {
      pf = stem_fetch(a.x, nxt, tid);
      STEM_FETCH_DY(nxt);
    }
    // the four pixels of a k-step are 4 apart (p = 16 seg + j + 4 g), so that the dy reads of the
    // four lane groups fall into different banks; wave w takes j = w of every segment
    const float* pl = plane[buf];
    const float* sd = sdy[buf];
    // all operands of a half band are read before its MFMAs, so the LDS latency is paid twice per
    // band instead of once per k-step
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float a0[8], a1[8], bv[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int p = (half * 8 + s) * 16 + wave + 4 * g;
        const float* win = pl + (p >> 5) * kRow + (p & 31) * 3;
        a0[s] = win[off0];
        a1[s] = win[off1];
        bv[s] = sd[p * DS + n];
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bv[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bv[s], acc1, 0, 0, 0);
      }
    }
  }
#undef STEM_FETCH_DY
  // lane holds dw[k = 16 t + 4 g + i][co = n]: combine the four waves, one atomic per value
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[wave][0][i][lane] = acc0[i];
    red[wave][1][i][lane] = acc1[i];
  }
  __syncthreads();
  float* dw = a.dw_part ? a.dw_part + (size_t)blockIdx.x * 432 : a.dw + (a.slot_stride ? (size_t)(blockIdx.x & (kStatSlots - 1)) * a.slot_stride : 0);
  for (int e = tid; e < 2 * 4 * 64; e += 256) {
    const int l = e & 63, i = (e >> 6) & 3, t = e >> 8;
    const int k = 16 * t + 4 * (l >> 4) + i, co = l & 15;
    const float s = (red[0][t][i][l] + red[1][t][i][l]) + (red[2][t][i][l] + red[3][t][i][l]);
    if (k < 27) {
      if (a.dw_part) dw[k * 16 + co] = s;            // plain stores: every element of the copy is written
      else if (s != 0.f) atomicAdd(dw + k * 16 + co, s);
    }
  }
}

template <typename K>
int stem_blocks_per_cu(K kernel) {
  int nb = 0;
  DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0));
  return nb < 1 ? 1 : nb;
}

int stem_grid(int total_tiles, int cap) {
  const int iters = ceil_div(total_tiles, cap);
  return ceil_div(total_tiles, iters);
}

}  // namespace

bool conv_stem_supported(int H, int W, int Cin, int Cout, int ks, int stride) {
  return H == kW && W == kW && Cin == 3 && Cout == 16 && ks == 3 && stride == 1;
}

void conv_stem_fwd(const ConvStemArgs& a, hipStream_t stream) {
  if (a.B < 1 || !a.x || !a.w || !a.y) throw std::runtime_error("conv_stem_fwd: null operand or empty batch");
  static const int bpc = stem_blocks_per_cu(conv_stem_fwd_kernel);
  launch("conv_stem_fwd", conv_stem_fwd_kernel, dim3(stem_grid(4 * a.B, 256 * (bpc > 4 ? 4 : bpc))), dim3(256), 0, stream, a);
}

static int stem_wgrad_grid(const ConvStemArgs& a) {
  if (a.dy_x) {
    static const int bpc = stem_blocks_per_cu(conv_stem_wgrad_kernel<true>);
    return stem_grid(4 * a.B, 256 * (bpc > 2 ? 2 : bpc));
  }
  static const int bpc = stem_blocks_per_cu(conv_stem_wgrad_kernel<false>);
  return stem_grid(4 * a.B, 256 * (bpc > 2 ? 2 : bpc));
}

int conv_stem_wgrad_copies(const ConvStemArgs& a) {
  if (a.B < 1) throw std::runtime_error("conv_stem_wgrad: empty batch");
  return stem_wgrad_grid(a);
}

void conv_stem_wgrad(const ConvStemArgs& a, hipStream_t stream) {
  if (a.B < 1 || !a.x || !a.dy || !(a.dw || a.dw_part)) throw std::runtime_error("conv_stem_wgrad: null operand or empty batch");
  if (a.dy_x) {
    if (!a.dy_bn.sums || !a.dy_bn.gamma || !a.dy_gsums) throw std::runtime_error("conv_stem_wgrad: the BatchNorm prologue needs its statistics");
    launch("conv_stem_wgrad", conv_stem_wgrad_kernel<true>, dim3(stem_wgrad_grid(a)), dim3(256), 0, stream, a);
    return;
  }
  launch("conv_stem_wgrad", conv_stem_wgrad_kernel<false>, dim3(stem_wgrad_grid(a)), dim3(256), 0, stream, a);
}

}  // namespace dtf
