// 5x5, stride 1, SAME convolution over 14x14 feature maps: the second convolution of the MNIST
// LeNet (32 -> 64 channels, reference: DistributedLeNet/src/mnist.py:119-123) and, with the taps
// flipped and the channel roles swapped, its data gradient (64 -> 32).
//
// The general implicit-GEMM kernel stages both operands through LDS in k-chunks with a barrier per
// chunk and reached ~30 % of the fp32 MFMA rate on this shape. Here:
//   * a workgroup's tile is IMGS whole images (196 pixels each): the zero-padded 18x18 input planes
//     sit in LDS for all 25 taps, so every input element is read from HBM exactly once and the tap
//     loop has no barrier;
//   * the 5*5*Cin*Cout filter (205 KB) does not fit in LDS next to them and does not need to: one
//     tap's 16 x Cin slice per wave is a few 16-byte loads per lane straight from L2 into the MFMA
//     A registers, fetched one tap ahead (a tap is >= 3500 MFMA cycles per wave);
//   * M = output channels, N = pixels, so a lane ends up with 4 consecutive output channels of one
//     pixel (one 16-byte store); k is permuted so one ds_read_b128 feeds four MFMAs;
//   * the next tile's input is prefetched into registers while the current one is multiplied.
// 196 is not a multiple of 16: pixel tiles run over the linearised pixels of the IMGS images
// (2 images = 24.5 tiles, 2 % padding) with per-lane window offsets computed once.
#include "common.h"
#include "conv.h"

namespace dtf {
namespace {

constexpr int kHW = 14;            // feature-map height and width
constexpr int kPW = kHW + 4;       // padded plane (halo 2 on every side)
constexpr int kImgPx = kHW * kHW;  // 196

// The 64-channel gradient at the second convolution's output is the max-pool backward of the
// gradient dp at the pooled tensor: dy[y][x][c] = dp[y/2][x/2][c] if (y, x) is the window's winner
// (arg, encoded as in maxpool2x2_fwd) and the pooled maximum p is positive (ReLU), else 0. With
// `pool_dp` set, the kernels below stage dy from (dp, arg, p) directly, one 16-byte chunk of a
// pooled cell feeding its four pixels, so the full-resolution gradient is never written.
constexpr int kCells = 49;                 // 7 x 7 pooled cells per image
constexpr int kCellChunks = kCells * 16;   // 16-byte chunks (4 of 64 channels) per image
constexpr int kNpc = (kCellChunks + 255) / 256;
struct PooledChunk {
  floatx4 dp, p;
  unsigned arg;
};
__device__ __forceinline__ void pooled_fetch(const float* dp, const unsigned char* arg, const float* p, int img, int cc, PooledChunk& o) {
  const size_t off = ((size_t)img * kCells + (cc >> 4)) * 64 + 4 * (cc & 15);
  o.dp = *reinterpret_cast<const floatx4*>(dp + off);
  o.p = *reinterpret_cast<const floatx4*>(p + off);
  o.arg = *reinterpret_cast<const unsigned*>(arg + off);
}
// value of window position pos (0..3) for the chunk
__device__ __forceinline__ float4 pooled_value(const PooledChunk& c, unsigned pos) {
  float4 v;
  v.x = ((c.arg & 0xffu) == pos && c.p.x > 0.f) ? c.dp.x : 0.f;
  v.y = (((c.arg >> 8) & 0xffu) == pos && c.p.y > 0.f) ? c.dp.y : 0.f;
  v.z = (((c.arg >> 16) & 0xffu) == pos && c.p.z > 0.f) ? c.dp.z : 0.f;
  v.w = ((c.arg >> 24) == pos && c.p.w > 0.f) ? c.dp.w : 0.f;
  return v;
}

template <int CIN, int COUT, int IMGS, int CSPLIT = 1>
struct C5 {
  static constexpr int CW = COUT / CSPLIT;           // output channels of one workgroup (blockIdx.y picks the slice)
  static constexpr int PS = CIN + 8;                 // LDS pixel stride (floats): +8 halves the bank conflicts of the pixel-operand reads against +4 (14-pixel rows never line up with the 16-lane read groups)
  static constexpr int PX = IMGS * kImgPx;           // pixels per tile
  static constexpr int NT = (PX + 15) / 16;          // 16-pixel MFMA tiles per workgroup tile
  static constexpr int MT = CW / 16;                 // 16-channel output tiles
  static constexpr int WPM = 4 / MT;                 // waves sharing one output-channel tile
  static constexpr int NTW = (NT + WPM - 1) / WPM;   // pixel tiles per wave
  static constexpr int KG = CIN / 16;                // 16-channel groups per tap
  static constexpr int C4 = CIN / 4;
  static constexpr int CHUNKS = PX * C4;             // 16-byte chunks of a tile's input
  static constexpr int NPF = (CHUNKS + 255) / 256;
  static constexpr int LDS_PLANE_FLOATS = IMGS * kPW * kPW * PS;
  static constexpr int OS = CW + 4;                  // pixel stride of the output tile parked in LDS (fused pool)
  static constexpr int LDS_OUT_FLOATS = PX * OS;
  static constexpr int LDS_FLOATS = LDS_PLANE_FLOATS;
  static_assert(CW == 16 || CW == 32 || CW == 64, "4 waves cover 1, 2 or 4 output-channel tiles");
  static_assert(CIN % 16 == 0, "input channels come in groups of 16");
};

// x [B,14,14,CIN]; w [25][COUT][CIN] (FLIP: tap t reads w[24-t]); y [B,14,14,COUT].
// CSPLIT = 2: two workgroups share an image, each computing half of the output channels from its
// own copy of the input planes - for batches with fewer images than CUs, where half of the chip
// would otherwise have no workgroup.
template <int CIN, int COUT, int IMGS, bool FLIP, bool BIAS_RELU, bool POOLED, bool POOLOUT, int CSPLIT>
__global__ __launch_bounds__(256) void conv5x5_tile_kernel(Conv5x5Args a) {
  DTF_ABORT_GUARD(a);
  using G = C5<CIN, COUT, IMGS, CSPLIT>;
  const int cbase = CSPLIT > 1 ? (int)blockIdx.y * G::CW : 0;   // first output channel of this workgroup
  static_assert(!POOLED || (CIN == 64 && IMGS == 1), "pooled staging is the data-gradient form");
  static_assert(!POOLOUT || (BIAS_RELU && COUT == 64), "the fused max pool follows the forward form");
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int mtile = wave % G::MT, nsub = wave / G::MT;
  const int co0 = mtile * 16;
  const int n_tiles = (a.B + IMGS - 1) / IMGS;

  // zero the planes once: the halo is never written again
  for (int i = tid; i < G::LDS_PLANE_FLOATS / 4; i += 256) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // window origin (top-left of the 5x5 footprint in the padded plane) of this lane's pixel in each
  // of the wave's pixel tiles, plus the lane's channel offset inside a 16-channel group
  int boff[G::NTW];
#pragma unroll
  for (int i = 0; i < G::NTW; ++i) {
    int q = (nsub * G::NTW + i) * 16 + n;
    if (q > G::PX - 1) q = G::PX - 1;
    const int img = q / kImgPx, r = q % kImgPx;
    boff[i] = ((img * kPW + r / kHW) * kPW + r % kHW) * G::PS + 4 * g;
  }
  // staging: chunk c of the tile's input is contiguous in global memory; its LDS home is fixed
  int sdst[G::NPF];
#pragma unroll
  for (int k = 0; k < G::NPF; ++k) {
    const int c = tid + 256 * k;
    const int px = (c < G::CHUNKS ? c : 0) / G::C4, c4 = c % G::C4;
    const int img = px / kImgPx, r = px % kImgPx;
    sdst[k] = ((img * kPW + r / kHW + 2) * kPW + r % kHW + 2) * G::PS + 4 * c4;
  }
  float4 pf[POOLED ? 1 : G::NPF];
  PooledChunk pc[POOLED ? kNpc : 1];
  int pdst[POOLED ? kNpc : 1];
  if constexpr (POOLED) {
#pragma unroll
    for (int k = 0; k < kNpc; ++k) {
      const int cc = min(tid + 256 * k, kCellChunks - 1);
      const int cell = cc >> 4;
      pdst[k] = ((2 * (cell / 7) + 2) * kPW + 2 * (cell % 7) + 2) * G::PS + 4 * (cc & 15);
    }
  }
  auto fetch = [&](int tile) {
    if constexpr (POOLED) {
#pragma unroll
      for (int k = 0; k < kNpc; ++k)
        if (tid + 256 * k < kCellChunks) pooled_fetch(a.pool_dp, a.pool_arg, a.pool_p, tile, tid + 256 * k, pc[k]);
    } else {
      const int valid = min(G::PX, (a.B - tile * IMGS) * kImgPx) * G::C4;
      const float4* src = reinterpret_cast<const float4*>(a.x + (size_t)tile * G::PX * CIN);
#pragma unroll
      for (int k = 0; k < G::NPF; ++k) {
        const int c = tid + 256 * k;
        pf[k] = c < valid ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (BIAS_RELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = a.bias[cbase + co0 + 4 * g + i];
  }
  // this lane's filter row: output channel co0 + n, input channels 16 j + 4 g + (0..3)
  const float* wrow = a.w + (size_t)(cbase + co0 + n) * CIN + 4 * g;
  constexpr int W_TAP = COUT * CIN;

  int tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();   // every wave is done reading the previous tile (and the zero fill landed)
    if constexpr (POOLED) {
#pragma unroll
      for (int k = 0; k < kNpc; ++k)
        if (tid + 256 * k < kCellChunks) {
          float* d = lds + pdst[k];
          *reinterpret_cast<float4*>(d) = pooled_value(pc[k], 0);
          *reinterpret_cast<float4*>(d + G::PS) = pooled_value(pc[k], 1);
          *reinterpret_cast<float4*>(d + kPW * G::PS) = pooled_value(pc[k], 2);
          *reinterpret_cast<float4*>(d + (kPW + 1) * G::PS) = pooled_value(pc[k], 3);
        }
    } else {
#pragma unroll
      for (int k = 0; k < G::NPF; ++k)
        if (tid + 256 * k < G::CHUNKS) *reinterpret_cast<float4*>(lds + sdst[k]) = pf[k];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);

    floatx4 acc[G::NTW];
#pragma unroll
    for (int i = 0; i < G::NTW; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};

    float4 wcur[G::KG], wnxt[G::KG];
    {
      const float* w0 = wrow + (size_t)(FLIP ? 24 : 0) * W_TAP;
#pragma unroll
      for (int j = 0; j < G::KG; ++j) wcur[j] = *reinterpret_cast<const float4*>(w0 + 16 * j);
    }
    int toff = 0;   // LDS offset of the tap inside the window
#pragma unroll 1
    for (int kh = 0; kh < 5; ++kh) {
#pragma unroll 1
      for (int kw = 0; kw < 5; ++kw) {
        const int tap = kh * 5 + kw;
        if (tap < 24) {
          const float* wn = wrow + (size_t)(FLIP ? 23 - tap : tap + 1) * W_TAP;
#pragma unroll
          for (int j = 0; j < G::KG; ++j) wnxt[j] = *reinterpret_cast<const float4*>(wn + 16 * j);
        }
        const float* base = lds + toff;
#pragma unroll
        for (int j = 0; j < G::KG; ++j) {
          // four pixel tiles at a time, so that consecutive MFMAs write different accumulators.
          // A wave whose last pixel tile lies beyond the image multiplies a clamped copy of the
          // last pixel instead of branching (the result is not stored).
#pragma unroll
          for (int i0 = 0; i0 < G::NTW; i0 += 4) {
            constexpr int kGroup = 4;
            float4 b[kGroup];
#pragma unroll
            for (int u = 0; u < kGroup; ++u)
              if (i0 + u < G::NTW) b[u] = *reinterpret_cast<const float4*>(base + boff[i0 + u] + 16 * j);
#pragma unroll
            for (int u = 0; u < kGroup; ++u)
              if (i0 + u < G::NTW) acc[i0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[j].x, b[u].x, acc[i0 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < kGroup; ++u)
              if (i0 + u < G::NTW) acc[i0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[j].y, b[u].y, acc[i0 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < kGroup; ++u)
              if (i0 + u < G::NTW) acc[i0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[j].z, b[u].z, acc[i0 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < kGroup; ++u)
              if (i0 + u < G::NTW) acc[i0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[j].w, b[u].w, acc[i0 + u], 0, 0, 0);
          }
        }
#pragma unroll
        for (int j = 0; j < G::KG; ++j) wcur[j] = wnxt[j];
        toff += G::PS;
      }
      toff += (kPW - 5) * G::PS;
    }

    // lane holds output channels co0 + 4g .. +3 of pixel 16 t + n
    const int valid_px = min(G::PX, (a.B - tile * IMGS) * kImgPx);
    if constexpr (POOLOUT) __syncthreads();   // the input planes are about to be overwritten by the outputs
    float* yt = a.y + (size_t)tile * G::PX * COUT + cbase + co0 + 4 * g;
#pragma unroll
    for (int i = 0; i < G::NTW; ++i) {
      const int q = (nsub * G::NTW + i) * 16 + n;
      if (q < valid_px) {
        float4 o;
        o.x = acc[i][0] + bias4[0];
        o.y = acc[i][1] + bias4[1];
        o.z = acc[i][2] + bias4[2];
        o.w = acc[i][3] + bias4[3];
        if constexpr (BIAS_RELU) {
          o.x = fmaxf(o.x, 0.f);
          o.y = fmaxf(o.y, 0.f);
          o.z = fmaxf(o.z, 0.f);
          o.w = fmaxf(o.w, 0.f);
        }
        if constexpr (POOLOUT) *reinterpret_cast<float4*>(lds + q * G::OS + co0 + 4 * g) = o;
        else *reinterpret_cast<float4*>(yt + (size_t)q * COUT) = o;
      }
    }
    if constexpr (POOLOUT) {
      // 2x2 / 2 max pool of the tile from LDS: pooled maximum + winner position (first maximum
      // wins, positions numbered (0,0) (0,1) (1,0) (1,1) as in maxpool2x2_fwd); the full-resolution
      // activation is not written.
      __syncthreads();
      const int cells_valid = min(IMGS, a.B - tile * IMGS) * kCells;
      constexpr int CH4 = G::CW / 4;   // 16-byte chunks per pooled cell of this workgroup's channel slice
      for (int cc = tid; cc < IMGS * kCells * CH4; cc += 256) {
        const int cell = cc / CH4, c4 = cc % CH4;
        if (cell >= cells_valid) break;
        const int img = cell / kCells, r = cell % kCells;
        const float* o0 = lds + (img * kImgPx + 2 * (r / 7) * kHW + 2 * (r % 7)) * G::OS + 4 * c4;
        const float4 v0 = *reinterpret_cast<const float4*>(o0);
        const float4 v1 = *reinterpret_cast<const float4*>(o0 + G::OS);
        const float4 v2 = *reinterpret_cast<const float4*>(o0 + kHW * G::OS);
        const float4 v3 = *reinterpret_cast<const float4*>(o0 + (kHW + 1) * G::OS);
        float4 best = v0;
        unsigned arg = 0;
        auto upd = [&](float& b, float v, unsigned pos, int e) {
          if (v > b) { b = v; arg = (arg & ~(0xffu << (8 * e))) | (pos << (8 * e)); }
        };
        upd(best.x, v1.x, 1, 0); upd(best.y, v1.y, 1, 1); upd(best.z, v1.z, 1, 2); upd(best.w, v1.w, 1, 3);
        upd(best.x, v2.x, 2, 0); upd(best.y, v2.y, 2, 1); upd(best.z, v2.z, 2, 2); upd(best.w, v2.w, 2, 3);
        upd(best.x, v3.x, 3, 0); upd(best.y, v3.y, 3, 1); upd(best.z, v3.z, 3, 2); upd(best.w, v3.w, 3, 3);
        const size_t off = ((size_t)tile * IMGS * kCells + cell) * COUT + cbase + 4 * c4;
        *reinterpret_cast<float4*>(a.pool_out + off) = best;
        *reinterpret_cast<unsigned*>(a.pool_arg_out + off) = arg;
      }
      // the planes' halo was overwritten: restore the zeros before the next tile is staged
      __syncthreads();
      for (int i = tid; i < G::LDS_PLANE_FLOATS / 4; i += 256) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// ---- filter gradient of the same layer: dw[tap][ci][co] += sum over pixels x[p + tap][ci] * dy[p][co]
//
// One image per workgroup tile: its zero-padded input plane and its output gradient sit in LDS, k
// runs over the 196 pixels (49 MFMA k-steps, no padding), and the 25 x 32 x 64 outputs live in
// the accumulators of the four waves for the whole kernel (wave = one 16-channel slice of ci x two
// of co, all 25 taps: 50 tiles, 200 registers), so the pixel loop has no barrier and no LDS
// writes. Per k-step a wave reads its two dy fragments once and one shifted x fragment per tap:
// 27 ds_read_b32 feed 50 MFMAs. Pixel strides are = 16 mod 64 banks, so the four pixel groups of
// a fragment read disjoint banks, also where a row of 14 pixels wraps. Blocks are persistent over
// images and add their result once, into the gradient copy of their XCD.
constexpr int kWgPsx = 48;   // x plane pixel stride (32 channels + 16)
constexpr int kWgPsy = 80;   // dy pixel stride (64 channels + 16)
constexpr int kWgXFloats = kPW * kPW * kWgPsx;
constexpr int kWgYFloats = kImgPx * kWgPsy;
constexpr int kWgXChunks = kImgPx * 8, kWgYChunks = kImgPx * 16;
constexpr int kWgNpfX = (kWgXChunks + 255) / 256, kWgNpfY = (kWgYChunks + 255) / 256;

// TSPLIT = 2: two workgroups per image, each with its own copy of the staged image and half of the
// 25 taps (13 and 12) - for batches with fewer images than CUs.
template <bool POOLED, int TSPLIT>
__global__ __launch_bounds__(256) void conv5x5_wgrad_kernel(Conv5x5WgradArgs a) {
  DTF_ABORT_GUARD(a);
  constexpr int NTAP = (25 + TSPLIT - 1) / TSPLIT;             // tap slots of this workgroup
  const int tap0 = TSPLIT > 1 ? (int)blockIdx.y * NTAP : 0;    // its first tap
  int tap_off[NTAP];                                           // window offset of every tap slot (uniform)
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tap = min(tap0 + t, 24);                         // a slot past the last tap reads tap 24 and is not used
    tap_off[t] = ((tap / 5) * kPW + tap % 5) * kWgPsx;
  }
  extern __shared__ __align__(16) float lds[];
  float* sx = lds;
  float* sy = lds + kWgXFloats;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int v = wave >> 1, h = wave & 1;   // ci slice 16v.., co slices 32h.. and 32h+16..

  for (int i = tid; i < kWgXFloats / 4; i += 256) reinterpret_cast<float4*>(sx)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  int xdst[kWgNpfX], ydst[kWgNpfY];
#pragma unroll
  for (int k = 0; k < kWgNpfX; ++k) {
    const int c = tid + 256 * k;
    const int px = (c < kWgXChunks ? c : 0) >> 3, c4 = c & 7;
    xdst[k] = ((px / kHW + 2) * kPW + px % kHW + 2) * kWgPsx + 4 * c4;
  }
#pragma unroll
  for (int k = 0; k < kWgNpfY; ++k) {
    const int c = tid + 256 * k;
    ydst[k] = ((c < kWgYChunks ? c : 0) >> 4) * kWgPsy + 4 * (c & 15);
  }
  // native vector types: assigning a float4 (a struct) from global memory is a memcpy into the
  // array, which then stays in scratch memory and makes the prefetch synchronous
  floatx4 pfx[kWgNpfX], pfy[POOLED ? 1 : kWgNpfY];
  PooledChunk pc[POOLED ? kNpc : 1];
  int pdst[POOLED ? kNpc : 1];
  if constexpr (POOLED) {
#pragma unroll
    for (int k = 0; k < kNpc; ++k) {
      const int cc = min(tid + 256 * k, kCellChunks - 1);
      const int cell = cc >> 4;
      pdst[k] = (2 * (cell / 7) * kHW + 2 * (cell % 7)) * kWgPsy + 4 * (cc & 15);
    }
  }
  auto fetch = [&](int img) {
    const floatx4* xs = reinterpret_cast<const floatx4*>(a.x + (size_t)img * kImgPx * 32);
#pragma unroll
    for (int k = 0; k < kWgNpfX; ++k)
      if (tid + 256 * k < kWgXChunks) pfx[k] = xs[tid + 256 * k];
    if constexpr (POOLED) {
#pragma unroll
      for (int k = 0; k < kNpc; ++k)
        if (tid + 256 * k < kCellChunks) pooled_fetch(a.pool_dp, a.pool_arg, a.pool_p, img, tid + 256 * k, pc[k]);
    } else {
      const floatx4* ys = reinterpret_cast<const floatx4*>(a.dy + (size_t)img * kImgPx * 64);
#pragma unroll
      for (int k = 0; k < kWgNpfY; ++k)
        if (tid + 256 * k < kWgYChunks) pfy[k] = ys[tid + 256 * k];
    }
  };

  floatx4 acc[NTAP][2];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) acc[t][0] = acc[t][1] = floatx4{0.f, 0.f, 0.f, 0.f};

  int img = blockIdx.x;
  if (img < a.B) fetch(img);
  for (; img < a.B; img += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWgNpfX; ++k)
      if (tid + 256 * k < kWgXChunks) *reinterpret_cast<floatx4*>(sx + xdst[k]) = pfx[k];
    if constexpr (POOLED) {
#pragma unroll
      for (int k = 0; k < kNpc; ++k)
        if (tid + 256 * k < kCellChunks) {
          float* d = sy + pdst[k];
          *reinterpret_cast<float4*>(d) = pooled_value(pc[k], 0);
          *reinterpret_cast<float4*>(d + kWgPsy) = pooled_value(pc[k], 1);
          *reinterpret_cast<float4*>(d + kHW * kWgPsy) = pooled_value(pc[k], 2);
          *reinterpret_cast<float4*>(d + (kHW + 1) * kWgPsy) = pooled_value(pc[k], 3);
        }
    } else {
#pragma unroll
      for (int k = 0; k < kWgNpfY; ++k)
        if (tid + 256 * k < kWgYChunks) *reinterpret_cast<floatx4*>(sy + ydst[k]) = pfy[k];
    }
    __syncthreads();
    if (img + (int)gridDim.x < a.B) fetch(img + gridDim.x);

    // This lane's pixel of k-step ks is 4 ks + g, walked through the rows of 14. The 27 operand
    // reads of k-step ks + 1 are issued before the 50 MFMAs of k-step ks (two register sets,
    // the loop body handles two k-steps), so no MFMA waits for LDS.
    int px_y = 0, px_x = g;
    auto load_step = [&](float (&av)[NTAP], float (&bv)[2]) {
      const float* yrow = sy + (px_y * kHW + px_x) * kWgPsy + 32 * h + n;
      bv[0] = yrow[0];
      bv[1] = yrow[16];
      const float* xwin = sx + (px_y * kPW + px_x) * kWgPsx + 16 * v + n;   // window origin of the pixel
      if constexpr (TSPLIT == 1) {
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
          for (int kw = 0; kw < 5; ++kw) av[kh * 5 + kw] = xwin[(kh * kPW + kw) * kWgPsx];   // immediates
      } else {
#pragma unroll
        for (int t = 0; t < NTAP; ++t) av[t] = xwin[tap_off[t]];
      }
      px_x += 4;
      if (px_x >= kHW) { px_x -= kHW; ++px_y; }
    };
    auto mfma_step = [&](const float (&av)[NTAP], const float (&bv)[2]) {
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[0], acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[1], acc[t][1], 0, 0, 0);
      }
    };
    constexpr int kSteps = kImgPx / 4;   // 49
    float a0[NTAP], a1[NTAP], b0[2], b1[2];
    load_step(a0, b0);
#pragma unroll 1
    for (int ks = 0; ks + 2 < kSteps; ks += 2) {
      load_step(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_step(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_step(a0, b0);   // k-step 48 (the loop covers 0..47)
  }

  // lane holds dw[tap][ci = 16v + 4g + i][co = 32h + 16j + n]
  const int lane_off = (16 * v + 4 * g) * 64 + 32 * h + n;
  if (a.partials) {
    float* dst = a.partials + (size_t)blockIdx.x * (25 * 2048) + (size_t)tap0 * 2048 + lane_off;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (tap0 + t < 25) {   // (a break would keep the loop rolled and the accumulators in scratch memory)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[t * 2048 + i * 64 + 16 * j] = acc[t][j][i];
      }
    }
    return;
  }
  float* dst = a.dw + (a.slot_stride ? (size_t)(blockIdx.x & (kStatSlots - 1)) * a.slot_stride : 0) + (size_t)tap0 * 2048 + lane_off;
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    if (tap0 + t < 25) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float s = acc[t][j][i];
          if (s != 0.f) atomicAdd(dst + t * 2048 + i * 64 + 16 * j, s);
        }
    }
  }
}

// [25][CIN][COUT] (HWIO filter) -> [25][COUT][CIN]
__global__ __launch_bounds__(256) void conv5x5_pack_kernel(Conv5x5PackArgs a) {
  const int per_tap = a.Cin * a.Cout;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 25 * per_tap) return;
  const int tap = i / per_tap, r = i % per_tap;
  const int co = r / a.Cin, ci = r % a.Cin;
  a.out[i] = a.w[(size_t)tap * per_tap + ci * a.Cout + co];
}

template <int CIN, int COUT, int IMGS, bool FLIP, bool BIAS_RELU, bool POOLED = false, bool POOLOUT = false, int CSPLIT = 1>
void launch_c5(const Conv5x5Args& a, hipStream_t stream) {
  using G = C5<CIN, COUT, IMGS, CSPLIT>;
  auto kernel = conv5x5_tile_kernel<CIN, COUT, IMGS, FLIP, BIAS_RELU, POOLED, POOLOUT, CSPLIT>;
  const size_t shmem = (size_t)(POOLOUT && G::LDS_OUT_FLOATS > G::LDS_PLANE_FLOATS ? G::LDS_OUT_FLOATS : G::LDS_PLANE_FLOATS) * sizeof(float);
  static const int bpc = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), shmem);
    int nb = 0;
    DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, shmem));
    return nb < 1 ? 1 : nb;
  }();
  const int n_tiles = (a.B + IMGS - 1) / IMGS;
  int cap = 256 * bpc / CSPLIT;
  if (cap < 1) cap = 1;
  const int iters = ceil_div(n_tiles, cap);
  const int grid = ceil_div(n_tiles, iters);
  launch("conv5x5_tile", kernel, dim3(grid, CSPLIT), dim3(256), shmem, stream, a);
}

}  // namespace

bool conv5x5_tile_supported(int H, int W, int Cin, int Cout) {
  return H == kHW && W == kHW && ((Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 32));
}

void conv5x5_tile(const Conv5x5Args& a, hipStream_t stream) {
  if (a.B < 1) throw std::runtime_error("conv5x5_tile: empty batch");
  if (a.Cin == 32 && a.Cout == 64 && !a.flip) {
    if (!a.bias) throw std::runtime_error("conv5x5_tile: the 32->64 forward form needs a bias");
    // two images per tile halve the padding of the last pixel tile; one image per tile keeps
    // every CU busy at small batches
    const bool two = a.imgs == 2 || (a.imgs == 0 && a.B >= 512);
    if (a.pool_out) {
      if (!a.pool_arg_out) throw std::runtime_error("conv5x5_tile: the fused pool writes maxima and winner positions");
      if (two) return launch_c5<32, 64, 2, false, true, false, true>(a, stream);
      // fewer images than CUs: two workgroups per image, half of the output channels each
      // (LeNet step at batch 128: 0.266 -> 0.247 ms; no difference once every CU has an image)
      if (a.B < 256) return launch_c5<32, 64, 1, false, true, false, true, 2>(a, stream);
      return launch_c5<32, 64, 1, false, true, false, true>(a, stream);
    }
    if (!a.y) throw std::runtime_error("conv5x5_tile: null output");
    if (two) return launch_c5<32, 64, 2, false, true>(a, stream);
    if (a.B < 256) return launch_c5<32, 64, 1, false, true, false, false, 2>(a, stream);
    return launch_c5<32, 64, 1, false, true>(a, stream);
  }
  if (a.Cin == 64 && a.Cout == 32 && a.flip) {
    if (a.pool_dp) {
      if (!a.pool_arg || !a.pool_p) throw std::runtime_error("conv5x5_tile: pooled staging needs dp, arg and p");
      if (a.B < 256) return launch_c5<64, 32, 1, true, false, true, false, 2>(a, stream);   // fewer images than CUs
      return launch_c5<64, 32, 1, true, false, true>(a, stream);
    }
    if (!a.x) throw std::runtime_error("conv5x5_tile: null input");
    if (a.B < 256) return launch_c5<64, 32, 1, true, false, false, false, 2>(a, stream);
    return launch_c5<64, 32, 1, true, false>(a, stream);
  }
  throw std::runtime_error("conv5x5_tile: unsupported channel counts / direction");
}

int conv5x5_wgrad_parts(int B) {
  const int cap = 256;   // one workgroup per CU (125 KB of LDS)
  return ceil_div(B, ceil_div(B, cap));
}

void conv5x5_wgrad(const Conv5x5WgradArgs& a, hipStream_t stream) {
  if (a.B < 1) throw std::runtime_error("conv5x5_wgrad: empty batch");
  const size_t shmem = (size_t)(kWgXFloats + kWgYFloats) * sizeof(float);
  static const bool once = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv5x5_wgrad_kernel<false, 1>), shmem);
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv5x5_wgrad_kernel<true, 1>), shmem);
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv5x5_wgrad_kernel<false, 2>), shmem);
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv5x5_wgrad_kernel<true, 2>), shmem);
    return true;
  }();
  (void)once;
  const int grid = conv5x5_wgrad_parts(a.B);
  // fewer images than CUs: two workgroups per image, 13 and 12 of the 25 taps each
  const bool split = a.B < 256;
  const dim3 g(grid, split ? 2 : 1);
  if (a.pool_dp) {
    if (!a.pool_arg || !a.pool_p) throw std::runtime_error("conv5x5_wgrad: pooled staging needs dp, arg and p");
    if (split) launch("conv5x5_wgrad", conv5x5_wgrad_kernel<true, 2>, g, dim3(256), shmem, stream, a);
    else launch("conv5x5_wgrad", conv5x5_wgrad_kernel<true, 1>, g, dim3(256), shmem, stream, a);
  } else {
    if (!a.dy) throw std::runtime_error("conv5x5_wgrad: null output gradient");
    if (split) launch("conv5x5_wgrad", conv5x5_wgrad_kernel<false, 2>, g, dim3(256), shmem, stream, a);
    else launch("conv5x5_wgrad", conv5x5_wgrad_kernel<false, 1>, g, dim3(256), shmem, stream, a);
  }
}

void conv5x5_pack(const Conv5x5PackArgs& a, hipStream_t stream) {
  const int total = 25 * a.Cin * a.Cout;
  launch("conv5x5_pack", conv5x5_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, a);
}

}  // namespace dtf
