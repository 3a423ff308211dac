// The LeNet's second convolution (5x5, SAME, 14x14 maps, 32 -> 64 channels; conv5x5_tile.hip) and its data
// gradient on the bf16 matrix pipe with fp32 operands split three ways ("bf16x6", conv_fprop_split.hip):
// six v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation. Opt-in; tile walk, fused bias + ReLU +
// 2x2 max pool (forward) and fused pool + ReLU backward while staging (data gradient) as in the exact
// kernel:
//   * a workgroup's tile is one whole image: its zero-padded 18x18 input sits in LDS for all 25 taps as
//     three bf16 planes (h, m, l parts; split while it is staged), pixels 96 bytes (32 channels) or 144
//     bytes (64 channels) apart;
//   * the filter is packed once per step into MFMA A fragments [tap][channel tile][K step][h, m, l][lane]
//     (for the data gradient with the taps reversed and the channel roles swapped); a wave's fragments of
//     a tap are three (six) 16-byte loads per lane straight from L2, fetched one tap ahead;
//   * M = output channels, N = pixels: a lane ends up with 4 consecutive channels of one pixel; four pixel
//     tiles run interleaved so that consecutive MFMAs write different accumulators (one wave per SIMD).
// (reference: DistributedLeNet/src/mnist.py:119-123 and the conv2d_backprop_input of its gradient)
#include "conv.h"
#include "split_common.h"

namespace dtf {
namespace {

constexpr int kHW = 14;            // feature-map height and width
constexpr int kPW = kHW + 4;       // padded plane (halo 2 on every side)
constexpr int kImgPx = kHW * kHW;  // 196
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
// value of window position pos (0..3) for the chunk: the pool's winner with a positive maximum passes dp
__device__ __forceinline__ floatx4 pooled_value(const PooledChunk& c, unsigned pos) {
  floatx4 v;
  v[0] = ((c.arg & 0xffu) == pos && c.p[0] > 0.f) ? c.dp[0] : 0.f;
  v[1] = (((c.arg >> 8) & 0xffu) == pos && c.p[1] > 0.f) ? c.dp[1] : 0.f;
  v[2] = (((c.arg >> 16) & 0xffu) == pos && c.p[2] > 0.f) ? c.dp[2] : 0.f;
  v[3] = ((c.arg >> 24) == pos && c.p[3] > 0.f) ? c.dp[3] : 0.f;
  return v;
}

template <int CIN, int COUT, int CSPLIT>
struct S5 {
  static constexpr int CW = COUT / CSPLIT;           // output channels of one workgroup (blockIdx.y picks the slice)
  static constexpr int KS = CIN / 32;                // K = 32 steps per tap
  static constexpr int PSB = CIN == 32 ? 96 : 144;   // bytes per pixel and plane: an odd number of 16-byte slots (+2, +1 of padding)
  // (16 pixels across rows of 14 make about every second 16-lane group of a ds_read_b128 two-way
  // conflicted; 128 bytes of row padding remove that at 32 channels and change no measured time)
  static constexpr int ROWB = kPW * PSB;             // bytes per plane row
  static constexpr int PLANE = kPW * ROWB;           // bytes per plane
  static constexpr int NT = (kImgPx + 15) / 16;      // 13 pixel tiles (the last one a quarter full)
  static constexpr int MT = CW / 16;                 // output-channel tiles of the workgroup
  static constexpr int MT_ALL = COUT / 16;
  static constexpr int MW = MT >= 2 ? 2 : 1;         // output-channel tiles per wave: each pixel operand read feeds MW tiles
  static constexpr int MG = MT / MW;                 // wave groups along the output channels
  static constexpr int WPM = 4 / MG;                 // waves sharing one group of output-channel tiles
  static constexpr int NTW = (NT + WPM - 1) / WPM;   // pixel tiles per wave
  static constexpr int C4 = CIN / 4;
  static constexpr int CHUNKS = kImgPx * C4;
  static constexpr int NPF = (CHUNKS + 255) / 256;
  static constexpr int OS = CW + 4;                  // pixel stride (floats) of the output tile parked in LDS (fused pool)
  static constexpr int OUT_BYTES = kImgPx * OS * 4;
  static constexpr int LDS_BYTES = 3 * PLANE > OUT_BYTES ? 3 * PLANE : OUT_BYTES;
  static_assert(CW == 16 || CW == 32 || CW == 64, "4 waves cover 1, 2 or 4 output-channel tiles");
  static_assert(CIN % 32 == 0 && LDS_BYTES <= (int)kLdsBudget, "whole K steps; one workgroup per CU");
};

// x [B,14,14,CIN]; wpk: conv5x5_split_pack's output; y [B,14,14,COUT].
template <int CIN, int COUT, bool BIAS_RELU, bool POOLED, bool POOLOUT, int CSPLIT>
__global__ __launch_bounds__(256) void conv5x5_split_kernel(Conv5x5Args a) {
  DTF_ABORT_GUARD(a);
  using G = S5<CIN, COUT, CSPLIT>;
  constexpr int KS = G::KS, PSB = G::PSB, ROWB = G::ROWB, PLANE = G::PLANE, NTW = G::NTW;
  const int cbase = CSPLIT > 1 ? (int)blockIdx.y * G::CW : 0;   // first output channel of this workgroup
  static_assert(!POOLED || CIN == 64, "pooled staging is the data-gradient form");
  static_assert(!POOLOUT || (BIAS_RELU && COUT == 64), "the fused max pool follows the forward form");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_b[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  constexpr int MW = G::MW;
  const int mtile = (wave % G::MG) * MW, nsub = wave / G::MG;   // first of the wave's MW output-channel tiles
  const int co0 = mtile * 16;

  // zero the planes once: the halo is never written again
  for (int i = tid; i < 3 * PLANE / 16; i += 256) reinterpret_cast<uintx4*>(lds_b)[i] = uintx4{0u, 0u, 0u, 0u};

  // window origin (top-left of the 5x5 footprint in the padded plane) of this lane's pixel in each of the
  // wave's pixel tiles, plus the lane group's channel slot (channels 8 q .. 8 q + 7 of a K step)
  int boff[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    int p = (nsub * NTW + i) * 16 + n;
    if (p > kImgPx - 1) p = kImgPx - 1;   // a tile beyond the image multiplies a copy of the last pixel (not stored)
    boff[i] = (p / kHW) * ROWB + (p % kHW) * PSB + 16 * q;
  }
  // staging: chunk c (4 channels of a pixel) of the image is contiguous in global memory; its LDS home is fixed
  int sdst[G::NPF];
#pragma unroll
  for (int k = 0; k < G::NPF; ++k) {
    const int c = tid + 256 * k;
    const int px = (c < G::CHUNKS ? c : 0) / G::C4, c4 = c % G::C4;
    sdst[k] = (px / kHW + 2) * ROWB + (px % kHW + 2) * PSB + 8 * c4;
  }
  floatx4 pf[POOLED ? 1 : G::NPF];
  PooledChunk pc[POOLED ? kNpc : 1];
  int pdst[POOLED ? kNpc : 1];
  if constexpr (POOLED) {
#pragma unroll
    for (int k = 0; k < kNpc; ++k) {
      const int cc = min(tid + 256 * k, kCellChunks - 1);
      const int cell = cc >> 4;
      pdst[k] = (2 * (cell / 7) + 2) * ROWB + (2 * (cell % 7) + 2) * PSB + 8 * (cc & 15);
    }
  }
  auto fetch = [&](int img) {
    if constexpr (POOLED) {
#pragma unroll
      for (int k = 0; k < kNpc; ++k)
        if (tid + 256 * k < kCellChunks) pooled_fetch(a.pool_dp, a.pool_arg, a.pool_p, img, tid + 256 * k, pc[k]);
    } else {
      const floatx4* src = reinterpret_cast<const floatx4*>(a.x + (size_t)img * kImgPx * CIN);
#pragma unroll
      for (int k = 0; k < G::NPF; ++k) {
        const int c = tid + 256 * k;
        pf[k] = c < G::CHUNKS ? src[c] : floatx4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto put = [&](int off, const floatx4& v) {
    uintx2 h, m, l;
    split3(v, h, m, l);
    *reinterpret_cast<uintx2*>(lds_b + off) = h;
    *reinterpret_cast<uintx2*>(lds_b + off + PLANE) = m;
    *reinterpret_cast<uintx2*>(lds_b + off + 2 * PLANE) = l;
  };

  float bias4[MW][4] = {};
  if constexpr (BIAS_RELU) {
#pragma unroll
    for (int j = 0; j < MW; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) bias4[j][i] = a.bias[cbase + co0 + 16 * j + 4 * q + i];
  }
  // this lane's filter fragments: [tap][channel tile][K step][part][lane] x 16 bytes
  const uintx4* wlane = reinterpret_cast<const uintx4*>(a.w) + (size_t)((cbase / 16 + mtile) * KS) * 3 * 64 + lane;
  constexpr int W_TAP = G::MT_ALL * KS * 3 * 64;   // 16-byte units per tap

  int img = blockIdx.x;
  if (img < a.B) fetch(img);
  for (; img < a.B; img += gridDim.x) {
    __syncthreads();   // every wave is done reading the previous image (and the zero fill landed)
    if constexpr (POOLED) {
#pragma unroll
      for (int k = 0; k < kNpc; ++k)
        if (tid + 256 * k < kCellChunks) {
          put(pdst[k], pooled_value(pc[k], 0));
          put(pdst[k] + PSB, pooled_value(pc[k], 1));
          put(pdst[k] + ROWB, pooled_value(pc[k], 2));
          put(pdst[k] + ROWB + PSB, pooled_value(pc[k], 3));
        }
    } else {
#pragma unroll
      for (int k = 0; k < G::NPF; ++k)
        if (tid + 256 * k < G::CHUNKS) put(sdst[k], pf[k]);
    }
    __syncthreads();
    if (img + (int)gridDim.x < a.B) fetch(img + gridDim.x);

    floatx4 acc[MW][NTW];
#pragma unroll
    for (int j = 0; j < MW; ++j)
#pragma unroll
      for (int i = 0; i < NTW; ++i) acc[j][i] = floatx4{0.f, 0.f, 0.f, 0.f};

    uintx4 wcur[MW * KS][3], wnxt[MW * KS][3];   // [tile j][K step] flattened: consecutive in the packed filter
#pragma unroll
    for (int ks = 0; ks < MW * KS; ++ks)
#pragma unroll
      for (int part = 0; part < 3; ++part) wcur[ks][part] = wlane[(ks * 3 + part) * 64];
    // A tap is U units of (K step, group of up to four pixel tiles); the pixel-operand reads of the next unit -
    // the first unit of the next tap after a tap's last - are in flight while this unit's MFMAs run.
    constexpr int kGroup = (KS * ((NTW + 3) / 4)) % 2 == 0 ? 4 : 2;
    constexpr int NG = (NTW + kGroup - 1) / kGroup;
    constexpr int U = KS * NG;
    static_assert(U % 2 == 0, "two operand buffers alternate with a fixed phase per tap");
    bf16x8 bh[2][kGroup], bm[2][kGroup], bl[2][kGroup];
    auto read_unit = [&](int u, int toff_, int buf) {
      const int ks = u / NG, i0 = (u % NG) * kGroup;
#pragma unroll
      for (int v = 0; v < kGroup; ++v)
        if (i0 + v < NTW) {
          const unsigned char* p = lds_b + toff_ + boff[i0 + v] + 64 * ks;
          bh[buf][v] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uintx4*>(p));
          bm[buf][v] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uintx4*>(p + PLANE));
          bl[buf][v] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uintx4*>(p + 2 * PLANE));
        }
    };
    int toff = 0;   // byte offset of the tap inside the window
    read_unit(0, 0, 0);
#pragma unroll 1
    for (int tap = 0; tap < 25; ++tap) {
      {
        // no branch around the prefetch (the last tap loads its own fragments again): with one, the compiler
        // has to assume the path without newer loads in flight and waits for these right away
        const uintx4* wn = wlane + (size_t)min(tap + 1, 24) * W_TAP;
#pragma unroll
        for (int ks = 0; ks < MW * KS; ++ks)
#pragma unroll
          for (int part = 0; part < 3; ++part) wnxt[ks][part] = wn[(ks * 3 + part) * 64];
      }
      const int toff_next = tap == 24 ? toff : toff + ((tap % 5) == 4 ? ROWB - 4 * PSB : PSB);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ks = u / NG, i0 = (u % NG) * kGroup, b = u & 1;
        if (u + 1 < U) read_unit(u + 1, toff, b ^ 1);
        else read_unit(0, toff_next, b ^ 1);   // unconditional for the same reason (the last tap reads its own again)
        __builtin_amdgcn_sched_barrier(0);
        // small terms first, then the large ones; the wave's tiles alternate
        auto term = [&](int wpart, const bf16x8 (&bp)[2][kGroup]) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < MW; ++j) {
            const bf16x8 wv = __builtin_bit_cast(bf16x8, wcur[j * KS + ks][wpart]);
#pragma unroll
            for (int v = 0; v < kGroup; ++v)
              if (i0 + v < NTW) acc[j][i0 + v] = DTF_MFMA_BF16(wv, bp[b][v], acc[j][i0 + v]);
          }
        };
        term(2, bh);
        term(0, bl);
        term(1, bm);
        term(1, bh);
        term(0, bm);
        term(0, bh);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ks = 0; ks < MW * KS; ++ks)
#pragma unroll
        for (int part = 0; part < 3; ++part) wcur[ks][part] = wnxt[ks][part];
      toff = toff_next;
    }

    // lane holds output channels co0 + 4 q .. + 3 of pixel 16 t + n
    if constexpr (POOLOUT) __syncthreads();   // the input planes are about to be overwritten by the outputs
    float* lds_f = reinterpret_cast<float*>(lds_b);
    float* yt = POOLOUT ? nullptr : a.y + (size_t)img * kImgPx * COUT + cbase + co0 + 4 * q;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int p = (nsub * NTW + i) * 16 + n;
      if (p < kImgPx) {
#pragma unroll
        for (int j = 0; j < MW; ++j) {
          floatx4 o = acc[j][i];
          if constexpr (BIAS_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r] + bias4[j][r], 0.f);
          }
          if constexpr (POOLOUT) *reinterpret_cast<floatx4*>(lds_f + p * G::OS + co0 + 16 * j + 4 * q) = o;
          else *reinterpret_cast<floatx4*>(yt + (size_t)p * COUT + 16 * j) = o;
        }
      }
    }
    if constexpr (POOLOUT) {
      // 2x2 / 2 max pool of the image from LDS: pooled maximum + winner position (first maximum wins,
      // positions numbered (0,0) (0,1) (1,0) (1,1) as in maxpool2x2_fwd); the activation is not written
      __syncthreads();
      constexpr int CH4 = G::CW / 4;   // 16-byte chunks per pooled cell of this workgroup's channel slice
      for (int cc = tid; cc < kCells * CH4; cc += 256) {
        const int cell = cc / CH4, c4 = cc % CH4;
        const float* o0 = lds_f + (2 * (cell / 7) * kHW + 2 * (cell % 7)) * G::OS + 4 * c4;
        const floatx4 v0 = *reinterpret_cast<const floatx4*>(o0);
        const floatx4 v1 = *reinterpret_cast<const floatx4*>(o0 + G::OS);
        const floatx4 v2 = *reinterpret_cast<const floatx4*>(o0 + kHW * G::OS);
        const floatx4 v3 = *reinterpret_cast<const floatx4*>(o0 + (kHW + 1) * G::OS);
        floatx4 best = v0;
        unsigned arg = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (v1[e] > best[e]) { best[e] = v1[e]; arg = (arg & ~(0xffu << (8 * e))) | (1u << (8 * e)); }
          if (v2[e] > best[e]) { best[e] = v2[e]; arg = (arg & ~(0xffu << (8 * e))) | (2u << (8 * e)); }
          if (v3[e] > best[e]) { best[e] = v3[e]; arg = (arg & ~(0xffu << (8 * e))) | (3u << (8 * e)); }
        }
        const size_t off = ((size_t)img * kCells + cell) * COUT + cbase + 4 * c4;
        *reinterpret_cast<floatx4*>(a.pool_out + off) = best;
        *reinterpret_cast<unsigned*>(a.pool_arg_out + off) = arg;
      }
      // the planes (halo included) were overwritten: restore the zeros before the next image is staged
      __syncthreads();
      for (int i = tid; i < 3 * PLANE / 16; i += 256) reinterpret_cast<uintx4*>(lds_b)[i] = uintx4{0u, 0u, 0u, 0u};
    }
  }
}

// HWIO filter [25][Cin][Cout] -> MFMA A fragments, bf16 element i of [tap][row tile][K step][part][lane][8]:
// forward: rows = output channels, K = input channels; flip (data gradient): tap t takes filter tap
// 24 - t, rows = input channels, K = output channels.
__global__ __launch_bounds__(256) void conv5x5_split_pack_kernel(Conv5x5PackArgs a) {
  const int flip = a.flip;
  const int rows = flip ? a.Cin : a.Cout, kdim = flip ? a.Cout : a.Cin;
  const int mt_all = rows / 16, ks_all = kdim / 32;
  const int total = 25 * mt_all * ks_all * 3 * 64 * 8;
  unsigned short* out = reinterpret_cast<unsigned short*>(a.out);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int j = i & 7, lane = (i >> 3) & 63, part = (i >> 9) % 3;
    const int ks = ((i >> 9) / 3) % ks_all, mt = ((i >> 9) / (3 * ks_all)) % mt_all, tap = (i >> 9) / (3 * ks_all * mt_all);
    const int row = mt * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + j;
    const float v = flip ? a.w[((size_t)(24 - tap) * a.Cin + row) * a.Cout + k] : a.w[((size_t)tap * a.Cin + k) * a.Cout + row];
    out[i] = split_part_bits(v, part);
  }
}

template <int CIN, int COUT, bool BIAS_RELU, bool POOLED = false, bool POOLOUT = false, int CSPLIT = 1>
void launch_s5(const Conv5x5Args& a, hipStream_t stream) {
  using G = S5<CIN, COUT, CSPLIT>;
  auto kernel = conv5x5_split_kernel<CIN, COUT, BIAS_RELU, POOLED, POOLOUT, CSPLIT>;
  const size_t shmem = (size_t)G::LDS_BYTES;
  static const bool ready = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), shmem);
    return true;
  }();
  (void)ready;
  int cap = 256 / CSPLIT;   // one workgroup per CU
  const int iters = ceil_div(a.B, cap);
  const int grid = ceil_div(a.B, iters);
  launch("conv5x5_split", kernel, dim3(grid, CSPLIT), dim3(256), shmem, stream, a);
}

}  // namespace

size_t conv5x5_split_packed_floats() { return (size_t)25 * 32 * 64 * 3 / 2; }

void conv5x5_split_pack(const Conv5x5PackArgs& a, hipStream_t stream) {
  if (a.Cin != 32 || a.Cout != 64) throw std::runtime_error("conv5x5_split_pack: the 5x5, 32 -> 64 filter");
  launch("conv5x5_split_pack", conv5x5_split_pack_kernel, dim3(150), dim3(256), 0, stream, a);
}

// Same interface as conv5x5_tile; a.w = conv5x5_split_pack's output (flip = a.flip), a.imgs is ignored.
void conv5x5_split(const Conv5x5Args& a, hipStream_t stream) {
  if (a.B < 1) throw std::runtime_error("conv5x5_split: empty batch");
  if (a.Cin == 32 && a.Cout == 64 && !a.flip) {
    if (!a.bias) throw std::runtime_error("conv5x5_split: the 32->64 forward form needs a bias");
    if (a.pool_out) {
      if (!a.pool_arg_out) throw std::runtime_error("conv5x5_split: the fused pool writes maxima and winner positions");
      if (a.B < 256) return launch_s5<32, 64, true, false, true, 2>(a, stream);   // fewer images than CUs: half of the channels each
      return launch_s5<32, 64, true, false, true>(a, stream);
    }
    if (!a.y) throw std::runtime_error("conv5x5_split: null output");
    if (a.B < 256) return launch_s5<32, 64, true, false, false, 2>(a, stream);
    return launch_s5<32, 64, true>(a, stream);
  }
  if (a.Cin == 64 && a.Cout == 32 && a.flip) {
    if (!a.y) throw std::runtime_error("conv5x5_split: null output");
    if (a.pool_dp) {
      if (!a.pool_arg || !a.pool_p) throw std::runtime_error("conv5x5_split: pooled staging needs dp, arg and p");
      if (a.B < 256) return launch_s5<64, 32, false, true, false, 2>(a, stream);
      return launch_s5<64, 32, false, true>(a, stream);
    }
    if (!a.x) throw std::runtime_error("conv5x5_split: null input");
    if (a.B < 256) return launch_s5<64, 32, false, false, false, 2>(a, stream);
    return launch_s5<64, 32, false>(a, stream);
  }
  throw std::runtime_error("conv5x5_split: unsupported channel counts / direction");
}

}  // namespace dtf
