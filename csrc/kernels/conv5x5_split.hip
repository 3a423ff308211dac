// The LeNet's second convolution (5x5, SAME, 14x14 maps, 32 -> 64 channels; conv5x5_tile.hip), its data
// gradient and (further down) its filter gradient on the bf16 matrix pipe with fp32 operands split three ways ("bf16x6", conv_fprop_split.hip):
// six v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation. Opt-in; tile walk, fused bias + ReLU +
// 2x2 max pool (forward) and fused pool + ReLU backward while staging (data gradient) as in the exact
// kernel:
//   * a workgroup's tile is one whole image: its zero-padded 18x18 input sits in LDS for all 25 taps as
//     three bf16 planes (h, m, l parts; split while it is staged), pixels 64 bytes (32 channels: 61 KB, two
//     workgroups per CU, so one stages or pools while the other multiplies) or 144 bytes (64 channels) apart;
//   * the filter is packed once per step into MFMA A fragments [tap][channel tile][K step][h, m, l][lane]
//     (for the data gradient with the taps reversed and the channel roles swapped); a wave's fragments of
//     a tap are three (six) 16-byte loads per lane straight from L2, fetched one tap ahead;
//   * M = output channels, N = pixels: a lane ends up with 4 consecutive channels of one pixel; a wave owns
//     two output-channel tiles and walks groups of pixel tiles, so every pixel-operand read feeds twelve
//     MFMAs and consecutive MFMAs write different accumulators.
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
  static constexpr int PSB = CIN == 32 ? 64 : 144;   // bytes per pixel and plane (64 channels: +1 sixteen-byte slot of padding)
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
__global__ __launch_bounds__(256, CIN == 32 ? 2 : 1) void conv5x5_split_kernel(Conv5x5Args a) {
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
    constexpr int kGroup = CIN == 32 ? 2 : ((KS * ((NTW + 3) / 4)) % 2 == 0 ? 4 : 2);
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

    // The next image is requested here, not ahead of the tap loop: vector memory operations retire in order, so
    // the first wait for a tap's filter fragments would also wait for these loads' trip to HBM.
    if (img + (int)gridDim.x < a.B) fetch(img + gridDim.x);

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

// ---- filter gradient of the same layer on the bf16 pipe: dw[tap][ci][co] += sum over pixels x[p + tap][ci] * dy[p][co]
//
// As in the exact kernel (conv5x5_tile.hip: conv5x5_wgrad_kernel) one image per workgroup tile sits in LDS, the
// 25 x 32 x 64 outputs live in the accumulators of the four waves for the whole kernel (wave = one 16-channel
// slice of ci x two of co, all taps: 50 tiles, 200 registers) and the pixel loop has no barrier. Here K = 32
// pixels per MFMA: 7 steps cover the 196 pixels (the 28 pixels past the end are zero in dy). Both operands have
// the pixels along K, so both come out of pixel-major bf16 planes through transposed reads
// (ds_read_b64_tr_b16: lane group q holds pixels 4 q .. 4 q + 3 and 16 + 4 q .. of the step, for x and dy alike):
//   x:  [part h, m, l][16-channel half][18 x 18 padded pixels][32 bytes]  - a tap is an immediate offset
//   dy: [part][16-channel tile of co][224 pixels][32 bytes]
// Per step and wave 150 + 12 transposed reads feed 300 MFMAs; the reads of tap t + 1 (after the last tap: of
// the next step's first) are in flight under the MFMAs of tap t.
constexpr int kWsXHalf = kPW * kPW * 32;          // bytes of one 16-channel half plane of x
constexpr int kWsXPart = 2 * kWsXHalf;
constexpr int kWsKSteps = (kImgPx + 31) / 32;     // 7
constexpr int kWsYPx = kWsKSteps * 32;            // 224
constexpr int kWsYTile = kWsYPx * 32;             // bytes of one 16-channel tile plane of dy
constexpr int kWsYPart = 4 * kWsYTile;
constexpr int kWsLds = 3 * kWsXPart + 3 * kWsYPart;
static_assert(kWsLds <= (int)kLdsBudget, "x and dy planes of one image");
constexpr int kWsXChunks = kImgPx * 8, kWsYChunks = kImgPx * 16;
constexpr int kWsNpfX = (kWsXChunks + 255) / 256, kWsNpfY = (kWsYChunks + 255) / 256;

// TSPLIT = 2: two workgroups per image, each with its own copy of the staged image and half of the 25 taps
// (13 and 12) - for batches with fewer images than CUs.
template <bool POOLED, int TSPLIT>
__global__ __launch_bounds__(256) void conv5x5_wgrad_split_kernel(Conv5x5WgradArgs a) {
  DTF_ABORT_GUARD(a);
  constexpr int NTAP = (25 + TSPLIT - 1) / TSPLIT;
  static_assert(NTAP % 2 == 1, "the last tap of a step uses the first operand buffer again");
  const int tap0 = TSPLIT > 1 ? (int)blockIdx.y * NTAP : 0;
  int tap_off[TSPLIT > 1 ? NTAP : 1];   // TSPLIT = 1: immediates
  if constexpr (TSPLIT > 1) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int tap = min(tap0 + t, 24);   // a slot past the last tap reads tap 24 and is not used
      tap_off[t] = ((tap / 5) * kPW + tap % 5) * 32;
    }
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_b[];
  unsigned char* sx = lds_b;
  unsigned char* sy = lds_b + 3 * kWsXPart;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int v = wave >> 1, h = wave & 1;   // ci slice 16 v .., co slices 32 h .. and 32 h + 16 ..
  const int tr_j = (lane >> 2) & 3, tr_c = lane & 3;   // transposed read: this lane points at pixel 4 q + tr_j, channels 4 tr_c ..

  // zero once: the halo of x and the pixels of dy past the image are never written
  for (int i = tid; i < kWsLds / 16; i += 256) reinterpret_cast<uintx4*>(lds_b)[i] = uintx4{0u, 0u, 0u, 0u};

  int xdst[kWsNpfX], ydst[POOLED ? 1 : kWsNpfY];
#pragma unroll
  for (int k = 0; k < kWsNpfX; ++k) {
    const int c = tid + 256 * k;
    const int px = (c < kWsXChunks ? c : 0) >> 3, c4 = c & 7;
    xdst[k] = (c4 >> 2) * kWsXHalf + ((px / kHW + 2) * kPW + px % kHW + 2) * 32 + 8 * (c4 & 3);
  }
  if constexpr (!POOLED) {
#pragma unroll
    for (int k = 0; k < kWsNpfY; ++k) {
      const int c = tid + 256 * k;
      ydst[k] = ((c & 15) >> 2) * kWsYTile + ((c < kWsYChunks ? c : 0) >> 4) * 32 + 8 * (c & 3);
    }
  }
  floatx4 pfx[kWsNpfX], pfy[POOLED ? 1 : kWsNpfY];
  PooledChunk pc[POOLED ? kNpc : 1];
  int pdst[POOLED ? kNpc : 1];
  if constexpr (POOLED) {
#pragma unroll
    for (int k = 0; k < kNpc; ++k) {
      const int cc = min(tid + 256 * k, kCellChunks - 1);
      const int cell = cc >> 4;
      pdst[k] = ((cc & 15) >> 2) * kWsYTile + (2 * (cell / 7) * kHW + 2 * (cell % 7)) * 32 + 8 * (cc & 3);
    }
  }
  auto fetch = [&](int img) {
    const floatx4* xs = reinterpret_cast<const floatx4*>(a.x + (size_t)img * kImgPx * 32);
#pragma unroll
    for (int k = 0; k < kWsNpfX; ++k)
      if (tid + 256 * k < kWsXChunks) pfx[k] = xs[tid + 256 * k];
    if constexpr (POOLED) {
#pragma unroll
      for (int k = 0; k < kNpc; ++k)
        if (tid + 256 * k < kCellChunks) pooled_fetch(a.pool_dp, a.pool_arg, a.pool_p, img, tid + 256 * k, pc[k]);
    } else {
      const floatx4* ys = reinterpret_cast<const floatx4*>(a.dy + (size_t)img * kImgPx * 64);
#pragma unroll
      for (int k = 0; k < kWsNpfY; ++k)
        if (tid + 256 * k < kWsYChunks) pfy[k] = ys[tid + 256 * k];
    }
  };
  auto put = [&](unsigned char* base, int part_stride, int off, const floatx4& val) {
    uintx2 ph, pm, pl;
    split3(val, ph, pm, pl);
    *reinterpret_cast<uintx2*>(base + off) = ph;
    *reinterpret_cast<uintx2*>(base + off + part_stride) = pm;
    *reinterpret_cast<uintx2*>(base + off + 2 * part_stride) = pl;
  };

  floatx4 acc[NTAP][2];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) acc[t][0] = acc[t][1] = floatx4{0.f, 0.f, 0.f, 0.f};

  // lane addresses of K step s: pixels 32 s + 4 q + tr_j and 16 more (x: clamped to the image; what a pixel past
  // the end multiplies is zero in dy)
  auto x_addr = [&](int p) {
    p = min(p, kImgPx - 1);
    return v * kWsXHalf + ((p / kHW) * kPW + p % kHW) * 32 + 8 * tr_c;
  };
  const int y_lane = 2 * h * kWsYTile + (4 * q + tr_j) * 32 + 8 * tr_c;

  int img = blockIdx.x;
  if (img < a.B) fetch(img);
  for (; img < a.B; img += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWsNpfX; ++k)
      if (tid + 256 * k < kWsXChunks) put(sx, kWsXPart, xdst[k], pfx[k]);
    if constexpr (POOLED) {
#pragma unroll
      for (int k = 0; k < kNpc; ++k)
        if (tid + 256 * k < kCellChunks) {
          put(sy, kWsYPart, pdst[k], pooled_value(pc[k], 0));
          put(sy, kWsYPart, pdst[k] + 32, pooled_value(pc[k], 1));
          put(sy, kWsYPart, pdst[k] + kHW * 32, pooled_value(pc[k], 2));
          put(sy, kWsYPart, pdst[k] + (kHW + 1) * 32, pooled_value(pc[k], 3));
        }
    } else {
#pragma unroll
      for (int k = 0; k < kWsNpfY; ++k)
        if (tid + 256 * k < kWsYChunks) put(sy, kWsYPart, ydst[k], pfy[k]);
    }
    __syncthreads();
    if (img + (int)gridDim.x < a.B) fetch(img + gridDim.x);

    uintx2 abuf[2][3][2], anext[3][2];          // [buffer][part][K half]
    uintx2 bcur[2][3][2], bnext[2][3][2];       // [co tile][part][K half]
    auto read_a = [&](int xa0, int xa1, int t, uintx2 (&dst)[3][2]) __attribute__((always_inline)) {
      int off;
      if constexpr (TSPLIT > 1) off = tap_off[t];
      else off = ((t / 5) * kPW + t % 5) * 32;
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        dst[part][0] = lds_read_tr(sx + xa0 + off + part * kWsXPart);
        dst[part][1] = lds_read_tr(sx + xa1 + off + part * kWsXPart);
      }
    };
    auto read_b = [&](int s, uintx2 (&dst)[2][3][2]) __attribute__((always_inline)) {
      const unsigned char* p = sy + y_lane + s * (32 * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          dst[j][part][0] = lds_read_tr(p + j * kWsYTile + part * kWsYPart);
          dst[j][part][1] = lds_read_tr(p + j * kWsYTile + part * kWsYPart + 16 * 32);
        }
    };
    int xa0 = x_addr(4 * q + tr_j), xa1 = x_addr(16 + 4 * q + tr_j);
    read_b(0, bcur);
    read_a(xa0, xa1, 0, abuf[0]);
#pragma unroll 1
    for (int s = 0; s < kWsKSteps; ++s) {
      const int sn = min(s + 1, kWsKSteps - 1);   // the last step reads its own operands again: no branch around a prefetch
      const int xn0 = x_addr(32 * sn + 4 * q + tr_j), xn1 = x_addr(32 * sn + 16 + 4 * q + tr_j);
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        if (t + 1 < NTAP) read_a(xa0, xa1, t + 1, abuf[(t + 1) & 1]);
        else read_a(xn0, xn1, 0, anext);
        if (t == NTAP / 2) read_b(sn, bnext);
        __builtin_amdgcn_sched_barrier(0);
        const uintx2 (&ab)[3][2] = abuf[t & 1];
        const bf16x8 ah = frag8(ab[0][0], ab[0][1]), am = frag8(ab[1][0], ab[1][1]), al = frag8(ab[2][0], ab[2][1]);
        // small terms first; the two co tiles alternate
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = DTF_MFMA_BF16(al, frag8(bcur[j][0][0], bcur[j][0][1]), acc[t][j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = DTF_MFMA_BF16(ah, frag8(bcur[j][2][0], bcur[j][2][1]), acc[t][j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = DTF_MFMA_BF16(am, frag8(bcur[j][1][0], bcur[j][1][1]), acc[t][j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = DTF_MFMA_BF16(am, frag8(bcur[j][0][0], bcur[j][0][1]), acc[t][j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = DTF_MFMA_BF16(ah, frag8(bcur[j][1][0], bcur[j][1][1]), acc[t][j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = DTF_MFMA_BF16(ah, frag8(bcur[j][0][0], bcur[j][0][1]), acc[t][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          abuf[0][part][kh] = anext[part][kh];
          bcur[0][part][kh] = bnext[0][part][kh];
          bcur[1][part][kh] = bnext[1][part][kh];
        }
      xa0 = xn0;
      xa1 = xn1;
    }
  }

  // lane holds dw[tap][ci = 16 v + 4 q + i][co = 32 h + 16 j + n]
  const int lane_off = (16 * v + 4 * q) * 64 + 32 * h + n;
  if (a.partials) {
    float* dst = a.partials + (size_t)blockIdx.x * (25 * 2048) + (size_t)tap0 * 2048 + lane_off;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (tap0 + t < 25) {
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
          const float sum = acc[t][j][i];
          if (sum != 0.f) atomicAdd(dst + t * 2048 + i * 64 + 16 * j, sum);
        }
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
  int cap = (CIN == 32 ? 512 : 256) / CSPLIT;
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

// Same interface as conv5x5_wgrad (conv5x5_wgrad_parts(B) partial results, per-XCD copies, pooled staging).
void conv5x5_wgrad_split(const Conv5x5WgradArgs& a, hipStream_t stream) {
  if (a.B < 1) throw std::runtime_error("conv5x5_wgrad_split: empty batch");
  const size_t shmem = (size_t)kWsLds;
  static const bool once = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv5x5_wgrad_split_kernel<false, 1>), shmem);
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv5x5_wgrad_split_kernel<true, 1>), shmem);
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv5x5_wgrad_split_kernel<false, 2>), shmem);
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv5x5_wgrad_split_kernel<true, 2>), shmem);
    return true;
  }();
  (void)once;
  const int grid = conv5x5_wgrad_parts(a.B);
  const bool halves = a.B < 256;   // fewer images than CUs: two workgroups per image, 13 and 12 of the 25 taps each
  const dim3 g(grid, halves ? 2 : 1);
  if (a.pool_dp) {
    if (!a.pool_arg || !a.pool_p) throw std::runtime_error("conv5x5_wgrad_split: pooled staging needs dp, arg and p");
    if (halves) launch("conv5x5_wgrad_split", conv5x5_wgrad_split_kernel<true, 2>, g, dim3(256), shmem, stream, a);
    else launch("conv5x5_wgrad_split", conv5x5_wgrad_split_kernel<true, 1>, g, dim3(256), shmem, stream, a);
  } else {
    if (!a.dy) throw std::runtime_error("conv5x5_wgrad_split: null output gradient");
    if (halves) launch("conv5x5_wgrad_split", conv5x5_wgrad_split_kernel<false, 2>, g, dim3(256), shmem, stream, a);
    else launch("conv5x5_wgrad_split", conv5x5_wgrad_split_kernel<false, 1>, g, dim3(256), shmem, stream, a);
  }
}

}  // namespace dtf
