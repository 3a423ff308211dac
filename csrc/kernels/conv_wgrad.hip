// Filter-gradient convolution on the exact-fp32 matrix cores of gfx950.
//
// GEMM view: D[ci][co] (one 16x16 tile per filter tap) += sum over pixels of
// act[pixel shifted by the tap][ci] * dout[pixel][co]; the MFMA k dimension walks pixels, four
// per instruction. A block stages the (BatchNorm+ReLU'd) input halo tile and the matching
// output-gradient tile in LDS, walks `tiles_per_block` tiles keeping every tap's partial
// filter gradient in accumulator registers, and flushes once through LDS with coalesced fp32
// atomics, so the atomic traffic is amortised over many pixels.
// Waves split the (ci-tile, co-tile) pairs; when a layer has fewer than four pairs the waves
// split the pixels instead and are summed in LDS before the flush.
// (reference op: TensorFlow conv2d_backprop_filter under opt.compute_gradients,
//  DistributedResNet/src/distributed_train.py:256)
#include "conv.h"

#include <cstdlib>

namespace dtf {
namespace {

__device__ __forceinline__ int round_up4(int v) { return (v + 3) & ~3; }

template <int KS, int S, int CP, int CO_T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvWgradArgs a) {
  DTF_ABORT_GUARD(a);
#define DTF_STAMP(k) do { if (a.dbg_stamps && threadIdx.x == 0) a.dbg_stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = wall_clock64(); } while (0)
  DTF_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CPP = CP + 4;
  constexpr int CO_B = CO_T * 16;
  constexpr int COP = CO_B + 4;
  constexpr int CI_T = (CP >= 16) ? CP / 16 : (KS * KS + 3) / 4;
  constexpr int TAPS = (CP >= 16) ? KS * KS : 1;
  constexpr int PAIRS = CI_T * CO_T;
  constexpr int WP = (PAIRS % 4 == 0) ? 4 : ((PAIRS % 2 == 0) ? 2 : 1);
  constexpr int WX = 4 / WP;
  constexpr int PPW = PAIRS / WP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  const int l15 = lane & 15;
  const int wp = wave % WP;
  const int wx = wave / WP;

  const int LH = (a.TH - 1) * S + KS;
  const int LW = (a.TW - 1) * S + KS;
  const int tile_px = a.TH * a.TW;
  const int px_pad = (tile_px + 3) & ~3;

  // The tile area (input halo + output-gradient tile) doubles as the flush scratch and is at
  // least one tap's worth of it: WX * CI_T*16 rows of CO_B + 4 floats.
  int tile_floats = round_up4(LH * LW * CPP) + px_pad * COP;
  if (tile_floats < WX * CI_T * 16 * (CO_B + 4)) tile_floats = WX * CI_T * 16 * (CO_B + 4);
  float* s_in = smem;                                  // [LH*LW][CPP]
  float* s_do = s_in + round_up4(LH * LW * CPP);       // [px_pad][COP]
  float* s_bn = smem + tile_floats;                    // scale[CP], shift[CP]

  const int co_blk0 = blockIdx.y * CO_B;
  const int tiles_img = a.tiles_y * a.tiles_x;
  const int total_tiles = tiles_img * a.B;

  if (a.in_mode == IN_BNRELU && tid < CP) {
    float sc = 0.f, sh = 0.f, m, r;
    if (tid < a.Cin) bn_coeffs(a.in_bn, tid, a.Cin, sc, sh, m, r);
    s_bn[tid] = sc;
    s_bn[CP + tid] = sh;
  }

  floatx4 acc[TAPS][PPW];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int k = 0; k < PPW; ++k) acc[t][k] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = blockIdx.x * a.tiles_per_block;
  int t_end = t_begin + a.tiles_per_block;
  if (t_end > total_tiles) t_end = total_tiles;

  for (int t = t_begin; t < t_end; ++t) {
    const int n = t / tiles_img;
    const int tile = t - n * tiles_img;
    const int oy0 = (tile / a.tiles_x) * a.TH;
    const int ox0 = (tile % a.tiles_x) * a.TW;
    __syncthreads();  // previous tile's readers are done (and s_bn is visible on the first pass)

    // ---- stage input halo tile (activation applied) ----
    {
      constexpr int C4 = CP / 4;
      const int npix = LH * LW;
      const int vy0 = oy0 * S - a.pad;
      const int vx0 = ox0 * S - a.pad;
      const bool vec = (a.Cin & 3) == 0;
      for (int i = tid; i < npix * C4; i += 256) {
        const int pix = i / C4;
        const int c4 = i - pix * C4;
        const int r = pix / LW;
        const int c = pix - r * LW;
        const int iy = vy0 + r, ix = vx0 + c;
        const bool ok = (iy >= 0) && (ix >= 0) && (iy < a.H) && (ix < a.W) && !(a.debug_skip & 2);
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const float* src = a.in + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + c4 * 4;
          if (vec) {
            if (c4 * 4 < a.Cin) v = *reinterpret_cast<const floatx4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c4 * 4 + e < a.Cin) v[e] = src[e];
          }
          if (a.in_mode == IN_BNRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ch = c4 * 4 + e;
              v[e] = fmaxf(v[e] * s_bn[ch] + s_bn[CP + ch], 0.f);
            }
          }
        }
        *reinterpret_cast<floatx4*>(s_in + pix * CPP + c4 * 4) = v;
      }
    }
    // ---- stage output-gradient tile (zeros outside the image / tile) ----
    {
      constexpr int O4 = CO_B / 4;
      for (int i = tid; i < px_pad * O4; i += 256) {
        const int p = i / O4;
        const int c4 = i - p * O4;
        const int ty = p / a.TW;
        const int tx = p - ty * a.TW;
        const int oy = oy0 + ty, ox = ox0 + tx;
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        const int co = co_blk0 + c4 * 4;
        if (p < tile_px && oy < a.Ho && ox < a.Wo && co < a.Cout && !(a.debug_skip & 2))
          v = *reinterpret_cast<const floatx4*>(a.dout + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout + co);
        *reinterpret_cast<floatx4*>(s_do + p * COP + c4 * 4) = v;
      }
    }
    __syncthreads();
    if (t == t_begin) DTF_STAMP(1);

    // ---- accumulate: k-steps of four pixels ----
    const int n_steps = px_pad >> 2;
    if (!(a.debug_skip & 1))
    for (int step = wx; step < n_steps; step += WX) {
      int p = step * 4 + q;
      const int pc = p < tile_px ? p : 0;
      const int ty = pc / a.TW;
      const int tx = pc - ty * a.TW;
      const int abase = ((ty * S) * LW + tx * S) * CPP;
      const float* dop = s_do + p * COP + l15;
      float bfrag[PPW];
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        const int co_t = (wp * PPW + k) % CO_T;
        bfrag[k] = dop[co_t * 16];
      }
      if constexpr (CP >= 16) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const int kh = tap / KS, kw = tap % KS;
          const float* ap = s_in + abase + (kh * LW + kw) * CPP + l15;
#pragma unroll
          for (int k = 0; k < PPW; ++k) {
            const int ci_t = (wp * PPW + k) / CO_T;
            const float af = ap[ci_t * 16];
            acc[tap][k] = DTF_MFMA16(af, bfrag[k], acc[tap][k]);
          }
        }
      } else {
        // CP == 4: row i of the tile is (tap = 4*g + i/4, channel = i%4).
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
          const int g = (wp * PPW + k) / CO_T;
          int tap = 4 * g + (l15 >> 2);
          if (tap >= KS * KS) tap = 0;
          const int kh = tap / KS, kw = tap - kh * KS;
          const float af = s_in[abase + (kh * LW + kw) * CPP + (l15 & 3)];
          acc[0][k] = DTF_MFMA16(af, bfrag[k], acc[0][k]);
        }
      }
    }
    if (t == t_begin) DTF_STAMP(2);
  }
  DTF_STAMP(3);

  if (a.debug_skip & 4) return;
  float* dw = a.dw + (size_t)(blockIdx.x & (kStatSlots - 1)) * a.dw_slot_stride;
  // ---- flush: groups of taps go through the (now free) tile area of LDS in one pass each: every
  // wave parks its accumulators, one barrier, then all threads add the pixel-split partials and
  // issue coalesced fp32 atomics that nobody waits for. A group is as many taps as fit.
  constexpr int ROWS = CI_T * 16;
  constexpr int RS = CO_B + 4;            // padded row: the four q groups of a wave hit different banks
  float* s_fl = smem;
  int tg = tile_floats / (WX * ROWS * RS);
  if (tg > TAPS) tg = TAPS;
  if (tg < 1) tg = 1;                     // the launcher sizes LDS for at least one tap
  for (int tap0 = 0; tap0 < TAPS; tap0 += tg) {
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {  // unrolled: acc[][] must be indexed statically
      if (tap < tap0 || tap >= tap0 + tg) continue;
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        const int pr = wp * PPW + k;
        const int ci_t = pr / CO_T, co_t = pr % CO_T;
        const floatx4 v = acc[tap][k];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s_fl[((wx * tg + (tap - tap0)) * ROWS + ci_t * 16 + 4 * q + r) * RS + co_t * 16 + l15] = v[r];
      }
    }
    __syncthreads();
    const int n_here = (TAPS - tap0 < tg ? TAPS - tap0 : tg) * ROWS * CO_B;
    for (int i = tid; i < n_here; i += 256) {
      const int col = i % CO_B;
      const int row = (i / CO_B) % ROWS;
      const int tl = i / (CO_B * ROWS);
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WX; ++w) v += s_fl[((w * tg + tl) * ROWS + row) * RS + col];
      const int co = co_blk0 + col;
      int tap_g, ci;
      if (CP >= 16) {
        tap_g = tap0 + tl;
        ci = row;
      } else {
        tap_g = 4 * (row >> 4) + ((row & 15) >> 2);
        ci = row & 3;
      }
      if (tap_g < KS * KS && ci < a.Cin && co < a.Cout && v != 0.f) atomicAdd(dw + ((size_t)tap_g * a.Cin + ci) * a.Cout + co, v);
    }
  }
  DTF_STAMP(15);
#undef DTF_STAMP
}

template <int KS, int S, int CP, int CO_T>
size_t launch_wgrad(const ConvWgradArgs& a, hipStream_t stream) {
  constexpr int CPP = CP + 4;
  constexpr int CO_B = CO_T * 16;
  constexpr int COP = CO_B + 4;
  constexpr int CI_T = (CP >= 16) ? CP / 16 : (KS * KS + 3) / 4;
  const int LH = (a.TH - 1) * S + KS, LW = (a.TW - 1) * S + KS;
  const int px_pad = (a.TH * a.TW + 3) & ~3;
  constexpr int PAIRS = CI_T * CO_T;
  constexpr int WX = 4 / ((PAIRS % 4 == 0) ? 4 : ((PAIRS % 2 == 0) ? 2 : 1));
  size_t tile_floats = (size_t)((LH * LW * CPP + 3) & ~3) + (size_t)px_pad * COP;
  const size_t one_tap = (size_t)WX * CI_T * 16 * (CO_B + 4);   // flush scratch for one tap (see the kernel)
  if (tile_floats < one_tap) tile_floats = one_tap;
  const size_t floats = tile_floats + 2 * CP;
  const size_t shmem = floats * sizeof(float);
  if (shmem > kLdsBudget) throw std::runtime_error("conv_wgrad: tile does not fit LDS");
  const int total_tiles = a.tiles_y * a.tiles_x * a.B;
  const int gx = ceil_div(total_tiles, a.tiles_per_block);
  const int gy = ceil_div(ceil_div(a.Cout, 16), CO_T);
  launch("conv_wgrad", conv_wgrad_kernel<KS, S, CP, CO_T>, dim3(gx, gy), dim3(256), shmem, stream, a);
  return shmem;
}

}  // namespace

#define DTF_WGRAD_CASE(KS_, S_, CP_)                                        \
  if (ks == KS_ && a.stride == S_ && cp == CP_) {                           \
    if (co_t == 4) return launch_wgrad<KS_, S_, CP_, 4>(a, stream);         \
    if (co_t == 2) return launch_wgrad<KS_, S_, CP_, 2>(a, stream);         \
    return launch_wgrad<KS_, S_, CP_, 1>(a, stream);                        \
  }

size_t conv_wgrad(ConvWgradArgs a, int ks, hipStream_t stream) {
  static const bool use_band = [] {
    const char* e = getenv("DTF_WGRAD_BAND");
    return !(e && e[0] == '0');
  }();
  if (use_band && a.TH <= 0) {
    if (const size_t r = conv_wgrad_band(a, ks, stream)) return r;
  }
  int cp;
  if (a.Cin <= 4) cp = 4;
  else if (a.Cin <= 16) cp = 16;
  else if (a.Cin <= 32) cp = 32;
  else if (a.Cin <= 64) cp = 64;
  else throw std::runtime_error("conv_wgrad: more than 64 input channels is not supported");
  if ((a.Cout & 3) != 0) throw std::runtime_error("conv_wgrad: Cout must be a multiple of 4");
  if (a.TH <= 0 || a.TW <= 0) {
    a.TW = a.Wo <= 32 ? a.Wo : 32;
    a.TH = a.Ho;
    while (a.TH * a.TW > 256 && a.TH > 1) a.TH = (a.TH + 1) / 2;
    a.tiles_x = ceil_div(a.Wo, a.TW);
    a.tiles_y = ceil_div(a.Ho, a.TH);
  }
  // co-tiles per block: keep the per-wave accumulator count (taps * pairs/4 * 4 regs) moderate.
  const int nt_total = ceil_div(a.Cout, 16);
  const int ci_t = cp >= 16 ? cp / 16 : (ks * ks + 3) / 4;
  const int taps = cp >= 16 ? ks * ks : 1;
  int co_t = 1;
  if (a.co_tiles > 0) {
    if (nt_total % a.co_tiles || (a.co_tiles != 1 && a.co_tiles != 2 && a.co_tiles != 4))
      throw std::runtime_error("conv_wgrad: co_tiles must be 1, 2 or 4 and divide the output tiles");
    co_t = a.co_tiles;
  } else
  for (int cand : {4, 2}) {
    if (nt_total % cand) continue;
    const int pairs = ci_t * cand;
    const int wp = (pairs % 4 == 0) ? 4 : ((pairs % 2 == 0) ? 2 : 1);
    if (taps * (pairs / wp) * 4 <= 80) { co_t = cand; break; }
  }
  const int total_tiles = a.tiles_y * a.tiles_x * a.B;
  if (a.tiles_per_block <= 0) {
    // Aim for about two waves of blocks over the 256 CUs per cout split.
    const int gy = nt_total / co_t;
    int target_blocks = 1024 / gy;
    if (target_blocks < 64) target_blocks = 64;
    a.tiles_per_block = ceil_div(total_tiles, target_blocks);
    if (a.tiles_per_block < 1) a.tiles_per_block = 1;
  }
  DTF_WGRAD_CASE(3, 1, 4)
  DTF_WGRAD_CASE(3, 1, 16)
  DTF_WGRAD_CASE(3, 2, 16)
  DTF_WGRAD_CASE(3, 1, 32)
  DTF_WGRAD_CASE(3, 2, 32)
  DTF_WGRAD_CASE(3, 1, 64)
  DTF_WGRAD_CASE(5, 1, 4)
  DTF_WGRAD_CASE(5, 1, 32)
  throw std::runtime_error("conv_wgrad: unsupported (ks, stride, Cin) combination");
}

}  // namespace dtf
