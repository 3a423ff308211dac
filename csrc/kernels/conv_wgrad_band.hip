// Filter gradient of the 3x3 ResNet convolutions: persistent, register-prefetched, band-tiled.
//
// The general kernel in conv_wgrad.hip spends most of its time outside the matrix cores: runtime
// index arithmetic while staging, one LDS round trip per MFMA with nothing in flight, and a flush
// that walks the nine filter taps through LDS one at a time, each tap waiting for its global
// atomics before the next barrier. This kernel fixes the tile geometry at compile time
// (output tile = a TH-row band of the full output width TW, a power of two), which makes every
// LDS offset an instruction immediate, and restructures the block:
//
//   * a block is persistent over a contiguous range of bands; the global loads of band i+1 are
//     issued into registers before the MFMA loop of band i and written to LDS after it;
//   * the MFMA loop is software pipelined: the A/B fragments of k-step s+1 are read from LDS
//     while the nine-tap MFMA group of step s executes;
//   * the flush is one pass: every wave drops its accumulators into (bank-conflict-free padded)
//     LDS, one barrier, then all threads sum the pixel-split partials and issue coalesced fp32
//     atomics that nobody waits for, into the gradient copy of the block's XCD.
//
// GEMM view as in conv_wgrad.hip: D[ci][co] per tap += sum over pixels x[pixel+tap][ci] * g[pixel][co],
// four pixels per v_mfma_f32_16x16x4_f32; the four pixels of a k-step are neighbours in one row.
// (reference op: TensorFlow conv2d_backprop_filter under opt.compute_gradients,
//  DistributedResNet/src/distributed_train.py:256)
#include "conv.h"

namespace dtf {
namespace {

template <int S, int CP, int CO_T, int TW, int TH>
struct BandGeom {
  static constexpr int CPP = CP + 4;
  static constexpr int CO_B = CO_T * 16;
  static constexpr int COP = CO_B + 4;
  static constexpr int CI_T = CP / 16;
  static constexpr int PAIRS = CI_T * CO_T;
  static constexpr int WP = (PAIRS % 4 == 0) ? 4 : ((PAIRS % 2 == 0) ? 2 : 1);
  static constexpr int WX = 4 / WP;
  static constexpr int PPW = PAIRS / WP;
  // distinct ci tiles / co tiles among one wave's pairs (pair p -> ci_t = p / CO_T, co_t = p % CO_T)
  static constexpr int NB = (PPW < CO_T) ? PPW : CO_T;
  static constexpr int NA = PPW / NB;
  static constexpr int LH = (TH - 1) * S + 3;
  static constexpr int LW = (TW - 1) * S + 3;
  static constexpr int C4 = CP / 4;
  static constexpr int O4 = CO_B / 4;
  static constexpr int IN_CHUNKS = LH * LW * C4;
  static constexpr int G_CHUNKS = TH * TW * O4;
  static constexpr int NPF_IN = (IN_CHUNKS + 255) / 256;
  static constexpr int NPF_G = (G_CHUNKS + 255) / 256;
  static constexpr int STEPS = TH * TW / 4;
  static constexpr int IN_FLOATS = (LH * LW * CPP + 3) & ~3;
  static constexpr int G_FLOATS = TH * TW * COP;
  static constexpr int RED_STRIDE = CO_B + 4;
  static constexpr int RED_FLOATS = WX * 9 * CI_T * 16 * RED_STRIDE;
  static constexpr int TILE_FLOATS = IN_FLOATS + G_FLOATS;
  static constexpr int LDS_FLOATS = (TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS) + 2 * CP;
};

template <int S, int CP, int CO_T, int TW, int TH>
__global__ __launch_bounds__(256, 2) void conv_wgrad_band_kernel(ConvWgradArgs a) {
  DTF_ABORT_GUARD(a);
  using G = BandGeom<S, CP, CO_T, TW, TH>;
  constexpr int CPP = G::CPP, CO_B = G::CO_B, COP = G::COP, CI_T = G::CI_T;
  constexpr int WP = G::WP, WX = G::WX, PPW = G::PPW, NA = G::NA, NB = G::NB;
  constexpr int LW = G::LW, C4 = G::C4, O4 = G::O4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                       // [LH*LW][CPP]
  float* s_do = smem + G::IN_FLOATS;        // [TH*TW][COP]
  float* s_bn = smem + G::LDS_FLOATS - 2 * CP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  const int l15 = lane & 15;
  const int wp = wave % WP;
  const int wx = wave / WP;

  const int gx = a.tiles_per_block;         // here: number of blocks that share one cout block
  const int bx = blockIdx.x % gx;
  const int co_blk0 = (blockIdx.x / gx) * CO_B;
  const int bands = a.Ho / TH;
  const int total = bands * a.B;
  // contiguous range of bands for this block
  const int per = total / gx, extra = total % gx;
  const int t_begin = bx * per + (bx < extra ? bx : extra);
  const int t_end = t_begin + per + (bx < extra ? 1 : 0);

  // ---- per-thread staging geometry, fixed for the whole kernel ---------------------------------
  int in_goff[G::NPF_IN], in_loff[G::NPF_IN], in_row[G::NPF_IN];
#pragma unroll
  for (int k = 0; k < G::NPF_IN; ++k) {
    const int i = tid + k * 256;
    const int pix = i / C4, c4 = i % C4;
    const int r = pix / LW, c = pix % LW;
    const int ix = c - a.pad;
    const bool ok = (i < G::IN_CHUNKS) && (ix >= 0) && (ix < a.W) && (c4 * 4 < a.Cin);
    in_goff[k] = (r * a.W + ix) * a.Cin + c4 * 4;
    in_loff[k] = pix * CPP + c4 * 4;
    in_row[k] = ok ? r : -100000;           // rows that can never be inside the image
  }
  int g_goff[G::NPF_G], g_loff[G::NPF_G];
#pragma unroll
  for (int k = 0; k < G::NPF_G; ++k) {
    const int i = tid + k * 256;
    const int p = i / O4, c4 = i % O4;
    const bool ok = (i < G::G_CHUNKS) && (co_blk0 + c4 * 4 < a.Cout);
    g_goff[k] = ok ? p * a.Cout + co_blk0 + c4 * 4 : -1;
    g_loff[k] = p * COP + c4 * 4;
  }

  floatx4 pf_in[G::NPF_IN], pf_g[G::NPF_G];
  unsigned okbits = 0;
  auto load_band = [&](int t) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const int vy0 = oy0 * S - a.pad;
    const float* in0 = a.in + ((long long)(n * a.H + vy0) * a.W) * a.Cin;
    const float* g0 = a.dout + ((long long)(n * a.Ho + oy0) * a.Wo) * a.Cout;
    okbits = 0;
#pragma unroll
    for (int k = 0; k < G::NPF_IN; ++k) {
      const int iy = vy0 + in_row[k];
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < a.H && !(DTF_DEBUG_SKIP(a) & 2)) {
        v = *reinterpret_cast<const floatx4*>(in0 + in_goff[k]);
        okbits |= 1u << k;
      }
      pf_in[k] = v;
    }
#pragma unroll
    for (int k = 0; k < G::NPF_G; ++k) {
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (g_goff[k] >= 0 && !(DTF_DEBUG_SKIP(a) & 2)) v = *reinterpret_cast<const floatx4*>(g0 + g_goff[k]);
      pf_g[k] = v;
    }
  };
  auto write_band = [&]() {
#pragma unroll
    for (int k = 0; k < G::NPF_IN; ++k) {
      if (tid + k * 256 < G::IN_CHUNKS) {
        floatx4 v = pf_in[k];
        if (a.in_mode == IN_BNRELU && ((okbits >> k) & 1u)) {
          const int ch = (tid % C4) * 4;   // chunk tid + 256 k: the same four channels for every k (256 % C4 == 0)
          const floatx4 sc = *reinterpret_cast<const floatx4*>(s_bn + ch);
          const floatx4 sh = *reinterpret_cast<const floatx4*>(s_bn + CP + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        }
        *reinterpret_cast<floatx4*>(s_in + in_loff[k]) = v;
      }
    }
#pragma unroll
    for (int k = 0; k < G::NPF_G; ++k)
      if (tid + k * 256 < G::G_CHUNKS) *reinterpret_cast<floatx4*>(s_do + g_loff[k]) = pf_g[k];
  };

  if (t_begin < t_end) load_band(t_begin);
  if (a.in_mode == IN_BNRELU && tid < CP) {
    float sc = 0.f, sh = 0.f, m, r;
    if (tid < a.Cin) bn_coeffs(a.in_bn, tid, a.Cin, sc, sh, m, r);
    s_bn[tid] = sc;
    s_bn[CP + tid] = sh;
  }

  floatx4 acc[9][PPW];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < PPW; ++k) acc[t][k] = floatx4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // s_bn visible

  // pair k of this wave: p = wp*PPW + k -> ci tile p / CO_T, co tile p % CO_T; with PPW <= CO_T the
  // pairs share one ci tile, otherwise they cover whole ci tiles with all co tiles.
  const int p0 = wp * PPW;
  const int laneA = (q * S) * CPP + l15 + (p0 / CO_T) * 16;
  const int laneB = q * COP + l15 + (p0 % CO_T) * 16;

  auto read_frags = [&](int step, float (&af)[9][NA], float (&bf)[NB]) {
    const int p = step * 4;
    const int row = p / TW, col = p % TW;
    const float* ap = s_in + ((row * S) * LW + col * S) * CPP + laneA;
    const float* bp = s_do + p * COP + laneB;
#pragma unroll
    for (int j = 0; j < NB; ++j) bf[j] = bp[j * 16];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < NA; ++i) af[tap][i] = ap[((tap / 3) * LW + (tap % 3)) * CPP + i * 16];
  };

  for (int t = t_begin; t < t_end; ++t) {
    write_band();
    __syncthreads();
    if (t + 1 < t_end) load_band(t + 1);   // in flight during the MFMA loop

    if (!(DTF_DEBUG_SKIP(a) & 1)) {
      // Two fragment sets alternate: while the nine-tap MFMA group of one k-step executes, the LDS
      // reads of the next k-step are already in flight. The scheduling barriers keep the compiler
      // from sinking the reads below the MFMAs (it otherwise waits for LDS once per k-step).
      static_assert((G::STEPS / WX) % 2 == 0, "k-steps per wave must be even");
      float af0[9][NA], bf0[NB], af1[9][NA], bf1[NB];
      auto mfma_group = [&](float (&af)[9][NA], float (&bf)[NB]) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[tap][i * NB + j] = DTF_MFMA16(af[tap][i], bf[j], acc[tap][i * NB + j]);
      };
      read_frags(wx, af0, bf0);
      for (int step = wx; step < G::STEPS; step += 2 * WX) {
        read_frags(step + WX, af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(step + 2 * WX < G::STEPS ? step + 2 * WX : step, af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // all waves are done with this band's LDS tile
  }

  if (DTF_DEBUG_SKIP(a) & 4) return;
  // ---- flush: accumulators -> LDS (one pass), sum the pixel-split partials, coalesced atomics ----
  constexpr int ROWS = CI_T * 16;
  constexpr int RS = G::RED_STRIDE;
  float* s_red = smem;   // [WX][9][ROWS][RS]
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int p = p0 + k;
      const int ci_t = p / CO_T, co_t = p % CO_T;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        s_red[((wx * 9 + tap) * ROWS + ci_t * 16 + 4 * q + r) * RS + co_t * 16 + l15] = acc[tap][k][r];
    }
  __syncthreads();
  float* dw = a.dw_part ? a.dw_part + (size_t)bx * (9 * a.Cin * a.Cout) : a.dw + (size_t)(blockIdx.x & (kStatSlots - 1)) * a.dw_slot_stride;
  for (int e = tid; e < 9 * ROWS * CO_B; e += 256) {
    const int col = e % CO_B;
    const int row = (e / CO_B) % ROWS;
    const int tap = e / (CO_B * ROWS);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WX; ++w) v += s_red[((w * 9 + tap) * ROWS + row) * RS + col];
    const int co = co_blk0 + col;
    if (row < a.Cin && co < a.Cout) {
      float* dst = dw + ((size_t)tap * a.Cin + row) * a.Cout + co;
      if (a.dw_part) *dst = v;
      else atomicAdd(dst, v);
    }
  }
}

template <int S, int CP, int CO_T, int TW, int TH>
size_t launch_band(ConvWgradArgs a, hipStream_t stream, int* copies) {
  using G = BandGeom<S, CP, CO_T, TW, TH>;
  const size_t shmem = (size_t)G::LDS_FLOATS * sizeof(float);
  const int gy = ceil_div(ceil_div(a.Cout, 16), CO_T);
  const int total = (a.Ho / TH) * a.B;
  // Two co-resident blocks per CU; `tiles_per_block` (bands walked per block) may override.
  int gx = a.tiles_per_block > 0 ? ceil_div(total, a.tiles_per_block) : 512 / gy;
  if (gx > total) gx = total;
  if (gx < 1) gx = 1;
  gx = (gx + 7) / 8 * 8;   // keeps (blockIdx.x % 8) equal for the blocks of one XCD across cout blocks
  if (gx > total) gx = total;
  a.tiles_per_block = gx;  // the kernel reads the number of blocks per cout block from this field
  if (copies) {
    *copies = gx;
    return 1;
  }
  launch("conv_wgrad_band", conv_wgrad_band_kernel<S, CP, CO_T, TW, TH>, dim3(gx * gy), dim3(256), shmem, stream, a);
  return shmem;
}

}  // namespace

// Returns 0 when the shape is not covered (the caller falls back to the general kernel).
static size_t band_dispatch(const ConvWgradArgs& a, int ks, hipStream_t stream, int* copies) {
  if (ks != 3 || a.dbias != nullptr || a.pad > 1 || (a.Cin & 15) || (a.Cout & 15)) return 0;
  if (a.Ho != a.Wo || a.H != a.W || a.H != a.Ho * a.stride) return 0;
  const int nt = a.Cout / 16;
  int co_t = a.co_tiles;
#define DTF_BAND(S_, CIN_, TW_, TH_, CO_DEFAULT)                                                   \
  if (a.stride == S_ && a.Cin == CIN_ && a.Wo == TW_ && a.Ho % TH_ == 0) {                         \
    if (co_t <= 0) co_t = CO_DEFAULT;                                                              \
    if (nt % co_t) return 0;                                                                       \
    if (co_t == 1) return launch_band<S_, CIN_, 1, TW_, TH_>(a, stream, copies);                   \
    if (co_t == 2) return launch_band<S_, CIN_, 2, TW_, TH_>(a, stream, copies);                   \
    return 0;                                                                                      \
  }
  DTF_BAND(1, 16, 32, 8, 1)
  DTF_BAND(2, 16, 16, 8, 1)
  DTF_BAND(1, 32, 16, 8, 1)
  DTF_BAND(2, 32, 8, 8, 1)
  DTF_BAND(1, 64, 8, 8, 1)
#undef DTF_BAND
  return 0;
}

size_t conv_wgrad_band(const ConvWgradArgs& a, int ks, hipStream_t stream) { return band_dispatch(a, ks, stream, nullptr); }

int conv_wgrad_band_copies(const ConvWgradArgs& a, int ks) {
  int copies = 0;
  return band_dispatch(a, ks, nullptr, &copies) ? copies : 0;
}

}  // namespace dtf
