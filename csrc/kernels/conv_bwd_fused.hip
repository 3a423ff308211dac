// Fused backward pass of a stride-1 3x3 convolution layer y = conv(relu(bn(x)), W): one kernel
// produces the masked data gradient, BatchNorm-backward's two reductions AND the filter gradient.
//
// Run separately (conv_fprop_band.hip as data gradient + conv_wgrad_band.hip), the two kernels
// read the output gradient dy twice and the layer input x twice (once as the filter gradient's
// activation, once as the ReLU mask). In the 16-channel stage of the CIFAR ResNet both are bound
// by per-CU memory throughput, so the second reads are pure cost. Here a block stages, per
// band of TH rows:
//     * the dy band with a one-pixel halo  -> LDS   (all C channels)
//     * the x band, raw, no halo           -> LDS   (this block's 16*NT channel slice)
// and uses them twice:
//   data gradient   g[p, ci] = 1[bn(x[p,ci]) > 0] * (sum_{tap,co} dy[p+tap-1, co] W'[tap][co][ci] + extra[p, ci])
//                   GEMM M = pixels, N = ci slice, K = 9 taps x C; dy tile read with ds_read_b128
//                   in the K-permuted order of conv_fprop.hip; mask and x-hat come from the LDS x tile;
//   filter gradient dW[tap][ci][co] += sum_p relu(bn(x[p, ci])) * dy[p - (tap - 1), co]
//                   re-associated over the pixels of x, so x needs no halo: the shifted dy comes
//                   from the halo tile (zero outside the image); GEMM M = ci, N = co, K = pixels;
//                   BatchNorm + ReLU are applied to the A fragment after the LDS read (two VALU
//                   instructions in the shadow of a 32-cycle MFMA).
// Blocks are persistent over bands (register prefetch of the next band); the ci dimension may be
// split over blocks (zsplit): a block then owns rows [ci slice] of dW and the same channels of g.
// With the optional prologue (ConvBwdFusedArgs::dy_x) the BatchNorm backward of the FOLLOWING
// layer is applied to dy while it is staged, so that tensor is never materialised either; with
// dy_res / dy_out the gradient of an identity (or option-A) shortcut is added and the sum written
// out once. The third prefetched tensor fits the register file of the 32- and 64-channel instances
// (232 / 222 VGPRs); the 16-channel instance runs over 4-row bands for it (189 VGPRs; with 8 rows
// it spills and is slower than the elementwise pass it replaces). Only the image columns of the
// dy halo tile are staged (exactly 5 chunks per thread for an 8-row band at every channel count);
// its two padding columns are zeroed once. Filter gradients are flushed once per block through LDS into the gradient
// copy of the block's XCD; statistics go to per-XCD slots.
// (reference ops: conv2d_backprop_input + conv2d_backprop_filter + the BatchNorm / ReLU gradients
//  under opt.compute_gradients, DistributedResNet/src/distributed_train.py:256)
#include "conv.h"

#include <cstdlib>

namespace dtf {
namespace {

template <int C, int NT, int TW, int TH>
struct BF {
  static constexpr int CPP = C + 4;            // dy halo tile pixel stride
  static constexpr int XC = NT * 16;           // x channels held by the block
  static constexpr int XPP = XC + 4;
  static constexpr int CG = C / 16;
  static constexpr int KG = 9 * CG;
  static constexpr int LH = TH + 2, LW = TW + 2;
  static constexpr int D_CHUNKS = LH * TW * (C / 4);   // the halo columns are outside the image for every band: zeroed once, never staged
  static constexpr int X_CHUNKS = TH * TW * (XC / 4);
  static constexpr int NPF_D = (D_CHUNKS + 255) / 256;
  static constexpr int NPF_X = (X_CHUNKS + 255) / 256;
  static constexpr int W_CHUNKS = KG * NT * 64;
  static constexpr int NW = (W_CHUNKS + 255) / 256;
  static constexpr int N_MT = TH * TW / 16;
  static constexpr int MTW = (N_MT >= 8 && NT <= 2) ? 2 : 1;
  static constexpr int ITER = N_MT / (4 * MTW);
  // filter gradient: (ci tile, co tile) pairs over the four waves
  static constexpr int CO_T = C / 16;
  static constexpr int PAIRS = NT * CO_T;
  static constexpr int WP = (PAIRS % 4 == 0) ? 4 : ((PAIRS % 2 == 0) ? 2 : 1);
  static constexpr int WX = 4 / WP;
  static constexpr int PPW = PAIRS / WP;
  static constexpr int STEPS = TH * TW / 4;
  static constexpr int D_FLOATS = (LH * LW * CPP + 3) & ~3;
  static constexpr int X_FLOATS = TH * TW * XPP;
  static constexpr int RED_STRIDE = C + 4;
  static constexpr int RED_FLOATS = WX * 9 * XC * RED_STRIDE;
  static constexpr int TILE_FLOATS = D_FLOATS + X_FLOATS;
  static constexpr int W_FLOATS = KG * NT * 256;
  static constexpr int LDS_FLOATS = (TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS) + W_FLOATS + 6 * XC + 3 * C;
  static_assert(PPW == 1, "one (ci, co) tile pair per wave");
  static_assert((TH * TW) % 16 == 0 && N_MT % (4 * MTW) == 0 && (STEPS / WX) % 2 == 0, "band must split evenly");
  static_assert(TW % 8 == 0 && (TH * TW) % 16 == 0, "k-steps are taken from 16-pixel segments of whole rows");
};

template <int C, int NT, int TW, int TH, bool DYBN, int RES = 0>   // RES: 0 none, 1 identity shortcut, 2 option-A shortcut
__global__ __launch_bounds__(256, 2) void conv_bwd_fused_kernel(ConvBwdFusedArgs a) {
  DTF_ABORT_GUARD(a);
  static_assert(!RES || DYBN, "the shortcut term rides on the BatchNorm prologue");
  using G = BF<C, NT, TW, TH>;
  constexpr int CPP = G::CPP, XC = G::XC, XPP = G::XPP, CG = G::CG, KG = G::KG, LW = G::LW;
  constexpr int MTW = G::MTW, ITER = G::ITER, CO_T = G::CO_T, WP = G::WP, WX = G::WX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_dy = smem;                                            // [LH*LW][CPP]
  float* s_x = smem + G::D_FLOATS;                               // [TH*TW][XPP]
  float* s_w = smem + (G::TILE_FLOATS > G::RED_FLOATS ? G::TILE_FLOATS : G::RED_FLOATS);
  float* s_co = s_w + G::W_FLOATS;                               // [4][XC]: scale, shift, mean, rstd
  float* s_stat = s_co + 4 * XC;                                 // [2][XC]
  float* s_dco = s_stat + 2 * XC;                                // [3][C]: dy_eff = ca*g' + cb*y + cc

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  const int l15 = lane & 15;

  constexpr int zsplit = CO_T / NT;
  const int gpz = gridDim.x / zsplit;
  const int bid = blockIdx.x;
  const int z = (bid >> 3) % zsplit;
  const int jblk = (bid / (8 * zsplit)) * 8 + (bid & 7);
  const int bands = a.H / TH;
  const int total_tiles = bands * a.B;
  const int c0 = z * XC;                     // first channel of the block's ci slice

  // ---- BatchNorm statistics of x first, then the filter slice, then the first band -------------
  const bool co_thread = tid < XC;
  double s1 = 0.0, s2 = 0.0;
  float gam = 0.f, bet = 0.f;
  if (co_thread) {
    s1 = stat_total(a.bn.sums, c0 + tid, C);
    s2 = stat_total(a.bn.sums, C + c0 + tid, C);
    gam = a.bn.gamma[c0 + tid];
    bet = a.bn.beta[c0 + tid];
  }
  double ds1 = 0.0, ds2 = 0.0, dg1 = 0.0, dg2 = 0.0;
  float dgam = 0.f;
  if (DYBN && tid < C) {
    ds1 = stat_total(a.dy_bn.sums, tid, C);
    ds2 = stat_total(a.dy_bn.sums, C + tid, C);
    dg1 = stat_total(a.dy_gsums, tid, C);
    dg2 = stat_total(a.dy_gsums, C + tid, C);
    dgam = a.dy_bn.gamma[tid];
  }
  floatx4 wreg[G::NW];
  {
    const floatx4* wsrc = reinterpret_cast<const floatx4*>(a.wpk_t);
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
      const int i = tid + k * 256;
      const int g = i / (NT * 64);
      const int rem = i % (NT * 64);
      wreg[k] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (i < G::W_CHUNKS) wreg[k] = wsrc[(g * CO_T + z * NT + (rem >> 6)) * 64 + (rem & 63)];
    }
  }
  // staging geometry: dy halo chunks (packed LDS offset | row+1) and x chunks
  int d_goff[G::NPF_D], d_lr[G::NPF_D];
#pragma unroll
  for (int k = 0; k < G::NPF_D; ++k) {
    const int i = tid + k * 256;
    const int ipx = i / (C / 4), c4 = i % (C / 4);
    const int r = ipx / TW, ix = ipx % TW;      // halo-tile row, image column (a.W == TW)
    const int pix = r * LW + ix + 1;
    const bool ok = i < G::D_CHUNKS;
    d_goff[k] = (r * a.W + ix) * C + c4 * 4;
    d_lr[k] = (pix * CPP + c4 * 4) | ((ok ? r + 1 : 0) << 16);
  }
  static_assert(G::D_FLOATS < 65536, "LDS offsets must fit 16 bits");
  int x_goff[G::NPF_X], x_loff[G::NPF_X];
#pragma unroll
  for (int k = 0; k < G::NPF_X; ++k) {
    const int i = tid + k * 256;
    const int pix = i / (XC / 4), c4 = i % (XC / 4);
    x_goff[k] = (i < G::X_CHUNKS) ? pix * C + c0 + c4 * 4 : -1;
    x_loff[k] = pix * XPP + c4 * 4;
  }

  floatx4 pf_d[G::NPF_D], pf_x[G::NPF_X];
  floatx4 pf_c[DYBN ? G::NPF_D : 1];
  floatx4 pf_r[RES ? G::NPF_D : 1];
  int r_goff[RES == 2 ? G::NPF_D : 1];   // option-A shortcut: offset of the chunk's source in the half-resolution gradient
  if (RES == 2) {
#pragma unroll
    for (int k = 0; k < G::NPF_D; ++k) {
      const int i = tid + k * 256;
      const int ipx = i / (C / 4), c4 = i % (C / 4);
      const int r = ipx / TW, ix = ipx % TW;
      r_goff[k] = (((r - 1) >> 1) * (TW / 2) + (ix >> 1)) * (2 * C) + C / 2 + c4 * 4;   // row r of the halo tile is image row oy0 - 1 + r
    }
  }
  unsigned okbits = 0;
  auto load_band = [&](int t) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const long long doff = ((long long)(n * a.H + oy0 - 1) * a.W) * C;
    const float* d0 = a.dy + doff;
    const float* x0 = a.x + ((long long)(n * a.H + oy0) * a.W) * C;
    okbits = 0;
#pragma unroll
    for (int k = 0; k < G::NPF_D; ++k) {
      const int rr = d_lr[k] >> 16;
      const int iy = oy0 - 1 + rr - 1;
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (rr > 0 && iy >= 0 && iy < a.H) {
        v = *reinterpret_cast<const floatx4*>(d0 + d_goff[k]);
        if (DYBN) {
          pf_c[k] = *reinterpret_cast<const floatx4*>(a.dy_x + doff + d_goff[k]);
          okbits |= 1u << k;
        }
      }
      pf_d[k] = v;
    }
#pragma unroll
    for (int k = 0; k < G::NPF_X; ++k) {
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (x_goff[k] >= 0) v = *reinterpret_cast<const floatx4*>(x0 + x_goff[k]);
      pf_x[k] = v;
    }
  };
  // The shortcut term of the next band is requested between the two MFMA phases: by then the data
  // gradient's registers are free, and the filter-gradient phase covers the latency.
  auto load_res = [&](int t) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const float* r0 = RES == 2 ? a.dy_res + ((long long)(n * (a.H / 2) + oy0 / 2) * (TW / 2)) * (2 * C)
                               : a.dy_res + ((long long)(n * a.H + oy0 - 1) * a.W) * C;
#pragma unroll
    for (int k = 0; k < G::NPF_D; ++k)
      if ((okbits >> k) & 1u) pf_r[k] = *reinterpret_cast<const floatx4*>(r0 + (RES == 2 ? r_goff[k] : d_goff[k]));
  };
  auto write_band = [&](long long doff) {
#pragma unroll
    for (int k = 0; k < G::NPF_D; ++k)
      if (tid + k * 256 < G::D_CHUNKS) {
        floatx4 v = pf_d[k];
        if (DYBN && ((okbits >> k) & 1u)) {
          // BatchNorm backward of the following layer, applied while staging (zero padding stays zero)
          static_assert(256 % (C / 4) == 0, "a thread's chunks share their channel group");
          const int ch = (tid % (C / 4)) * 4;   // the same four channels for every chunk of the thread
          const floatx4 ca = *reinterpret_cast<const floatx4*>(s_dco + ch);
          const floatx4 cb = *reinterpret_cast<const floatx4*>(s_dco + C + ch);
          const floatx4 cc = *reinterpret_cast<const floatx4*>(s_dco + 2 * C + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ca[e] * v[e] + (cb[e] * pf_c[k][e] + cc[e]);
          if (RES) {
            if (RES == 2) v += 0.25f * pf_r[k];
            else v += pf_r[k];
            // the band's own rows (not the halo) of this block's channel slice leave as the block input gradient
            const int row = (d_lr[k] >> 16) - 1;
            if (row >= 1 && row <= TH && (unsigned)(ch - c0) < (unsigned)XC) *reinterpret_cast<floatx4*>(a.dy_out + doff + d_goff[k]) = v;
          }
        }
        *reinterpret_cast<floatx4*>(s_dy + (d_lr[k] & 0xffff)) = v;
      }
#pragma unroll
    for (int k = 0; k < G::NPF_X; ++k)
      if (tid + k * 256 < G::X_CHUNKS) *reinterpret_cast<floatx4*>(s_x + x_loff[k]) = pf_x[k];
  };

  // Tile walk. Static: t, t + gpz, ... Dynamic (a.tile_counters): the first two tiles are the static
  // ones; further tiles of XCD x are x + 8k with k handed out by that XCD's counter. Thread 0 asks
  // for a tile one whole iteration before the block needs it and publishes it through LDS.
  const int xcd = bid & 7;
  int* ctr = a.tile_counters ? a.tile_counters + z * 8 + xcd : nullptr;
  __shared__ int s_next_tile;
  auto take_tile = [&]() -> int {   // thread 0 only
    const int k = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 2 * (gpz >> 3);
    return xcd + 8 * k;
  };
  int t = jblk;
  int t_next = t + gpz;
  int r_fetch = total_tiles;        // thread 0: the tile after t_next
  if (ctr && tid == 0) r_fetch = take_tile();
  if (t < total_tiles) {
    load_band(t);
    if (RES) load_res(t);
  }

  if (co_thread) {
    const double mean = s1 * (double)a.bn.inv_count;
    double var = s2 * (double)a.bn.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float r = rsqrtf((float)var + a.bn.eps);
    const float m = (float)mean;
    const float sc = gam * r;
    s_co[tid] = sc;
    s_co[XC + tid] = bet - m * sc;
    s_co[2 * XC + tid] = m;
    s_co[3 * XC + tid] = r;
  }
  if (DYBN && tid < C) {
    const double inv = (double)a.dy_bn.inv_count;
    const double mean = ds1 * inv;
    double var = ds2 * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = rsqrtf((float)var + a.dy_bn.eps);
    const float k1 = dgam * rstd;
    const float mg = (float)(dg1 * inv), mgx = (float)(dg2 * inv);
    s_dco[tid] = k1;
    s_dco[C + tid] = -k1 * mgx * rstd;
    s_dco[2 * C + tid] = k1 * (mgx * (float)mean * rstd - mg);
    if (blockIdx.x == 0) {
      if (a.dy_dbeta) a.dy_dbeta[tid] = (float)dg1;
      if (a.dy_dgamma) a.dy_dgamma[tid] = (float)dg2;
    }
  }
#pragma unroll
  for (int k = 0; k < G::NW; ++k)
    if (tid + k * 256 < G::W_CHUNKS) reinterpret_cast<floatx4*>(s_w)[tid + k * 256] = wreg[k];
  // left / right halo column of the dy tile: zero padding for every band
  for (int i = tid; i < G::LH * 2 * CPP; i += 256) {
    const int r = i / (2 * CPP), side = (i / CPP) & 1, ch = i % CPP;
    s_dy[(r * LW + (side ? LW - 1 : 0)) * CPP + ch] = 0.f;
  }
  __syncthreads();

  // ---- per-lane constants -----------------------------------------------------------------------
  // data gradient: pixel l15 of an m-tile; dy halo pixel (r+1, c+1); output offset
  const int lane_a = (((l15 / TW) + 0) * LW + (l15 % TW)) * CPP + 4 * q;   // tap (kh, kw) adds (kh*LW + kw)*CPP
  const int lane_x = l15 * XPP + 4 * q;                                     // x tile: [pixel][XPP], channels 4q.. of tile j
  const int lane_o = l15 * C + c0 + 4 * q;
  // filter gradient: this wave's (ci tile, co tile) pair and pixel split
  const int wp = wave % WP, wx = wave / WP;
  const int ci_t = wp / CO_T, co_t = wp % CO_T;
  // The four pixels of k-step (seg, j) are 16 seg + j + 4 q' (q' = lane group), not four neighbours:
  // pixel strides are C + 4 = 4 mod 16 floats, so pixels 4 apart start 16 banks apart and the 64
  // lanes of a fragment read (4 pixels x 16 channels) touch 64 different banks.
  const int dq = 4 * q;
  const int laneA = dq * XPP + ci_t * 16 + l15;               // x[pixel][ci]
  // dy[pixel - (tap - 1)]: halo pixel (r + 2 - kh, c + 2 - kw)
  const int laneB = ((dq / TW) * LW + dq % TW) * CPP + co_t * 16 + l15;
  const float w_sc = s_co[ci_t * 16 + l15], w_sh = s_co[XC + ci_t * 16 + l15];

  floatx4 accw[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) accw[tap] = floatx4{0.f, 0.f, 0.f, 0.f};
  floatx4 st1[NT], st2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { st1[j] = floatx4{0.f, 0.f, 0.f, 0.f}; st2[j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

  auto read_wfrags = [&](int step, float& af, float (&bf)[9]) {
    const int p = (step >> 2) * 16 + (step & 3);   // pixel of lane group 0
    const int row = p / TW, col = p % TW;
    af = s_x[p * XPP + laneA];
    const float* bp = s_dy + ((row + 2) * LW + col + 2) * CPP + laneB;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) bf[tap] = bp[-((tap / 3) * LW + (tap % 3)) * CPP];
  };

  while (t < total_tiles) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const long long obase = ((long long)(n * a.H + oy0) * TW) * C;

    write_band(((long long)(n * a.H + oy0 - 1) * a.W) * C);
    if (ctr && tid == 0) s_next_tile = r_fetch;
    __syncthreads();
    const int t_after = ctr ? s_next_tile : t_next + gpz;
    if (ctr && tid == 0 && t_after < total_tiles) r_fetch = take_tile();   // lands during the MFMA phases
    if (t_next < total_tiles) load_band(t_next);   // in flight during both MFMA phases

    // ---- data gradient: M = pixels, N = ci slice, K = taps x C --------------------------------------
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int mb = wave * MTW + it * 4 * MTW;
      floatx4 acc[MTW][NT], ev[MTW][NT];
      int aoff[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int p0 = (mb + i) * 16;
        aoff[i] = lane_a + ((p0 / TW) * LW + (p0 % TW)) * CPP;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
          ev[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
          if (a.extra) ev[i][j] = *reinterpret_cast<const floatx4*>(a.extra + obase + (long long)p0 * C + lane_o + j * 16);
        }
      }
      floatx4 av[2][MTW], bv[2][NT];
      auto read_group = [&](int g, floatx4 (&ao)[MTW], floatx4 (&bo)[NT]) {
        const int tap = g / CG, cg = g % CG;
        const int toff = ((tap / 3) * LW + (tap % 3)) * CPP + cg * 16;
#pragma unroll
        for (int i = 0; i < MTW; ++i) ao[i] = *reinterpret_cast<const floatx4*>(s_dy + aoff[i] + toff);
#pragma unroll
        for (int j = 0; j < NT; ++j) bo[j] = *reinterpret_cast<const floatx4*>(s_w + (g * NT + j) * 256 + lane * 4);
      };
      read_group(0, av[0], bv[0]);
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        if (g + 1 < KG) read_group(g + 1, av[(g + 1) & 1], bv[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = DTF_MFMA16(bv[g & 1][j][s], av[g & 1][i][s], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue: mask and x-hat from the LDS x tile, statistics, 16-byte stores
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int p0 = (mb + i) * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          floatx4 v = acc[i][j] + ev[i][j];
          const floatx4 x = *reinterpret_cast<const floatx4*>(s_x + p0 * XPP + lane_x + j * 16);
          const floatx4 esc = *reinterpret_cast<const floatx4*>(s_co + j * 16 + 4 * q);
          const floatx4 esh = *reinterpret_cast<const floatx4*>(s_co + XC + j * 16 + 4 * q);
          const floatx4 emu = *reinterpret_cast<const floatx4*>(s_co + 2 * XC + j * 16 + 4 * q);
          const floatx4 ers = *reinterpret_cast<const floatx4*>(s_co + 3 * XC + j * 16 + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = (x[r] * esc[r] + esh[r] > 0.f) ? v[r] : 0.f;
            st1[j][r] += v[r];
            st2[j][r] += v[r] * ((x[r] - emu[r]) * ers[r]);
          }
          *reinterpret_cast<floatx4*>(a.g + obase + (long long)p0 * C + lane_o + j * 16) = v;
        }
      }
    }

    if (RES && t_next < total_tiles) load_res(t_next);
    __builtin_amdgcn_sched_barrier(0);
    // ---- filter gradient: M = ci, N = co, K = pixels of the band ---------------------------------------
    {
      float af0, bf0[9], af1, bf1[9];
      auto mfma_group = [&](float af, float (&bf)[9]) {
        const float act = fmaxf(af * w_sc + w_sh, 0.f);   // BatchNorm + ReLU on the A fragment
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) accw[tap] = DTF_MFMA16(act, bf[tap], accw[tap]);
      };
      read_wfrags(wx, af0, bf0);
      for (int step = wx; step < G::STEPS; step += 2 * WX) {
        read_wfrags(step + WX, af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        read_wfrags(step + 2 * WX < G::STEPS ? step + 2 * WX : step, af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // all waves are done with this band's LDS tiles
    t = t_next;
    t_next = t_after;
  }

  // ---- statistics (the filter slice in LDS is no longer needed: its space parks the wave rows) ------
  if (a.gsums) block_stat_flush<NT, 4>(s_w, st1, st2, a.gsums, C, c0, kStatGridBwd);
  // ---- filter gradient flush: accumulators -> LDS, sum the pixel-split partials, coalesced atomics ----
  constexpr int RS = G::RED_STRIDE;
  float* s_red = smem;   // [WX][9][XC][RS]; the tile buffers are free (barrier above)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      s_red[((wx * 9 + tap) * XC + ci_t * 16 + 4 * q + r) * RS + co_t * 16 + l15] = accw[tap][r];
  __syncthreads();
  // with dw_part: plain stores into copy jblk of the whole gradient (this block's rows of it)
  float* dw = a.dw_part ? a.dw_part + (size_t)jblk * (9 * C * C) : a.dw + (size_t)(blockIdx.x & (kStatSlots - 1)) * a.dw_slot_stride;
  for (int e = tid; e < 9 * XC * C; e += 256) {
    const int col = e % C;
    const int row = (e / C) % XC;
    const int tap = e / (C * XC);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WX; ++w) v += s_red[((w * 9 + tap) * XC + row) * RS + col];
    float* dst = dw + ((size_t)tap * C + c0 + row) * C + col;
    if (a.dw_part) *dst = v;
    else atomicAdd(dst, v);
  }
}

// copies != nullptr: no launch, *copies = workgroups per channel split (= copies of the gradient
// filled through ConvBwdFusedArgs::dw_part)
template <int C, int NT, int TW, int TH, bool DYBN, int RES = 0>
size_t launch_bf(const ConvBwdFusedArgs& a, hipStream_t stream, int* copies) {
  using G = BF<C, NT, TW, TH>;
  const size_t shmem = (size_t)G::LDS_FLOATS * sizeof(float);
  constexpr int zsplit = (C / 16) / NT;
  const int total_tiles = (a.H / TH) * a.B;
  static const int bpc = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_bwd_fused_kernel<C, NT, TW, TH, DYBN, RES>), shmem);
    int n = 0;
    DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_bwd_fused_kernel<C, NT, TW, TH, DYBN, RES>, 256, shmem));
    return n < 1 ? 1 : n;
  }();
  int gpz_max = (256 * bpc) / zsplit;
  // Every workgroup ends with a 9*16*C-element gradient flush and starts by loading its filter
  // slice: with fewer than 4 bands per workgroup one workgroup per CU is faster than two
  // (batch 128: 0.83 -> 0.78 ms per ResNet-20 step, batch 256: 1.18 -> 1.17, no change from 512 on).
  if (total_tiles / gpz_max < 4 && bpc > 1) gpz_max = 256 / zsplit;
  if (gpz_max < 8) gpz_max = 8;
  const int iters = ceil_div(total_tiles, gpz_max);
  int gpz = ceil_div(total_tiles, iters);
  gpz = (gpz + 7) / 8 * 8;
  if (copies) {
    *copies = gpz;
    return 0;
  }
  launch("conv_bwd_fused", conv_bwd_fused_kernel<C, NT, TW, TH, DYBN, RES>, dim3(gpz * zsplit), dim3(256), shmem, stream, a);
  return shmem;
}

}  // namespace

bool conv_bwd_fused_supported(int B, int H, int W, int C) {
  (void)B;
  return H == W && H % 8 == 0 && ((C == 16 && W == 32) || (C == 32 && W == 16) || (C == 64 && W == 8));
}

static size_t bf_dispatch(const ConvBwdFusedArgs& a, hipStream_t stream, int* copies) {
  if (!conv_bwd_fused_supported(a.B, a.H, a.W, a.C)) throw std::runtime_error("conv_bwd_fused: unsupported shape");
  if (!a.dy || !a.x || !a.wpk_t || !a.g || !a.dw || !a.bn.sums) throw std::runtime_error("conv_bwd_fused: null operand");
  if (a.dy_x) {
    if (!a.dy_bn.sums || !a.dy_gsums) throw std::runtime_error("conv_bwd_fused: the BatchNorm prologue needs its statistics");
    if (a.dy_res) {
      if (!a.dy_out) throw std::runtime_error("conv_bwd_fused: dy_res needs dy_out");
      if (a.dy_res_mode == 1) {
        if (a.C == 16) return launch_bf<16, 1, 32, 4, true, 2>(a, stream, copies);
        if (a.C == 32) return launch_bf<32, 1, 16, 8, true, 2>(a, stream, copies);
        throw std::runtime_error("conv_bwd_fused: no option-A shortcut behind a 64-channel stage");
      }
      if (a.C == 16) return launch_bf<16, 1, 32, 4, true, 1>(a, stream, copies);
      if (a.C == 32) return launch_bf<32, 1, 16, 8, true, 1>(a, stream, copies);
      return launch_bf<64, 1, 8, 8, true, 1>(a, stream, copies);
    }
    if (a.C == 16) return launch_bf<16, 1, 32, 8, true>(a, stream, copies);
    if (a.C == 32) return launch_bf<32, 1, 16, 8, true>(a, stream, copies);
    return launch_bf<64, 1, 8, 8, true>(a, stream, copies);
  }
  if (a.C == 16) return launch_bf<16, 1, 32, 8, false>(a, stream, copies);
  if (a.C == 32) return launch_bf<32, 1, 16, 8, false>(a, stream, copies);
  return launch_bf<64, 1, 8, 8, false>(a, stream, copies);
}

size_t conv_bwd_fused(const ConvBwdFusedArgs& a, hipStream_t stream) { return bf_dispatch(a, stream, nullptr); }

int conv_bwd_fused_copies(const ConvBwdFusedArgs& a) {
  int copies = 0;
  bf_dispatch(a, nullptr, &copies);
  return copies;
}

}  // namespace dtf
