// Fused backward pass of a stride-1 3x3 layer (conv_bwd_fused.hip) on the bf16 matrix pipe with both
// operands of both products split three ways ("bf16x6", see conv_fprop_split.hip), fp32 accumulation.
// Opt-in; interface, prologues (BatchNorm backward of the following layer, shortcut gradient),
// epilogue, statistics and gradient flush are those of the exact-fp32 kernel.
//
// Per band a workgroup (one 16-channel slice of the layer's input channels) stages
//     * the output gradient dy with a one-pixel halo, all C channels, as three bf16 planes (h, m, l)
//     * relu(bn(x)) of its channel slice, no halo, as three bf16 planes of 32 bytes per pixel
// and runs two products from them:
//   data gradient   M = pixels, N = the 16 input channels, K = taps x C: the forward split kernel's
//                   product over the flipped filter. The h and m parts of the filter slice stay in
//                   registers, the l part is read from LDS; the ReLU mask and x-hat come from the raw x
//                   in global memory (just staged by this workgroup, an L2 hit).
//   filter gradient M = input channels, N = output channels, K = pixels of the band. Both operands
//                   are stored pixel-major (channels contiguous), the MFMA wants K = pixels contiguous per
//                   lane: ds_read_b64_tr_b16 transposes 4 pixels x 16 channels per 16 lanes on the way
//                   out of LDS. A K = 32 step takes pixels 32 s .. 32 s + 31 of the band; lane group g
//                   holds pixels 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 of it, so that the 8 pixels a
//                   32-lane half reads at a time are neighbours in a row of the halo tile and, at the
//                   planes' pixel strides (32, 96 bytes), fall on 64 different banks.
// (reference ops: conv2d_backprop_input + conv2d_backprop_filter + the BatchNorm / ReLU gradients
//  under opt.compute_gradients, DistributedResNet/src/distributed_train.py:256)
#include "conv.h"
#include "split_common.h"

#include <cstdlib>

namespace dtf {
namespace {

template <int C, int TW, int TH>
struct BSplit {
  using PL = SplitPlanes<C, TW, TH>;
  static constexpr int KS = split_ksteps(C);
  static constexpr int CO_T = C / 16;
  static constexpr int LH = TH + 2, LW = TW + 2;
  static constexpr int XPB = 32;                        // bytes per pixel of an x plane: the block's 16 channels
  static constexpr int XPLANE = TH * TW * XPB;
  static constexpr int D_CHUNKS = LH * TW * (C / 4);    // the halo columns are zeroed once, never staged
  static constexpr int X_CHUNKS = TH * TW * 4;
  static constexpr int NPF_D = (D_CHUNKS + 255) / 256;
  static constexpr int NPF_X = (X_CHUNKS + 255) / 256;
  static constexpr int N_MT = TH * TW / 16;
  static constexpr int PD = N_MT / 2;                   // pixel tiles per data-gradient wave (waves 0-1)
  static constexpr int PG = 2;                          // of which a group of two shares each filter fragment read
  static constexpr int WL_UNITS = KS * 64;              // 16-byte units of the l part of the block's filter slice
  static constexpr int NWL = (WL_UNITS + 255) / 256;
  // filter gradient (waves 2-3): one output-channel tile each, or one tile and half of the pixels each
  static constexpr int WP = CO_T;
  static constexpr int WX = 2 / WP;
  static constexpr int WSTEPS = TH * TW / 32;           // K = 32 steps of a band
  static constexpr int RED_STRIDE = C + 4;
  static constexpr int RED_BYTES = WX * 9 * 16 * RED_STRIDE * 4;
  static constexpr int TILE_BYTES = 3 * PL::PLANE + 3 * XPLANE;
  static constexpr int BUF_BYTES = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
  static constexpr int LDS_BYTES = BUF_BYTES + WL_UNITS * 16 + (6 * 16 + 3 * C) * 4;
  static_assert(CO_T == 1 || CO_T == 2, "one output-channel tile per filter-gradient wave");
  static_assert(PD % PG == 0 && WSTEPS % WX == 0, "band must split evenly");
  static_assert(TW == 16 || TW == 32, "a K step of the filter gradient is one or two whole rows");
  static_assert(256 % (C / 4) == 0, "a thread's chunks share their channel group");
  static_assert(PL::PLANE < 65536, "LDS offsets must fit 16 bits");
};

// RES: 0 none, 1 identity shortcut, 2 option-A shortcut. STAMPS: the timing instrumentation of
// bench/diag/split_bwd_timeline.py, an instantiation of its own (in the production kernel its pointer and
// counters cost seven spilled registers, each reloaded with a wait inside the band loop).
template <int C, int TW, int TH, bool DYBN, int RES, bool STAMPS = false>
__global__ __launch_bounds__(256, 2) void conv_bwd_split_kernel(ConvBwdFusedArgs a) {
  DTF_ABORT_GUARD(a);
  static_assert(!RES || DYBN, "the shortcut term rides on the BatchNorm prologue");
  using G = BSplit<C, TW, TH>;
  using PL = typename G::PL;
  constexpr int KS = G::KS, CO_T = G::CO_T, LW = G::LW, PSB = PL::PSB, ROWB = PL::ROWB, PLANE = PL::PLANE;
  constexpr int XPB = G::XPB, XPLANE = G::XPLANE, WX = G::WX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* s_dy = smem_b;                                          // [3][LH][ROWB], pixels PSB apart
  unsigned char* s_x = smem_b + 3 * PLANE;                               // [3][TH*TW][XPB]
  uintx4* s_wl = reinterpret_cast<uintx4*>(smem_b + G::BUF_BYTES);       // [KS][64]
  float* s_co = reinterpret_cast<float*>(smem_b + G::BUF_BYTES + G::WL_UNITS * 16);   // [4][16]: scale, shift, mean, rstd
  float* s_dco = s_co + 6 * 16;                                          // [3][C]: dy_eff = ca*g' + cb*y + cc

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  const int l15 = lane & 15;

  constexpr int zsplit = CO_T;
  const int gpz = gridDim.x / zsplit;
  const int bid = blockIdx.x;
  const int z = (bid >> 3) % zsplit;
  const int jblk = (bid / (8 * zsplit)) * 8 + (bid & 7);
  const int bands = a.H / TH;
  const int total_tiles = bands * a.B;
  const int c0 = z * 16;                     // first channel of the block's input-channel slice

  // ---- statistics first, then the filter slice, then the first band ----------------------------------
  const bool co_thread = tid < 16;
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
  // filter slice of the data gradient, [K step][channel tile][h, m, l][lane] x 16 bytes; its l part goes to LDS
  const uintx4* wsrc = reinterpret_cast<const uintx4*>(a.wpk_t);
  uintx4 wl_reg[G::NWL];
#pragma unroll
  for (int k = 0; k < G::NWL; ++k) {
    const int i = tid + k * 256;
    wl_reg[k] = uintx4{0u, 0u, 0u, 0u};
    if (i < G::WL_UNITS) wl_reg[k] = wsrc[(((i >> 6) * CO_T + z) * 3 + 2) * 64 + (i & 63)];
  }

  // Staging geometry. Chunk k of a thread is chunk tid + 256 k of the band: with 256 / (C / 4) pixels per
  // round and TW | that, every round moves down RPK rows in the same column and channel group, so one
  // global offset, one LDS offset and one row per thread describe all of a thread's chunks.
  constexpr int RPK = 256 / (C / 4) / TW;            // halo-tile rows per round of 256 chunks
  static_assert(256 % ((C / 4) * TW) == 0, "a round of chunks is whole rows");
  const int d_r0 = (tid / (C / 4)) / TW, d_ix = (tid / (C / 4)) % TW, d_c4 = tid % (C / 4);
  const int d_goff0 = (d_r0 * TW + d_ix) * C + d_c4 * 4;        // a.W == TW
  const int d_loff0 = d_r0 * ROWB + (d_ix + 1) * PSB + d_c4 * 8;
  auto d_row = [&](int k) -> int { return d_r0 + k * RPK; };    // halo-tile row of chunk k (image row oy0 - 1 + that)
  auto d_ok = [&](int k) -> bool { return (k + 1) * 256 <= G::D_CHUNKS || d_row(k) < G::LH; };
  auto d_goff = [&](int k) -> int { return d_goff0 + k * RPK * TW * C; };
  auto d_loff = [&](int k) -> int { return d_loff0 + k * RPK * ROWB; };
  constexpr int XPK = 64;                            // pixels per round of 256 x chunks
  const int x_goff0 = (tid >> 2) * C + c0 + (tid & 3) * 4;
  const int x_loff0 = (tid >> 2) * XPB + (tid & 3) * 8;
  static_assert(G::X_CHUNKS % 256 == 0, "x chunks fill whole rounds");

  floatx4 pf_d[G::NPF_D], pf_x[G::NPF_X];
  floatx4 pf_c[DYBN ? G::NPF_D : 1];
  floatx4 pf_r[RES ? G::NPF_D : 1];
  // option-A shortcut: the chunk's source in the half-resolution gradient [H/2][TW/2][2C], relative to the
  // band's first pooled row; halo-tile row r is image row oy0 - 1 + r and TH, RPK are even
  static_assert(RES != 2 || (RPK % 2 == 0 && TH % 2 == 0), "pooled rows advance evenly");
  const int r_goff0 = (((d_r0 - 1) >> 1) * (TW / 2) + (d_ix >> 1)) * (2 * C) + C / 2 + d_c4 * 4;
  auto r_goff = [&](int k) -> int { return r_goff0 + k * (RPK / 2) * (TW / 2) * (2 * C); };
  unsigned okbits = 0;
  auto load_band = [&](int t) __attribute__((always_inline)) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const long long doff = ((long long)(n * a.H + oy0 - 1) * a.W) * C;
    const float* d0 = a.dy + doff;
    const float* x0 = a.x + ((long long)(n * a.H + oy0) * a.W) * C;
    okbits = 0;
#pragma unroll
    for (int k = 0; k < G::NPF_D; ++k) {
      const int iy = oy0 - 1 + d_row(k);
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (d_ok(k) && iy >= 0 && iy < a.H) {
        v = *reinterpret_cast<const floatx4*>(d0 + d_goff(k));
        if (DYBN) {
          pf_c[k] = *reinterpret_cast<const floatx4*>(a.dy_x + doff + d_goff(k));
          okbits |= 1u << k;
        }
      }
      pf_d[k] = v;
    }
#pragma unroll
    for (int k = 0; k < G::NPF_X; ++k) {
      pf_x[k] = *reinterpret_cast<const floatx4*>(x0 + x_goff0 + k * XPK * C);
    }
  };
  auto load_res = [&](int t) __attribute__((always_inline)) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const float* r0 = RES == 2 ? a.dy_res + ((long long)(n * (a.H / 2) + oy0 / 2) * (TW / 2)) * (2 * C)
                               : a.dy_res + ((long long)(n * a.H + oy0 - 1) * a.W) * C;
#pragma unroll
    for (int k = 0; k < G::NPF_D; ++k)
      if ((okbits >> k) & 1u) pf_r[k] = *reinterpret_cast<const floatx4*>(r0 + (RES == 2 ? r_goff(k) : d_goff(k)));
  };
  auto write_band = [&](long long doff) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < G::NPF_D; ++k)
      if (d_ok(k)) {
        floatx4 v = pf_d[k];
        if (DYBN && ((okbits >> k) & 1u)) {
          // BatchNorm backward of the following layer, applied while staging (zero padding stays zero)
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
            const int row = d_row(k);
            if (row >= 1 && row <= TH && (unsigned)(ch - c0) < 16u) *reinterpret_cast<floatx4*>(a.dy_out + doff + d_goff(k)) = v;
          }
        }
        uintx2 h, m, l;
        split3(v, h, m, l);
        unsigned char* dst = s_dy + d_loff(k);
        *reinterpret_cast<uintx2*>(dst) = h;
        *reinterpret_cast<uintx2*>(dst + PLANE) = m;
        *reinterpret_cast<uintx2*>(dst + 2 * PLANE) = l;
      }
#pragma unroll
    for (int k = 0; k < G::NPF_X; ++k)
      {
        // the filter gradient's activation: BatchNorm + ReLU, then the split
        const int ch = (tid & 3) * 4;
        const floatx4 sc = *reinterpret_cast<const floatx4*>(s_co + ch);
        const floatx4 sh = *reinterpret_cast<const floatx4*>(s_co + 16 + ch);
        floatx4 v = pf_x[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        uintx2 h, m, l;
        split3(v, h, m, l);
        unsigned char* dst = s_x + x_loff0 + k * XPK * XPB;
        *reinterpret_cast<uintx2*>(dst) = h;
        *reinterpret_cast<uintx2*>(dst + XPLANE) = m;
        *reinterpret_cast<uintx2*>(dst + 2 * XPLANE) = l;
      }
  };

  int t = jblk;
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
    s_co[16 + tid] = bet - m * sc;
    s_co[32 + tid] = m;
    s_co[48 + tid] = r;
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
  for (int k = 0; k < G::NWL; ++k)
    if (tid + k * 256 < G::WL_UNITS) s_wl[tid + k * 256] = wl_reg[k];
  // left / right halo column of the dy planes: zero padding for every band
  for (int i = tid; i < 3 * G::LH * 2 * (PSB / 4); i += 256) {
    const int w = i % (PSB / 4), side = (i / (PSB / 4)) & 1, r = (i / (2 * (PSB / 4))) % G::LH, plane = i / (2 * (PSB / 4) * G::LH);
    reinterpret_cast<unsigned*>(s_dy + plane * PLANE + r * ROWB + (side ? LW - 1 : 0) * PSB)[w] = 0u;
  }
  __syncthreads();

  // ---- wave roles and per-lane constants ----------------------------------------------------------------
  // Waves 0-1 compute the data gradient of the whole band, waves 2-3 its filter gradient, at the same
  // time: a data-gradient wave keeps the filter fragments in registers and no filter-gradient
  // accumulators, a filter-gradient wave the other way round, so both fit the register file next to
  // the next band's prefetch (a wave doing both phases in turn spills the prefetched values, which
  // makes the loads synchronous). The MFMA counts of the two roles are equal (32 channels) or 10 : 9.
  const bool dgrad_wave = wave < 2;
  const int rw = wave & 1;                   // index within the role
  // data gradient: B-operand byte offsets as in conv_fprop_split.hip (channel slot of the lane group +
  // the K step's tap shift; with 16 channels lane groups 2-3 read the tap after that of groups 0-1)
  const int q_slot = 16 * (C == 16 ? (q & 1) : q);
  const int d1 = (C == 16 && q >= 2) ? PSB : 0;
  const int dx = (C == 16 && q >= 2) ? ROWB - 2 * PSB : 0;
  auto kconst = [](int ks) constexpr -> int {
    const int tap = C == 32 ? ks : 2 * ks;
    return (tap / 3) * ROWB + (tap % 3) * PSB;
  };
  auto kvar = [&](int ks) -> int { return C != 16 ? 0 : (ks == 1 ? dx : (ks == 4 ? 0 : d1)); };
  const int lane_o = l15 * C + c0 + 4 * q;
  constexpr int PD = G::PD, PG = G::PG;      // pixel tiles per data-gradient wave, and per group sharing a filter fragment
  // byte offset of pixel tile 0 of the wave inside a plane; tile i of the wave adds tile_off(i)
  const int poff0 = ((rw * PD * 16 + l15) / TW) * ROWB + ((rw * PD * 16 + l15) % TW) * PSB + q_slot;
  auto tile_off = [](int i) constexpr -> int { return TW >= 16 ? ((i * 16) / TW) * ROWB + ((i * 16) % TW) * PSB : (i * 16 / TW) * ROWB; };
  // filter gradient: this wave's output-channel tile and share of the band's K steps; transposed-read
  // lane addresses: lane 4 j + i of a group points at pixel 4 g + j of the step, channels 4 i .. 4 i + 3
  const int wp = CO_T == 1 ? 0 : rw, wx = CO_T == 1 ? rw : 0;
  const int tr_j = (lane >> 2) & 3, tr_c = lane & 3;
  const int lane_xa = (4 * q + tr_j) * XPB + tr_c * 8;
  const int lane_db = (TW == 8 ? (q >> 1) * ROWB + (4 * (q & 1) + tr_j) * PSB : (4 * q + tr_j) * PSB) + wp * 32 + tr_c * 8;

  floatx4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
  constexpr int RS = G::RED_STRIDE;
  float* s_red = reinterpret_cast<float*>(smem_b);   // [WX][9][16][RS], after the last band

  // The persistent band loop, once per role: stage, barrier, request the next band, the role's MFMA phase, barrier.
  auto band_loop = [&](auto&& phase) __attribute__((always_inline)) {
    int t_next = t + gpz;
    int band_no = 0;
    unsigned long long* stamps = nullptr;
    if constexpr (STAMPS) stamps = a.dbg_stamps ? a.dbg_stamps + ((size_t)blockIdx.x * 4 + wave) * 48 : nullptr;
    auto stamp = [&](int k) {   // timing experiments only (bench/diag/split_bwd_timeline.py)
      if constexpr (STAMPS) {
        if (stamps && lane == 0 && band_no < 7) stamps[band_no * 6 + k] = wall_clock64();
      }
    };
    if constexpr (STAMPS) {
      if (stamps && lane == 0) {   // where the wave runs: HW_ID (wave, SIMD, CU, SE) and XCC_ID, raw (bench/diag/wg_placement.py)
        stamps[42] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        stamps[43] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      }
    }
    while (t < total_tiles) {
      const int n = t / bands;
      const int oy0 = (t - n * bands) * TH;
      stamp(0);
      write_band(((long long)(n * a.H + oy0 - 1) * a.W) * C);
      stamp(1);
      __syncthreads();
      stamp(2);
      if (t_next < total_tiles) {   // in flight during the MFMA phase
        load_band(t_next);
        if (RES) load_res(t_next);
      }
      stamp(3);
      phase(((long long)(n * a.H + oy0) * TW) * C);
      stamp(4);
      __syncthreads();   // all waves are done with this band's planes
      stamp(5);
      ++band_no;
      t = t_next;
      t_next += gpz;
    }
  };

  if (dgrad_wave) {
    // h and m parts of the filter slice stay in registers
    uintx4 wh[KS], wm[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wh[ks] = wsrc[((ks * CO_T + z) * 3 + 0) * 64 + lane];
      wm[ks] = wsrc[((ks * CO_T + z) * 3 + 1) * 64 + lane];
    }
    band_loop([&](long long obase) __attribute__((always_inline)) {
      // ---- data gradient: M = pixels, N = the block's input channels, K = taps x C ------------------------
      static_assert((PG * 16) % TW == 0, "a group of pixel tiles is whole rows");
#pragma unroll 1
      for (int grp = 0; grp < PD / PG; ++grp) {
        const unsigned char* s_grp = s_dy + poff0 + grp * ((PG * 16) / TW) * ROWB;
        // mask / x-hat source of the group's pixel tiles (the extra gradient of a first block is read in the epilogue)
        floatx4 cur_xv[PG], acc[PG];
#pragma unroll
        for (int i = 0; i < PG; ++i) {
          cur_xv[i] = *reinterpret_cast<const floatx4*>(a.x + obase + (long long)((rw * PD + grp * PG + i) * 16) * C + lane_o);
          acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
        // Walk (K step, pixel tile); the operand reads of the next unit are in flight while the six MFMAs of this one run.
        constexpr int UNITS = KS * PG;
        uintx4 ah[2], am[2], al[2], wl[2];
        auto read_unit = [&](int u, int buf) {
          const int ks = u / PG, i = u % PG;
          const unsigned char* p = s_grp + tile_off(i) + kvar(ks) + kconst(ks);
          ah[buf] = *reinterpret_cast<const uintx4*>(p);
          am[buf] = *reinterpret_cast<const uintx4*>(p + PLANE);
          al[buf] = *reinterpret_cast<const uintx4*>(p + 2 * PLANE);
          if (i == 0) wl[ks & 1] = s_wl[ks * 64 + lane];
        };
        read_unit(0, 0);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
          const int ks = u / PG, i = u % PG, b = u & 1;
          if (u + 1 < UNITS) read_unit(u + 1, b ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 whv = __builtin_bit_cast(bf16x8, wh[ks]);
          const bf16x8 wmv = __builtin_bit_cast(bf16x8, wm[ks]);
          const bf16x8 wlv = __builtin_bit_cast(bf16x8, wl[ks & 1]);
          const bf16x8 ahv = __builtin_bit_cast(bf16x8, ah[b]), amv = __builtin_bit_cast(bf16x8, am[b]), alv = __builtin_bit_cast(bf16x8, al[b]);
          // small terms first, then the large ones
          acc[i] = DTF_MFMA_BF16(wlv, ahv, acc[i]);
          acc[i] = DTF_MFMA_BF16(whv, alv, acc[i]);
          acc[i] = DTF_MFMA_BF16(wmv, amv, acc[i]);
          acc[i] = DTF_MFMA_BF16(wmv, ahv, acc[i]);
          acc[i] = DTF_MFMA_BF16(whv, amv, acc[i]);
          acc[i] = DTF_MFMA_BF16(whv, ahv, acc[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: ReLU mask and x-hat from the raw x, statistics, 16-byte stores
        __builtin_amdgcn_sched_barrier(0);   // the coefficient reads stay behind the MFMA loop: registers
        const floatx4 esc = *reinterpret_cast<const floatx4*>(s_co + 4 * q), esh = *reinterpret_cast<const floatx4*>(s_co + 16 + 4 * q);
        const floatx4 emu = *reinterpret_cast<const floatx4*>(s_co + 32 + 4 * q), ers = *reinterpret_cast<const floatx4*>(s_co + 48 + 4 * q);
#pragma unroll
        for (int i = 0; i < PG; ++i) {
          const int p0 = (rw * PD + grp * PG + i) * 16;
          floatx4 v = acc[i];
          if (a.extra) v += *reinterpret_cast<const floatx4*>(a.extra + obase + (long long)p0 * C + lane_o);
          const floatx4 x = cur_xv[i];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = (x[r] * esc[r] + esh[r] > 0.f) ? v[r] : 0.f;
            st1[r] += v[r];
            st2[r] += v[r] * ((x[r] - emu[r]) * ers[r]);
          }
          *reinterpret_cast<floatx4*>(a.g + obase + (long long)p0 * C + lane_o) = v;
        }
      }
    });
  } else {
    floatx4 accw[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) accw[tap] = floatx4{0.f, 0.f, 0.f, 0.f};
    band_loop([&](long long) __attribute__((always_inline)) {
      // ---- filter gradient: M = input channels, N = output channels, K = pixels of the band --------------------
      constexpr int MYSTEPS = G::WSTEPS / WX;
      constexpr int WUNITS = MYSTEPS * 9;
      uintx2 xa[3][2];        // A: relu(bn(x)) parts of the current step
      uintx2 db[2][3][2];     // B: dy parts of the current / next (step, tap)
      // The band's sum is formed in accumulators of its own and added to the running total once: the
      // total then takes one rounding per band instead of one per MFMA (a few hundred per workgroup),
      // which is what the error of a sum over 10^5 pixels is made of.
      floatx4 accb[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) accb[tap] = floatx4{0.f, 0.f, 0.f, 0.f};
      auto read_a = [&](int ms) {
        const unsigned char* p = s_x + lane_xa + (wx + ms * WX) * 32 * XPB;
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          xa[part][0] = lds_read_tr(p + part * XPLANE);
          xa[part][1] = lds_read_tr(p + part * XPLANE + 16 * XPB);
        }
      };
      auto read_b = [&](int u, int buf) {
        const int ms = u / 9, tap = u % 9;
        const int kh = tap / 3, kw = tap % 3;
        // dy[pixel - (tap - 1)]: halo position (row + 2 - kh, col + 2 - kw); a step is 32 / TW whole rows
        constexpr int SROWS = 32 / TW > 0 ? 32 / TW : 1;
        const unsigned char* p0 = s_dy + lane_db + (wx + ms * WX) * (SROWS * ROWB) + (2 - kh) * ROWB + (2 - kw) * PSB;
        const unsigned char* p1 = p0 + (TW == 32 ? 16 * PSB : (16 / TW) * ROWB);
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          db[buf][part][0] = lds_read_tr(p0 + part * PLANE);
          db[buf][part][1] = lds_read_tr(p1 + part * PLANE);
        }
      };
      read_a(0);
      read_b(0, 0);
#pragma unroll
      for (int u = 0; u < WUNITS; ++u) {
        const int tap = u % 9, b = u & 1;
        if (u + 1 < WUNITS) read_b(u + 1, b ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 xh = frag8(xa[0][0], xa[0][1]), xm = frag8(xa[1][0], xa[1][1]), xl = frag8(xa[2][0], xa[2][1]);
        const bf16x8 dh = frag8(db[b][0][0], db[b][0][1]), dm = frag8(db[b][1][0], db[b][1][1]), dl = frag8(db[b][2][0], db[b][2][1]);
        accb[tap] = DTF_MFMA_BF16(xl, dh, accb[tap]);
        accb[tap] = DTF_MFMA_BF16(xh, dl, accb[tap]);
        accb[tap] = DTF_MFMA_BF16(xm, dm, accb[tap]);
        accb[tap] = DTF_MFMA_BF16(xm, dh, accb[tap]);
        accb[tap] = DTF_MFMA_BF16(xh, dm, accb[tap]);
        accb[tap] = DTF_MFMA_BF16(xh, dh, accb[tap]);
        __builtin_amdgcn_sched_barrier(0);
        if (tap == 8 && u + 1 < WUNITS) read_a(u / 9 + 1);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) accw[tap] += accb[tap];
    });
    // accumulators -> LDS; the planes are free (barrier at the end of the last band)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        s_red[((wx * 9 + tap) * 16 + 4 * q + r) * RS + wp * 16 + l15] = accw[tap][r];
  }

  // ---- statistics (the filter's l part in LDS is no longer needed: its space parks the wave rows) --------
  if (a.gsums) {
    floatx4 s1v[1] = {st1}, s2v[1] = {st2};   // zeros from the filter-gradient waves
    block_stat_flush<1, 4>(reinterpret_cast<float*>(s_wl), s1v, s2v, a.gsums, C, c0, kStatGridBwd);
  }
  // ---- filter gradient flush: sum the pixel-split partials, coalesced stores / atomics ---------------------
  __syncthreads();
  // with dw_part: plain stores into copy jblk of the whole gradient (this block's rows of it)
  float* dw = a.dw_part ? a.dw_part + (size_t)jblk * (9 * C * C) : a.dw + (size_t)(blockIdx.x & (kStatSlots - 1)) * a.dw_slot_stride;
  for (int e = tid; e < 9 * 16 * C; e += 256) {
    const int col = e % C;
    const int row = (e / C) % 16;
    const int tap = e / (C * 16);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WX; ++w) v += s_red[((w * 9 + tap) * 16 + row) * RS + col];
    float* dst = dw + ((size_t)tap * C + c0 + row) * C + col;
    if (a.dw_part) *dst = v;
    else atomicAdd(dst, v);
  }
}

// copies != nullptr: no launch, *copies = workgroups per channel split (= copies of the gradient
// filled through ConvBwdFusedArgs::dw_part)
template <int C, int TW, int TH, bool DYBN, int RES = 0, bool STAMPS = false>
size_t launch_bs(const ConvBwdFusedArgs& a, hipStream_t stream, int* copies) {
  using G = BSplit<C, TW, TH>;
  const size_t shmem = (size_t)G::LDS_BYTES;
  static_assert(2 * G::LDS_BYTES <= (int)kLdsBudget, "two workgroups per CU");
  constexpr int zsplit = C / 16;
  const int total_tiles = (a.H / TH) * a.B;
  static const int bpc = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_bwd_split_kernel<C, TW, TH, DYBN, RES, STAMPS>), shmem);
    int n = 0;
    DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_bwd_split_kernel<C, TW, TH, DYBN, RES, STAMPS>, 256, shmem));
    return n < 1 ? 1 : (n > 2 ? 2 : n);
  }();
  int gpz_max = (256 * bpc) / zsplit;
  if (total_tiles / gpz_max < 4 && bpc > 1) gpz_max = 256 / zsplit;   // as conv_bwd_fused.hip: few bands, one workgroup per CU
  if (gpz_max < 8) gpz_max = 8;
  const int iters = ceil_div(total_tiles, gpz_max);
  int gpz = ceil_div(total_tiles, iters);
  gpz = (gpz + 7) / 8 * 8;
  if (copies) {
    *copies = gpz;
    return 0;
  }
  launch("conv_bwd_split", conv_bwd_split_kernel<C, TW, TH, DYBN, RES, STAMPS>, dim3(gpz * zsplit), dim3(256), shmem, stream, a);
  return shmem;
}

}  // namespace

bool conv_bwd_split_supported(int B, int H, int W, int C) {
  (void)B;
  return (H == W && H % 8 == 0 && ((C == 16 && W == 32) || (C == 32 && W == 16))) || conv_bwd_split_pc_supported(B, H, W, C);
}

static size_t bs_dispatch(const ConvBwdFusedArgs& a, hipStream_t stream, int* copies) {
  if (!conv_bwd_split_supported(a.B, a.H, a.W, a.C)) throw std::runtime_error("conv_bwd_split: unsupported shape");
  if (!a.dy || !a.x || !a.wpk_t || !a.g || !a.dw || !a.bn.sums) throw std::runtime_error("conv_bwd_split: null operand");
  if (a.tile_counters) throw std::runtime_error("conv_bwd_split: no dynamic band scheduling");
  if (a.dy_x && (!a.dy_bn.sums || !a.dy_gsums)) throw std::runtime_error("conv_bwd_split: the BatchNorm prologue needs its statistics");
  if (a.dy_res && (!a.dy_x || !a.dy_out)) throw std::runtime_error("conv_bwd_split: dy_res needs dy_x and dy_out");
  // Producer / consumer waves (conv_bwd_split_pc.hip): always at 64 channels; at 16 channels for the
  // shortcut variants (three prefetched tensors: 8-row bands there against 4-row bands here, 90-93 us
  // against 96-98 us in the step at batch 1024). Everything else is faster with two workgroups per CU
  // (16 channels + BatchNorm prologue 93 against 110 us, 32 channels 60 against 76 us).
  // DTF_BWD_SPLIT_PC=16,32 forces that organisation for a channel count (measurements).
  static const int pc_mask = [] {
    int m = 64;
    if (const char* e = std::getenv("DTF_BWD_SPLIT_PC"))
      for (const char* p = e; *p;) {
        if (*p >= '0' && *p <= '9') {
          const int v = std::atoi(p);
          if (v == 16 || v == 32) m |= v;
          while (*p >= '0' && *p <= '9') ++p;
        } else {
          ++p;
        }
      }
    return m;
  }();
  if ((a.C & pc_mask) || (a.C == 16 && a.dy_res)) return conv_bwd_split_pc(a, stream, copies);
  if (a.dy_x) {
    if (!a.dy_bn.sums || !a.dy_gsums) throw std::runtime_error("conv_bwd_split: the BatchNorm prologue needs its statistics");
    if (a.dy_res) {
      if (!a.dy_out) throw std::runtime_error("conv_bwd_split: dy_res needs dy_out");
      // three prefetched tensors: 4-row bands keep them in registers (8-row bands spill 40-50 of them)
      if (a.dy_res_mode == 1) {
        if (a.C == 16) return launch_bs<16, 32, 4, true, 2>(a, stream, copies);
        return launch_bs<32, 16, 4, true, 2>(a, stream, copies);
      }
      if (a.C == 16) return launch_bs<16, 32, 4, true, 1>(a, stream, copies);
      return launch_bs<32, 16, 4, true, 1>(a, stream, copies);
    }
    if (a.dbg_stamps) {
      if (a.C != 16) throw std::runtime_error("conv_bwd_split: timing stamps exist for the 16-channel BatchNorm-prologue kernel only");
      return launch_bs<16, 32, 8, true, 0, true>(a, stream, copies);
    }
    if (a.C == 16) return launch_bs<16, 32, 8, true>(a, stream, copies);
    return launch_bs<32, 16, 8, true>(a, stream, copies);
  }
  if (a.C == 16) return launch_bs<16, 32, 8, false>(a, stream, copies);
  return launch_bs<32, 16, 8, false>(a, stream, copies);
}

size_t conv_bwd_split(const ConvBwdFusedArgs& a, hipStream_t stream) { return bs_dispatch(a, stream, nullptr); }

int conv_bwd_split_copies(const ConvBwdFusedArgs& a) {
  int copies = 0;
  bs_dispatch(a, nullptr, &copies);
  return copies;
}

}  // namespace dtf
