// Fused backward pass of a stride-1 3x3 layer on the bf16 matrix pipe with split operands (as
// conv_bwd_split.hip: same products, prologues, epilogue, statistics and gradient flush), organised as
// producer and consumer waves for the layers whose filter slice does not fit next to a prefetch:
//
//   one workgroup of 8 waves per CU, two sets of staged planes in LDS
//   waves 0-1  data gradient of the band in set j % 2: the h AND m parts of the filter slice (K steps x 2
//              x 4 registers: 144 at 64 channels) stay in registers, the l part in LDS; no prefetch
//              registers, no vector-memory traffic besides the epilogue's
//   waves 2-3  filter gradient of the same band (transposed reads, conv_bwd_split.hip): one or two
//              output-channel tiles each
//   waves 4-7  producers: write band j + 1 (held in registers since two iterations ago) into set
//              (j + 1) % 2 - BatchNorm backward, shortcut term, BatchNorm + ReLU, three-way split - and
//              request band j + 3 (memory latency is longer than one band's MFMA phase, so two bands are
//              in flight); one barrier per band
// so the matrix waves never wait for memory or for staging, and staging costs them only the issue
// slots it takes on their SIMD. (A 64-channel filter slice streamed from L2 instead would queue its
// refills behind the prefetch loads - vector memory operations retire in order - and a second
// workgroup per CU leaves no LDS for it.)
// (reference ops: conv2d_backprop_input + conv2d_backprop_filter + the BatchNorm / ReLU gradients
//  under opt.compute_gradients, DistributedResNet/src/distributed_train.py:256)
#include "conv.h"
#include "split_common.h"

#include <cstdlib>

namespace dtf {
namespace {

template <int C, int TW, int TH>
struct BPC {
  using PL = SplitPlanes<C, TW, TH>;
  static constexpr int KS = split_ksteps(C);
  static constexpr int CO_T = C / 16;
  static constexpr int LH = TH + 2, LW = TW + 2;
  static constexpr int XPB = 32;                        // bytes per pixel of an x plane: the block's 16 channels
  static constexpr int XPLANE = TH * TW * XPB;
  static constexpr int D_CHUNKS = LH * TW * (C / 4);    // the halo columns are zeroed once, never staged
  static constexpr int X_CHUNKS = TH * TW * 4;
  static constexpr int NPF_D = (D_CHUNKS + 255) / 256;  // per producer thread (256 of them)
  static constexpr int NPF_X = (X_CHUNKS + 255) / 256;
  static constexpr int N_MT = TH * TW / 16;
  static constexpr int PD = N_MT / 2;                   // pixel tiles per data-gradient wave
  static constexpr int PG = 2;                          // of which a group of two shares each filter fragment read
  static constexpr int WL_UNITS = KS * 64;              // 16-byte units of the l part of the block's filter slice
  static constexpr int CPW = CO_T > 2 ? CO_T / 2 : 1;   // output-channel tiles per filter-gradient wave
  static constexpr int WX = CO_T == 1 ? 2 : 1;          // or halves of the pixels
  static constexpr int WSTEPS = TH * TW / 32;           // K = 32 steps of a band
  static constexpr int RED_STRIDE = C + 4;
  static constexpr int RED_BYTES = WX * 9 * 16 * RED_STRIDE * 4;
  static constexpr int SET_BYTES = 3 * PL::PLANE + 3 * XPLANE;   // one set of planes
  static constexpr int LDS_BYTES = 2 * SET_BYTES + WL_UNITS * 16 + (6 * 16 + 3 * C) * 4;
  static_assert(2 * SET_BYTES >= RED_BYTES, "the gradient flush reuses the planes");
  static_assert(PD % PG == 0 && WSTEPS % WX == 0, "band must split evenly");
  static_assert(256 % ((C / 4) * TW) == 0 && X_CHUNKS % 256 == 0, "a round of chunks is whole rows");
  static_assert(LDS_BYTES <= (int)kLdsBudget, "one workgroup per CU");
};

template <int C, int TW, int TH, bool DYBN, int RES>   // RES: 0 none, 1 identity shortcut, 2 option-A shortcut
__global__ __launch_bounds__(512, 1) void conv_bwd_split_pc_kernel(ConvBwdFusedArgs a) {
  DTF_ABORT_GUARD(a);
  static_assert(!RES || DYBN, "the shortcut term rides on the BatchNorm prologue");
  using G = BPC<C, TW, TH>;
  using PL = typename G::PL;
  constexpr int KS = G::KS, CO_T = G::CO_T, LW = G::LW, PSB = PL::PSB, ROWB = PL::ROWB, PLANE = PL::PLANE;
  constexpr int XPB = G::XPB, XPLANE = G::XPLANE, WX = G::WX, CPW = G::CPW, SET = G::SET_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  // set s: dy planes [3][LH][ROWB] at smem_b + s * SET, x planes [3][TH*TW][XPB] behind them
  uintx4* s_wl = reinterpret_cast<uintx4*>(smem_b + 2 * SET);           // [KS][64]
  float* s_co = reinterpret_cast<float*>(smem_b + 2 * SET + G::WL_UNITS * 16);   // [4][16]: scale, shift, mean, rstd
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
  const int t0 = jblk;

  // ---- common prologue: BatchNorm coefficients, the filter's l part and the zero halo columns -----------
  if (tid < 16) {
    const double s1 = stat_total(a.bn.sums, c0 + tid, C), s2 = stat_total(a.bn.sums, C + c0 + tid, C);
    const double mean = s1 * (double)a.bn.inv_count;
    double var = s2 * (double)a.bn.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float r = rsqrtf((float)var + a.bn.eps);
    const float m = (float)mean;
    const float sc = a.bn.gamma[c0 + tid] * r;
    s_co[tid] = sc;
    s_co[16 + tid] = a.bn.beta[c0 + tid] - m * sc;
    s_co[32 + tid] = m;
    s_co[48 + tid] = r;
  }
  if (DYBN && tid >= 64 && tid < 64 + C) {
    const int c = tid - 64;
    const double ds1 = stat_total(a.dy_bn.sums, c, C), ds2 = stat_total(a.dy_bn.sums, C + c, C);
    const double dg1 = stat_total(a.dy_gsums, c, C), dg2 = stat_total(a.dy_gsums, C + c, C);
    const double inv = (double)a.dy_bn.inv_count;
    const double mean = ds1 * inv;
    double var = ds2 * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = rsqrtf((float)var + a.dy_bn.eps);
    const float k1 = a.dy_bn.gamma[c] * rstd;
    const float mg = (float)(dg1 * inv), mgx = (float)(dg2 * inv);
    s_dco[c] = k1;
    s_dco[C + c] = -k1 * mgx * rstd;
    s_dco[2 * C + c] = k1 * (mgx * (float)mean * rstd - mg);
    if (blockIdx.x == 0) {
      if (a.dy_dbeta) a.dy_dbeta[c] = (float)dg1;
      if (a.dy_dgamma) a.dy_dgamma[c] = (float)dg2;
    }
  }
  // filter slice of the data gradient, [K step][channel tile][h, m, l][lane] x 16 bytes
  const uintx4* wsrc = reinterpret_cast<const uintx4*>(a.wpk_t);
  for (int i = tid; i < G::WL_UNITS; i += 512) s_wl[i] = wsrc[(((i >> 6) * CO_T + z) * 3 + 2) * 64 + (i & 63)];
  for (int i = tid; i < 2 * 3 * G::LH * 2 * (PSB / 4); i += 512) {
    const int w = i % (PSB / 4), side = (i / (PSB / 4)) & 1, r = (i / (2 * (PSB / 4))) % G::LH;
    const int plane = (i / (2 * (PSB / 4) * G::LH)) % 3, set = i / (2 * (PSB / 4) * G::LH * 3);
    reinterpret_cast<unsigned*>(smem_b + set * SET + plane * PLANE + r * ROWB + (side ? LW - 1 : 0) * PSB)[w] = 0u;
  }
  __syncthreads();

  floatx4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
  constexpr int RS = G::RED_STRIDE;
  float* s_red = reinterpret_cast<float*>(smem_b);   // [WX][9][16][RS], after the last band
  const int rw = wave & 1;                           // index within a matrix role

  if (wave >= 4) {
    // ================= producers: 256 threads stage every band ===========================================
    const int ts = tid & 255;
    // Chunk k of a thread is chunk ts + 256 k of the band: every round of 256 chunks moves down RPK rows in
    // the same column and channel group, so one global offset, one LDS offset and one row describe them all.
    constexpr int RPK = 256 / (C / 4) / TW;
    const int d_r0 = (ts / (C / 4)) / TW, d_ix = (ts / (C / 4)) % TW, d_c4 = ts % (C / 4);
    const int d_goff0 = (d_r0 * TW + d_ix) * C + d_c4 * 4;        // a.W == TW
    const int d_loff0 = d_r0 * ROWB + (d_ix + 1) * PSB + d_c4 * 8;
    auto d_row = [&](int k) -> int { return d_r0 + k * RPK; };    // halo-tile row of chunk k (image row oy0 - 1 + that)
    auto d_ok = [&](int k) -> bool { return (k + 1) * 256 <= G::D_CHUNKS || d_row(k) < G::LH; };
    auto d_goff = [&](int k) -> int { return d_goff0 + k * RPK * TW * C; };
    auto d_loff = [&](int k) -> int { return d_loff0 + k * RPK * ROWB; };
    constexpr int XPK = 64;                            // pixels per round of 256 x chunks
    const int x_goff0 = (ts >> 2) * C + c0 + (ts & 3) * 4;
    const int x_loff0 = (ts >> 2) * XPB + (ts & 3) * 8;
    // option-A shortcut: the chunk's source in the half-resolution gradient [H/2][TW/2][2C], relative to
    // the band's first pooled row
    static_assert(RES != 2 || (RPK % 2 == 0 && TH % 2 == 0), "pooled rows advance evenly");
    const int r_goff0 = (((d_r0 - 1) >> 1) * (TW / 2) + (d_ix >> 1)) * (2 * C) + C / 2 + d_c4 * 4;
    auto r_goff = [&](int k) -> int { return r_goff0 + k * (RPK / 2) * (TW / 2) * (2 * C); };

    // a band in flight: its chunks in registers between the loads and the staging
    struct Pf {
      floatx4 d[G::NPF_D], x[G::NPF_X];
      floatx4 c[DYBN ? G::NPF_D : 1];
      floatx4 r[RES ? G::NPF_D : 1];
      unsigned ok;
    };
    Pf pfa, pfb;   // two bands are in flight: memory latency is longer than one band's MFMA phase
    auto load_band = [&](int t, Pf& pf) __attribute__((always_inline)) {
      const int n = t / bands;
      const int oy0 = (t - n * bands) * TH;
      const long long doff = ((long long)(n * a.H + oy0 - 1) * a.W) * C;
      const float* d0 = a.dy + doff;
      const float* x0 = a.x + ((long long)(n * a.H + oy0) * a.W) * C;
      const float* r0 = !RES ? nullptr
                             : (RES == 2 ? a.dy_res + ((long long)(n * (a.H / 2) + oy0 / 2) * (TW / 2)) * (2 * C) : a.dy_res + doff);
      pf.ok = 0;
#pragma unroll
      for (int k = 0; k < G::NPF_D; ++k) {
        const int iy = oy0 - 1 + d_row(k);
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (d_ok(k) && iy >= 0 && iy < a.H) {
          v = *reinterpret_cast<const floatx4*>(d0 + d_goff(k));
          if (DYBN) {
            pf.c[k] = *reinterpret_cast<const floatx4*>(a.dy_x + doff + d_goff(k));
            pf.ok |= 1u << k;
          }
          if (RES) pf.r[k] = *reinterpret_cast<const floatx4*>(r0 + (RES == 2 ? r_goff(k) : d_goff(k)));
        }
        pf.d[k] = v;
      }
#pragma unroll
      for (int k = 0; k < G::NPF_X; ++k) pf.x[k] = *reinterpret_cast<const floatx4*>(x0 + x_goff0 + k * XPK * C);
    };
    auto write_band = [&](int t, unsigned char* set, const Pf& pf) __attribute__((always_inline)) {
      const int n = t / bands;
      const int oy0 = (t - n * bands) * TH;
      const long long doff = ((long long)(n * a.H + oy0 - 1) * a.W) * C;
      unsigned char* s_dy = set;
      unsigned char* s_x = set + 3 * PLANE;
#pragma unroll
      for (int k = 0; k < G::NPF_D; ++k)
        if (d_ok(k)) {
          floatx4 v = pf.d[k];
          if (DYBN && ((pf.ok >> k) & 1u)) {
            // BatchNorm backward of the following layer, applied while staging (zero padding stays zero)
            const int ch = d_c4 * 4;
            const floatx4 ca = *reinterpret_cast<const floatx4*>(s_dco + ch);
            const floatx4 cb = *reinterpret_cast<const floatx4*>(s_dco + C + ch);
            const floatx4 cc = *reinterpret_cast<const floatx4*>(s_dco + 2 * C + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ca[e] * v[e] + (cb[e] * pf.c[k][e] + cc[e]);
            if (RES) {
              if (RES == 2) v += 0.25f * pf.r[k];
              else v += pf.r[k];
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
      for (int k = 0; k < G::NPF_X; ++k) {
        // the filter gradient's activation: BatchNorm + ReLU, then the split
        const int ch = (ts & 3) * 4;
        const floatx4 sc = *reinterpret_cast<const floatx4*>(s_co + ch);
        const floatx4 sh = *reinterpret_cast<const floatx4*>(s_co + 16 + ch);
        floatx4 v = pf.x[k];
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
    // first band into set 0; bands two and three into the registers
    if (t0 < total_tiles) {
      load_band(t0, pfa);
      if (t0 + gpz < total_tiles) load_band(t0 + gpz, pfb);
      write_band(t0, smem_b, pfa);
      if (t0 + 2 * gpz < total_tiles) load_band(t0 + 2 * gpz, pfa);
    }
    __syncthreads();   // set 0 is staged
    // iteration on band t: stage band t + 1 (requested two iterations ago) into the other set, request band t + 3
    for (int t = t0; t < total_tiles; t += 2 * gpz) {
      if (t + gpz < total_tiles) {
        write_band(t + gpz, smem_b + SET, pfb);
        if (t + 3 * gpz < total_tiles) load_band(t + 3 * gpz, pfb);
      }
      __syncthreads();   // the matrix waves are done with set 0, set 1 is staged
      if (t + gpz >= total_tiles) break;
      if (t + 2 * gpz < total_tiles) {
        write_band(t + 2 * gpz, smem_b, pfa);
        if (t + 4 * gpz < total_tiles) load_band(t + 4 * gpz, pfa);
      }
      __syncthreads();   // the matrix waves are done with set 1, set 0 is staged
    }
  } else if (wave < 2) {
    // ================= data gradient: M = pixels, N = the block's input channels, K = taps x C =============
    uintx4 wh[KS], wm[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wh[ks] = wsrc[((ks * CO_T + z) * 3 + 0) * 64 + lane];
      wm[ks] = wsrc[((ks * CO_T + z) * 3 + 1) * 64 + lane];
    }
    // B-operand byte offsets as in conv_fprop_split.hip (channel slot of the lane group + the K step's
    // tap shift; with 16 channels lane groups 2-3 read the tap after that of groups 0-1)
    const int q_slot = 16 * (C == 16 ? (q & 1) : q);
    const int d1 = (C == 16 && q >= 2) ? PSB : 0;
    const int dx = (C == 16 && q >= 2) ? ROWB - 2 * PSB : 0;
    auto kconst = [](int ks) constexpr -> int {
      const int tap = C == 64 ? ks >> 1 : (C == 32 ? ks : 2 * ks);
      return (tap / 3) * ROWB + (tap % 3) * PSB + (C == 64 ? 64 * (ks & 1) : 0);
    };
    auto kvar = [&](int ks) -> int { return C != 16 ? 0 : (ks == 1 ? dx : (ks == 4 ? 0 : d1)); };
    const int lane_o = l15 * C + c0 + 4 * q;
    constexpr int PD = G::PD, PG = G::PG;
    const int poff0 = ((rw * PD * 16 + l15) / TW) * ROWB + ((rw * PD * 16 + l15) % TW) * PSB + q_slot;
    auto tile_off = [](int i) constexpr -> int { return ((i * 16) / TW) * ROWB + ((i * 16) % TW) * PSB; };
    static_assert((PG * 16) % TW == 0, "a group of pixel tiles is whole rows");
    __syncthreads();   // set 0 is staged
    int set = 0;
    for (int t = t0; t < total_tiles; t += gpz) {
      const int n = t / bands;
      const int oy0 = (t - n * bands) * TH;
      const long long obase = ((long long)(n * a.H + oy0) * TW) * C;
      const unsigned char* s_dy = smem_b + set * SET;
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
        // Walk the K steps; the operand reads of the next step are in flight while the twelve MFMAs of this
        // one run. The two pixel tiles alternate: with one matrix wave per SIMD an MFMA that accumulates into
        // the previous MFMA's result would wait for it.
        static_assert(PG == 2, "two accumulators alternate");
        uintx4 ah[2][PG], am[2][PG], al[2][PG], wl[2];
        auto read_step = [&](int ks, int buf) {
#pragma unroll
          for (int i = 0; i < PG; ++i) {
            const unsigned char* p = s_grp + tile_off(i) + kvar(ks) + kconst(ks);
            ah[buf][i] = *reinterpret_cast<const uintx4*>(p);
            am[buf][i] = *reinterpret_cast<const uintx4*>(p + PLANE);
            al[buf][i] = *reinterpret_cast<const uintx4*>(p + 2 * PLANE);
          }
          wl[buf] = s_wl[ks * 64 + lane];
        };
        read_step(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int b = ks & 1;
          if (ks + 1 < KS) read_step(ks + 1, b ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 whv = __builtin_bit_cast(bf16x8, wh[ks]);
          const bf16x8 wmv = __builtin_bit_cast(bf16x8, wm[ks]);
          const bf16x8 wlv = __builtin_bit_cast(bf16x8, wl[b]);
          // small terms first, then the large ones
#pragma unroll
          for (int i = 0; i < PG; ++i) acc[i] = DTF_MFMA_BF16(wlv, __builtin_bit_cast(bf16x8, ah[b][i]), acc[i]);
#pragma unroll
          for (int i = 0; i < PG; ++i) acc[i] = DTF_MFMA_BF16(whv, __builtin_bit_cast(bf16x8, al[b][i]), acc[i]);
#pragma unroll
          for (int i = 0; i < PG; ++i) acc[i] = DTF_MFMA_BF16(wmv, __builtin_bit_cast(bf16x8, am[b][i]), acc[i]);
#pragma unroll
          for (int i = 0; i < PG; ++i) acc[i] = DTF_MFMA_BF16(wmv, __builtin_bit_cast(bf16x8, ah[b][i]), acc[i]);
#pragma unroll
          for (int i = 0; i < PG; ++i) acc[i] = DTF_MFMA_BF16(whv, __builtin_bit_cast(bf16x8, am[b][i]), acc[i]);
#pragma unroll
          for (int i = 0; i < PG; ++i) acc[i] = DTF_MFMA_BF16(whv, __builtin_bit_cast(bf16x8, ah[b][i]), acc[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: ReLU mask and x-hat from the raw x, statistics, 16-byte stores
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
      __syncthreads();
      set ^= 1;
    }
  } else {
    // ================= filter gradient: M = input channels, N = output channels, K = pixels of the band =====
    // transposed-read lane addresses: lane 4 j + i of a group points at pixel 4 g + j of the step, channels 4 i .. 4 i + 3
    const int wp0 = CO_T == 1 ? 0 : rw * CPW, wx = CO_T == 1 ? rw : 0;
    const int tr_j = (lane >> 2) & 3, tr_c = lane & 3;
    const int lane_xa = (4 * q + tr_j) * XPB + tr_c * 8;
    const int lane_db = (TW == 8 ? (q >> 1) * ROWB + (4 * (q & 1) + tr_j) * PSB : (4 * q + tr_j) * PSB) + wp0 * 32 + tr_c * 8;
    floatx4 accw[CPW][9];
#pragma unroll
    for (int cp = 0; cp < CPW; ++cp)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) accw[cp][tap] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();   // set 0 is staged
    int set = 0;
    for (int t = t0; t < total_tiles; t += gpz) {
      const unsigned char* s_dy = smem_b + set * SET;
      const unsigned char* s_x = s_dy + 3 * PLANE;
      constexpr int MYSTEPS = G::WSTEPS / WX;
      // A unit is one (step, tap, channel tile): six MFMAs into one accumulator. Two units run interleaved
      // (two accumulators alternate, see the data gradient); the reads of the next pair are in flight meanwhile.
      constexpr int UPS = 9 * CPW;                 // units per step
      constexpr int WUNITS = MYSTEPS * UPS;
      constexpr int WPAIRS = (WUNITS + 1) / 2;
      uintx2 xa[2][3][2];     // A: relu(bn(x)) parts of the current / next step
      uintx2 db[2][2][3][2];  // B: dy parts of the current / next pair of units
      // the band's sum in accumulators of its own, added to the running total once (conv_bwd_split.hip)
      floatx4 accb[CPW][9];
#pragma unroll
      for (int cp = 0; cp < CPW; ++cp)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) accb[cp][tap] = floatx4{0.f, 0.f, 0.f, 0.f};
      auto read_a = [&](int ms) {
        const unsigned char* p = s_x + lane_xa + (wx + ms * WX) * 32 * XPB;
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          xa[ms & 1][part][0] = lds_read_tr(p + part * XPLANE);
          xa[ms & 1][part][1] = lds_read_tr(p + part * XPLANE + 16 * XPB);
        }
      };
      auto read_b = [&](int u, int buf, int slot) {
        const int cp = u % CPW, tap = (u / CPW) % 9, ms = u / UPS;
        const int kh = tap / 3, kw = tap % 3;
        // dy[pixel - (tap - 1)]: halo position (row + 2 - kh, col + 2 - kw); a step is 32 / TW whole rows
        constexpr int SROWS = 32 / TW > 0 ? 32 / TW : 1;
        const unsigned char* p0 = s_dy + lane_db + cp * 32 + (wx + ms * WX) * (SROWS * ROWB) + (2 - kh) * ROWB + (2 - kw) * PSB;
        const unsigned char* p1 = p0 + (TW == 32 ? 16 * PSB : (16 / TW) * ROWB);
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          db[buf][slot][part][0] = lds_read_tr(p0 + part * PLANE);
          db[buf][slot][part][1] = lds_read_tr(p1 + part * PLANE);
        }
      };
      auto read_pair = [&](int pr, int buf) {
        read_b(2 * pr, buf, 0);
        if (2 * pr + 1 < WUNITS) read_b(2 * pr + 1, buf, 1);
        // the first unit of a step brings the step's A fragments along
        if ((2 * pr) % UPS == 0) read_a((2 * pr) / UPS);
        else if ((2 * pr + 1) % UPS == 0 && 2 * pr + 1 < WUNITS) read_a((2 * pr + 1) / UPS);
      };
      read_pair(0, 0);
#pragma unroll
      for (int pr = 0; pr < WPAIRS; ++pr) {
        const int b = pr & 1;
        if (pr + 1 < WPAIRS) read_pair(pr + 1, b ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const int u0 = 2 * pr, u1 = 2 * pr + 1;
        const bool two = u1 < WUNITS;
        const int cp0 = u0 % CPW, tap0 = (u0 / CPW) % 9, ms0 = u0 / UPS;
        const int cp1 = u1 % CPW, tap1 = (u1 / CPW) % 9, ms1 = two ? u1 / UPS : ms0;
        bf16x8 x0[3], x1[3], d0[3], d1v[3];
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          x0[part] = frag8(xa[ms0 & 1][part][0], xa[ms0 & 1][part][1]);
          x1[part] = frag8(xa[ms1 & 1][part][0], xa[ms1 & 1][part][1]);
          d0[part] = frag8(db[b][0][part][0], db[b][0][part][1]);
          d1v[part] = frag8(db[b][1][part][0], db[b][1][part][1]);
        }
        // (x part, dy part) of the six partial products, small terms first
        constexpr int XP[6] = {2, 0, 1, 1, 0, 0}, DP[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          accb[cp0][tap0] = DTF_MFMA_BF16(x0[XP[k]], d0[DP[k]], accb[cp0][tap0]);
          if (two) accb[cp1][tap1] = DTF_MFMA_BF16(x1[XP[k]], d1v[DP[k]], accb[cp1][tap1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int cp = 0; cp < CPW; ++cp)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) accw[cp][tap] += accb[cp][tap];
      __syncthreads();
      set ^= 1;
    }
    // accumulators -> LDS; the planes are free (barrier at the end of the last band)
#pragma unroll
    for (int cp = 0; cp < CPW; ++cp)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s_red[((wx * 9 + tap) * 16 + 4 * q + r) * RS + (wp0 + cp) * 16 + l15] = accw[cp][tap][r];
  }

  // ---- statistics (the filter's l part in LDS is no longer needed: its space parks the wave rows) --------
  if (a.gsums) {
    floatx4 s1v[1] = {st1}, s2v[1] = {st2};   // zeros from all but the data-gradient waves
    block_stat_flush<1, 8>(reinterpret_cast<float*>(s_wl), s1v, s2v, a.gsums, C, c0, kStatGridBwd);
  }
  // ---- filter gradient flush: sum the pixel-split partials, coalesced stores / atomics ---------------------
  __syncthreads();
  // with dw_part: plain stores into copy jblk of the whole gradient (this block's rows of it)
  float* dw = a.dw_part ? a.dw_part + (size_t)jblk * (9 * C * C) : a.dw + (size_t)(blockIdx.x & (kStatSlots - 1)) * a.dw_slot_stride;
  for (int e = tid; e < 9 * 16 * C; e += 512) {
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
template <int C, int TW, int TH, bool DYBN, int RES = 0>
size_t launch_bpc(const ConvBwdFusedArgs& a, hipStream_t stream, int* copies) {
  using G = BPC<C, TW, TH>;
  const size_t shmem = (size_t)G::LDS_BYTES;
  constexpr int zsplit = C / 16;
  const int total_tiles = (a.H / TH) * a.B;
  static const bool ready = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_bwd_split_pc_kernel<C, TW, TH, DYBN, RES>), shmem);
    return true;
  }();
  (void)ready;
  int gpz_max = 256 / zsplit;   // one workgroup per CU
  if (gpz_max < 8) gpz_max = 8;
  const int iters = ceil_div(total_tiles, gpz_max);
  int gpz = ceil_div(total_tiles, iters);
  gpz = (gpz + 7) / 8 * 8;
  if (copies) {
    *copies = gpz;
    return 0;
  }
  launch("conv_bwd_split_pc", conv_bwd_split_pc_kernel<C, TW, TH, DYBN, RES>, dim3(gpz * zsplit), dim3(512), shmem, stream, a);
  return shmem;
}

}  // namespace

bool conv_bwd_split_pc_supported(int B, int H, int W, int C) {
  (void)B;
  return H == W && H % 8 == 0 && ((C == 16 && W == 32) || (C == 32 && W == 16) || (C == 64 && W == 8));
}

size_t conv_bwd_split_pc(const ConvBwdFusedArgs& a, hipStream_t stream, int* copies) {
  if (!conv_bwd_split_pc_supported(a.B, a.H, a.W, a.C)) throw std::runtime_error("conv_bwd_split_pc: unsupported shape");
  if (a.dy_x) {
    if (a.dy_res) {
      if (a.dy_res_mode == 1) {
        if (a.C == 16) return launch_bpc<16, 32, 8, true, 2>(a, stream, copies);
        if (a.C == 32) return launch_bpc<32, 16, 8, true, 2>(a, stream, copies);
        throw std::runtime_error("conv_bwd_split: no option-A shortcut behind a 64-channel stage");
      }
      if (a.C == 16) return launch_bpc<16, 32, 8, true, 1>(a, stream, copies);
      if (a.C == 32) return launch_bpc<32, 16, 8, true, 1>(a, stream, copies);
      return launch_bpc<64, 8, 8, true, 1>(a, stream, copies);
    }
    if (a.C == 16) return launch_bpc<16, 32, 8, true>(a, stream, copies);
    if (a.C == 32) return launch_bpc<32, 16, 8, true>(a, stream, copies);
    return launch_bpc<64, 8, 8, true>(a, stream, copies);
  }
  if (a.C == 16) return launch_bpc<16, 32, 8, false>(a, stream, copies);
  if (a.C == 32) return launch_bpc<32, 16, 8, false>(a, stream, copies);
  return launch_bpc<64, 8, 8, false>(a, stream, copies);
}

}  // namespace dtf
