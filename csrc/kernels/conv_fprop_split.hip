// Stride-1 3x3 convolution of the 16- and 32-channel ResNet layers on the bf16 matrix pipe with
// fp32 operands split three ways ("bf16x6"), fp32 accumulation. Opt-in; the exact-fp32 kernel of the
// same shapes is conv_fprop_band.hip and this file mirrors its interface, prologue and epilogues.
//
// Why: v_mfma_f32_16x16x4_f32 runs at the vector fp32 rate (64 FLOP/clk/SIMD); v_mfma_f32_16x16x32_bf16
// does 16 times the FLOPs per clock. An fp32 value x is cut into three bf16 values, each rounded to
// nearest: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), so |m| <= 2^-9 |x|, |l| <= 2^-18 |x| and
// x - (h + m + l) <= 2^-27 |x|; a product of two such values into the six partial products whose weight
// is at least 2^-18 of the whole:
//     x y  ~=  h h' + (h m' + m h') + (h l' + l h' + m m')          (dropped: m l', l m', l l' <= 2^-26 |x y|)
// Every partial product of two bf16 values is exact in fp32 and the MFMA accumulates in fp32, so the
// result differs from the fp32 kernel's by the dropped terms and by the order of the additions only.
// Six bf16 MFMAs of K = 32 replace eight fp32 MFMAs of K = 4: 96 instead of 256 matrix-pipe cycles.
//
// What it costs: every operand is three values, so the LDS traffic per FLOP is four times the fp32
// kernel's. The filter fragments of a wave (its 16 output channels, all taps: h and m parts) therefore
// live in REGISTERS for the whole kernel - a wave keeps one output-channel tile and walks four pixel
// tiles - and only the small l part and the activations are read from LDS: three ds_read_b128 per six
// MFMAs, half of the LDS read bandwidth with every SIMD busy. Activations are split while they are
// staged (11 vector instructions per pair of values on top of BatchNorm + ReLU).
//   K order: 32 channels: one tap per K = 32 step. 16 channels: two taps per step (lane groups 0-1 read
//   tap 2s, groups 2-3 tap 2s + 1); the tenth, non-existent tap has zero filter values. 64 channels:
//   half a tap per step.
//   64 channels: a wave's filter fragments (18 steps x 3 parts) fit neither its registers nor - next to
//   the activation planes of two workgroups - the LDS; the filter (221 KB for all four output-channel
//   tiles) stays in L2 and every wave streams its fragments from there straight into MFMA operand
//   registers, three K steps ahead of their use (54 KB per wave and image).
// (reference ops: conv2d / conv2d_backprop_input, DistributedResNet/src/resnet.py:99,121,150)
#include "conv.h"
#include "split_common.h"

#include <cstdlib>

namespace dtf {
namespace {

template <int CP, int TW, int TH>
struct FSplit {
  static constexpr int NT = CP / 16;                 // output-channel tiles (Cout == Cin == CP)
  static constexpr int KS = split_ksteps(CP);        // K = 32 steps: 5 (16 channels, tap 9 is padding), 9, 18
  static constexpr bool STREAM = CP == 64;           // filter fragments streamed from L2 instead of held in registers
  static constexpr int C4 = CP / 4;
  static constexpr int LH = TH + 2, LW = TW + 2;
  using PL = SplitPlanes<CP, TW, TH>;
  static constexpr int SLOTS = PL::SLOTS, PSB = PL::PSB, ROWB = PL::ROWB, PLANE = PL::PLANE;
  static constexpr int IN_CHUNKS = LH * TW * C4;     // only the image columns are staged
  static constexpr int NPF = (IN_CHUNKS + 255) / 256;
  static constexpr int N_MT = TH * TW / 16;
  static constexpr int MG = 4 / NT;                  // groups of pixel tiles over the waves
  static constexpr int P = N_MT / MG;                // pixel tiles per wave
  static constexpr int WL_UNITS = STREAM ? 0 : KS * NT * 64;   // 16-byte units of the filter's l part
  static constexpr int NWL = (WL_UNITS + 255) / 256;
  static constexpr int LDS_BYTES = 3 * PLANE + WL_UNITS * 16 + (2 * CP + 7 * NT * 16) * 4;
  static_assert(NT == 1 || NT == 2 || NT == 4, "16, 32 or 64 channels");
  static_assert(N_MT % MG == 0 && P % 2 == 0 && (TW % 16 == 0 || 16 % TW == 0), "pixel tiles are whole rows or parts of one");
  static_assert(256 % C4 == 0, "a thread's chunks share their channel group");
};

template <int CP, int TW, int TH, bool BWD>
__global__ __launch_bounds__(256, 2) void conv_fprop_split_kernel(ConvFpropArgs a) {
  DTF_ABORT_GUARD(a);
  using G = FSplit<CP, TW, TH>;
  constexpr int NT = G::NT, KS = G::KS, C4 = G::C4, LW = G::LW, PSB = G::PSB, ROWB = G::ROWB, PLANE = G::PLANE, P = G::P;
  constexpr bool STREAM = G::STREAM;
  constexpr int WDEPTH = 2;          // K steps a streamed filter fragment is requested ahead of its use
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* s_act = smem_b;                                        // [3][LH][ROWB], pixels PSB apart
  uintx4* s_wl = reinterpret_cast<uintx4*>(smem_b + 3 * PLANE);         // [KS][NT][64]
  float* s_bn = reinterpret_cast<float*>(smem_b + 3 * PLANE + G::WL_UNITS * 16);   // in-BN scale[CP], shift[CP]
  float* s_stat = s_bn + 2 * CP;                                        // [2][NT*16] (block_stat_flush scratch lives in s_act)
  float* s_ebn = s_stat + 2 * NT * 16;                                  // [5][NT*16]: scale, shift, mean, rstd, bias

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  const int l15 = lane & 15;
  const int ct = wave % NT;          // this wave's output-channel tile
  const int mg = wave / NT;          // and group of pixel tiles

  const int gpz = gridDim.x;
  const int jblk = blockIdx.x;
  const int bands = a.Ho / TH;
  const int total_tiles = bands * a.B;

  // ---- 1. statistics loads first ------------------------------------------------------------------
  const bool in_bn_thread = (a.in_mode == IN_BNRELU) && tid < CP;
  double ib_s1 = 0.0, ib_s2 = 0.0;
  float ib_g = 0.f, ib_b = 0.f;
  if (in_bn_thread) {
    ib_s1 = stat_total(a.in_bn.sums, tid, CP);
    ib_s2 = stat_total(a.in_bn.sums, CP + tid, CP);
    ib_g = a.in_bn.gamma[tid];
    ib_b = a.in_bn.beta[tid];
  }
  const BnRef& ebn = BWD ? a.mask_bn : a.res_bn;
  const bool e_bn_thread = (BWD || a.res_mode == RES_BNRELU) && tid < NT * 16;
  double eb_s1 = 0.0, eb_s2 = 0.0;
  float eb_g = 0.f, eb_b = 0.f;
  if (e_bn_thread) {
    eb_s1 = stat_total(ebn.sums, tid, CP);
    eb_s2 = stat_total(ebn.sums, CP + tid, CP);
    eb_g = ebn.gamma[tid];
    eb_b = ebn.beta[tid];
  }

  // ---- 2. filter fragments: h and m parts into registers, l part into LDS ---------------------------
  const uintx4* wsrc = reinterpret_cast<const uintx4*>(a.wpk);   // [KS][NT][3][64 lanes] x 16 bytes
  uintx4 wh[STREAM ? 1 : KS], wm[STREAM ? 1 : KS];
  uintx4 wl_reg[STREAM ? 1 : G::NWL];
  if (!STREAM) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wh[ks] = wsrc[((ks * NT + ct) * 3 + 0) * 64 + lane];
      wm[ks] = wsrc[((ks * NT + ct) * 3 + 1) * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < G::NWL; ++k) {
      const int i = tid + k * 256;
      wl_reg[k] = uintx4{0u, 0u, 0u, 0u};
      if (i < G::WL_UNITS) wl_reg[k] = wsrc[((i >> 6) * 3 + 2) * 64 + (i & 63)];
    }
  }
  // streamed: fragment (ks, part) of this lane = uniform base + (ks * NT * 3 + part) KB + w_voff bytes: a scalar
  // base and one 32-bit lane offset, so the 54 addresses of a tile cost no vector registers
  const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(a.wpk);
  const unsigned w_voff = (unsigned)((ct * 3 * 64 + lane) * 16);
  auto wfrag = [&](int ks, int part) -> uintx4 {
    return *reinterpret_cast<const uintx4*>(wbytes + (size_t)((ks * NT * 3 + part) * 1024) + w_voff);
  };

  // ---- 3. per-thread halo geometry and the first tile -------------------------------------------------
  // chunk i = tid + 256 k of the tile: 256 chunks are RPK whole rows, so a thread's chunks keep their column
  // and channel group and sit RPK rows apart - three registers instead of three per chunk
  constexpr int RPK = 256 / (C4 * TW);
  static_assert(256 % (C4 * TW) == 0, "256 chunks are whole rows");
  const int in_r0 = (tid / C4) / TW;
  const int in_goff0 = (in_r0 * a.W + (tid / C4) % TW) * CP + (tid % C4) * 4;
  const int in_loff0 = in_r0 * ROWB + ((tid / C4) % TW + 1) * PSB + (tid % C4) * 8;
  const int in_gstep = RPK * a.W * CP;
  // B-operand byte offset inside a plane = pixel-tile base (per lane, poff below) + channel slot of the
  // lane group + the K step's tap shift. The tap shift is a compile-time constant (an instruction
  // immediate) except with 16 channels, where lane groups 2-3 read the tap after that of groups 0-1:
  // one pixel further (d1), two pixels short of a row further (dx, steps whose second tap starts a
  // filter row), or the same tap again (the padding step).
  const int q_slot = 16 * (CP == 16 ? (q & 1) : q);
  const int d1 = (CP == 16 && q >= 2) ? PSB : 0;
  const int dx = (CP == 16 && q >= 2) ? ROWB - 2 * PSB : 0;
  auto kconst = [](int ks) constexpr -> int {
    const int tap = CP == 64 ? ks >> 1 : (CP == 32 ? ks : 2 * ks);
    return (tap / 3) * ROWB + (tap % 3) * PSB + (CP == 64 ? 64 * (ks & 1) : 0);
  };
  auto kvar = [&](int ks) -> int { return CP != 16 ? 0 : (ks == 1 ? dx : (ks == 4 ? 0 : d1)); };
  const int lane_o = l15 * CP + ct * 16 + 4 * q;
  const bool want_res = !BWD && ((a.res_mode == RES_IDENT) || (a.res_mode == RES_BNRELU));

  floatx4 pf[G::NPF];
  unsigned okbits = 0;
  auto load_tile = [&](int t) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const int base_row = oy0 - 1;
    const float* in0 = a.in + ((long long)(n * a.H + base_row) * a.W) * CP;
    okbits = 0;
#pragma unroll
    for (int k = 0; k < G::NPF; ++k) {
      const int iy = base_row + in_r0 + RPK * k;
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < a.H && tid + k * 256 < G::IN_CHUNKS) {
        v = *reinterpret_cast<const floatx4*>(in0 + in_goff0 + k * in_gstep);
        okbits |= 1u << k;
      }
      pf[k] = v;
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int k = 0; k < G::NPF; ++k) {
      if (tid + k * 256 < G::IN_CHUNKS) {
        floatx4 v = pf[k];
        if (!BWD && a.in_mode == IN_BNRELU && ((okbits >> k) & 1u)) {
          const int ch = (tid % C4) * 4;
          const floatx4 sc = *reinterpret_cast<const floatx4*>(s_bn + ch);
          const floatx4 sh = *reinterpret_cast<const floatx4*>(s_bn + CP + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        }
        uintx2 h, m, l;
        split3(v, h, m, l);
        unsigned char* dst = s_act + in_loff0 + k * (RPK * ROWB);
        *reinterpret_cast<uintx2*>(dst) = h;
        *reinterpret_cast<uintx2*>(dst + PLANE) = m;
        *reinterpret_cast<uintx2*>(dst + 2 * PLANE) = l;
      }
    }
  };

  int t = jblk;
  if (t < total_tiles) load_tile(t);

  // ---- 4. coefficients, the filter's l part and the zero halo columns into LDS -----------------------
  if (tid < CP) {
    float sc = 0.f, sh = 0.f;
    if (in_bn_thread) {
      const double mean = ib_s1 * (double)a.in_bn.inv_count;
      double var = ib_s2 * (double)a.in_bn.inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = rsqrtf((float)var + a.in_bn.eps);
      sc = ib_g * rstd;
      sh = ib_b - (float)mean * sc;
    }
    s_bn[tid] = sc;
    s_bn[CP + tid] = sh;
  }
  if (tid < NT * 16) {
    float sc = 1.f, sh = 0.f, m = 0.f, r = 1.f;
    if (e_bn_thread) {
      const double mean = eb_s1 * (double)ebn.inv_count;
      double var = eb_s2 * (double)ebn.inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      r = rsqrtf((float)var + ebn.eps);
      m = (float)mean;
      sc = eb_g * r;
      sh = eb_b - m * sc;
    }
    s_ebn[tid] = sc;
    s_ebn[NT * 16 + tid] = sh;
    s_ebn[2 * NT * 16 + tid] = m;
    s_ebn[3 * NT * 16 + tid] = r;
  }
  if (!STREAM) {
#pragma unroll
    for (int k = 0; k < G::NWL; ++k)
      if (tid + k * 256 < G::WL_UNITS) s_wl[tid + k * 256] = wl_reg[k];
  }
  for (int i = tid; i < 3 * G::LH * 2 * (PSB / 4); i += 256) {
    const int w = i % (PSB / 4), side = (i / (PSB / 4)) & 1, r = (i / (2 * (PSB / 4))) % G::LH, plane = i / (2 * (PSB / 4) * G::LH);
    reinterpret_cast<unsigned*>(s_act + plane * PLANE + r * ROWB + (side ? LW - 1 : 0) * PSB)[w] = 0u;
  }
  __syncthreads();

  const float* e_lane = s_ebn + ct * 16 + 4 * q;   // this lane's four output channels in each coefficient row
  floatx4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};   // statistics of this wave's output-channel tile

  // pixel-tile base offsets (bytes) of this wave inside a plane
  int poff[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int p = (mg * P + i) * 16 + l15;
    poff[i] = (p / TW) * ROWB + (p % TW) * PSB + q_slot;
  }

  for (; t < total_tiles; t += gpz) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const long long obase = ((long long)(n * a.Ho + oy0) * TW) * CP;

    write_tile();
    __syncthreads();
    // streamed filter fragments: the first WDEPTH steps are requested here, the rest from inside the MFMA loop
    uintx4 wq[STREAM ? WDEPTH : 1][3];
    if (STREAM) {
#pragma unroll
      for (int d = 0; d < WDEPTH; ++d)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) wq[d][sp] = wfrag(d, sp);
    }
    // epilogue operands of THIS tile (residual / mask source / extra gradient): used after the MFMA phase
    // (the filter fragments already take the registers a tile-ahead prefetch would need). They are
    // requested BEFORE the next tile: vector memory operations retire in order, so waiting for a load
    // issued behind the prefetch would wait for the whole prefetch - a memory round trip per tile.
    floatx4 cur_rv[P], cur_xv[BWD ? P : 1];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const long long oo = obase + (long long)((mg * P + i) * 16) * CP + lane_o;
      cur_rv[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (want_res) cur_rv[i] = *reinterpret_cast<const floatx4*>(a.res + oo);
      if (BWD) {
        cur_xv[i] = *reinterpret_cast<const floatx4*>(a.mask_src + oo);
        if (a.extra) cur_rv[i] = *reinterpret_cast<const floatx4*>(a.extra + oo);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!STREAM && t + gpz < total_tiles) load_tile(t + gpz);   // in flight during the MFMA phase

    floatx4 acc[P];
#pragma unroll
    for (int i = 0; i < P; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};

    // Walk (K step, pair of pixel tiles); the operand reads of the next pair are in flight while the
    // twelve MFMAs of this pair run.
    constexpr int PAIRS = P / 2;
    constexpr int STEPS = KS * PAIRS;
    uintx4 ah[2][2], am[2][2], al[2][2], wl[2];
    auto read_step = [&](int s, int buf) {
      const int ks = s / PAIRS, pr = s % PAIRS;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* p = s_act + poff[2 * pr + i] + kvar(ks) + kconst(ks);
        ah[buf][i] = *reinterpret_cast<const uintx4*>(p);
        am[buf][i] = *reinterpret_cast<const uintx4*>(p + PLANE);
        al[buf][i] = *reinterpret_cast<const uintx4*>(p + 2 * PLANE);
      }
      if (!STREAM) wl[buf] = s_wl[(ks * NT + ct) * 64 + lane];
    };
    read_step(0, 0);
#pragma clang loop unroll(full)
    for (int s = 0; s < STEPS; ++s) {
      const int ks = s / PAIRS, pr = s % PAIRS, b = s & 1;
      if (s + 1 < STEPS) read_step(s + 1, b ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 whv = __builtin_bit_cast(bf16x8, STREAM ? wq[ks % WDEPTH][0] : wh[STREAM ? 0 : ks]);
      const bf16x8 wmv = __builtin_bit_cast(bf16x8, STREAM ? wq[ks % WDEPTH][1] : wm[STREAM ? 0 : ks]);
      const bf16x8 wlv = __builtin_bit_cast(bf16x8, STREAM ? wq[ks % WDEPTH][2] : wl[b]);
      // small terms first, then the large ones; the two pixel tiles alternate
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 * pr + i] = DTF_MFMA_BF16(wlv, __builtin_bit_cast(bf16x8, ah[b][i]), acc[2 * pr + i]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 * pr + i] = DTF_MFMA_BF16(whv, __builtin_bit_cast(bf16x8, al[b][i]), acc[2 * pr + i]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 * pr + i] = DTF_MFMA_BF16(wmv, __builtin_bit_cast(bf16x8, am[b][i]), acc[2 * pr + i]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 * pr + i] = DTF_MFMA_BF16(wmv, __builtin_bit_cast(bf16x8, ah[b][i]), acc[2 * pr + i]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 * pr + i] = DTF_MFMA_BF16(whv, __builtin_bit_cast(bf16x8, am[b][i]), acc[2 * pr + i]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 * pr + i] = DTF_MFMA_BF16(whv, __builtin_bit_cast(bf16x8, ah[b][i]), acc[2 * pr + i]);
      __builtin_amdgcn_sched_barrier(0);
      if (STREAM && pr == PAIRS - 1 && ks + WDEPTH < KS) {   // this step's ring slot is free: request step ks + WDEPTH
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) wq[ks % WDEPTH][sp] = wfrag(ks + WDEPTH, sp);
      }
      // streamed fragments: the next tile is requested behind the last MFMA step - ahead of the ring's requests,
      // every wait for a fragment would also wait for the tile's trip to HBM (vector memory operations retire
      // in order), and its registers would be live through the whole MFMA phase (64 channels: 8x8x64 forward
      // 40.9 -> 31.0 us at batch 1024, 60 spilled registers -> 5, with the ring two deep instead of three)
      if (STREAM && s == STEPS - 1) {
        if (t + gpz < total_tiles) load_tile(t + gpz);
      }
    }

    // epilogue coefficients: read here, behind the MFMA phase, instead of living in registers through it
    __builtin_amdgcn_sched_barrier(0);
    const floatx4 e_scale = *reinterpret_cast<const floatx4*>(e_lane);
    const floatx4 e_shift = *reinterpret_cast<const floatx4*>(e_lane + NT * 16);
    floatx4 e_mean = {0.f, 0.f, 0.f, 0.f}, e_rstd = {0.f, 0.f, 0.f, 0.f};
    if (BWD) {
      e_mean = *reinterpret_cast<const floatx4*>(e_lane + 2 * NT * 16);
      e_rstd = *reinterpret_cast<const floatx4*>(e_lane + 3 * NT * 16);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int p0 = (mg * P + i) * 16;
      floatx4 v = acc[i];
      if (!BWD) {
        if (a.res_mode == RES_IDENT) {
          v += cur_rv[i];
        } else if (a.res_mode == RES_BNRELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += fmaxf(cur_rv[i][r] * e_scale[r] + e_shift[r], 0.f);
        } else if (a.res_mode == RES_SHORTCUT_A) {
          // option-A shortcut: 2x2 average pool of the block input, zero-padded in channels
          const int co = ct * 16 + 4 * q;
          constexpr int cres = CP >> 1, cq = CP >> 2;
          if (co >= cq && co < cq + cres) {
            const int p = p0 + l15;
            const int oy = oy0 + p / TW, ox = p % TW;
            const float* rp = a.res + ((long long)(n * a.Ho * 2 + oy * 2) * (2 * TW) + ox * 2) * cres + (co - cq);
            const floatx4 r0 = *reinterpret_cast<const floatx4*>(rp);
            const floatx4 r1 = *reinterpret_cast<const floatx4*>(rp + cres);
            const floatx4 r2 = *reinterpret_cast<const floatx4*>(rp + (long long)(2 * TW) * cres);
            const floatx4 r3 = *reinterpret_cast<const floatx4*>(rp + (long long)(2 * TW) * cres + cres);
            v += (r0 + r1 + r2 + r3) * 0.25f;
          }
        }
        s1 += v;
        s2 += v * v;
      } else {
        const floatx4 x = cur_xv[i];
        if (a.extra) v += cur_rv[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = (x[r] * e_scale[r] + e_shift[r] > 0.f) ? v[r] : 0.f;
          s1[r] += v[r];
          s2[r] += v[r] * ((x[r] - e_mean[r]) * e_rstd[r]);
        }
      }
      *reinterpret_cast<floatx4*>(a.out + obase + (long long)p0 * CP + lane_o) = v;
    }
    __syncthreads();   // all waves are done reading the planes before the next tile overwrites them
  }

  if (a.out_sums) {
    floatx4 st1[NT], st2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      st1[j] = j == ct ? s1 : floatx4{0.f, 0.f, 0.f, 0.f};
      st2[j] = j == ct ? s2 : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    block_stat_flush<NT, 4>(reinterpret_cast<float*>(smem_b), st1, st2, a.out_sums, CP, 0, BWD ? kStatGridBwd : kStatGridFwd);
  }
}

template <int CP, int TW, int TH, bool BWD>
size_t launch_fsplit(const ConvFpropArgs& a, hipStream_t stream) {
  using G = FSplit<CP, TW, TH>;
  const size_t shmem = (size_t)G::LDS_BYTES;
  const int total_tiles = (a.Ho / TH) * a.B;
  static const int bpc = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_fprop_split_kernel<CP, TW, TH, BWD>), shmem);
    int n = 0;
    DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_fprop_split_kernel<CP, TW, TH, BWD>, 256, shmem));
    return n < 1 ? 1 : (n > 2 ? 2 : n);
  }();
  int gpz_max = 256 * bpc;
  const int iters = ceil_div(total_tiles, gpz_max);
  int gpz = ceil_div(total_tiles, iters);
  gpz = (gpz + 7) / 8 * 8;
  launch("conv_fprop_split", conv_fprop_split_kernel<CP, TW, TH, BWD>, dim3(gpz), dim3(256), shmem, stream, a);
  return shmem;
}

// ---- filter packing: HWIO fp32 -> [K step][cout tile][h, m, l][lane][8 bf16] (split_pack.h) ---------------
__global__ __launch_bounds__(256) void pack_split_kernel(PackArgs a) {
  const int total = (int)split_packed_bf16(a.Cin);
  unsigned short* out = reinterpret_cast<unsigned short*>(a.out);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
    out[i] = split_packed_element(a.w, a.Cin, a.transpose_flip, i);
}

}  // namespace

size_t packed_split_floats(int C) { return (size_t)(split_packed_bf16(C) / 2); }

void pack_split_weights(const PackArgs& a, hipStream_t stream) {
  if (a.ks != 3 || a.Cin != a.Cout || (a.Cin != 16 && a.Cin != 32 && a.Cin != 64)) throw std::runtime_error("pack_split_weights: 3x3, C -> C channels, C in 16, 32, 64");
  launch("pack_split_weights", pack_split_kernel, dim3(64), dim3(256), 0, stream, a);
}

// Returns 0 when the shape is not covered.
size_t conv_fprop_split(const ConvFpropArgs& a, hipStream_t stream) {
  if (a.Cin != a.Cout || a.stride != 1 || a.upsample || a.pad != 1 || a.H != a.Ho || a.W != a.Wo || a.H != a.W) return 0;
  const bool bwd = a.epi_mode == EPI_MASK;
  if (bwd ? (a.in_mode != IN_RAW || a.res_mode != RES_NONE) : (a.epi_mode != EPI_PLAIN)) return 0;
  if (a.res_mode == RES_SHORTCUT_A && a.Cin == 16) return 0;   // the network has no such layer
  if (a.Cin == 16 && a.W == 32 && a.H % 8 == 0) return bwd ? launch_fsplit<16, 32, 8, true>(a, stream) : launch_fsplit<16, 32, 8, false>(a, stream);
  if (a.Cin == 32 && a.W == 16 && a.H % 8 == 0) return bwd ? launch_fsplit<32, 16, 8, true>(a, stream) : launch_fsplit<32, 16, 8, false>(a, stream);
  if (a.Cin == 64 && a.W == 8 && a.H == 8) return bwd ? launch_fsplit<64, 8, 8, true>(a, stream) : launch_fsplit<64, 8, 8, false>(a, stream);
  return 0;
}

}  // namespace dtf
