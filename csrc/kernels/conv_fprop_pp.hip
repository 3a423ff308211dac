// Stride-1 3x3 convolution (forward and masked data gradient) as an 8-wave ping-pong.
//
// conv_fprop_band.hip keeps the matrix pipes busy only ~70 % of a kernel: the waves of a block
// walk stage -> barrier -> MFMA -> epilogue together, the co-resident blocks started together and
// stay in step, so the staging / epilogue work of a CU (VALU, LDS writes, global stores) and its
// MFMA work take turns instead of overlapping.
//
// Here a block is two groups of four waves (one wave of each group on every SIMD). Each group
// owns its own LDS tile buffer and its own sequence of output tiles, and runs the same
// stage -> MFMA -> epilogue pipeline as the band kernel - but the two groups are held exactly half
// a period apart by the block barriers they share:
//
//     interval 2i   : group 0  MFMA(tile 2i)                     | group 1  epilogue(2i-1), stage(2i+1)
//     interval 2i+1 : group 0  epilogue(2i), stage(2i+2)          | group 1  MFMA(2i+1)
//
// so in every interval each SIMD has one wave issuing MFMAs and one wave doing everything else;
// accumulators stay in registers across the barrier between a group's MFMA and epilogue
// intervals. Geometry, K permutation, operand double buffering, prologue/epilogue fusion and
// statistics are those of conv_fprop_band.hip.
// (reference ops: conv2d / conv2d_backprop_input, DistributedResNet/src/resnet.py:99,121,150)
#include "conv.h"

#include <cstdlib>

namespace dtf {
namespace {

template <int CP, int NT, int TW, int TH>
struct PP {
  static constexpr int CPP = CP + 4;
  static constexpr int CG = CP / 16;
  static constexpr int KG = 9 * CG;
  static constexpr int C4 = CP / 4;
  static constexpr int LH = TH + 2;
  static constexpr int LW = TW + 2;
  static constexpr int IN_CHUNKS = LH * LW * C4;
  static constexpr int NPF = (IN_CHUNKS + 255) / 256;        // per thread of one group
  static constexpr int W_CHUNKS = KG * NT * 64;
  static constexpr int NW = (W_CHUNKS + 511) / 512;          // per thread of the block
  static constexpr int N_MT = TH * TW / 16;
  static constexpr int MTW = (N_MT >= 8 && NT <= 2) ? 2 : 1;
  static constexpr int ITER = N_MT / (4 * MTW);
  static constexpr int IN_FLOATS = (LH * LW * CPP + 3) & ~3;
  static constexpr int W_FLOATS = KG * NT * 256;
  static constexpr int LDS_FLOATS = 2 * IN_FLOATS + W_FLOATS + 2 * CP + 7 * NT * 16;
  static_assert((TH * TW) % 16 == 0 && N_MT % (4 * MTW) == 0, "a band must split evenly over four waves");
};

template <int CP, int NT, int TW, int TH, bool BWD>
__global__ __launch_bounds__(512, 2) void conv_fprop_pp_kernel(ConvFpropArgs a) {
  DTF_ABORT_GUARD(a);
  using G = PP<CP, NT, TW, TH>;
  constexpr int CPP = G::CPP, CG = G::CG, KG = G::KG, C4 = G::C4, LW = G::LW, MTW = G::MTW, ITER = G::ITER;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;          // wave group: 0 or 1
  const int hw = wave & 3;             // wave within its group
  const int htid = tid & 255;          // thread within its group
  const int q = lane >> 4;
  const int l15 = lane & 15;

  float* s_in = smem + half * G::IN_FLOATS;           // this group's tile buffer
  float* s_w = smem + 2 * G::IN_FLOATS;
  float* s_bn = s_w + G::W_FLOATS;
  float* s_stat = s_bn + 2 * CP;
  float* s_ebn = s_stat + 2 * NT * 16;

  const int zsplit = a.nt_total / NT;
  const int gpz = gridDim.x / zsplit;
  const int bid = blockIdx.x;
  const int z = (bid >> 3) % zsplit;
  const int jblk = (bid / (8 * zsplit)) * 8 + (bid & 7);
  const int bands = a.Ho / TH;
  const int total_tiles = bands * a.B;
  // tiles of this block: jblk, jblk + gpz, ...; group h takes the ones with index = h (mod 2)
  const int n_blk = jblk < total_tiles ? (total_tiles - jblk + gpz - 1) / gpz : 0;

  // ---- statistics first (see conv_fprop_band.hip for the ordering argument) -----------------------
  const bool in_bn_thread = !BWD && (a.in_mode == IN_BNRELU) && tid < CP;
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
  const int e_co = z * NT * 16 + tid;
  double eb_s1 = 0.0, eb_s2 = 0.0;
  float eb_g = 0.f, eb_b = 0.f;
  if (e_bn_thread) {
    eb_s1 = stat_total(ebn.sums, e_co, a.Cout);
    eb_s2 = stat_total(ebn.sums, a.Cout + e_co, a.Cout);
    eb_g = ebn.gamma[e_co];
    eb_b = ebn.beta[e_co];
  }
  float e_bias_v = 0.f;
  if (!BWD && (a.epi_mode == EPI_BIAS_RELU || a.epi_mode == EPI_BIAS) && tid < NT * 16) e_bias_v = a.bias[e_co];

  floatx4 wreg[G::NW];
  {
    const floatx4* wsrc = reinterpret_cast<const floatx4*>(a.wpk);
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
      const int i = tid + k * 512;
      const int g = i / (NT * 64);
      const int rem = i % (NT * 64);
      wreg[k] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (i < G::W_CHUNKS) wreg[k] = wsrc[(g * a.nt_total + z * NT + (rem >> 6)) * 64 + (rem & 63)];
    }
  }

  // ---- per-thread halo geometry of the group's staging work ------------------------------------
  // in_lr packs the LDS offset (low 16 bits) and the halo row + 1 (high bits; 0 = never valid)
  int in_goff[G::NPF], in_lr[G::NPF];
#pragma unroll
  for (int k = 0; k < G::NPF; ++k) {
    const int i = htid + k * 256;
    const int pix = i / C4, c4 = i % C4;
    const int r = pix / LW, c = pix % LW;
    const int ix = c - 1;
    const bool ok = (i < G::IN_CHUNKS) && (ix >= 0) && (ix < a.W);
    in_goff[k] = (r * a.W + ix) * CP + c4 * 4;
    in_lr[k] = (pix * CPP + c4 * 4) | ((ok ? r + 1 : 0) << 16);
  }
  static_assert(G::IN_FLOATS < 65536, "LDS offsets must fit 16 bits");
  const int lane_a = ((l15 / TW) * LW + (l15 % TW)) * CPP + 4 * q;
  const int lane_o = l15 * a.Cout + z * NT * 16 + 4 * q;
  const bool want_res = !BWD && ((a.res_mode == RES_IDENT) || (a.res_mode == RES_BNRELU));

  floatx4 pre_rv[ITER][MTW][NT], pre_xv[ITER][MTW][NT];
  floatx4 pf[G::NPF];
  unsigned okbits = 0;
  auto tile_origin = [&](int k, int& n, int& oy0) {
    const int t = jblk + k * gpz;
    n = t / bands;
    oy0 = (t - n * bands) * TH;
  };
  auto load_tile = [&](int k) {
    int n, oy0;
    tile_origin(k, n, oy0);
    const int base_row = oy0 - 1;
    {
      const long long obase = ((long long)(n * a.Ho + oy0) * TW) * a.Cout;
#pragma unroll
      for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const long long oo = obase + (long long)((hw * MTW + it * 4 * MTW + i) * 16) * a.Cout + lane_o;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            pre_rv[it][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            pre_xv[it][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (want_res) pre_rv[it][i][j] = *reinterpret_cast<const floatx4*>(a.res + oo + j * 16);
            if (BWD) pre_xv[it][i][j] = *reinterpret_cast<const floatx4*>(a.mask_src + oo + j * 16);
          }
        }
    }
    const float* in0 = a.in + ((long long)(n * a.H + base_row) * a.W) * CP;
    okbits = 0;
#pragma unroll
    for (int k2 = 0; k2 < G::NPF; ++k2) {
      const int rr = in_lr[k2] >> 16;          // halo row + 1, 0 = padding column / no chunk
      const int iy = base_row + rr - 1;
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (rr > 0 && iy >= 0 && iy < a.H) {
        v = *reinterpret_cast<const floatx4*>(in0 + in_goff[k2]);
        okbits |= 1u << k2;
      }
      pf[k2] = v;
    }
  };
  floatx4 cur_rv[ITER][MTW][NT], cur_xv[ITER][MTW][NT];
  auto stage_tile = [&]() {   // prefetched halo -> this group's LDS buffer; operands -> cur
#pragma unroll
    for (int k = 0; k < G::NPF; ++k) {
      if (htid + k * 256 < G::IN_CHUNKS) {
        floatx4 v = pf[k];
        if (!BWD && a.in_mode == IN_BNRELU && ((okbits >> k) & 1u)) {
          const int ch = (in_lr[k] & 0xffff) % CPP;
          const floatx4 sc = *reinterpret_cast<const floatx4*>(s_bn + ch);
          const floatx4 sh = *reinterpret_cast<const floatx4*>(s_bn + CP + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        }
        *reinterpret_cast<floatx4*>(s_in + (in_lr[k] & 0xffff)) = v;
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          cur_rv[it][i][j] = pre_rv[it][i][j];
          cur_xv[it][i][j] = pre_xv[it][i][j];
        }
  };

  if (half < n_blk) load_tile(half);

  // ---- coefficients and the filter slice into LDS ---------------------------------------------------
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
    s_ebn[4 * NT * 16 + tid] = e_bias_v;
  }
#pragma unroll
  for (int k = 0; k < G::NW; ++k)
    if (tid + k * 512 < G::W_CHUNKS) reinterpret_cast<floatx4*>(s_w)[tid + k * 512] = wreg[k];
  __syncthreads();

  floatx4 st1[NT], st2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { st1[j] = floatx4{0.f, 0.f, 0.f, 0.f}; st2[j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

  floatx4 acc[ITER][MTW][NT];

  auto mfma_tile = [&]() {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      int aoff[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int p0 = (hw * MTW + it * 4 * MTW + i) * 16;
        aoff[i] = lane_a + ((p0 / TW) * LW + (p0 % TW)) * CPP;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[it][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
      }
      floatx4 av[2][MTW], bv[2][NT];
      auto read_group = [&](int g, floatx4 (&ao)[MTW], floatx4 (&bo)[NT]) {
        const int tap = g / CG, cg = g % CG;
        const int toff = ((tap / 3) * LW + (tap % 3)) * CPP + cg * 16;
#pragma unroll
        for (int i = 0; i < MTW; ++i) ao[i] = *reinterpret_cast<const floatx4*>(s_in + aoff[i] + toff);
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
            for (int j = 0; j < NT; ++j) acc[it][i][j] = DTF_MFMA16(bv[g & 1][j][s], av[g & 1][i][s], acc[it][i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  auto epilogue_tile = [&](int k) {
    int n, oy0;
    tile_origin(k, n, oy0);
    const long long obase = ((long long)(n * a.Ho + oy0) * TW) * a.Cout;
#pragma unroll
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int mt = hw * MTW + it * 4 * MTW + i;
        const long long oo = obase + (long long)(mt * 16) * a.Cout + lane_o;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          floatx4 v = acc[it][i][j];
          // per-channel coefficients of this lane's four channels, read from LDS where needed
          const int ec = j * 16 + 4 * q;
          if (!BWD) {
            if (a.res_mode == RES_IDENT) {
              v += cur_rv[it][i][j];
            } else if (a.res_mode == RES_BNRELU) {
              const floatx4 esc = *reinterpret_cast<const floatx4*>(s_ebn + ec);
              const floatx4 esh = *reinterpret_cast<const floatx4*>(s_ebn + NT * 16 + ec);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += fmaxf(cur_rv[it][i][j][r] * esc[r] + esh[r], 0.f);
            } else if (a.res_mode == RES_SHORTCUT_A) {
              const int co = (z * NT + j) * 16 + 4 * q;
              const int cres = a.Cout >> 1, cq = a.Cout >> 2;
              if (co >= cq && co < cq + cres) {
                const int p = mt * 16 + l15;
                const int oy = oy0 + p / TW, ox = p % TW;
                const float* rp = a.res + ((long long)(n * a.Ho * 2 + oy * 2) * (2 * TW) + ox * 2) * cres + (co - cq);
                const floatx4 r0 = *reinterpret_cast<const floatx4*>(rp);
                const floatx4 r1 = *reinterpret_cast<const floatx4*>(rp + cres);
                const floatx4 r2 = *reinterpret_cast<const floatx4*>(rp + (long long)(2 * TW) * cres);
                const floatx4 r3 = *reinterpret_cast<const floatx4*>(rp + (long long)(2 * TW) * cres + cres);
                v += (r0 + r1 + r2 + r3) * 0.25f;
              }
            }
            if (a.epi_mode == EPI_BIAS_RELU || a.epi_mode == EPI_BIAS) {
              const floatx4 eb = *reinterpret_cast<const floatx4*>(s_ebn + 4 * NT * 16 + ec);
              v += eb;
              if (a.epi_mode == EPI_BIAS_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
              }
            }
            st1[j] += v;
            st2[j] += v * v;
          } else {
            const floatx4 x = cur_xv[it][i][j];
            if (a.extra) v += *reinterpret_cast<const floatx4*>(a.extra + oo + j * 16);   // first block only: not prefetched
            const floatx4 esc = *reinterpret_cast<const floatx4*>(s_ebn + ec);
            const floatx4 esh = *reinterpret_cast<const floatx4*>(s_ebn + NT * 16 + ec);
            const floatx4 emu = *reinterpret_cast<const floatx4*>(s_ebn + 2 * NT * 16 + ec);
            const floatx4 ers = *reinterpret_cast<const floatx4*>(s_ebn + 3 * NT * 16 + ec);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = (x[r] * esc[r] + esh[r] > 0.f) ? v[r] : 0.f;
              st1[j][r] += v[r];
              st2[j][r] += v[r] * ((x[r] - emu[r]) * ers[r]);
            }
          }
          *reinterpret_cast<floatx4*>(a.out + oo + j * 16) = v;
        }
      }
  };

  // ---- prologue interval: group 0 stages tile 0 ------------------------------------------------------
  if (half == 0 && n_blk > 0) {
    stage_tile();
    if (2 < n_blk) load_tile(2);
  }
  __syncthreads();

  // ---- interval `it`: the group with it = half (mod 2) computes tile `it`; the other group finishes
  // tile it-1 and stages tile it+1 (whose loads were issued two intervals ago) ------------------------
  for (int it = 0; it <= n_blk; ++it) {
    if ((it & 1) == half) {
      if (it < n_blk) mfma_tile();
    } else {
      if (it >= 1) epilogue_tile(it - 1);
      if (it + 1 < n_blk) {
        stage_tile();
        if (it + 3 < n_blk) load_tile(it + 3);
      }
    }
    __syncthreads();
  }

  if (a.out_sums) {
    block_stat_flush<NT, 8>(smem, st1, st2, a.out_sums, a.Cout, z * NT * 16, BWD ? kStatGridBwd : kStatGridFwd);
  }
}

template <int CP, int NT, int TW, int TH, bool BWD>
size_t launch_pp(const ConvFpropArgs& a, hipStream_t stream) {
  using G = PP<CP, NT, TW, TH>;
  const size_t shmem = (size_t)G::LDS_FLOATS * sizeof(float);
  const int zsplit = a.nt_total / NT;
  const int total_tiles = (a.Ho / TH) * a.B;
  static const int bpc = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_fprop_pp_kernel<CP, NT, TW, TH, BWD>), shmem);
    int n = 0;
    DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_fprop_pp_kernel<CP, NT, TW, TH, BWD>, 512, shmem));
    return n < 1 ? 1 : n;
  }();
  int gpz_max = (256 * bpc) / zsplit;
  if (gpz_max < 8) gpz_max = 8;
  // a block wants an even number of tiles (one per wave group and interval pair)
  const int iters = ceil_div(ceil_div(total_tiles, gpz_max), 2) * 2;
  int gpz = ceil_div(total_tiles, iters);
  gpz = (gpz + 7) / 8 * 8;
  launch("conv_fprop_pp", conv_fprop_pp_kernel<CP, NT, TW, TH, BWD>, dim3(gpz * zsplit), dim3(512), shmem, stream, a);
  return shmem;
}

}  // namespace

// Returns 0 when the shape is not covered.
size_t conv_fprop_pp(const ConvFpropArgs& a_in, int ks, hipStream_t stream) {
  if (ks != 3 || a_in.upsample || a_in.stride != 1 || a_in.pad != 1 || (a_in.debug_skip & ~64)) return 0;
  const bool force = (a_in.debug_skip & 64) != 0;   // tests: use this kernel whatever the tile count
  ConvFpropArgs a = a_in;
  if (a.Cin != a.Cout || a.Ho != a.Wo || a.H != a.W || a.H != a.Ho) return 0;
  const bool bwd = a.epi_mode == EPI_MASK;
  if (bwd && (a.in_mode != IN_RAW || a.res_mode != RES_NONE)) return 0;
  a.nt_total = a.Cout / 16;
  // at least two tiles per block and wave group, otherwise the band kernel's smaller blocks win
  const int tiles = (a.Ho / 8) * a.B;
#define DTF_PP(C_, TW_, NT_)                                                                       \
  if (a.Cin == C_ && a.Wo == TW_ && a.Ho % 8 == 0 && (force || tiles >= 4 * 256 / (C_ / 16 / NT_))) {        \
    if (bwd) return launch_pp<C_, NT_, TW_, 8, true>(a, stream);                                   \
    return launch_pp<C_, NT_, TW_, 8, false>(a, stream);                                           \
  }
  DTF_PP(16, 32, 1)
  DTF_PP(32, 16, 2)
  DTF_PP(64, 8, 1)
#undef DTF_PP
  return 0;
}

}  // namespace dtf
