// Forward-shaped 3x3 convolution of the ResNet layers with compile-time band geometry.
//
// Same GEMM mapping, K permutation, swapped-operand epilogue and fused prologue / epilogues as
// conv_fprop.hip, and persistent like conv_fprop_persist.hip, but the output tile is fixed at
// compile time to a TH-row band of the full output width TW (a power of two):
//
//   * every thread computes the global and LDS offsets of its share of the input halo ONCE; per
//     tile the staging code is one address add and one row-bounds test per 16-byte chunk (the
//     runtime-geometry kernels spend ~1000 VALU instructions per wave on div/mod, which at two to
//     three waves per SIMD is as long as the whole MFMA phase);
//   * LDS tap offsets are instruction immediates, output pixels of a band are contiguous in
//     memory and always inside the image, so the epilogue has no index arithmetic or bounds tests;
//   * at block start the loads are issued in dependency order (BatchNorm statistics, filter
//     slice, first tile) before anything waits, so the start-up costs one memory round trip.
//
// UPS instances read the input as if it were zero-upsampled by two (data gradient of a
// stride-2 layer); BWD instances apply the ReLU mask and accumulate BatchNorm-backward's sums.
// (reference ops: conv2d / conv2d_backprop_input, DistributedResNet/src/resnet.py:99,121,150)
#include "conv.h"
#include "lds_banking.h"

#include <cstdlib>

namespace dtf {
namespace {

template <int S, int CP, int NT, int TW, int TH, bool UPS>
struct FBand {
  static constexpr int CG = CP / 16;
  static constexpr int KG = 9 * CG;
  static constexpr int C4 = CP / 4;
  static constexpr int LH = (TH - 1) * S + 3;
  static constexpr int LW = (TW - 1) * S + 3;
  // per-pixel stride of the staged tile: the padding (4 or 8 floats) that keeps the 16-lane groups
  // of the A-operand ds_read_b128 off each other's banks (lds_banking.h)
  static constexpr int CPP = CP + lds_best_pad(CP, TW, LW, S);
  // stride 1: the halo tile's first and last column lie outside the image for every band; they are
  // zeroed once and only the TW image columns are staged (8-row bands: exactly 5 chunks per thread)
  static constexpr bool INTERIOR = (S == 1) && !UPS;
  static constexpr int IN_CHUNKS = LH * (INTERIOR ? TW : LW) * C4;
  static constexpr int NPF = (IN_CHUNKS + 255) / 256;
  static constexpr int W_CHUNKS = KG * NT * 64;            // float4s of the filter slice
  static constexpr int NW = (W_CHUNKS + 255) / 256;
  static constexpr int N_MT = TH * TW / 16;
  // (four m-tiles per wave for the 16-channel instance - one filter read per 16 MFMAs instead of
  // per 8 - measured no different: 54.8 vs 55.1 us at batch 1024)
  static constexpr int MTW = (N_MT >= 8 && NT <= 2) ? 2 : 1;
  static constexpr int ITER = N_MT / (4 * MTW);          // m-tile groups per wave and tile
  static_assert(N_MT % (4 * MTW) == 0, "every wave gets the same number of m-tile groups");
  static constexpr int IN_FLOATS = (LH * LW * CPP + 3) & ~3;
  static constexpr int W_FLOATS = KG * NT * 256;
  static constexpr int LDS_FLOATS = IN_FLOATS + W_FLOATS + 2 * CP + 7 * NT * 16;
  static_assert((TH * TW) % 16 == 0, "a band must be whole 16-pixel MFMA tiles");
  static_assert(!UPS || (S == 1 && TH % 2 == 0), "zero-upsampled input implies stride 1 and even bands");
};

template <int S, int CP, int NT, int TW, int TH, bool UPS, bool BWD>
__global__ __launch_bounds__(256, 2) void conv_fprop_band_kernel(ConvFpropArgs a) {
  DTF_ABORT_GUARD(a);
  using G = FBand<S, CP, NT, TW, TH, UPS>;
  constexpr int CPP = G::CPP, CG = G::CG, KG = G::KG, C4 = G::C4, LW = G::LW, MTW = G::MTW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = s_in + G::IN_FLOATS;
  float* s_bn = s_w + G::W_FLOATS;       // in-BN scale[CP], shift[CP]
  float* s_stat = s_bn + 2 * CP;         // [2][NT*16]
  float* s_ebn = s_stat + 2 * NT * 16;   // [5][NT*16]: scale, shift, mean, rstd, bias

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  const int l15 = lane & 15;

  const int zsplit = a.nt_total / NT;
  const int gpz = gridDim.x / zsplit;   // blocks that share one cout split
  const int bid = blockIdx.x;
  const int z = (bid >> 3) % zsplit;
  const int jblk = (bid / (8 * zsplit)) * 8 + (bid & 7);
  const int bands = a.Ho / TH;
  const int total_tiles = bands * a.B;

  // ---- 1. statistics loads first: their consumers wait for nothing issued after them ----------
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

  // ---- 2. filter slice into registers ----------------------------------------------------------
  floatx4 wreg[G::NW];
  {
    const floatx4* wsrc = reinterpret_cast<const floatx4*>(a.wpk);
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
      const int i = tid + k * 256;
      const int g = i / (NT * 64);
      const int rem = i % (NT * 64);
      wreg[k] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (i < G::W_CHUNKS) wreg[k] = wsrc[(g * a.nt_total + z * NT + (rem >> 6)) * 64 + (rem & 63)];
    }
  }

  // ---- 3. per-thread halo geometry (fixed for the kernel) and the first tile --------------------
  int in_goff[G::NPF], in_loff[G::NPF], in_row[G::NPF];
#pragma unroll
  for (int k = 0; k < G::NPF; ++k) {
    const int i = tid + k * 256;
    const int c4 = i % C4;
    const int r = G::INTERIOR ? (i / C4) / TW : (i / C4) / LW;
    const int c = G::INTERIOR ? (i / C4) % TW + 1 : (i / C4) % LW;   // pad = 1 for the stride-1 shapes
    const int pix = r * LW + c;
    int rr = r, ix = c - a.pad;
    bool ok = i < G::IN_CHUNKS;
    if (UPS) {
      const int vr = r - a.pad;
      ok = ok && ((vr & 1) == 0) && ((ix & 1) == 0);
      rr = vr >> 1;
      ix >>= 1;
    }
    ok = ok && (ix >= 0) && (ix < a.W);
    in_goff[k] = (rr * a.W + ix) * CP + c4 * 4;
    in_loff[k] = pix * CPP + c4 * 4;
    in_row[k] = ok ? rr : -100000;
  }
  // lane-constant parts of the LDS pixel address and of the output offset (pixel l15 of an m-tile)
  const int lane_a = (((l15 / TW) * S) * LW + (l15 % TW) * S) * CPP + 4 * q;
  const int lane_o = l15 * a.Cout + z * NT * 16 + 4 * q;
  const bool want_res = !BWD && ((a.res_mode == RES_IDENT) || (a.res_mode == RES_BNRELU));

  // Epilogue operands (residual / mask source / extra gradient) of a tile are requested together
  // with its input halo, one whole tile ahead of their use.
  constexpr int ITER = G::ITER;
  floatx4 pre_rv[ITER][MTW][NT], pre_xv[ITER][MTW][NT];
  floatx4 pf[G::NPF];
  unsigned okbits = 0;
  auto load_tile = [&](int t) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const int base_row = UPS ? (oy0 >> 1) : oy0 * S - a.pad;
    {
      const long long obase = ((long long)(n * a.Ho + oy0) * TW) * a.Cout;
#pragma unroll
      for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const long long oo = obase + (long long)((wave * MTW + it * 4 * MTW + i) * 16) * a.Cout + lane_o;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            pre_rv[it][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            pre_xv[it][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (!(DTF_DEBUG_SKIP(a) & 4)) {
              if (want_res) pre_rv[it][i][j] = *reinterpret_cast<const floatx4*>(a.res + oo + j * 16);
              if (BWD) {
                pre_xv[it][i][j] = *reinterpret_cast<const floatx4*>(a.mask_src + oo + j * 16);
                if (a.extra) pre_rv[it][i][j] = *reinterpret_cast<const floatx4*>(a.extra + oo + j * 16);
              }
            }
          }
        }
    }
    const float* in0 = a.in + ((long long)(n * a.H + base_row) * a.W) * CP;
    okbits = 0;
#pragma unroll
    for (int k = 0; k < G::NPF; ++k) {
      const int iy = base_row + in_row[k];
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < a.H && !(DTF_DEBUG_SKIP(a) & 2)) {
        v = *reinterpret_cast<const floatx4*>(in0 + in_goff[k]);
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
          // chunk tid + 256 k carries channels 4 ((tid + 256 k) % C4) ..: the same four for every k
          static_assert(256 % C4 == 0, "a thread's chunks share their channel group");
          const int ch = (tid % C4) * 4;
          const floatx4 sc = *reinterpret_cast<const floatx4*>(s_bn + ch);
          const floatx4 sh = *reinterpret_cast<const floatx4*>(s_bn + CP + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        }
        *reinterpret_cast<floatx4*>(s_in + in_loff[k]) = v;
      }
    }
  };

  int t = jblk;
  if (t < total_tiles) load_tile(t);

  // ---- 4. coefficients and filter slice into LDS --------------------------------------------------
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
    if (tid + k * 256 < G::W_CHUNKS) reinterpret_cast<floatx4*>(s_w)[tid + k * 256] = wreg[k];
  if (G::INTERIOR) {
    for (int i = tid; i < G::LH * 2 * CPP; i += 256) {
      const int r = i / (2 * CPP), side = (i / CPP) & 1, ch = i % CPP;
      s_in[(r * LW + (side ? LW - 1 : 0)) * CPP + ch] = 0.f;
    }
  }
  __syncthreads();

  float e_scale[NT][4], e_shift[NT][4], e_mean[NT][4], e_rstd[NT][4], e_bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = j * 16 + 4 * q + r;
      e_scale[j][r] = s_ebn[c];
      e_shift[j][r] = s_ebn[NT * 16 + c];
      e_mean[j][r] = s_ebn[2 * NT * 16 + c];
      e_rstd[j][r] = s_ebn[3 * NT * 16 + c];
      e_bias[j][r] = s_ebn[4 * NT * 16 + c];
    }
  }
  floatx4 st1[NT], st2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { st1[j] = floatx4{0.f, 0.f, 0.f, 0.f}; st2[j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

  for (; t < total_tiles; t += gpz) {
    const int n = t / bands;
    const int oy0 = (t - n * bands) * TH;
    const long long obase = ((long long)(n * a.Ho + oy0) * TW) * a.Cout;

    write_tile();
    floatx4 cur_rv[ITER][MTW][NT], cur_xv[ITER][MTW][NT];
#pragma unroll
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          cur_rv[it][i][j] = pre_rv[it][i][j];
          cur_xv[it][i][j] = pre_xv[it][i][j];
        }
    __syncthreads();
    if (t + gpz < total_tiles) load_tile(t + gpz);   // in flight during the MFMA phase

#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int mb = wave * MTW + it * 4 * MTW;
      floatx4 acc[MTW][NT];
      floatx4 rv[MTW][NT], xv[MTW][NT];
      int aoff[MTW];
      long long ooff[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int p0 = (mb + i) * 16;   // first pixel of the m-tile (wave-uniform)
        aoff[i] = lane_a + ((((p0 / TW) * S) * LW + (p0 % TW) * S) * CPP);
        ooff[i] = obase + (long long)p0 * a.Cout + lane_o;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
          rv[i][j] = cur_rv[it][i][j];
          xv[i][j] = cur_xv[it][i][j];
        }
      }

      if (!(DTF_DEBUG_SKIP(a) & 1)) {
        // Operand fragments are double buffered in registers: the LDS reads of k-group g+1 are in
        // flight while the MFMAs of k-group g execute; the scheduling barriers pin that order.
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
              for (int j = 0; j < NT; ++j) acc[i][j] = DTF_MFMA16(bv[g & 1][j][s], av[g & 1][i][s], acc[i][j]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }

#pragma unroll
      for (int i = 0; i < MTW; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          floatx4 v = acc[i][j];
          if (!BWD) {
            if (a.res_mode == RES_IDENT) {
              v += rv[i][j];
            } else if (a.res_mode == RES_BNRELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += fmaxf(rv[i][j][r] * e_scale[j][r] + e_shift[j][r], 0.f);
            } else if (a.res_mode == RES_SHORTCUT_A) {
              // option-A shortcut: 2x2 average pool of the block input, zero-padded in channels
              const int co = (z * NT + j) * 16 + 4 * q;
              const int cres = a.Cout >> 1, cq = a.Cout >> 2;
              if (co >= cq && co < cq + cres) {
                const int p = (mb + i) * 16 + l15;
                const int oy = oy0 + p / TW, ox = p % TW;
                const float* rp = a.res + ((long long)(n * a.Ho * 2 + oy * 2) * (2 * TW) + ox * 2) * cres + (co - cq);
                const floatx4 r0 = *reinterpret_cast<const floatx4*>(rp);
                const floatx4 r1 = *reinterpret_cast<const floatx4*>(rp + cres);
                const floatx4 r2 = *reinterpret_cast<const floatx4*>(rp + (long long)(2 * TW) * cres);
                const floatx4 r3 = *reinterpret_cast<const floatx4*>(rp + (long long)(2 * TW) * cres + cres);
                v += (r0 + r1 + r2 + r3) * 0.25f;
              }
            }
            if (a.epi_mode == EPI_BIAS_RELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] + e_bias[j][r], 0.f);
            } else if (a.epi_mode == EPI_BIAS) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += e_bias[j][r];
            }
            st1[j] += v;
            st2[j] += v * v;
          } else {
            const floatx4 x = xv[i][j];
            if (a.extra) v += rv[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = (x[r] * e_scale[j][r] + e_shift[j][r] > 0.f) ? v[r] : 0.f;
              st1[j][r] += v[r];
              st2[j][r] += v[r] * ((x[r] - e_mean[j][r]) * e_rstd[j][r]);
            }
          }
          if (!(DTF_DEBUG_SKIP(a) & 4)) *reinterpret_cast<floatx4*>(a.out + ooff[i] + j * 16) = v;
        }
      }
    }
    __syncthreads();   // all waves are done reading s_in before the next tile overwrites it
  }

  if (a.out_sums && !(DTF_DEBUG_SKIP(a) & 8)) {
    block_stat_flush<NT, 4>(smem, st1, st2, a.out_sums, a.Cout, z * NT * 16, BWD ? kStatGridBwd : kStatGridFwd);
  }
}

template <int S, int CP, int NT, int TW, int TH, bool UPS, bool BWD>
size_t launch_fband(const ConvFpropArgs& a, hipStream_t stream) {
  using G = FBand<S, CP, NT, TW, TH, UPS>;
  const size_t shmem = (size_t)G::LDS_FLOATS * sizeof(float);
  const int zsplit = a.nt_total / NT;
  const int total_tiles = (a.Ho / TH) * a.B;
  // co-resident blocks per CU as the runtime sees them (LDS and the registers the instance really uses)
  static const int bpc = [&] {
    const char* e = getenv("DTF_CONV_BPC");
    if (e && atoi(e) > 0) return atoi(e);
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_fprop_band_kernel<S, CP, NT, TW, TH, UPS, BWD>), shmem);
    int n = 0;
    DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_fprop_band_kernel<S, CP, NT, TW, TH, UPS, BWD>, 256, shmem));
    // the 16- and 64-channel instances would fit three workgroups per CU; a third one measures
    // slower (ResNet-20 step at batch 1024: 2.83 ms with two everywhere, 2.99 ms with three)
    return n < 1 ? 1 : (n > 2 ? 2 : n);
  }();
  int gpz_max = (256 * bpc) / zsplit;
  if (gpz_max < 8) gpz_max = 8;
  // even work: iterations = ceil(tiles / gpz_max), then the smallest multiple-of-8 grid that needs no more
  const int iters = ceil_div(total_tiles, gpz_max);
  int gpz = ceil_div(total_tiles, iters);
  gpz = (gpz + 7) / 8 * 8;
  launch("conv_fprop_band", conv_fprop_band_kernel<S, CP, NT, TW, TH, UPS, BWD>, dim3(gpz * zsplit), dim3(256), shmem,
         stream, a);
  return shmem;
}

}  // namespace

// Returns 0 when the shape is not covered (the caller falls back to the runtime-geometry kernels).
size_t conv_fprop_band(const ConvFpropArgs& a_in, int ks, hipStream_t stream) {
  if (ks != 3) return 0;
  ConvFpropArgs a = a_in;
  if ((a.Cin & 15) || (a.Cout & 15) || a.Ho != a.Wo || a.H != a.W) return 0;
  const bool bwd = a.epi_mode == EPI_MASK;
  if (bwd && (a.in_mode != IN_RAW || a.res_mode != RES_NONE)) return 0;
  const bool ups = a.upsample != 0;
  const int s = ups ? 1 : a.stride;
  if (ups ? (a.Ho != 2 * a.H || a.pad != 2) : (a.H != a.Ho * s || a.pad != (s == 1 ? 1 : 0))) return 0;
  a.nt_total = a.Cout / 16;

#define DTF_FBAND(S_, CIN_, COUT_, TW_, TH_, NT_, UPS_, BWD_)                                               \
  if (s == S_ && a.Cin == CIN_ && a.Cout == COUT_ && a.Wo == TW_ && ups == UPS_ && bwd == BWD_ && a.Ho % TH_ == 0) \
    return launch_fband<S_, CIN_, NT_, TW_, TH_, UPS_, BWD_>(a, stream);
  // forward
  DTF_FBAND(1, 16, 16, 32, 8, 1, false, false)
  DTF_FBAND(2, 16, 32, 16, 8, 2, false, false)
  DTF_FBAND(1, 32, 32, 16, 8, 2, false, false)
  DTF_FBAND(2, 32, 64, 8, 8, 2, false, false)
  DTF_FBAND(1, 64, 64, 8, 8, 1, false, false)
  // data gradient (filters packed flipped/transposed, so Cin/Cout are those of the gradient conv)
  DTF_FBAND(1, 16, 16, 32, 8, 1, false, true)
  DTF_FBAND(1, 32, 16, 32, 8, 1, true, true)
  DTF_FBAND(1, 32, 32, 16, 8, 2, false, true)
  DTF_FBAND(1, 64, 32, 16, 4, 1, true, true)
  DTF_FBAND(1, 64, 64, 8, 8, 1, false, true)
#undef DTF_FBAND
  return 0;
}

}  // namespace dtf
