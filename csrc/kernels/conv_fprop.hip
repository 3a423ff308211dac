// Implicit-GEMM convolution on the exact-fp32 matrix cores of gfx950.
//
// GEMM view: M = output pixels of one spatial tile, N = output channels, K = taps x input
// channels. One 256-thread block owns a TH x TW output tile of one image and NT 16-channel
// output tiles. It stages
//   * the input halo tile into LDS, applying the producer's BatchNorm + ReLU on the way in
//     (pre-activation ResNet order, reference: DistributedResNet/src/resnet.py:117-121), and
//   * its slice of the pre-packed filter, already in MFMA B-operand order,
// then every wave walks its 16-pixel M-tiles issuing v_mfma_f32_16x16x4_f32. The K order is
// permuted so that one lane's operands for four consecutive k-steps are four consecutive
// channels of one pixel: a single ds_read_b128 feeds four MFMAs for A and for B alike.
// LDS pixels are padded by 4 floats so those 16-byte reads stay at most 2-way bank conflicted.
// The epilogue fuses the residual add (identity / BN+ReLU'd / option-A shortcut), bias+ReLU,
// the ReLU-mask of the backward pass, and the per-channel statistics the next BatchNorm
// (or this BatchNorm's backward) needs, accumulated with fp64 atomics.
#include "conv.h"

#include <cstdlib>

namespace dtf {

namespace {

template <int KS, int CP>
struct KGroups {
  static constexpr int value = (CP >= 16) ? KS * KS * (CP / 16) : (KS * KS + 3) / 4;
};

__device__ __forceinline__ int round_up4(int v) { return (v + 3) & ~3; }

template <int KS, int S, int CP, int NT, int MTW>
__global__ __launch_bounds__(256) void conv_fprop_kernel(ConvFpropArgs a) {
  DTF_ABORT_GUARD(a);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CPP = CP + 4;
  constexpr int KG = KGroups<KS, CP>::value;
  constexpr int CG = (CP >= 16) ? CP / 16 : 1;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  const int l15 = lane & 15;

  // Block decode. Blocks b and b+8 tend to share an XCD (and so an L2): keep the cout-split
  // siblings of one spatial tile 8 apart so they re-read the same input tile from one L2.
  const int zsplit = a.nt_total / NT;
  const int tiles_img = a.tiles_y * a.tiles_x;
  const int total_tiles = tiles_img * a.B;
  const int bid = blockIdx.x;
  const int t = (bid / (8 * zsplit)) * 8 + (bid & 7);
  const int z = (bid >> 3) % zsplit;
  if (t >= total_tiles) return;
  const int n = t / tiles_img;
  const int tile = t - n * tiles_img;
  const int oy0 = (tile / a.tiles_x) * a.TH;
  const int ox0 = (tile % a.tiles_x) * a.TW;

  const int LH = (a.TH - 1) * S + KS;
  const int LW = (a.TW - 1) * S + KS;

  float* s_in = smem;
  float* s_w = s_in + round_up4(LH * LW * CPP);
  float* s_bn = s_w + KG * NT * 256;      // scale[CP], shift[CP]
  float* s_stat = s_bn + 2 * CP;          // [2][NT*16]

  // ---- prologue: BatchNorm coefficients of the input --------------------------------
  if (a.in_mode == IN_BNRELU) {
    if (tid < CP) {
      float sc = 0.f, sh = 0.f, m, r;
      if (tid < a.Cin) bn_coeffs(a.in_bn, tid, a.Cin, sc, sh, m, r);
      s_bn[tid] = sc;
      s_bn[CP + tid] = sh;
    }
  }

  // ---- stage the filter slice (already in B-operand order) --------------------------
  {
    const floatx4* wsrc = reinterpret_cast<const floatx4*>(a.wpk);
    floatx4* wdst = reinterpret_cast<floatx4*>(s_w);
    for (int i = tid; i < KG * NT * 64; i += 256) {
      const int g = i / (NT * 64);
      const int rem = i - g * (NT * 64);
      const int j = rem >> 6;
      const int l = rem & 63;
      wdst[i] = wsrc[(g * a.nt_total + z * NT + j) * 64 + l];
    }
  }
  __syncthreads();

  // ---- stage the input halo tile ----------------------------------------------------
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
      int iy = vy0 + r, ix = vx0 + c;
      bool ok = (iy >= 0) && (ix >= 0);
      if (a.upsample) {
        ok = ok && ((iy & 1) == 0) && ((ix & 1) == 0);
        iy >>= 1;
        ix >>= 1;
      }
      ok = ok && (iy < a.H) && (ix < a.W);
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
  __syncthreads();

  // ---- per-lane epilogue coefficients: the MFMA operands are swapped (filter as A, pixels as
  // B), so a lane's four accumulator registers are output channels 4q..4q+3 of ONE pixel (l15):
  // the epilogue moves 16 contiguous bytes per lane and 16 consecutive pixels per instruction.
  float e_scale[NT][4], e_shift[NT][4], e_mean[NT][4], e_rstd[NT][4], e_bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = (z * NT + j) * 16 + 4 * q + r;
      e_scale[j][r] = 1.f; e_shift[j][r] = 0.f; e_mean[j][r] = 0.f; e_rstd[j][r] = 1.f; e_bias[j][r] = 0.f;
      if (co < a.Cout) {
        if (a.res_mode == RES_BNRELU)
          bn_coeffs(a.res_bn, co, a.Cout, e_scale[j][r], e_shift[j][r], e_mean[j][r], e_rstd[j][r]);
        if (a.epi_mode == EPI_MASK)
          bn_coeffs(a.mask_bn, co, a.Cout, e_scale[j][r], e_shift[j][r], e_mean[j][r], e_rstd[j][r]);
        if (a.epi_mode == EPI_BIAS_RELU || a.epi_mode == EPI_BIAS) e_bias[j][r] = a.bias[co];
      }
    }
  }
  floatx4 st1[NT], st2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { st1[j] = floatx4{0.f, 0.f, 0.f, 0.f}; st2[j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

  // ---- main loop over this wave's M-tiles ------------------------------------------
  const int tile_px = a.TH * a.TW;
  const int n_mt = (tile_px + 15) >> 4;
  for (int mb = wave * MTW; mb < n_mt; mb += 4 * MTW) {
    floatx4 acc[MTW][NT];
    int aoff[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      int p = (mb + i) * 16 + l15;
      if (p >= tile_px) p = 0;
      const int ty = p / a.TW;
      const int tx = p - ty * a.TW;
      aoff[i] = ((ty * S) * LW + tx * S) * CPP;
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    }

    if constexpr (CP >= 16) {
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int kh = tap / KS, kw = tap % KS;
        const int toff = (kh * LW + kw) * CPP + 4 * q;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
          const int g = tap * CG + cg;
          floatx4 av[MTW], bv[NT];
#pragma unroll
          for (int i = 0; i < MTW; ++i)
            av[i] = *reinterpret_cast<const floatx4*>(s_in + aoff[i] + toff + cg * 16);
#pragma unroll
          for (int j = 0; j < NT; ++j)
            bv[j] = *reinterpret_cast<const floatx4*>(s_w + (g * NT + j) * 256 + lane * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[i][j] = DTF_MFMA16(bv[j][s], av[i][s], acc[i][j]);
        }
      }
    } else {
      // CP == 4: a k-group is four taps (lane quarter q) x four channels (k-step s).
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        int tap = 4 * g + q;
        if (tap >= KS * KS) tap = 0;  // packed weights are zero there
        const int kh = tap / KS, kw = tap - kh * KS;
        const int toff = (kh * LW + kw) * CPP;
        floatx4 av[MTW], bv[NT];
#pragma unroll
        for (int i = 0; i < MTW; ++i) av[i] = *reinterpret_cast<const floatx4*>(s_in + aoff[i] + toff);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bv[j] = *reinterpret_cast<const floatx4*>(s_w + (g * NT + j) * 256 + lane * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = DTF_MFMA16(bv[j][s], av[i][s], acc[i][j]);
      }
    }

    // ---- epilogue: lane holds channels 4q..4q+3 (per N-tile) of pixel l15 of each M-tile
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int p = (mb + i) * 16 + l15;
      const int ty = p / a.TW;
      const int tx = p - ty * a.TW;
      const int oy = oy0 + ty, ox = ox0 + tx;
      const bool pok = (p < tile_px) && (oy < a.Ho) && (ox < a.Wo);
      const size_t pbase = ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = (z * NT + j) * 16 + 4 * q;
        if (!pok || co >= a.Cout) continue;
        floatx4 v = acc[i][j];
        const size_t idx = pbase + co;
        if (a.res_mode == RES_IDENT) {
          v += *reinterpret_cast<const floatx4*>(a.res + idx);
        } else if (a.res_mode == RES_BNRELU) {
          const floatx4 rv = *reinterpret_cast<const floatx4*>(a.res + idx);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += fmaxf(rv[r] * e_scale[j][r] + e_shift[j][r], 0.f);
        } else if (a.res_mode == RES_SHORTCUT_A) {
          const int cres = a.Cout >> 1, cq = a.Cout >> 2;
          if (co >= cq && co < cq + cres) {
            const int RW = a.Wo * 2;
            const float* rp = a.res + ((size_t)(n * a.Ho * 2 + oy * 2) * RW + ox * 2) * cres + (co - cq);
            const floatx4 r0 = *reinterpret_cast<const floatx4*>(rp);
            const floatx4 r1 = *reinterpret_cast<const floatx4*>(rp + cres);
            const floatx4 r2 = *reinterpret_cast<const floatx4*>(rp + (size_t)RW * cres);
            const floatx4 r3 = *reinterpret_cast<const floatx4*>(rp + (size_t)RW * cres + cres);
            v += (r0 + r1 + r2 + r3) * 0.25f;
          }
        }
        if (a.epi_mode == EPI_BIAS_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] + e_bias[j][r], 0.f);
        } else if (a.epi_mode == EPI_BIAS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += e_bias[j][r];
        } else if (a.epi_mode == EPI_MASK) {
          const floatx4 x = *reinterpret_cast<const floatx4*>(a.mask_src + idx);
          if (a.extra) v += *reinterpret_cast<const floatx4*>(a.extra + idx);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = (x[r] * e_scale[j][r] + e_shift[j][r] > 0.f) ? v[r] : 0.f;
            st1[j][r] += v[r];
            st2[j][r] += v[r] * ((x[r] - e_mean[j][r]) * e_rstd[j][r]);
          }
        }
        if (a.epi_mode != EPI_MASK) {
          st1[j] += v;
          st2[j] += v * v;
        }
        *reinterpret_cast<floatx4*>(a.out + idx) = v;
      }
    }
  }

  // ---- statistics: reduce over the 16 pixel lanes, then waves (LDS), then blocks (fp64 atomics)
  if (a.out_sums) {
    block_stat_flush<NT, 4>(smem, st1, st2, a.out_sums, a.Cout, z * NT * 16, a.epi_mode == EPI_MASK ? kStatGridBwd : kStatGridFwd);
  }
}

template <int KS, int S, int CP, int NT, int MTW>
size_t launch_fprop(const ConvFpropArgs& a, hipStream_t stream) {
  constexpr int CPP = CP + 4;
  constexpr int KG = KGroups<KS, CP>::value;
  const int LH = (a.TH - 1) * S + KS, LW = (a.TW - 1) * S + KS;
  const size_t floats = (size_t)((LH * LW * CPP + 3) & ~3) + (size_t)KG * NT * 256 + 2 * CP + 2 * NT * 16;
  const size_t shmem = floats * sizeof(float);
  if (shmem > kLdsBudget) throw std::runtime_error("conv_fprop: tile does not fit LDS");
  const int zsplit = a.nt_total / NT;
  const int total_tiles = a.tiles_y * a.tiles_x * a.B;
  const int t8 = (total_tiles + 7) / 8 * 8;
  launch("conv_fprop", conv_fprop_kernel<KS, S, CP, NT, MTW>, dim3(t8 * zsplit), dim3(256), shmem, stream, a);
  return shmem;
}

int pad_cin(int cin) {
  if (cin <= 4) return 4;
  if (cin <= 16) return 16;
  if (cin <= 32) return 32;
  if (cin <= 64) return 64;
  throw std::runtime_error("conv: more than 64 input channels is not supported");
}

size_t lds_bytes(int ks, int s, int cp, int nt, int th, int tw) {
  const int cpp = cp + 4;
  const int kg = (cp >= 16) ? ks * ks * (cp / 16) : (ks * ks + 3) / 4;
  const int lh = (th - 1) * s + ks, lw = (tw - 1) * s + ks;
  return ((size_t)((lh * lw * cpp + 3) & ~3) + (size_t)kg * nt * 256 + 2 * cp + 2 * nt * 16) * 4;
}

}  // namespace

// Choose the output tile and the number of 16-channel output tiles per block so that two
// blocks fit a CU's 160 KiB LDS when possible, preferring tiles of 128-256 pixels.
void conv_fprop_plan(ConvFpropArgs& a, int ks) {
  const int cp = pad_cin(a.Cin);
  const int s = a.upsample ? 1 : a.stride;
  a.nt_total = ceil_div(a.Cout, 16);
  int tw = a.Wo <= 32 ? a.Wo : 32;
  int th = a.Ho;
  while (th * tw > 256 && th > 1) th = (th + 1) / 2;
  a.TW = tw;
  a.TH = th;
  a.tiles_x = ceil_div(a.Wo, tw);
  a.tiles_y = ceil_div(a.Ho, th);
}

namespace {
int choose_nt(const ConvFpropArgs& a, int ks, int cp, int s) {
  // Largest NT in {4,2,1} dividing nt_total whose LDS footprint allows two blocks per CU;
  // else the largest that fits at all.
  const int cands[3] = {4, 2, 1};
  for (int nt : cands) {
    if (a.nt_total % nt) continue;
    if (lds_bytes(ks, s, cp, nt, a.TH, a.TW) <= 80 * 1024) return nt;
  }
  for (int nt : cands) {
    if (a.nt_total % nt) continue;
    if (lds_bytes(ks, s, cp, nt, a.TH, a.TW) <= kLdsBudget) return nt;
  }
  throw std::runtime_error("conv_fprop: no tiling fits LDS");
}
}  // namespace

#define DTF_FPROP_CASE(KS_, S_, CP_)                                                      \
  if (ks == KS_ && s == S_ && cp == CP_) {                                                \
    if (nt == 4) return launch_fprop<KS_, S_, CP_, 4, 1>(a, stream);                      \
    if (nt == 2) return mtw2 ? launch_fprop<KS_, S_, CP_, 2, 2>(a, stream)                \
                             : launch_fprop<KS_, S_, CP_, 2, 1>(a, stream);               \
    return mtw2 ? launch_fprop<KS_, S_, CP_, 1, 2>(a, stream)                             \
                : launch_fprop<KS_, S_, CP_, 1, 1>(a, stream);                            \
  }

size_t conv_fprop_persist(const ConvFpropArgs& a_in, int ks, hipStream_t stream);

static bool persist_enabled() {
  static const bool on = [] {
    const char* e = getenv("DTF_CONV_PERSIST");
    return !(e && e[0] == '0');
  }();
  return on;
}

size_t conv_fprop(const ConvFpropArgs& a_in, int ks, hipStream_t stream) {
  // The 3x3 ResNet layer shapes run on the compile-time band-geometry kernel, other 3x3 shapes on
  // the persistent runtime-geometry kernel; everything else (and callers that pin a tiling) on
  // the one-tile-per-block kernel below. DTF_CONV_BAND=0 / DTF_CONV_PERSIST=0 disable a tier.
  static const bool band_on = [] {
    const char* e = getenv("DTF_CONV_BAND");
    return !(e && e[0] == '0');
  }();
  if (a_in.TH <= 0 && band_on) {
    if (a_in.upsample) {
      if (size_t r = conv_dgrad_s2(a_in, ks, stream)) return r;
    }
    // The 8-wave ping-pong variant measures the same as the band kernel standalone (60.6 vs 64.7,
    // 49.4 vs 49.0, 49.4 vs 49.2 us at batch 1024) and, owning a whole CU, it cannot share the CU
    // with the filter-gradient kernel on the side stream (3.45 vs 3.06 ms/step): opt-in only.
    static const bool pp_on = [] {
      const char* e = getenv("DTF_CONV_PP");
      return e && e[0] == '1';
    }();
    if (pp_on || (a_in.debug_skip & 64)) {
      if (size_t r = conv_fprop_pp(a_in, ks, stream)) return r;
    }
    if (size_t r = conv_fprop_band(a_in, ks, stream)) return r;
  }
  if (a_in.TH <= 0 && persist_enabled()) {
    if (size_t r = conv_fprop_persist(a_in, ks, stream)) return r;
  }
  ConvFpropArgs a = a_in;
  if (a.TH <= 0 || a.TW <= 0) conv_fprop_plan(a, ks);
  const int cp = pad_cin(a.Cin);
  const int s = a.upsample ? 1 : a.stride;
  if (a.nt_total <= 0) a.nt_total = ceil_div(a.Cout, 16);
  if ((a.Cin & 3) != 0 && a.Cin > 4) throw std::runtime_error("conv_fprop: Cin must be <=4 or a multiple of 4");
  if (a.Cout & 3) throw std::runtime_error("conv_fprop: Cout must be a multiple of 4");
  // Shrink the tile until the smallest configuration fits.
  while (lds_bytes(ks, s, cp, 1, a.TH, a.TW) > kLdsBudget && a.TH > 1) {
    a.TH = (a.TH + 1) / 2;
    a.tiles_y = ceil_div(a.Ho, a.TH);
  }
  // Prefer a tiling that leaves room for two blocks per CU.
  while (lds_bytes(ks, s, cp, 1, a.TH, a.TW) > 80 * 1024 && a.TH * a.TW > 64 && a.TH > 1) {
    a.TH = (a.TH + 1) / 2;
    a.tiles_y = ceil_div(a.Ho, a.TH);
  }
  const int nt = choose_nt(a, ks, cp, s);
  const int n_mt = ceil_div(a.TH * a.TW, 16);
  const bool mtw2 = n_mt >= 8;
  DTF_FPROP_CASE(3, 1, 4)
  DTF_FPROP_CASE(3, 1, 16)
  DTF_FPROP_CASE(3, 2, 16)
  DTF_FPROP_CASE(3, 1, 32)
  DTF_FPROP_CASE(3, 2, 32)
  DTF_FPROP_CASE(3, 1, 64)
  DTF_FPROP_CASE(5, 1, 4)
  DTF_FPROP_CASE(5, 1, 32)
  DTF_FPROP_CASE(5, 1, 64)
  throw std::runtime_error("conv_fprop: unsupported (ks, stride, Cin) combination");
}

// -------------------------------------------------------------------------------------
// Filter packing

namespace {
__global__ void pack_kernel(PackArgs a) {
  const int ks = a.ks;
  // Effective (fprop-shaped) filter dims.
  const int cin_e = a.transpose_flip ? a.Cout : a.Cin;
  const int cout_e = a.transpose_flip ? a.Cin : a.Cout;
  int cp = cin_e <= 4 ? 4 : (cin_e <= 16 ? 16 : (cin_e <= 32 ? 32 : 64));
  const int cg_n = cp >= 16 ? cp / 16 : 1;
  const int kg_n = cp >= 16 ? ks * ks * cg_n : (ks * ks + 3) / 4;
  const int nt_n = (cout_e + 15) / 16;
  const int total = kg_n * nt_n * 256;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int s = i & 3;
    const int l = (i >> 2) & 63;
    const int j = (i >> 8) % nt_n;
    const int g = (i >> 8) / nt_n;
    const int q = l >> 4, nn = l & 15;
    int tap, ci;
    if (cp >= 16) {
      tap = g / cg_n;
      ci = (g % cg_n) * 16 + 4 * q + s;
    } else {
      tap = 4 * g + q;
      ci = s;
    }
    const int co = j * 16 + nn;
    float v = 0.f;
    if (tap < ks * ks && ci < cin_e && co < cout_e) {
      const int kh = tap / ks, kw = tap % ks;
      if (a.transpose_flip) {
        // W'[kh][kw][ci_e][co_e] = W[ks-1-kh][ks-1-kw][co_e][ci_e]
        v = a.w[(((ks - 1 - kh) * ks + (ks - 1 - kw)) * a.Cin + co) * a.Cout + ci];
      } else {
        v = a.w[((kh * ks + kw) * a.Cin + ci) * a.Cout + co];
      }
    }
    a.out[i] = v;
  }
}
}  // namespace

size_t packed_weight_floats(int ks, int Cin, int Cout, int transpose_flip) {
  const int cin_e = transpose_flip ? Cout : Cin;
  const int cout_e = transpose_flip ? Cin : Cout;
  const int cp = pad_cin(cin_e);
  const int kg = cp >= 16 ? ks * ks * (cp / 16) : (ks * ks + 3) / 4;
  return (size_t)kg * ceil_div(cout_e, 16) * 256;
}

void pack_conv_weights(const PackArgs& a, hipStream_t stream) {
  const size_t total = packed_weight_floats(a.ks, a.Cin, a.Cout, a.transpose_flip);
  const int blocks = (int)((total + 255) / 256);
  launch("pack_conv_weights", pack_kernel, dim3(blocks > 512 ? 512 : blocks), dim3(256), 0, stream, a);
}

}  // namespace dtf
