// Persistent, software-pipelined variant of the implicit-GEMM convolution (3x3 ResNet shapes).
//
// The one-tile-per-block kernel in conv_fprop.hip is latency bound: every block walks
// load -> barrier -> MFMA -> store once, and because the whole grid is resident at the same time
// all blocks sit in the same phase, so HBM and the matrix cores take turns. Here a block is
// persistent: it stages its filter slice and BatchNorm coefficients once, then loops over output
// tiles; while the MFMAs of tile i run, the global loads of tile i+1 are already in flight into
// registers (and the residual / mask operands of tile i were issued before its MFMA loop), so the
// matrix cores only wait for LDS. The grid is sized to the number of co-resident blocks and to
// an even number of tiles per block.
//
// Everything else (GEMM mapping, K permutation, swapped-operand epilogue, fused prologue and
// epilogues, fp64 statistics) is identical to conv_fprop.hip; both kernels share ConvFpropArgs.
#include "conv.h"

namespace dtf {
namespace {

__device__ __forceinline__ int round_up4(int v) { return (v + 3) & ~3; }

template <int KS, int S, int CP, int NT, int MTW, int NPF>
__global__ __launch_bounds__(256, (CP <= 16 && NT == 1) ? 3 : 2) void conv_fprop_persist_kernel(ConvFpropArgs a) {
  DTF_ABORT_GUARD(a);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CPP = CP + 4;
  constexpr int CG = (CP >= 16) ? CP / 16 : 1;
  constexpr int KG = (CP >= 16) ? KS * KS * CG : (KS * KS + 3) / 4;
  constexpr int C4 = CP / 4;

#define DTF_STAMP(k) do { if (a.dbg_stamps && threadIdx.x == 0) a.dbg_stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
  DTF_STAMP(0);
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
  const int tiles_img = a.tiles_y * a.tiles_x;
  const int total_tiles = tiles_img * a.B;

  const int LH = (a.TH - 1) * S + KS;
  const int LW = (a.TW - 1) * S + KS;
  const int n_chunks = LH * LW * C4;

  float* s_in = smem;
  float* s_w = s_in + round_up4(LH * LW * CPP);
  float* s_bn = s_w + KG * NT * 256;
  float* s_stat = s_bn + 2 * CP;
  float* s_ebn = s_stat + 2 * NT * 16;

  // ---- prefetch registers for one input halo tile -------------------------------------------
  floatx4 pf[NPF];
  unsigned okbits = 0;
  const bool vec = (a.Cin & 3) == 0;

  auto load_tile = [&](int t) {
    const int n = t / tiles_img;
    const int tile = t - n * tiles_img;
    const int vy0 = (tile / a.tiles_x) * a.TH * S - a.pad;
    const int vx0 = (tile % a.tiles_x) * a.TW * S - a.pad;
    okbits = 0;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int i = tid + k * 256;
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < n_chunks) {
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
        ok = ok && (iy < a.H) && (ix < a.W) && (c4 * 4 < a.Cin) && !(a.debug_skip & 2);
        if (ok) {
          const float* src = a.in + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + c4 * 4;
          if (vec) {
            v = *reinterpret_cast<const floatx4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c4 * 4 + e < a.Cin) v[e] = src[e];
          }
          okbits |= 1u << k;
        }
      }
      pf[k] = v;
    }
  };

  auto write_tile = [&]() {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int i = tid + k * 256;
      if (i < n_chunks) {
        const int pix = i / C4;
        const int c4 = i - pix * C4;
        floatx4 v = pf[k];
        if (a.in_mode == IN_BNRELU && ((okbits >> k) & 1u)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ch = c4 * 4 + e;
            v[e] = fmaxf(v[e] * s_bn[ch] + s_bn[CP + ch], 0.f);
          }
        }
        *reinterpret_cast<floatx4*>(s_in + pix * CPP + c4 * 4) = v;
      }
    }
  };

  int t = jblk;
  if (t < total_tiles) load_tile(t);

  // ---- one-time staging: BatchNorm coefficients and the filter slice ---------------------------
  if (a.in_mode == IN_BNRELU && tid < CP) {
    float sc = 0.f, sh = 0.f, m, r;
    if (tid < a.Cin) bn_coeffs(a.in_bn, tid, a.Cin, sc, sh, m, r);
    s_bn[tid] = sc;
    s_bn[CP + tid] = sh;
  }
  {
    const floatx4* wsrc = reinterpret_cast<const floatx4*>(a.wpk);
    floatx4* wdst = reinterpret_cast<floatx4*>(s_w);
    for (int i = tid; i < KG * NT * 64; i += 256) {
      const int g = i / (NT * 64);
      const int rem = i - g * (NT * 64);
      wdst[i] = wsrc[(g * a.nt_total + z * NT + (rem >> 6)) * 64 + (rem & 63)];
    }
  }

  // Epilogue BatchNorm coefficients: computed once per block by NT*16 threads, handed to the
  // lanes through LDS (every lane needs the four channels 4q..4q+3 of each of its N tiles).
  if (tid < NT * 16) {
    const int co = z * NT * 16 + tid;
    float sc = 1.f, sh = 0.f, m = 0.f, r = 1.f, bias = 0.f;
    if (co < a.Cout) {
      if (a.res_mode == RES_BNRELU) bn_coeffs(a.res_bn, co, a.Cout, sc, sh, m, r);
      if (a.epi_mode == EPI_MASK) bn_coeffs(a.mask_bn, co, a.Cout, sc, sh, m, r);
      if (a.epi_mode == EPI_BIAS_RELU || a.epi_mode == EPI_BIAS) bias = a.bias[co];
    }
    s_ebn[tid] = sc;
    s_ebn[NT * 16 + tid] = sh;
    s_ebn[2 * NT * 16 + tid] = m;
    s_ebn[3 * NT * 16 + tid] = r;
    s_ebn[4 * NT * 16 + tid] = bias;
  }
  floatx4 st1[NT], st2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { st1[j] = floatx4{0.f, 0.f, 0.f, 0.f}; st2[j] = floatx4{0.f, 0.f, 0.f, 0.f}; }
  __syncthreads();   // s_bn / s_ebn ready
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
  DTF_STAMP(1);
  int it = 0;

  const int tile_px = a.TH * a.TW;
  const int n_mt = (tile_px + 15) >> 4;
  const bool want_res = (a.res_mode == RES_IDENT) || (a.res_mode == RES_BNRELU);
  const bool want_mask = (a.epi_mode == EPI_MASK);

  for (; t < total_tiles; t += gpz) {
    const int n = t / tiles_img;
    const int tile = t - n * tiles_img;
    const int oy0 = (tile / a.tiles_x) * a.TH;
    const int ox0 = (tile % a.tiles_x) * a.TW;

    write_tile();
    __syncthreads();
    if (it < 3) DTF_STAMP(2 + 4 * it);
    if (t + gpz < total_tiles) load_tile(t + gpz);   // in flight during the MFMA phase

    for (int mb = wave * MTW; mb < n_mt; mb += 4 * MTW) {
      floatx4 acc[MTW][NT];
      floatx4 rv[MTW][NT], xv[MTW][NT];
      int aoff[MTW];
      size_t pbase[MTW];
      bool pok[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int p = (mb + i) * 16 + l15;
        const int pc = p < tile_px ? p : 0;
        const int ty = pc / a.TW;
        const int tx = pc - ty * a.TW;
        aoff[i] = ((ty * S) * LW + tx * S) * CPP;
        const int oy = oy0 + ty, ox = ox0 + tx;
        pok[i] = (p < tile_px) && (oy < a.Ho) && (ox < a.Wo);
        pbase[i] = ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
          rv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
          xv[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
          const int co = (z * NT + j) * 16 + 4 * q;
          // epilogue operands are requested now and consumed after the MFMA loop
          if (pok[i] && co < a.Cout && !(a.debug_skip & 4)) {
            if (want_res) rv[i][j] = *reinterpret_cast<const floatx4*>(a.res + pbase[i] + co);
            if (want_mask) {
              xv[i][j] = *reinterpret_cast<const floatx4*>(a.mask_src + pbase[i] + co);
              if (a.extra) rv[i][j] = *reinterpret_cast<const floatx4*>(a.extra + pbase[i] + co);
            }
          }
        }
      }

      if (it < 3 && mb == wave * MTW) DTF_STAMP(3 + 4 * it);
      if (!(a.debug_skip & 1)) {
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
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          int tap = 4 * g + q;
          if (tap >= KS * KS) tap = 0;
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
      }
      if (it < 3 && mb == wave * MTW) DTF_STAMP(4 + 4 * it);

#pragma unroll
      for (int i = 0; i < MTW; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int co = (z * NT + j) * 16 + 4 * q;
          if (!pok[i] || co >= a.Cout) continue;
          floatx4 v = acc[i][j];
          const size_t idx = pbase[i] + co;
          if (a.res_mode == RES_IDENT) {
            v += rv[i][j];
          } else if (a.res_mode == RES_BNRELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += fmaxf(rv[i][j][r] * e_scale[j][r] + e_shift[j][r], 0.f);
          } else if (a.res_mode == RES_SHORTCUT_A) {
            const int cres = a.Cout >> 1, cq = a.Cout >> 2;
            if (co >= cq && co < cq + cres) {
              const int p = (mb + i) * 16 + l15;
              const int ty = p / a.TW, tx = p - ty * a.TW;
              const int oy = oy0 + ty, ox = ox0 + tx;
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
            const floatx4 x = xv[i][j];
            if (a.extra) v += rv[i][j];
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
          if (!(a.debug_skip & 4) || (idx == 0 && v[0] == 12345.f)) *reinterpret_cast<floatx4*>(a.out + idx) = v;
        }
      }
    }
    __syncthreads();   // all waves are done reading s_in before the next tile overwrites it
    if (it < 3) DTF_STAMP(5 + 4 * it);
    ++it;
  }

  DTF_STAMP(14);
  if (a.out_sums && !(a.debug_skip & 8)) {
    block_stat_flush<NT, 4>(smem, st1, st2, a.out_sums, a.Cout, z * NT * 16, a.epi_mode == EPI_MASK ? kStatGridBwd : kStatGridFwd);
  }
  DTF_STAMP(15);
#undef DTF_STAMP
}

size_t lds_bytes_p(int ks, int s, int cp, int nt, int th, int tw) {
  const int cpp = cp + 4;
  const int kg = (cp >= 16) ? ks * ks * (cp / 16) : (ks * ks + 3) / 4;
  const int lh = (th - 1) * s + ks, lw = (tw - 1) * s + ks;
  return ((size_t)((lh * lw * cpp + 3) & ~3) + (size_t)kg * nt * 256 + 2 * cp + 7 * nt * 16) * 4;
}

template <int KS, int S, int CP, int NT, int MTW, int NPF>
size_t launch_persist(const ConvFpropArgs& a, hipStream_t stream) {
  const size_t shmem = lds_bytes_p(KS, S, CP, NT, a.TH, a.TW);
  const int zsplit = a.nt_total / NT;
  const int total_tiles = a.tiles_y * a.tiles_x * a.B;
  // Co-resident blocks per CU: LDS-limited, and capped by the register budget the kernel was
  // compiled for (__launch_bounds__: 3 waves per SIMD for the shallow single-N-tile kernels, else 2).
  const int reg_bpc = (CP <= 16 && NT == 1) ? 3 : 2;
  int bpc = (int)((kLdsBudget) / shmem);
  if (bpc > reg_bpc) bpc = reg_bpc;
  if (bpc < 1) bpc = 1;
  int gpz_max = (256 * bpc) / zsplit;
  if (gpz_max < 8) gpz_max = 8;
  // Even work: iterations = ceil(tiles / gpz_max), then the smallest gpz that needs no more.
  const int iters = ceil_div(total_tiles, gpz_max);
  int gpz = ceil_div(total_tiles, iters);
  gpz = (gpz + 7) / 8 * 8;
  launch("conv_fprop_persist", conv_fprop_persist_kernel<KS, S, CP, NT, MTW, NPF>, dim3(gpz * zsplit), dim3(256), shmem,
         stream, a);
  return shmem;
}

}  // namespace

// Returns 0 when the shape is not covered (the caller falls back to the one-tile kernel).
size_t conv_fprop_persist(const ConvFpropArgs& a_in, int ks, hipStream_t stream) {
  if (ks != 3) return 0;
  ConvFpropArgs a = a_in;
  const int s = a.upsample ? 1 : a.stride;
  int cp;
  if (a.Cin <= 4) cp = 4;
  else if (a.Cin <= 16) cp = 16;
  else if (a.Cin <= 32) cp = 32;
  else if (a.Cin <= 64) cp = 64;
  else return 0;
  if ((a.Cin & 3) != 0 && a.Cin > 4) return 0;
  if (a.Cout & 15) return 0;
  a.nt_total = a.Cout / 16;
  // Output tile: up to 32 wide, 128 (deep inputs) to 256 (shallow inputs) pixels.
  a.TW = a.Wo <= 32 ? a.Wo : 32;
  const int target_px = (cp >= 64 || s == 2) ? 128 : 256;
  a.TH = a.Ho;
  while (a.TH * a.TW > target_px && a.TH > 1) a.TH = (a.TH + 1) / 2;
  a.tiles_x = ceil_div(a.Wo, a.TW);
  a.tiles_y = ceil_div(a.Ho, a.TH);
  const int LH = (a.TH - 1) * s + 3, LW = (a.TW - 1) * s + 3;
  const int n_chunks = LH * LW * (cp / 4);
  const int npf = ceil_div(n_chunks, 256);
  const int n_mt = ceil_div(a.TH * a.TW, 16);
  // NT: all output channels in one block when the filter slice leaves room for two blocks per CU.
  int nt = a.nt_total;
  while (nt > 1 && (lds_bytes_p(3, s, cp, nt, a.TH, a.TW) > 80 * 1024 || nt == 3)) nt >>= 1;
  if (a.nt_total % nt) return 0;
  if (lds_bytes_p(3, s, cp, nt, a.TH, a.TW) > kLdsBudget) return 0;
  const bool mtw2 = (n_mt >= 8) && nt <= 2;

#define DTF_PCASE(S_, CP_, NT_, MTW_, NPF_)                                                         \
  if (s == S_ && cp == CP_ && nt == NT_ && (mtw2 ? 2 : 1) == MTW_ && npf <= NPF_)                    \
    return launch_persist<3, S_, CP_, NT_, MTW_, NPF_>(a, stream);

  DTF_PCASE(1, 4, 1, 2, 2)      // stem 3->16 @32x32
  DTF_PCASE(1, 16, 1, 2, 6)     // 16->16 @32x32 (forward and data gradient)
  DTF_PCASE(2, 16, 2, 2, 10)    // 16->32 stride 2
  DTF_PCASE(2, 16, 2, 1, 10)
  DTF_PCASE(1, 32, 2, 2, 6)     // 32->32 @16x16
  DTF_PCASE(1, 32, 2, 1, 6)
  DTF_PCASE(1, 32, 1, 2, 12)    // 32->16 data gradient of the stride-2 layer (zero-upsampled input)
  DTF_PCASE(2, 32, 2, 1, 10)    // 32->64 stride 2
  DTF_PCASE(2, 32, 4, 1, 10)
  DTF_PCASE(1, 64, 1, 1, 8)     // 64->64 @8x8
  DTF_PCASE(1, 64, 2, 1, 8)
  DTF_PCASE(1, 64, 1, 2, 12)    // 64->32 data gradient of the stride-2 layer
  DTF_PCASE(1, 64, 2, 2, 12)
  DTF_PCASE(1, 64, 2, 1, 12)
#undef DTF_PCASE
  return 0;
}

}  // namespace dtf
