// Data gradient of a 3x3 stride-2 convolution (the two stage transitions of the CIFAR ResNet),
// decomposed by output parity.
//
// Treating it as a stride-1 convolution over the zero-upsampled output gradient (what the
// forward-shaped kernels do with `upsample`) multiplies three zeros for every useful product.
// With y[o] = sum_k x[2o + k] w[k] (TensorFlow SAME on an even size pads only after), the gradient
// of input pixel i = 2p + a only sees taps k = a (mod 2):
//     dx[2p+a, 2q+b] = sum_{kh in K(a), kw in K(b)} dy[p - (kh-a)/2, q - (kw-b)/2] . w[kh, kw]^T
// with K(0) = {0, 2}, K(1) = {1}: four small correlations (4, 2, 2 and 1 taps) over the SAME
// staged dy tile (one-pixel halo above / left), nine taps in total instead of thirty-six.
//
// One m-tile is 16 pixels of the half-resolution grid; for each of the four parity classes the
// block runs that class's taps on the matrix cores and writes 16 output pixels at stride 2.
// The epilogue is the masked data-gradient epilogue of conv_fprop_band.hip (ReLU mask from the
// forward input, BatchNorm-backward sums into per-XCD slots). Filters come packed for the
// data-gradient convolution (flipped and transposed), tap t' = (2-kh)*3 + (2-kw).
// (reference op: conv2d_backprop_input under opt.compute_gradients,
//  DistributedResNet/src/distributed_train.py:256)
#include "conv.h"

#include <cstdlib>

namespace dtf {
namespace {

template <int CP, int NT, int TW, int TH>
struct DS2 {
  static constexpr int CPP = CP + 4;
  static constexpr int CG = CP / 16;
  static constexpr int KG = 9 * CG;
  static constexpr int C4 = CP / 4;
  static constexpr int LH = TH + 1;
  static constexpr int LW = TW + 1;
  static constexpr int IN_CHUNKS = LH * LW * C4;
  static constexpr int NPF = (IN_CHUNKS + 255) / 256;
  static constexpr int W_CHUNKS = KG * NT * 64;
  static constexpr int NW = (W_CHUNKS + 255) / 256;
  static constexpr int N_MT = TH * TW / 16;
  static constexpr int IN_FLOATS = (LH * LW * CPP + 3) & ~3;
  static constexpr int W_FLOATS = KG * NT * 256;
  static constexpr int LDS_FLOATS = IN_FLOATS + W_FLOATS + 6 * NT * 16;
  static_assert((TH * TW) % 16 == 0, "a band must be whole 16-pixel MFMA tiles");
};

template <int CP, int NT, int TW, int TH>
__global__ __launch_bounds__(256, 2) void conv_dgrad_s2_kernel(ConvFpropArgs a) {
  DTF_ABORT_GUARD(a);
  using G = DS2<CP, NT, TW, TH>;
  constexpr int CPP = G::CPP, CG = G::CG, C4 = G::C4, LW = G::LW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = s_in + G::IN_FLOATS;
  float* s_stat = s_w + G::W_FLOATS;     // [2][NT*16]
  float* s_ebn = s_stat + 2 * NT * 16;   // [4][NT*16]: scale, shift, mean, rstd of the mask BatchNorm

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  const int l15 = lane & 15;

  const int zsplit = a.nt_total / NT;
  const int gpz = gridDim.x / zsplit;
  const int bid = blockIdx.x;
  const int z = (bid >> 3) % zsplit;
  const int jblk = (bid / (8 * zsplit)) * 8 + (bid & 7);
  const int bands = a.H / TH;             // bands of the half-resolution (dy) grid
  const int total_tiles = bands * a.B;

  // statistics first, then the filter slice, then the first tile (see conv_fprop_band.hip)
  const bool e_thread = tid < NT * 16;
  const int e_co = z * NT * 16 + tid;
  double eb_s1 = 0.0, eb_s2 = 0.0;
  float eb_g = 0.f, eb_b = 0.f;
  if (e_thread) {
    eb_s1 = stat_total(a.mask_bn.sums, e_co, a.Cout);
    eb_s2 = stat_total(a.mask_bn.sums, a.Cout + e_co, a.Cout);
    eb_g = a.mask_bn.gamma[e_co];
    eb_b = a.mask_bn.beta[e_co];
  }
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
  int in_goff[G::NPF], in_loff[G::NPF], in_row[G::NPF];
#pragma unroll
  for (int k = 0; k < G::NPF; ++k) {
    const int i = tid + k * 256;
    const int pix = i / C4, c4 = i % C4;
    const int r = pix / LW, c = pix % LW;
    const bool ok = (i < G::IN_CHUNKS) && c >= 1;     // column -1 is padding
    in_goff[k] = ((r - 1) * a.W + (c - 1)) * CP + c4 * 4;
    in_loff[k] = pix * CPP + c4 * 4;
    in_row[k] = ok ? r - 1 : -100000;
  }
  floatx4 pf[G::NPF];
  auto load_tile = [&](int t) {
    const int n = t / bands;
    const int p0 = (t - n * bands) * TH;
    const float* in0 = a.in + ((long long)(n * a.H + p0) * a.W) * CP;
#pragma unroll
    for (int k = 0; k < G::NPF; ++k) {
      const int iy = p0 + in_row[k];
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < a.H) v = *reinterpret_cast<const floatx4*>(in0 + in_goff[k]);
      pf[k] = v;
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int k = 0; k < G::NPF; ++k)
      if (tid + k * 256 < G::IN_CHUNKS) *reinterpret_cast<floatx4*>(s_in + in_loff[k]) = pf[k];
  };

  int t = jblk;
  if (t < total_tiles) load_tile(t);

  if (e_thread) {
    const double mean = eb_s1 * (double)a.mask_bn.inv_count;
    double var = eb_s2 * (double)a.mask_bn.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float r = rsqrtf((float)var + a.mask_bn.eps);
    const float m = (float)mean;
    const float sc = eb_g * r;
    s_ebn[tid] = sc;
    s_ebn[NT * 16 + tid] = eb_b - m * sc;
    s_ebn[2 * NT * 16 + tid] = m;
    s_ebn[3 * NT * 16 + tid] = r;
  }
#pragma unroll
  for (int k = 0; k < G::NW; ++k)
    if (tid + k * 256 < G::W_CHUNKS) reinterpret_cast<floatx4*>(s_w)[tid + k * 256] = wreg[k];
  __syncthreads();

  float e_scale[NT][4], e_shift[NT][4], e_mean[NT][4], e_rstd[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = j * 16 + 4 * q + r;
      e_scale[j][r] = s_ebn[c];
      e_shift[j][r] = s_ebn[NT * 16 + c];
      e_mean[j][r] = s_ebn[2 * NT * 16 + c];
      e_rstd[j][r] = s_ebn[3 * NT * 16 + c];
    }
  floatx4 st1[NT], st2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { st1[j] = floatx4{0.f, 0.f, 0.f, 0.f}; st2[j] = floatx4{0.f, 0.f, 0.f, 0.f}; }

  // half-resolution pixel l15 of an m-tile: LDS address of dy[p, q] (halo offset +1, +1) and the
  // offset of output pixel (2p, 2q)
  const int pr_l = l15 / TW, pc_l = l15 % TW;
  const int lane_a = ((pr_l + 1) * LW + pc_l + 1) * CPP + 4 * q;
  const int Wo = 2 * a.W;
  const int lane_o = ((2 * pr_l) * Wo + 2 * pc_l) * a.Cout + z * NT * 16 + 4 * q;

  for (; t < total_tiles; t += gpz) {
    const int n = t / bands;
    const int p0 = (t - n * bands) * TH;
    const long long obase = ((long long)(n * a.Ho + 2 * p0) * Wo) * a.Cout;

    write_tile();
    __syncthreads();
    if (t + gpz < total_tiles) load_tile(t + gpz);

    for (int mt = wave; mt < G::N_MT; mt += 4) {
      const int pm = mt * 16;   // first half-resolution pixel of the m-tile (wave-uniform)
      const int aoff = lane_a + ((pm / TW) * LW + (pm % TW)) * CPP;
      const long long ooff = obase + (long long)((2 * (pm / TW)) * Wo + 2 * (pm % TW)) * a.Cout + lane_o;
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
        const int pa = cls >> 1, pb = cls & 1;          // output row / column parity
        const long long o = ooff + (long long)(pa * Wo + pb) * a.Cout;
        floatx4 xv[NT], rv[NT], acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
          rv[j] = floatx4{0.f, 0.f, 0.f, 0.f};
          xv[j] = *reinterpret_cast<const floatx4*>(a.mask_src + o + j * 16);
          if (a.extra) rv[j] = *reinterpret_cast<const floatx4*>(a.extra + o + j * 16);
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          if ((kh & 1) != pa) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if ((kw & 1) != pb) continue;
            const int dr = (kh - pa) / 2, dc = (kw - pb) / 2;       // dy[p - dr, q - dc]
            const int toff = -(dr * LW + dc) * CPP;
            const int tap = (2 - kh) * 3 + (2 - kw);                // index in the flipped packed filter
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
              const int g = tap * CG + cg;
              const floatx4 av = *reinterpret_cast<const floatx4*>(s_in + aoff + toff + cg * 16);
              floatx4 bv[NT];
#pragma unroll
              for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const floatx4*>(s_w + (g * NT + j) * 256 + lane * 4);
#pragma unroll
              for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = DTF_MFMA16(bv[j][s], av[s], acc[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          floatx4 v = acc[j];
          const floatx4 x = xv[j];
          if (a.extra) v += rv[j];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = (x[r] * e_scale[j][r] + e_shift[j][r] > 0.f) ? v[r] : 0.f;
            st1[j][r] += v[r];
            st2[j][r] += v[r] * ((x[r] - e_mean[j][r]) * e_rstd[j][r]);
          }
          *reinterpret_cast<floatx4*>(a.out + o + j * 16) = v;
        }
      }
    }
    __syncthreads();
  }

  if (a.out_sums) block_stat_flush<NT, 4>(smem, st1, st2, a.out_sums, a.Cout, z * NT * 16, kStatGridBwd);
}

template <int CP, int NT, int TW, int TH>
size_t launch_ds2(const ConvFpropArgs& a, hipStream_t stream) {
  using G = DS2<CP, NT, TW, TH>;
  const size_t shmem = (size_t)G::LDS_FLOATS * sizeof(float);
  const int zsplit = a.nt_total / NT;
  const int total_tiles = (a.H / TH) * a.B;
  static const int bpc = [&] {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_dgrad_s2_kernel<CP, NT, TW, TH>), shmem);
    int n = 0;
    DTF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_dgrad_s2_kernel<CP, NT, TW, TH>, 256, shmem));
    return n < 1 ? 1 : n;
  }();
  int gpz_max = (256 * bpc) / zsplit;
  if (gpz_max < 8) gpz_max = 8;
  const int iters = ceil_div(total_tiles, gpz_max);
  int gpz = ceil_div(total_tiles, iters);
  gpz = (gpz + 7) / 8 * 8;
  launch("conv_dgrad_s2", conv_dgrad_s2_kernel<CP, NT, TW, TH>, dim3(gpz * zsplit), dim3(256), shmem, stream, a);
  return shmem;
}

}  // namespace

// Covers the call the forward-shaped API makes for a stride-2 layer's data gradient: `upsample`,
// pad 2, masked epilogue. Returns 0 for anything else.
size_t conv_dgrad_s2(const ConvFpropArgs& a_in, int ks, hipStream_t stream) {
  if (ks != 3 || !a_in.upsample || a_in.pad != 2 || a_in.epi_mode != EPI_MASK) return 0;
  if (a_in.in_mode != IN_RAW || a_in.res_mode != RES_NONE) return 0;
  ConvFpropArgs a = a_in;
  if (a.H != a.W || a.Ho != 2 * a.H || a.Wo != 2 * a.W || (a.Cout & 15)) return 0;
  a.nt_total = a.Cout / 16;
  if (a.Cin == 32 && a.Cout == 16 && a.W == 16 && a.H % 4 == 0) return launch_ds2<32, 1, 16, 4>(a, stream);
  if (a.Cin == 64 && a.Cout == 32 && a.W == 8 && a.H % 8 == 0) return launch_ds2<64, 1, 8, 8>(a, stream);
  return 0;
}

}  // namespace dtf
