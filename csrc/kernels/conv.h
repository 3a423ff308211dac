// Host API of the implicit-GEMM convolution kernels (NHWC activations, HWIO filters, fp32).
//
// The reference obtains these from TensorFlow's conv2d / conv2d_backprop_{input,filter} ops
// (reference: DistributedResNet/src/resnet.py:99,121,150 and DistributedLeNet/src/mnist.py:107-123).
#pragma once
#include "common.h"

// The `debug_skip` switches of the band kernels (timing experiments: parts of a kernel switched
// off at run time) cost branches and default-value moves in every tile, so they only exist in
// builds made with DTF_CONV_DEBUG=1 (build_native.py reads it from the environment).
#ifndef DTF_CONV_DEBUG
#define DTF_CONV_DEBUG 0
#endif
#define DTF_DEBUG_SKIP(a) (DTF_CONV_DEBUG ? (a).debug_skip : 0)

namespace dtf {

enum InMode : int { IN_RAW = 0, IN_BNRELU = 1 };
enum ResMode : int { RES_NONE = 0, RES_IDENT = 1, RES_BNRELU = 2, RES_SHORTCUT_A = 3 };
enum EpiMode : int { EPI_PLAIN = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_BIAS = 3 };

// One forward-shaped convolution: out = conv(in', W) [+ residual] with optional fused
// prologue (in' = relu(bn(in))) and epilogues. Also serves as the data-gradient kernel:
// dgrad of a stride-1 conv is a stride-1 conv with flipped/transposed filters, dgrad of a
// stride-2 conv is the same over the zero-upsampled gradient (`upsample`=1).
struct ConvFpropArgs {
  const float* in;    // [B,H,W,Cin]
  const float* wpk;   // packed filters, see pack_conv_weights
  float* out;         // [B,Ho,Wo,Cout]
  int B, H, W, Cin;
  int Ho, Wo, Cout;
  int TH, TW, tiles_y, tiles_x;  // output tile and tile grid per image
  int pad;                       // rows/cols of implicit zero padding before the input
  int stride;                    // 1 or 2 (must match the template instance)
  int upsample;                  // 1: `in` is read as if zero-upsampled by 2 (stride-2 dgrad)
  int nt_total;                  // 16-channel output tiles in the packed filter (= ceil(Cout/16))
  int debug_skip;                // timing experiments only: 1 = no MFMA, 2 = no input loads, 4 = no epilogue memory traffic, 8 = no statistics flush; 64 = force the ping-pong kernel (tests)
  unsigned long long* dbg_stamps;  // timing experiments only: [blocks][16] wall-clock stamps of thread 0 (nullable)
  // prologue
  int in_mode;
  BnRef in_bn;
  // residual (forward)
  int res_mode;
  const float* res;  // RES_IDENT/RES_BNRELU: [B,Ho,Wo,Cout]; RES_SHORTCUT_A: [B,2Ho,2Wo,Cout/2]
  BnRef res_bn;
  // epilogue
  int epi_mode;
  const float* bias;      // EPI_BIAS_RELU / EPI_BIAS: [Cout]
  const float* mask_src;  // EPI_MASK: x such that the mask is bn(x) > 0, [B,Ho,Wo,Cout]
  BnRef mask_bn;
  const float* extra;     // EPI_MASK: optional gradient added before masking, [B,Ho,Wo,Cout]
  // statistics accumulated over the written tensor (nullable, [kStatSlots][2*Cout]): for forward
  // the per-channel sum and sum of squares of `out`; for EPI_MASK the sums of g and g*xhat.
  double* out_sums;
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};

// ks in {3,5}; cin_pad in {4,16,32,64}; launches over all B images.
// Returns the dynamic LDS bytes used (for reporting), throws on unsupported shapes.
size_t conv_fprop(const ConvFpropArgs& a, int ks, hipStream_t stream);
// Compile-time band-geometry variant for the 3x3 ResNet shapes; returns 0 when the shape is not covered.
size_t conv_fprop_band(const ConvFpropArgs& a, int ks, hipStream_t stream);
// 8-wave ping-pong variant of the stride-1 band kernel (two wave groups half a period apart); 0 = not covered.
size_t conv_fprop_pp(const ConvFpropArgs& a, int ks, hipStream_t stream);
// Parity-decomposed data gradient of a 3x3 stride-2 layer (the `upsample` call); 0 = not covered.
size_t conv_dgrad_s2(const ConvFpropArgs& a, int ks, hipStream_t stream);
// Picks (TH, TW, tiles, cout split) for the given problem; fills the tiling fields of `a`.
void conv_fprop_plan(ConvFpropArgs& a, int ks);

// Packs HWIO filters [ks,ks,Cin,Cout] into the MFMA B-operand layout
//   packed[kgroup][ntile][lane 0..63][4]
// `transpose_flip`=1 packs the data-gradient filter W'[kh][kw][co][ci] = W[ks-1-kh][ks-1-kw][ci][co].
struct PackArgs {
  const float* w;
  float* out;
  int ks, Cin, Cout;   // of the forward filter
  int transpose_flip;
};
void pack_conv_weights(const PackArgs& a, hipStream_t stream);
size_t packed_weight_floats(int ks, int Cin, int Cout, int transpose_flip);
// Opt-in "bf16x6" variant of the stride-1 band kernel for the 16- and 32-channel layers (conv_fprop_split.hip):
// fp32 operands cut exactly into three bf16 values each, six partial products per product on the bf16
// matrix pipe, fp32 accumulation. `wpk` is then the output of pack_split_weights. 0 = not covered.
size_t conv_fprop_split(const ConvFpropArgs& a, hipStream_t stream);
void pack_split_weights(const PackArgs& a, hipStream_t stream);   // ks = 3, Cin == Cout in {16, 32}
size_t packed_split_floats(int C);

// Filter gradient: dW[kh,kw,ci,co] += sum_{n,y,x} act(in)[n, y*s+kh-pad, x*s+kw-pad, ci] * dout[n,y,x,co]
// accumulated with fp32 atomics into `dw` (HWIO), which the caller zeroes beforehand. With
// `dw_slot_stride` > 0 a workgroup adds into the copy dw + (blockIdx.x % kStatSlots) * stride, so
// the atomics of one XCD never leave its L2; the caller sums the kStatSlots copies afterwards.
struct ConvWgradArgs {
  const float* in;     // [B,H,W,Cin]
  const float* dout;   // [B,Ho,Wo,Cout]
  float* dw;           // [ks,ks,Cin,Cout]
  int B, H, W, Cin;
  int Ho, Wo, Cout;
  int TH, TW, tiles_y, tiles_x;
  int pad, stride;
  int in_mode;
  BnRef in_bn;
  int tiles_per_block;   // tiles of the flattened (image, tile) sequence walked by one block
  float* dbias;          // nullable: [Cout] += sum of dout (LeNet bias gradient)
  long long dw_slot_stride;  // floats between the per-XCD copies of dw (0: one copy)
  // Band kernel only, non-null: no atomics. Workgroup b of a cout block stores its part of the sum
  // into copy b of conv_wgrad_band_copies() copies of the whole filter gradient, `dw_part` + b * ks*ks*Cin*Cout
  // (the cout blocks fill disjoint columns of a copy; every element of every copy is written);
  // part_reduce adds the copies up in a fixed order, so the gradient has the same bits on every run.
  float* dw_part;
  int co_tiles;          // 16-channel output tiles per block (0: chosen by the launcher)
  int debug_skip;        // timing experiments only: 1 = no MFMA, 2 = no loads, 4 = no flush
  unsigned long long* dbg_stamps;  // timing experiments only: [blocks][16] wall-clock stamps of thread 0
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
size_t conv_wgrad(ConvWgradArgs a, int ks, hipStream_t stream);
// Band-tiled persistent variant for the 3x3 ResNet shapes; returns 0 when the shape is not covered.
size_t conv_wgrad_band(const ConvWgradArgs& a, int ks, hipStream_t stream);
// Copies of the filter gradient that call would fill through `dw_part` (0: shape not covered).
int conv_wgrad_band_copies(const ConvWgradArgs& a, int ks);


// Whole backward pass of one stride-1 3x3 layer y = conv(relu(bn(x)), W) in one kernel:
//   g   = 1[bn(x) > 0] * (conv(dy, W') + extra)      (gradient at the BatchNorm output, W' = flipped/transposed W)
//   gsums += (sum g, sum g * xhat)                    (per-XCD statistic slots)
//   dw  += sum_p relu(bn(x))[p + tap - 1] * dy[p]      (into the gradient copy of the block's XCD)
struct ConvBwdFusedArgs {
  const float* dy;       // [B,H,W,C]
  const float* x;        // [B,H,W,C] the layer's (pre-BatchNorm) input
  const float* wpk_t;    // data-gradient packing of W (pack_conv_weights with transpose_flip = 1)
  float* g;              // [B,H,W,C]
  float* dw;             // [3,3,C,C] HWIO, copy 0 of kStatSlots copies
  long long dw_slot_stride;
  // Non-null: no atomics (see ConvWgradArgs::dw_part). The workgroups that share a tile sequence
  // (one per channel split) fill copy j of conv_bwd_fused_copies() copies of the whole [3,3,C,C]
  // gradient with plain stores; part_reduce sums the copies in a fixed order.
  float* dw_part;
  BnRef bn;              // BatchNorm of x
  double* gsums;         // [kStatSlots][2*C], nullable
  const float* extra;    // [B,H,W,C] gradient added before masking, nullable
  int B, H, W, C;
  // Optional BatchNorm-backward prologue: `dy` then holds the masked gradient g' at the output of
  // the BatchNorm that follows this layer's convolution, and the kernel reads
  //     dy_eff = gamma*rstd * (g' - mean(g') - xhat(y) * mean(g'*xhat(y)))
  // while staging, from g', the convolution output y (`dy_x`), that BatchNorm's statistics
  // (`dy_bn`) and its backward sums (`dy_gsums`): the BatchNorm input gradient is never written
  // to memory. Block 0 also writes that BatchNorm's dgamma / dbeta.
  const float* dy_x;       // nullable = no prologue
  BnRef dy_bn;
  const double* dy_gsums;  // [kStatSlots][2*C]
  float* dy_dgamma;        // [C], nullable
  float* dy_dbeta;         // [C], nullable
  // With the prologue, optionally dy_eff += dy_res (the gradient arriving over an identity
  // shortcut) and dy_eff is written to dy_out: the residual block's input gradient, which the
  // block before needs again, is produced here instead of by a separate elementwise pass.
  const float* dy_res;     // [B,H,W,C], nullable
  float* dy_out;           // [B,H,W,C], required with dy_res
  // dy_res_mode 1: the block after this layer's halves the resolution and doubles the channels
  // (option-A shortcut: 2x2 average pool, channels zero-padded by C/2 on each side); dy_res is
  // then its output gradient [B,H/2,W/2,2C] and the term is dy_res[n, y/2, x/2, C/2 + c] / 4.
  int dy_res_mode;
  // Optional dynamic tile scheduling: kBwdTileCounters zeroed ints. A block takes its first two
  // bands statically and every further one from the counter of its XCD (and channel split), so
  // when some CUs are shared with another kernel (an all-reduce of an earlier gradient bucket, a
  // neighbour process) the remaining blocks pick up the work instead of the grid waiting for the
  // slow blocks' fixed share. nullptr = static striding.
  int* tile_counters;
  // timing experiments only (conv_bwd_split.hip): [blocks][4 waves][48]: for the first 7 bands six wall-clock
  // stamps of lane 0 - band start, staged, past the barrier, prefetch issued, MFMA phase done, past the
  // second barrier - then at [42], [43] the wave's HW_ID and XCC_ID registers
  unsigned long long* dbg_stamps;
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
constexpr int kBwdTileCounters = 32;   // 8 XCDs x up to 4 channel splits
bool conv_bwd_fused_supported(int B, int H, int W, int C);
size_t conv_bwd_fused(const ConvBwdFusedArgs& a, hipStream_t stream);
int conv_bwd_fused_copies(const ConvBwdFusedArgs& a);
// Opt-in "bf16x6" variant (conv_bwd_split.hip, conv_bwd_split_pc.hip): both products on the
// bf16 matrix pipe with operands split three ways, fp32 accumulation. `wpk_t` is then the output of
// pack_split_weights with transpose_flip = 1; no dynamic band scheduling.
bool conv_bwd_split_supported(int B, int H, int W, int C);
size_t conv_bwd_split(const ConvBwdFusedArgs& a, hipStream_t stream);
int conv_bwd_split_copies(const ConvBwdFusedArgs& a);
// 64-channel instance: producer / consumer waves, one workgroup per CU (conv_bwd_split_pc.hip); reached
// through conv_bwd_split. copies != nullptr: no launch, *copies = gradient copies the launch would fill.
bool conv_bwd_split_pc_supported(int B, int H, int W, int C);
size_t conv_bwd_split_pc(const ConvBwdFusedArgs& a, hipStream_t stream, int* copies);

// ---- stem of the CIFAR ResNet: 3x3, 3 -> 16 channels, SAME, 32x32 images (conv_stem.hip)
// forward: y = conv(x, w), sums (nullable) += per-channel (sum, sum of squares) of y in the
// statistic-slot layout; gradient: dw [3,3,3,16] += filter gradient from dy, with slot_stride != 0
// into the copy of the workgroup's XCD. w / dw are the plain HWIO filter (no packing).
struct ConvStemArgs {
  const float* x;    // [B,32,32,3]
  const float* w;    // [27][16], forward
  float* y;          // [B,32,32,16], forward
  double* sums;      // [kStatSlots][32], forward, nullable
  const float* dy;   // [B,32,32,16], gradient
  float* dw;         // [27][16], gradient
  float* dw_part;    // gradient, non-null: workgroup b stores its sum at dw_part + b * 432 instead (conv_stem_wgrad_copies() copies)
  int slot_stride;
  int B;
  // Optional BatchNorm-backward prologue of the gradient kernel (as ConvBwdFusedArgs::dy_x): `dy`
  // is the masked gradient at the output of the BatchNorm that follows the stem, `dy_x` the stem
  // output; the BatchNorm input gradient is formed while dy is staged and never written. Block 0
  // writes that BatchNorm's dgamma / dbeta.
  const float* dy_x;       // [B,32,32,16], nullable = no prologue
  BnRef dy_bn;
  const double* dy_gsums;  // [kStatSlots][32]
  float* dy_dgamma;        // [16], nullable
  float* dy_dbeta;         // [16], nullable
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
bool conv_stem_supported(int H, int W, int Cin, int Cout, int ks, int stride);
void conv_stem_fwd(const ConvStemArgs& a, hipStream_t stream);
void conv_stem_wgrad(const ConvStemArgs& a, hipStream_t stream);
int conv_stem_wgrad_copies(const ConvStemArgs& a);

// ---- 5x5 stride-1 SAME convolution over 14x14 maps, whole images per workgroup (conv5x5_tile.hip)
// Forward form (flip = 0): Cin 32 -> Cout 64, y = relu(conv(x, w) + bias), w = conv5x5_pack of
// the HWIO filter ([25][Cout][Cin]). Data-gradient form (flip = 1): Cin 64 -> Cout 32 over the
// output gradient, w = the HWIO filter itself (its [tap][in][out] rows are the [tap][Cout][Cin]
// rows of the transposed problem; the kernel walks the taps backwards), no bias / activation.
struct Conv5x5Args {
  const float* x;     // [B,14,14,Cin]
  const float* w;     // [25][Cout][Cin]
  const float* bias;  // [Cout], forward form only
  float* y;           // [B,14,14,Cout]
  int B, Cin, Cout;
  int flip;
  int imgs;           // images per workgroup tile: 0 = choose by batch, 1 or 2 (forward form)
  // Data-gradient form only, instead of x: the gradient dp [B,7,7,64] at the 2x2 max pool that
  // follows the convolution, the pool's winner positions and its (ReLU) maxima; the kernel stages
  // the pool + ReLU backward of dp on the fly.
  const float* pool_dp;
  const unsigned char* pool_arg;
  const float* pool_p;
  // Forward form only, instead of y: the 2x2 / 2 max pool of the activation, [B,7,7,64] maxima and
  // winner positions (maxpool2x2_fwd's encoding); the activation itself is not written.
  float* pool_out;
  unsigned char* pool_arg_out;
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
struct Conv5x5PackArgs {
  const float* w;   // [25][Cin][Cout]
  float* out;       // [25][Cout][Cin]
  int Cin, Cout;
  int flip;         // conv5x5_split_pack only: pack the data-gradient operand (taps reversed, channel roles swapped)
};
// Opt-in split-operand ("bf16x6") form of conv5x5_tile, forward and data gradient (conv5x5_split.hip;
// the filter gradient's form is conv5x5_wgrad_split below):
// `w` is then conv5x5_split_pack's output for the same direction (float-typed storage of
// conv5x5_split_packed_floats() elements holding bf16 values), `imgs` is ignored.
void conv5x5_split(const Conv5x5Args& a, hipStream_t stream);
void conv5x5_split_pack(const Conv5x5PackArgs& a, hipStream_t stream);   // a.w = the HWIO filter [5,5,32,64]
size_t conv5x5_split_packed_floats();
// Filter gradient of that layer: dw [25][32][64] (HWIO) += sum_p x[p + tap] (x) dy[p]; with
// slot_stride != 0 into the copy of the workgroup's XCD (dw + slot * slot_stride).
struct Conv5x5WgradArgs {
  const float* x;    // [B,14,14,32]
  const float* dy;   // [B,14,14,64]
  float* dw;
  int slot_stride;
  int B;
  // Instead of dy (see Conv5x5Args): pooled gradient, winner positions, pooled maxima.
  const float* pool_dp;
  const unsigned char* pool_arg;
  const float* pool_p;
  // Non-null: workgroup b stores its whole partial result at partials + b * 25*32*64 with plain
  // stores instead of adding it to dw (no atomics; sum the conv5x5_wgrad_parts(B) copies with
  // partial_sum). Every element is written, so the buffer needs no clearing.
  float* partials;
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
int conv5x5_wgrad_parts(int B);
void conv5x5_wgrad(const Conv5x5WgradArgs& a, hipStream_t stream);
// Opt-in split-operand ("bf16x6") form of conv5x5_wgrad (conv5x5_split.hip): same arguments and results layout.
void conv5x5_wgrad_split(const Conv5x5WgradArgs& a, hipStream_t stream);
bool conv5x5_tile_supported(int H, int W, int Cin, int Cout);
void conv5x5_tile(const Conv5x5Args& a, hipStream_t stream);
void conv5x5_pack(const Conv5x5PackArgs& a, hipStream_t stream);

}  // namespace dtf
