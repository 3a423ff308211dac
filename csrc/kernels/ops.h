// Host API of the non-convolution kernels (BatchNorm, classifier head, loss, optimizers,
// elementwise, pooling, dropout, GEMM, data augmentation).
#pragma once
#include "common.h"

namespace dtf {

// ---- BatchNorm (batch statistics in training AND evaluation; eps 0.001) ------------------
// reference: DistributedResNet/src/resnet.py:65-84 (tf.nn.moments over axes [0,1,2] +
// tf.nn.batch_normalization; no moving averages).
struct BnStatsArgs {
  const float* x;   // [N, C] flattened NHWC
  double* sums;     // [kStatSlots][2*C] += sum, sum of squares
  long long n_rows;
  int C;
};
void bn_stats(const BnStatsArgs& a, hipStream_t stream);

struct BnApplyArgs {
  const float* x;
  float* y;
  BnRef bn;
  long long n_rows;
  int C;
  int relu;
};
void bn_apply(const BnApplyArgs& a, hipStream_t stream);

// Masked gradient g = dy * 1[bn(x) > 0] with the two per-channel reductions BatchNorm's
// backward needs (sum g, sum g*xhat), for use when no convolution epilogue produces them.
struct BnReluBwdReduceArgs {
  const float* dy;
  const float* x;
  float* g;         // may alias dy
  BnRef bn;
  double* gsums;    // [kStatSlots][2*C] += sum g, sum g*xhat
  long long n_rows;
  int C;
  int relu;         // 0: no mask (plain BatchNorm)
};
void bn_relu_bwd_reduce(const BnReluBwdReduceArgs& a, hipStream_t stream);

enum BnBwdRes : int { BWD_RES_NONE = 0, BWD_RES_IDENT = 1, BWD_RES_SHORTCUT_A = 2 };
// dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) [+ residual gradient]; block 0 also
// writes dgamma = sum g*xhat and dbeta = sum g.
struct BnBwdApplyArgs {
  const float* g;
  const float* x;
  float* dx;            // may alias g
  BnRef bn;
  const double* gsums;  // [kStatSlots][2*C]
  float* dgamma;        // nullable
  float* dbeta;         // nullable
  long long n_rows;
  int C;
  int res_mode;
  const float* res;     // IDENT: [n_rows, C]; SHORTCUT_A: gradient of the block output [B,H/2,W/2,2C]
  int H, W;             // spatial dims of x (needed for SHORTCUT_A)
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
void bn_bwd_apply(const BnBwdApplyArgs& a, hipStream_t stream);

// ---- classifier head: BN + ReLU + global average pool + FC + softmax cross-entropy --------
// reference: DistributedResNet/src/resnet.py:209-215, DistributedResNet/src/distributed_train.py:119-129
struct HeadFwdArgs {
  const float* x;       // [B, HW, C]
  BnRef bn;
  const float* fc_w;    // [C, K]
  const float* fc_b;    // [K]
  const int* labels;    // [B]
  float* feat;          // [B, C]   pooled features (saved for backward)
  float* logits;        // [B, K]
  float* dlogits;       // [B, K]   (softmax - onehot) / B
  float* loss;          // [slots] += mean cross-entropy; image n adds to element n % slots, the loss is their sum
  int* correct;         // [slots] += number of top-1 hits, spread the same way (nullable)
  int B, HW, C, K;
  int slots;            // power of two; one atomic target for 1024 blocks serialises them (~25 ns each)
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
void head_fwd(const HeadFwdArgs& a, hipStream_t stream);

struct HeadBwdArgs {
  const float* x;        // [B, HW, C]
  BnRef bn;
  const float* fc_w;     // [C, K]
  const float* feat;     // [B, C]
  const float* dlogits;  // [B, K]
  float* g;              // [B, HW, C]  masked gradient wrt bn output
  double* gsums;         // [kStatSlots][2*C] += sum g, sum g*xhat
  float* d_fc_w;         // [C, K] +=
  float* d_fc_b;         // [K] +=
  int B, HW, C, K;
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
void head_bwd(const HeadBwdArgs& a, hipStream_t stream);

// ---- softmax cross-entropy on logits (LeNet path) -----------------------------------------
// reference: DistributedLeNet/src/mnist.py:149-164
struct SoftmaxXentArgs {
  const float* logits;  // [B, K]
  const int* labels;    // [B]
  float* dlogits;       // [B, K] nullable
  float* loss;          // [1] +=
  int* correct;         // [1] += nullable
  int B, K;
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
void softmax_xent(const SoftmaxXentArgs& a, hipStream_t stream);

// Fully connected classifier of the MNIST network in one launch (see fc_head.hip): bias + ReLU +
// dropout of the hidden layer, the output layer, softmax cross-entropy, and the whole way back to
// the gradient at the hidden layer's pre-activation. Gradient outputs are all given or all null
// (evaluation). keep_prob >= 1 means no dropout.
struct FcHeadArgs {
  const float* h_raw;   // [B, H] first layer's product, before its bias
  const float* b1;      // [H]
  const float* w2;      // [H, K]
  const float* b2;      // [K]
  const int* labels;    // [B]
  float* logits;        // [B, K]
  float* loss;          // [1] += mean loss
  int* correct;         // [1] += top-1 hits, nullable
  float* dh;            // [B, H] gradient at the hidden pre-activation, nullable
  float* dw2;           // [H, K] +=
  float* db2;           // [K] +=
  float* db1;           // [H] +=
  int B, H, K;
  float keep_prob;
  unsigned long long seed, offset;
  const unsigned long long* offset_dev;   // nullable, added to offset at run time (see DropoutArgs)
  int max_blocks;   // 0: the launcher's choice; a workgroup walks tiles of eight samples
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
void fc_head(const FcHeadArgs& a, hipStream_t stream);

// ---- optimizers over the flat parameter arena ---------------------------------------------
// hyper (device): [0]=lr_t  [1]=beta1  [2]=beta2  [3]=eps  [4]=grad_scale  [5]=momentum  [6]=weight_decay
// Adam follows tf.train.AdamOptimizer (reference: DistributedLeNet/src/distributed_train.py:164):
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (computed on the host), m,v updates, p -= lr_t*m/(sqrt(v)+eps).
struct FlatOptArgs {
  float* p;
  const float* g;
  float* m;
  float* v;           // unused by momentum SGD
  const float* hyper;
  long long n;
};
void fused_adam(const FlatOptArgs& a, hipStream_t stream);
void fused_momentum_sgd(const FlatOptArgs& a, hipStream_t stream);

// Writes up to 8 floats and up to 8 64-bit integers, passed BY VALUE, to device memory in stream
// order: the step-dependent scalars of a replayed program (learning rate, gradient scale, step
// counter) without a host staging buffer that a later step could overwrite before the copy ran.
constexpr int kSetScalarsMax = 8;
struct SetScalarsArgs {
  float* f_dst; unsigned long long* u_dst;
  float f[kSetScalarsMax]; unsigned long long u[kSetScalarsMax];
  int nf, nu;
};
void set_scalars(const SetScalarsArgs& a, hipStream_t stream);

// ---- elementwise / flat ---------------------------------------------------------------------
struct FlatScaleArgs { float* x; const float* scale_dev; float scale; long long n; };
void flat_scale(const FlatScaleArgs& a, hipStream_t stream);   // x *= (scale_dev ? *scale_dev : scale)
struct AxpyArgs { float* y; const float* x; float alpha; long long n; };
void flat_axpy(const AxpyArgs& a, hipStream_t stream);         // y += alpha*x

// y[0:n] += sum of the kStatSlots copies slots[r*slot_stride + (0:n)] (per-XCD filter-gradient
// copies, see ConvWgradArgs::dw_slot_stride); the copies are zeroed where they held anything.
struct SlotReduceArgs { float* y; float* slots; long long n; long long slot_stride; };
void slot_reduce(const SlotReduceArgs& a, hipStream_t stream);
// Fixed-order sum of per-workgroup gradient copies (ConvBwdFusedArgs::dw_part and friends), all
// tensors of a gradient bucket in one launch: slots[k][dst_off + i] = part[k][i] + part[k + 8][i] +
// part[k + 16][i] + ... for k < kStatSlots, stored (not added); slot_reduce then adds the eight
// slots in its own fixed order. Nothing depends on which workgroup finished first, so two runs
// from the same state produce the same bits.
struct PartEntry {
  const float* part;     // [copies][4 * numel4]
  long long dst_off;     // float offset of the tensor inside a slot (multiple of 4)
  int copies;
  int numel4;            // float4s per copy
  int first_item;        // work items of the entries before this one
  int pad;
};
struct PartReduceArgs { const PartEntry* table; int n_entries; float* slots; long long slot_stride; };
void part_reduce(const PartReduceArgs& a, int total_items, hipStream_t stream);
// y[0:n] += sum over p < parts of ws[p*stride + (0:n)] (per-workgroup partial results written with
// plain stores; n and stride multiples of 4).
struct PartialSumArgs { float* y; const float* ws; long long n; int parts; long long stride; AbortRef abort; };
void partial_sum(const PartialSumArgs& a, hipStream_t stream);

struct AddReluArgs { const float* a; const float* b; float* out; long long n; int relu_first; };
void add_relu_fwd(const AddReluArgs& a, hipStream_t stream);   // relu_first ? relu(a)+b : relu(a+b)
struct AddReluBwdArgs { const float* dout; const float* a; const float* b; float* da; float* db; long long n; int relu_first; };
void add_relu_bwd(const AddReluBwdArgs& a, hipStream_t stream);

// option-A shortcut: avg_pool 2x2/2 VALID + zero-pad channels C/2 each side
// reference: DistributedResNet/src/resnet.py:159-163
struct ShortcutAArgs { const float* x; float* y; int B, H, W, C; };   // y: [B,H/2,W/2,2C]
void shortcut_a_fwd(const ShortcutAArgs& a, hipStream_t stream);
struct ShortcutABwdArgs { const float* dy; float* dx; int B, H, W, C; int accumulate; };
void shortcut_a_bwd(const ShortcutABwdArgs& a, hipStream_t stream);

// max_pool 2x2 stride 2 SAME (reference: DistributedLeNet/src/mnist.py:115-118)
struct MaxPoolArgs { const float* x; float* y; unsigned char* argmax; int B, H, W, C; };  // y: [B,ceil(H/2),ceil(W/2),C]
void maxpool2x2_fwd(const MaxPoolArgs& a, hipStream_t stream);
struct MaxPoolBwdArgs { const float* dy; const unsigned char* argmax; const float* y; float* dx; int B, H, W, C; int relu_mask; };
void maxpool2x2_bwd(const MaxPoolBwdArgs& a, hipStream_t stream);

// First LeNet layer fused with its 2x2/2 max pool (lenet_conv1.hip): 28x28x1 images, 5x5x1x32 HWIO
// filter, SAME padding. Forward writes only the pooled ReLU maximum p [B,14,14,32] and the winner
// position arg (same encoding as maxpool2x2_fwd). Backward adds the filter gradient [25][32] and
// the bias gradient [32] computed from the pooled gradient dp; with slot_stride != 0 they go to
// the copy of the block's XCD (dw + slot * slot_stride, dbias + slot * slot_stride).
struct Conv1PoolFwdArgs { const float* x; const float* w; const float* bias; float* p; unsigned char* arg; int B; AbortRef abort; };
void conv1_pool_fwd(const Conv1PoolFwdArgs& a, hipStream_t stream);
struct Conv1PoolBwdArgs { const float* x; const float* dp; const unsigned char* arg; const float* p; float* dw; float* dbias;
                          int slot_stride; int B; AbortRef abort; };
void conv1_pool_bwd(const Conv1PoolBwdArgs& a, hipStream_t stream);

// dropout with a counter-based Philox-4x32-10 stream: mask regenerated from (seed, offset)
// in backward (reference: DistributedLeNet/src/mnist.py:139-140, keep_prob 0.5, seed 66478)
struct DropoutArgs { const float* x; float* y; long long n; float keep_prob; unsigned long long seed; unsigned long long offset;
                     const unsigned long long* offset_dev; AbortRef abort; };   // offset_dev (nullable): added to offset, read at run time so a replayed program draws a new mask per step
void dropout_apply(const DropoutArgs& a, hipStream_t stream);   // same kernel for fwd (x) and bwd (dy)

// fp32 GEMM on MFMA: C[M,N] = op(A) * op(B) (+bias[N]) (relu) ; row-major storage
struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K;
  int lda, ldb, ldc;
  int trans_a, trans_b;   // trans_a: A stored [K,M]; trans_b: B stored [N,K]
  int relu;
  int accumulate;         // C += result
  int variant, splits;    // 0: the launcher's choice (see gemm.hip); tuning runs set them
  AbortRef abort;   // short-circuit reference, filled in by the launch dispatcher
};
void gemm_f32(const GemmArgs& a, hipStream_t stream);

// out[C] += column sums of x [n_rows, C]. With relu_of (same shape): of x * 1[relu_of > 0], the
// gradient through a ReLU whose output was relu_of; that masked tensor is also stored to `masked`
// when given (which may be x itself): ReLU backward and the bias gradient in one pass.
struct ColSumArgs { const float* x; float* out; long long n_rows; int C; const float* relu_of; float* masked; AbortRef abort; };
void col_sum(const ColSumArgs& a, hipStream_t stream);
struct ReluMaskArgs { const float* dy; const float* y; float* dx; long long n; AbortRef abort; };   // dx = dy * 1[y>0]
void relu_bwd(const ReluMaskArgs& a, hipStream_t stream);

// evaluation / summaries
struct Top1Args { const float* logits; const int* labels; int* correct; int B, K; };
void top1_count(const Top1Args& a, hipStream_t stream);
struct ZeroFracArgs { const float* x; unsigned long long* zeros; long long n; };
void zero_count(const ZeroFracArgs& a, hipStream_t stream);
// Histogram over caller-supplied ascending bucket limits (tf.summary.histogram semantics: a value
// lands in the first bucket whose limit is greater than it; reference: activation_summary,
// DistributedResNet/src/resnet.py:13-20). counts [nb] += per-bucket counts; moments [2] += (sum,
// sum of squares); minmax [2] = order-preserving integer keys of the minimum / maximum (start them
// at INT_MAX / INT_MIN; decode with histogram_key_to_float).
struct HistogramArgs { const float* x; long long n; const double* limits; int nb; unsigned long long* counts; double* moments; int* minmax; };
void tf_histogram(const HistogramArgs& a, hipStream_t stream);

// CIFAR augmentation on device: random crop from the zero-padded image + per-image whitening
// reference: DistributedResNet/src/cifar10_input.py:111-143
struct CropWhitenArgs {
  const float* src;     // [N, HP, WP, C] padded images
  const int* index;     // [B] source image per output
  const int* offs;      // [B,2] crop offsets (y, x); nullable = (0,0)
  float* out;           // [B, H, W, C]
  int B, HP, WP, H, W, C;
  int whiten;
};
void crop_whiten(const CropWhitenArgs& a, hipStream_t stream);

// Packs every conv filter of a model in one launch. The table lives in device memory:
// per entry {w_off, out_off, ks, Cin, Cout, transpose_flip, n_floats, first_block}; transpose_flip 2 / 3 =
// the split-operand layout of pack_split_weights (plain / data-gradient filter), n_floats = packed_split_floats.
struct PackEntry { long long w_off; long long out_off; int ks, Cin, Cout, transpose_flip; int n_floats; int first_block; };
struct PackMultiArgs { const float* params; float* packed; const PackEntry* table; int n_entries; };
void pack_multi(const PackMultiArgs& a, int total_blocks, hipStream_t stream);

// Zeroes up to kZeroMultiMax buffers in one launch (the per-step clears of a training program:
// statistics, counters, loss, gradients). Every buffer is 16-byte aligned; sizes count 16-byte chunks.
constexpr int kZeroMultiMax = 8;
struct ZeroMultiArgs { void* ptr[kZeroMultiMax]; unsigned long long chunks[kZeroMultiMax]; int count; };
void zero_multi(const ZeroMultiArgs& a, hipStream_t stream);

}  // namespace dtf
