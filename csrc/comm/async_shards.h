// Asynchronous SGD over one-sided, staleness-bounded parameter shards, device side.
//
// Replica r owns shard r of the flat parameter vector: the authoritative parameters of that range
// (a fine-grained buffer every peer maps) and its optimizer slots. A worker iteration is, on the
// worker's stream and with no host synchronisation:
//   async_pull     one-sided read of every owner's shard (and of its version word) into the local
//                  parameter arena
//   forward + backward
//   async_push     one-sided write of each gradient slice into the owner's per-sender inbox, then a
//                  doorbell word carrying the version the gradient was computed on
// and, on a side stream of the OWNER, launched every fraction of a millisecond by a service thread,
//   async_service  one sweep over the doorbells of its shard: a pushed gradient is applied (Adam /
//                  momentum SGD over the shard, learning rate from the shard's version) iff
//                  version - pushed_version <= tau, else dropped: the rule of TensorFlow's
//                  ConditionalAccumulator (reference: backup_worker_optimizer.py:44-51,241-246)
//                  with the bound as a parameter. The sweep is a short kernel of its own, so pushes
//                  are applied while the owner's forward / backward pass is running - between two
//                  of its kernels - and even while the owner's training thread is not running at
//                  all. It is deliberately NOT a resident polling kernel: the convolution kernels
//                  are persistent grids sized to two workgroups per CU with up to 243 vector
//                  registers per lane, and a resident workgroup on a CU keeps one of the two from
//                  being placed until the grid's tail (measured: +20 % on every convolution).
// Consistency of a pull: it takes no lock. The version words are read first (a kernel of their
// own), then the shards are copied while the owners' sweeps may be writing them, so one pull can
// hold parameters of several versions of a shard - never older than the recorded version, possibly
// newer, and possibly a shard that is part-way through an update. That is the consistency of the
// reference's workers reading variables from the parameter server while it applies other workers'
// gradients, and what the staleness bound is defined against (the recorded version is a lower bound).
// The optimizer slots of a shard live in peer-mapped buffers next to the authoritative parameters,
// so the chief can copy the whole optimizer state of every owner into its arena for a checkpoint
// without stopping anybody (DeviceAsyncShards.gather_state).
// Nobody waits for another replica's compute; a push waits (bounded) only for its own previous push
// to the same owner to be consumed, and is dropped if that does not happen in time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dtf {

constexpr int kAsyncMaxPeers = 8;
constexpr int kAsyncHistBins = 16;     // staleness histogram: 0..14, 15 = more
constexpr int kAsyncMaxBlocks = 64;    // blocks of a launch must be co-resident

enum AsyncWord : int {
  AW_VERSION = 0,        // owner: updates applied to this shard
  AW_DOORBELL = 1,       // [sender] pushed version + 1 of a gradient waiting in the inbox, 0 = empty
  AW_SVC_SEQ = 9,        // owner: launch sequence of the service kernel
  AW_ERROR = 10,
  AW_SVC_COUNTER = 11,   // rendezvous state of the service kernel
  AW_SVC_TAG = 12,       // 4
  AW_SVC_VAL = 16,       // 4
  AW_WRK_COUNTER = 20,   // rendezvous state of the worker-stream kernels (push)
  AW_WRK_TAG = 21,       // 4
  AW_WRK_VAL = 25,       // 4
  AW_WRK_SEQ = 29,
  AW_PULLED = 30,        // [owner] version read by the last pull
  AW_STATS = 38,         // applied, dropped (stale), pushed, push given up, + kAsyncHistBins staleness bins
  AW_COUNT = AW_STATS + 4 + kAsyncHistBins,
};
constexpr size_t kAsyncCtlBytes = (size_t)AW_COUNT * 16 * sizeof(unsigned long long);

struct AsyncArgs {
  unsigned long long* ctl[kAsyncMaxPeers];   // control blocks (own + peer mappings)
  float* shard[kAsyncMaxPeers];              // authoritative parameters of every shard
  float* inbox[kAsyncMaxPeers];              // owner o's inboxes: [world][shard_floats], this pointer = o's base
  float* params;                             // local arena (padded): world * shard_floats floats
  const float* grads;                        // local gradient, same layout
  float* m;                                  // optimizer slots of the OWN shard: shard_floats floats each
  float* v;
  unsigned long long* status;                // host-visible mirror (nullable): [0..7] pulled versions, [8..] stats
  long long shard_floats;                    // floats per shard, multiple of 4
  int rank, world;
  int tau;                                   // staleness bound
  int opt;                                   // 0 Adam, 1 momentum SGD
  float lr0, decay;
  int decay_steps;
  float beta1, beta2, eps, momentum, weight_decay;
  long long timeout_ticks;                   // bound of any wait (100 MHz ticks)
  long long push_wait_ticks;                 // how long a push waits for its inbox slot
};

void async_pull(const AsyncArgs& a, int blocks, hipStream_t stream);
void async_push(const AsyncArgs& a, int blocks, hipStream_t stream);
void async_service(const AsyncArgs& a, int blocks, hipStream_t stream);

}  // namespace dtf
