// Device side of the update log: non-blocking gradient aggregation over peer-mapped HBM.
//
// Every replica applies the same sequence of reduced gradients ("updates"); update u is open while
// gradients computed on version u arrive (one 64-bit compare-exchange on the coordinator's word),
// is closed by the arrival that completes the quorum or by the chief's interval timer, and stays
// readable in the owners' memory for kUlogRing updates, so a replica that was not waiting - a
// straggler - drops its stale gradient and applies what it missed without anybody having waited
// for it (reference semantics: the first M fresh gradients make the update and ALL workers are
// released, DistributedLeNet/src/backup_worker_experiment/backup_worker_optimizer.py:241-307;
// timer-driven updates, DistributedResNet/src/sync_replicas_optimizer_modified/
// sync_replicas_optimizer_modified.py:207-214). The protocol is the one of
// distributed_tensorflow_amd/parallel/update_log.py; here arrival, reduction, gathering and the
// optimizer run inside two stream-ordered launches per step with no host involvement at all:
//   ulog_begin  (one thread)  back-pressure, interval take-back, staging decision
//   ulog_step   (grid)        stage -> arrive -> [close] -> per closed update: reduce own shard,
//                             publish, gather the owners' shards fused with Adam / momentum SGD
// and the chief's timer issues ulog_close_begin + ulog_close_reduce on a stream of its own.
//
// Memory: per replica one fine-grained control block (64-bit words, one per 128-byte line), one
// staging buffer (n floats) and one ring of kUlogRing reduced vectors (n floats each), all
// peer-mapped through HIP IPC. Flags are system-scope atomics; data is written before a
// system-scope fence + flag and read after the flag + an acquire fence.
//
// Who waits for whom: a replica whose gradient was accepted waits for the update to close and for
// the owners (= the contributors) to publish their shards; a replica in ULOG_SKIP mode waits for
// the current update to close; nobody else waits for anything but its own back-pressure (a
// contributor never runs more than kUlogRing updates ahead of the slowest live replica). Every
// wait is bounded by `timeout_ticks`; a timeout raises the replica's error word and ends the
// launch, it never spins forever.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dtf {

constexpr int kUlogMaxPeers = 8;
constexpr int kUlogRing = 16;
constexpr int kUlogWordStride = 16;   // 64-bit words per control word: one 128-byte line each
constexpr int kUlogMaxBlocks = 128;   // all blocks of ulog_step must be co-resident

enum UlogWord : int {
  UW_OPEN = 0,        // coordinator (rank 0's block): u<<32 | frozen<<31 | count<<16 | contributor mask
  UW_ABORT = 1,       // every block: e+1 once update e has all the gradients it will take
  UW_VERSION = 2,     // own block: updates this replica has applied
  UW_SEQ = 3,         // own: launch sequence of ulog_begin / ulog_step
  UW_ERROR = 4,       // own: nonzero after a wait timed out
  UW_PENDING = 5,     // own, interval mode: (u+1)<<16 | gradients staged for the still-open update u
  UW_COUNTER = 6,     // own: grid rendezvous of ulog_step
  UW_COUNTER2 = 7,    // own: grid rendezvous of ulog_close_reduce
  UW_CLOSE_DEC = 8,   // own: update the close kernels work on, +1 (0 = nothing)
  UW_D0 = 9,          // own: ulog_begin's decision for ulog_step
  UW_DEC_TAG = 10,    // own: 4 decision tags (seq<<16 | k)
  UW_DEC_VAL = 14,    // own: 4 decision values
  UW_APPLIED = 18,    // every block: applied[p], pushed by replica p
  UW_HIST_TAG = 26,   // every block: kUlogRing closed-update tags (u+1)<<32 | count<<16 | mask
  UW_HIST_AUX = UW_HIST_TAG + kUlogRing,   // closer_reduces<<63 | owners<<32 | float scale bits
  UW_READY = UW_HIST_AUX + kUlogRing,      // ready[slot][p] = u+1, pushed by owner p
  UW_STATS = UW_READY + kUlogRing * kUlogMaxPeers,   // own: see UlogStat
  UW_DEAD = UW_STATS + 8,                  // own block: mask of replicas the watchdog gave up on
  UW_COUNT = UW_DEAD + 1,
};
constexpr size_t kUlogCtlBytes = (size_t)UW_COUNT * kUlogWordStride * sizeof(unsigned long long);

// Counters kept in the own control block and mirrored to `status` (host-visible) after every launch.
enum UlogStat : int {
  US_SEQ = 0,          // launches completed (written last)
  US_VERSION = 1,      // updates applied
  US_ACCEPTED = 2,     // gradients that entered an update
  US_DROPPED = 3,      // gradients that were stale / late / beyond the staleness bound
  US_APPLIED = 4,      // updates applied by the last launch
  US_LAST = 5,         // last launch: bit 0 accepted, bit 1 closed the update, bits 8.. arrival order
  US_ERROR = 6,        // error word
  US_CONTRIB = 7,      // contributor count of the last applied update
};

enum UlogMode : int { ULOG_CONTRIBUTE = 0, ULOG_OPEN = 1, ULOG_CATCH_UP = 2, ULOG_SKIP = 3 };
enum UlogOpt : int { ULOG_ADAM = 0, ULOG_MOMENTUM = 1 };

struct UlogArgs {
  unsigned long long* ctl[kUlogMaxPeers];   // control blocks; ctl[rank] is local, the others peer mappings
  float* stage[kUlogMaxPeers];              // staging buffers, same convention
  float* red[kUlogMaxPeers];                // rings of reduced vectors [kUlogRing][n]
  const float* grads;                       // this replica's gradient, n floats, local
  float* p;                                 // parameters and optimizer slots, n floats each, local
  float* m;
  float* v;                                 // Adam only
  unsigned long long* status;               // [8] host-visible mirror of the UlogStat counters (nullable)
  long long n;                              // floats, multiple of 4
  int rank, world;
  int mode;                                 // UlogMode
  int quorum;                               // ULOG_CONTRIBUTE: the arrival that brings the count to it closes the update
  int abort_at;                             // > 0: the abort word rises to e+1 when the count reaches it
  float scale;                              // what the reduced sum is multiplied by; <= 0: 1 / count
  float weight;                             // multiplies this replica's gradient when it is staged
  long long max_version;                    // no contribution to an update at or beyond it
  int max_staleness;                        // ULOG_OPEN: drop when the open update is further ahead; < 0: never
  int opt;                                  // UlogOpt
  float lr0, decay;                         // lr(u) = lr0 * decay^floor(u / decay_steps)
  int decay_steps;
  float beta1, beta2, eps, momentum, weight_decay;
  long long timeout_ticks;                  // 100 MHz wall-clock ticks a wait may take
};

void ulog_step(const UlogArgs& a, int blocks, hipStream_t stream);    // ulog_begin + ulog_step
void ulog_close(const UlogArgs& a, int blocks, hipStream_t stream);   // ulog_close_begin + ulog_close_reduce
// Initialises a freshly allocated (zeroed) control block for a run that starts at version u0.
void ulog_init_ctl(unsigned long long* ctl, long long u0, hipStream_t stream);
// ORs `mask` into the block's dead-replica mask (replicas the back-pressure no longer waits for).
void ulog_mark_dead(unsigned long long* ctl, unsigned mask, hipStream_t stream);
// Copies `count` control words starting at `first` to the host (synchronous; for end-of-run
// agreement, tests and the watchdog, never on the training path).
void ulog_peek(const unsigned long long* ctl, int first, int count, unsigned long long* out);
// Overwrites one control word from the host (synchronous; tests and fault injection only).
void ulog_poke(unsigned long long* ctl, int word, unsigned long long value);
// The same store from a one-thread kernel on `stream`, `delay_ticks` (100 MHz) after the kernel
// started: lands while other streams' kernels run (tests raise the abort word in the middle of a
// step this way).
void ulog_poke_async(unsigned long long* ctl, int word, unsigned long long value, long long delay_ticks, hipStream_t stream);

}  // namespace dtf
