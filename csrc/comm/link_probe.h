// First-contact probe for peer-mapped memory: two ranks exchange a flag word - and a payload - over
// each other's fine-grained HBM with exactly the primitives the update log and the asynchronous
// shards are built from (device_sync.h: st_rel / ld_acq / cas at system scope, one control word per
// 128-byte line). Between two GPUs the traffic crosses xGMI; between two processes on one GPU the
// same code runs through the same IPC mappings. It answers, in microseconds and GB/s, the three
// questions a multi-GPU bring-up starts with: does a release store to a peer become visible to its
// acquire load, does the payload written before it arrive intact, and is a compare-exchange on a
// peer's word atomic against the peer's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dtf {

enum LinkProbeWord : int { LP_FLAG = 0, LP_COUNTER = 1, LP_WORDS = 2 };
constexpr size_t kLinkProbeCtlBytes = (size_t)LP_WORDS * 16 * sizeof(unsigned long long);

struct LinkProbeArgs {
  unsigned long long* own_ctl;    // this rank's control words (LP_FLAG, LP_COUNTER), fine-grained
  unsigned long long* peer_ctl;   // the peer's, through the IPC mapping
  float* own_buf;                 // payload landing zone of this rank: payload_floats floats
  float* peer_buf;
  unsigned long long* counter;    // the shared counter both ranks compare-exchange on (rank 0's LP_COUNTER)
  long long payload_floats;       // multiple of 4; 0 = flag only
  int iters;                      // round trips
  int cas_iters;                  // compare-exchange increments per rank
  int role;                       // 0 starts every round trip, 1 answers
  long long timeout_ticks;        // bound of every wait (100 MHz ticks)
  // out, host-visible: [0] ticks of the flag-only round trips, [1] ticks of the round trips with
  // payload, [2] payload words that did not hold the expected value, [3] 1 = a wait timed out,
  // [4] compare-exchange retries
  unsigned long long* result;
};

// One workgroup of 256 threads on `stream`; both ranks must launch it (the waits are bounded).
void link_probe(const LinkProbeArgs& a, hipStream_t stream);

}  // namespace dtf
