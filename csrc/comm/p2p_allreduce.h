// Two-shot direct all-reduce over peer-mapped buffers (one node, up to 8 GPUs on an xGMI mesh).
//
// RCCL's ring needs 2(N-1) serial hops for a gradient of a few MB, which is latency bound on a
// point-to-point mesh where every GPU has a private link to every other GPU. Here each rank
// reduces ONE shard by reading it from all peers at once (all links busy, one hop), then gathers
// the other reduced shards (second hop). Per-rank weights make it a masked / weighted
// all-reduce: a weight of zero drops that replica's gradient without changing who takes part
// (backup-worker and least-squares-weighted aggregation, reference:
// DistributedLeNet/src/backup_worker_optimizer.py:241-270,
// Distributed_test/src/sync_replicas_optimizer_modified/back_up.py:339-349).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace dtf {

constexpr int kMaxPeers = 8;
constexpr int kFlagStride = 32;   // uint32 per flag slot: one 128-byte line per (phase, peer)
constexpr int kFlagPhases = 4;    // staged, reduced, done, (block counters)
constexpr size_t kFlagBlockBytes = (size_t)kFlagPhases * kMaxPeers * kFlagStride * sizeof(uint32_t);

struct P2PAllReduceArgs {
  float* xbuf[kMaxPeers];       // exchange buffers, xbuf[rank] is local, the others peer mappings
  uint32_t* flags[kMaxPeers];   // flag blocks, same convention
  float weight[kMaxPeers];      // contribution weight of every rank (0 = masked out)
  float* data;                  // in: this rank's gradient, out: the weighted sum (local memory)
  long long n;                  // floats, multiple of 4
  int rank, world;
  uint32_t seq;                 // strictly increasing call number, identical on all ranks
  long long timeout_ticks;      // 100 MHz wall-clock ticks a wait may take before the call gives up
  int* error;                   // set to 1 when a wait timed out (local memory)
};

void p2p_allreduce(const P2PAllReduceArgs& a, int blocks, hipStream_t stream);

// Peer-shareable device allocations (uncached for remote agents when `fine_grained`).
void* ipc_alloc(size_t bytes, bool fine_grained);
void ipc_free(void* p);
std::string ipc_export(void* p);             // opaque handle bytes for another process
void* ipc_import(const std::string& handle);  // maps a peer's allocation into this process
void ipc_release(void* p);

}  // namespace dtf
