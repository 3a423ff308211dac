// Shared-memory control board: the single-node control plane of the straggler-mitigation
// strategies.
//
// The reference coordinates workers through TensorFlow FIFOQueues on the parameter server
// (token barrier, stop queue; reference: backup_worker_optimizer.py:274-307,
// soft_kill_optimizer.py:220-236,306-316) and a Twisted RPC mesh for the ready barrier and
// timing telemetry (reference: DistributedLeNet/src/timeout_manager.py:18-77,198-211). With one
// process per GPU on one node, all of that state fits in one POSIX shared-memory segment of
// lock-free atomics:
//   * per-step arrival counters: `arrive(step)` returns this replica's arrival order, which is
//     what decides membership in "the first M gradients" (backup workers) without any message;
//   * per-step stop flags and deadlines polled by stragglers between backward segments
//     (soft-kill / interval aggregation);
//   * the global step, per-rank ready flags (startup barrier), per-rank tokens, heartbeats;
//   * a ring of per-(rank, iteration) timestamps for the worker-time CDF telemetry.
// Steps index a ring of kRing slots; a slot is recycled kRing steps later, and a slot's tag
// tells readers which step it currently belongs to.
#pragma once
#include <atomic>
#include <cstdint>
#include <string>

namespace dtf {

constexpr int kBoardMaxRanks = 64;
constexpr int kBoardRing = 64;
constexpr int kBoardTimeRing = 1024;

struct BoardStepSlot {
  std::atomic<int64_t> tag;          // step this slot currently serves (-1 = free)
  std::atomic<int32_t> arrivals;     // replicas that have arrived
  std::atomic<int32_t> stop;         // nonzero: enough gradients arrived / deadline passed
  std::atomic<int64_t> start_ns;     // wall-clock ns at which the step window opened (0 = unset)
  std::atomic<uint64_t> mask;        // bit r set: replica r's gradient is part of the update
  std::atomic<int32_t> mask_final;   // 1 once `mask` is frozen
  std::atomic<int32_t> pad;
};

struct BoardTimeRec {
  std::atomic<int64_t> iter;         // -1 = empty
  std::atomic<int64_t> t_dequeue_ns;
  std::atomic<int64_t> t_finish_ns;
};

struct BoardLayout {
  uint64_t magic;
  int32_t world;
  int32_t version;
  std::atomic<int64_t> global_step;
  std::atomic<int32_t> should_stop;
  std::atomic<int32_t> generation;                  // bumped by barriers
  std::atomic<int32_t> barrier_count;
  std::atomic<int32_t> ready[kBoardMaxRanks];
  std::atomic<int64_t> heartbeat_ns[kBoardMaxRanks];
  std::atomic<int64_t> tokens[kBoardMaxRanks];      // per-replica token queue depth
  std::atomic<int64_t> token_step[kBoardMaxRanks];  // step carried by the newest token
  std::atomic<int64_t> local_step[kBoardMaxRanks];
  // async parameter shards: version of each owner's shard and per-(owner, sender) doorbells
  std::atomic<int64_t> shard_version[kBoardMaxRanks];
  std::atomic<int64_t> doorbell[kBoardMaxRanks][kBoardMaxRanks];   // [owner][sender]: local_step+1 of a pending push, 0 = empty
  BoardStepSlot steps[kBoardRing];
  BoardTimeRec times[kBoardMaxRanks][kBoardTimeRing];
};

class Board {
 public:
  // Creates (rank 0 / create=true) or attaches to the named segment.
  Board(const std::string& name, int world, bool create);
  ~Board();
  Board(const Board&) = delete;
  Board& operator=(const Board&) = delete;

  static void unlink(const std::string& name);
  static int64_t now_ns();

  int world() const { return lay_->world; }

  // ---- startup barrier (reference: timeout_manager.py launch_manager spin-until-all-ready)
  void mark_ready(int rank);
  bool all_ready() const;
  bool wait_all_ready(double timeout_s) const;
  // Sense-reversing barrier over all ranks (host side).
  bool barrier(double timeout_s);

  // ---- global step / stop
  int64_t global_step() const;
  void set_global_step(int64_t s);
  int64_t advance_global_step();
  void request_stop();
  bool should_stop() const;

  // ---- per-step arrival protocol
  // Opens the step window if nobody did yet and returns its start time.
  int64_t open_step(int64_t step);
  // Registers this replica's gradient; returns its 0-based arrival order. When the order
  // reaches `quorum`-1 the stop flag is raised (quorum <= 0: never). Replicas arriving with
  // order < quorum (or any, if quorum <= 0) are added to the mask.
  int arrive(int64_t step, int rank, int quorum);
  // Registers a replica that gave up (short-circuited backward): counted, never in the mask.
  int arrive_aborted(int64_t step, int rank);
  int arrivals(int64_t step) const;
  bool stop_requested(int64_t step) const;
  void raise_stop(int64_t step);
  int64_t step_start_ns(int64_t step) const;
  // Freezes and returns the participation mask (idempotent).
  uint64_t freeze_mask(int64_t step);
  uint64_t mask(int64_t step) const;
  // Blocks until all `world` replicas have arrived (or aborted) at `step`.
  bool wait_arrivals(int64_t step, int count, double timeout_s) const;

  // ---- tokens (reference: sync_token_q / per-worker sync_token_q_%d)
  void put_tokens(int rank, int64_t n, int64_t step);
  bool take_token(int rank, double timeout_s, int64_t* step_out);
  int64_t token_depth(int rank) const;
  void set_local_step(int rank, int64_t s);
  int64_t local_step(int rank) const;

  // ---- heartbeats
  void heartbeat(int rank);
  int64_t last_heartbeat_ns(int rank) const;

  // ---- async parameter shards
  int64_t shard_version(int owner) const;
  int64_t bump_shard_version(int owner);
  bool ring_doorbell(int owner, int sender, int64_t local_step);   // false if the previous push is still pending
  int64_t poll_doorbell(int owner, int sender) const;              // -1 if empty, else local_step
  void clear_doorbell(int owner, int sender);

  // ---- timing telemetry
  void post_times(int rank, int64_t iter, int64_t t_dequeue_ns, int64_t t_finish_ns);
  bool get_times(int rank, int64_t iter, int64_t* t_dequeue_ns, int64_t* t_finish_ns) const;

 private:
  BoardStepSlot& slot(int64_t step) const;
  std::string name_;
  BoardLayout* lay_ = nullptr;
  int fd_ = -1;
  bool owner_ = false;
};

}  // namespace dtf
