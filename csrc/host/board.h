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
constexpr int kAggRing = 16;       // closed updates kept for replicas that lag behind
constexpr int kAggMaxRanks = 16;   // width of the contributor / owner masks of the update log

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

// One closed update of the aggregation log (see the "update log" block in class Board).
struct BoardAggEntry {
  std::atomic<uint64_t> tag;       // (u+1)<<32 | count<<16 | contributor mask; 0 = never closed
  std::atomic<uint32_t> owners;    // replicas that hold a reduced shard of this update
  std::atomic<uint32_t> scale_bits;  // float: what the reduced SUM is multiplied by (1/M, 1/count, 1/sum of weights)
  std::atomic<int64_t> ready[kAggMaxRanks];   // ready[p] = u+1 once owner p has published its reduced shard
};

struct BoardLayout {
  uint64_t magic;
  int32_t world;
  int32_t version;
  std::atomic<uint64_t> nonce;                      // identifies the run that created the segment
  std::atomic<int64_t> save_step;                   // iteration at which every replica joins a checkpoint gather (-1 = none)
  // update log
  std::atomic<uint64_t> agg_open;                   // u<<32 | frozen<<31 | count<<16 | contributor mask
  std::atomic<int64_t> agg_abort;                   // e+1 once update e has all the gradients it will take
  std::atomic<int64_t> agg_applied[kAggMaxRanks];   // version (updates applied) of every replica
  std::atomic<uint32_t> agg_dead;                   // replicas the watchdog has given up on
  BoardAggEntry agg_hist[kAggRing];
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
  // nonce != 0: the creator stamps it into the segment and an attacher keeps re-opening the name
  // until it finds that stamp (a segment left behind by an earlier run carries another one).
  Board(const std::string& name, int world, bool create, uint64_t nonce = 0);
  ~Board();
  Board(const Board&) = delete;
  Board& operator=(const Board&) = delete;

  static void unlink(const std::string& name);
  static int64_t now_ns();

  int world() const { return lay_->world; }
  const std::string& name() const { return name_; }

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
  // A worker whose helper thread (interval timer, shard service) failed stops the job and leaves its
  // rank for the others' watchdogs to name; -1 = nobody did.
  void report_failure(int rank);
  int failed_rank() const;

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

  // ---- heartbeats: wall-clock ns of the replica's last sign of life; 0 = not started yet,
  // kDeparted = it finished its work and left on purpose (never to be taken for dead)
  static constexpr int64_t kDeparted = INT64_MAX;
  void heartbeat(int rank);
  int64_t last_heartbeat_ns(int rank) const;
  void depart(int rank);

  // ---- async parameter shards
  int64_t shard_version(int owner) const;
  int64_t bump_shard_version(int owner);
  void set_shard_version(int owner, int64_t version);              // a resumed run continues the restored step count
  bool ring_doorbell(int owner, int sender, int64_t local_step);   // false if the previous push is still pending
  int64_t poll_doorbell(int owner, int sender) const;              // -1 if empty, else local_step
  void clear_doorbell(int owner, int sender);

  // ---- checkpoint rendezvous (sharded optimizer state must be gathered by all replicas at the
  // same iteration; the chief announces the iteration two steps ahead)
  void request_save(int64_t iteration);
  int64_t save_step() const;

  // ---- update log: non-blocking aggregation (backup workers, interval, soft-kill, weighted).
  // Replicas apply the same sequence of reduced gradients ("updates"). Update u is OPEN while
  // gradients computed on version u arrive; it is CLOSED by the arrival that completes the quorum
  // or by the chief's interval timer; a closed update records who contributed, who holds the
  // reduced shards and the scale, and stays readable for kAggRing updates so that a replica that
  // was not waiting (a straggler) can apply it later. Nobody ever waits for a non-contributor.
  struct AggArrival {
    int status;      // 0 = accepted, 1 = stale (update e is already closed), 2 = late (quorum already full)
    int order;       // gradients accepted before this one (status 0)
    int closed;      // 1 if this arrival completed the quorum and closed the update
    int64_t open_u;  // update that was open when the call returned
  };
  void agg_reset(int64_t u0);
  // Registers `add` gradients of replica `rank` for update e. quorum > 0: the arrival that brings
  // the count to quorum closes the update (owners = contributors, scale as given, or 1/count when
  // scale <= 0); abort_at > 0: the abort word is raised to e+1 when the count reaches it.
  // any_update: accept whatever update is open (interval mode) instead of only e.
  AggArrival agg_arrive(int64_t e, int rank, int add, int quorum, int abort_at, float scale, bool any_update);
  // Interval mode: withdraws this replica's pending contribution from the still-open update e.
  bool agg_take_back(int64_t e, int rank, int count);
  // Chief timer: closes the open update if it holds at least one gradient; the closer owns the
  // whole reduced vector. Returns the closed update or -1.
  int64_t agg_close(int owner_rank);
  int64_t agg_open_update() const;
  int agg_open_count() const;
  uint32_t agg_open_mask() const;
  // Closed-update record; false while update u is not closed (or already recycled: *recycled).
  bool agg_entry(int64_t u, int* count, uint32_t* mask, uint32_t* owners, float* scale, bool* recycled) const;
  bool agg_wait_closed(int64_t u, double timeout_s) const;
  void agg_publish_ready(int64_t u, int owner);
  bool agg_is_ready(int64_t u, int owner) const;
  bool agg_wait_ready(int64_t u, uint32_t owners, double timeout_s) const;
  void agg_set_applied(int rank, int64_t version);
  int64_t agg_applied(int rank) const;
  int64_t agg_min_applied() const;   // over replicas not marked dead
  bool agg_wait_min_applied(int64_t version, double timeout_s) const;
  int64_t agg_abort() const;
  void agg_raise_abort(int64_t value);
  void agg_mark_dead(int rank);
  uint32_t agg_dead() const;

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
