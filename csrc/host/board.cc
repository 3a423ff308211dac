#include "board.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <limits>
#include <new>
#include <stdexcept>
#include <thread>

namespace dtf {
namespace {
constexpr uint64_t kMagic = 0x6474665f626f6172ull;  // "dtf_boar"
constexpr int64_t kInit = std::numeric_limits<int64_t>::min();

inline void relax(int spins) {
  if (spins < 64)
    std::this_thread::yield();
  else
    std::this_thread::sleep_for(std::chrono::microseconds(50));
}
}  // namespace

int64_t Board::now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

Board::Board(const std::string& name, int world, bool create, uint64_t nonce) : name_(name), owner_(create) {
  if (world < 1 || world > kBoardMaxRanks) throw std::runtime_error("Board: world size out of range");
  const std::string path = name[0] == '/' ? name : "/" + name;
  name_ = path;
  if (create) {
    ::shm_unlink(path.c_str());
    fd_ = ::shm_open(path.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd_ < 0) throw std::runtime_error("Board: shm_open(create) failed for " + path);
    if (::ftruncate(fd_, sizeof(BoardLayout)) != 0) throw std::runtime_error("Board: ftruncate failed");
  } else {
    // The creator may still be setting up, or the name may still point at the segment of an
    // earlier run (killed before it could unlink): keep re-opening the NAME until the segment
    // behind it is initialised and, when a nonce is given, stamped with it.
    for (int attempt = 0; attempt < 6000; ++attempt) {
      fd_ = ::shm_open(path.c_str(), O_RDWR, 0600);
      if (fd_ >= 0) {
        struct stat st;
        if (::fstat(fd_, &st) == 0 && (size_t)st.st_size >= sizeof(BoardLayout)) {
          void* q = ::mmap(nullptr, sizeof(BoardLayout), PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
          if (q != MAP_FAILED) {
            BoardLayout* l = static_cast<BoardLayout*>(q);
            if (l->magic == kMagic && (nonce == 0 || l->nonce.load(std::memory_order_acquire) == nonce)) {
              lay_ = l;
              break;
            }
            ::munmap(q, sizeof(BoardLayout));
          }
        }
        ::close(fd_);
        fd_ = -1;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    if (!lay_) throw std::runtime_error("Board: no initialised segment of this run found at " + path);
    if (lay_->world != world) throw std::runtime_error("Board: world size mismatch with the creator");
    return;
  }
  void* p = ::mmap(nullptr, sizeof(BoardLayout), PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
  if (p == MAP_FAILED) throw std::runtime_error("Board: mmap failed");
  lay_ = static_cast<BoardLayout*>(p);
  if (create) {
    // ftruncate gives zero pages; zero is a valid initial state for every atomic except tags.
    lay_->world = world;
    lay_->version = 1;
    for (auto& s : lay_->steps) s.tag.store(-1, std::memory_order_relaxed);
    for (auto& r : lay_->times)
      for (auto& t : r) t.iter.store(-1, std::memory_order_relaxed);
    lay_->save_step.store(-1, std::memory_order_relaxed);
    lay_->nonce.store(nonce, std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_release);
    lay_->magic = kMagic;
  }
}

Board::~Board() {
  if (lay_) ::munmap(lay_, sizeof(BoardLayout));
  if (fd_ >= 0) ::close(fd_);
  if (owner_) ::shm_unlink(name_.c_str());
}

void Board::unlink(const std::string& name) {
  const std::string path = name[0] == '/' ? name : "/" + name;
  ::shm_unlink(path.c_str());
}

// ---- startup ---------------------------------------------------------------------------------
void Board::mark_ready(int rank) { lay_->ready[rank].store(1, std::memory_order_release); }

bool Board::all_ready() const {
  for (int r = 0; r < lay_->world; ++r)
    if (!lay_->ready[r].load(std::memory_order_acquire)) return false;
  return true;
}

bool Board::wait_all_ready(double timeout_s) const {
  const int64_t deadline = now_ns() + int64_t(timeout_s * 1e9);
  for (int spins = 0; !all_ready(); ++spins) {
    if (now_ns() > deadline) return false;
    relax(spins);
  }
  return true;
}

bool Board::barrier(double timeout_s) {
  const int gen = lay_->generation.load(std::memory_order_acquire);
  const int arrived = lay_->barrier_count.fetch_add(1, std::memory_order_acq_rel) + 1;
  if (arrived == lay_->world) {
    lay_->barrier_count.store(0, std::memory_order_relaxed);
    lay_->generation.fetch_add(1, std::memory_order_acq_rel);
    return true;
  }
  const int64_t deadline = now_ns() + int64_t(timeout_s * 1e9);
  for (int spins = 0; lay_->generation.load(std::memory_order_acquire) == gen; ++spins) {
    if (now_ns() > deadline) return false;
    relax(spins);
  }
  return true;
}

// ---- global step -------------------------------------------------------------------------------
int64_t Board::global_step() const { return lay_->global_step.load(std::memory_order_acquire); }
void Board::set_global_step(int64_t s) { lay_->global_step.store(s, std::memory_order_release); }
int64_t Board::advance_global_step() { return lay_->global_step.fetch_add(1, std::memory_order_acq_rel) + 1; }
void Board::request_stop() { lay_->should_stop.store(1, std::memory_order_release); }
bool Board::should_stop() const { return lay_->should_stop.load(std::memory_order_acquire) != 0; }
void Board::report_failure(int rank) { lay_->should_stop.store(2 + rank, std::memory_order_release); }
int Board::failed_rank() const {
  const int32_t v = lay_->should_stop.load(std::memory_order_acquire);
  return v >= 2 ? v - 2 : -1;
}

// ---- per-step protocol ---------------------------------------------------------------------------
BoardStepSlot& Board::slot(int64_t step) const {
  if (step < 0) throw std::runtime_error("Board: negative step");
  BoardStepSlot& s = lay_->steps[step % kBoardRing];
  for (int spins = 0;; ++spins) {
    int64_t t = s.tag.load(std::memory_order_acquire);
    if (t == step) return s;
    if (t == kInit) {
      relax(spins);
      continue;
    }
    if (t > step) throw std::runtime_error("Board: step slot already recycled (replica lags more than the ring)");
    if (s.tag.compare_exchange_strong(t, kInit, std::memory_order_acq_rel)) {
      s.arrivals.store(0, std::memory_order_relaxed);
      s.stop.store(0, std::memory_order_relaxed);
      s.start_ns.store(0, std::memory_order_relaxed);
      s.mask.store(0, std::memory_order_relaxed);
      s.mask_final.store(0, std::memory_order_relaxed);
      s.tag.store(step, std::memory_order_release);
      return s;
    }
  }
}

int64_t Board::open_step(int64_t step) {
  BoardStepSlot& s = slot(step);
  int64_t expected = 0;
  const int64_t now = now_ns();
  if (s.start_ns.compare_exchange_strong(expected, now, std::memory_order_acq_rel)) return now;
  return expected;
}

int Board::arrive(int64_t step, int rank, int quorum) {
  BoardStepSlot& s = slot(step);
  const int order = s.arrivals.fetch_add(1, std::memory_order_acq_rel);
  if (quorum <= 0 || order < quorum) {
    if (!s.mask_final.load(std::memory_order_acquire)) s.mask.fetch_or(uint64_t(1) << rank, std::memory_order_acq_rel);
  }
  if (quorum > 0 && order + 1 >= quorum) s.stop.store(1, std::memory_order_release);
  return order;
}

int Board::arrive_aborted(int64_t step, int rank) {
  (void)rank;
  BoardStepSlot& s = slot(step);
  return s.arrivals.fetch_add(1, std::memory_order_acq_rel);
}

int Board::arrivals(int64_t step) const { return slot(step).arrivals.load(std::memory_order_acquire); }
bool Board::stop_requested(int64_t step) const { return slot(step).stop.load(std::memory_order_acquire) != 0; }
void Board::raise_stop(int64_t step) { slot(step).stop.store(1, std::memory_order_release); }
int64_t Board::step_start_ns(int64_t step) const { return slot(step).start_ns.load(std::memory_order_acquire); }

uint64_t Board::freeze_mask(int64_t step) {
  BoardStepSlot& s = slot(step);
  s.mask_final.store(1, std::memory_order_release);
  return s.mask.load(std::memory_order_acquire);
}
uint64_t Board::mask(int64_t step) const { return slot(step).mask.load(std::memory_order_acquire); }

bool Board::wait_arrivals(int64_t step, int count, double timeout_s) const {
  BoardStepSlot& s = slot(step);
  const int64_t deadline = now_ns() + int64_t(timeout_s * 1e9);
  for (int spins = 0; s.arrivals.load(std::memory_order_acquire) < count; ++spins) {
    if (now_ns() > deadline) return false;
    relax(spins);
  }
  return true;
}

// ---- tokens ----------------------------------------------------------------------------------------
void Board::put_tokens(int rank, int64_t n, int64_t step) {
  lay_->token_step[rank].store(step, std::memory_order_release);
  lay_->tokens[rank].fetch_add(n, std::memory_order_acq_rel);
}

bool Board::take_token(int rank, double timeout_s, int64_t* step_out) {
  const int64_t deadline = now_ns() + int64_t(timeout_s * 1e9);
  for (int spins = 0;; ++spins) {
    int64_t n = lay_->tokens[rank].load(std::memory_order_acquire);
    while (n > 0) {
      if (lay_->tokens[rank].compare_exchange_weak(n, n - 1, std::memory_order_acq_rel)) {
        if (step_out) *step_out = lay_->token_step[rank].load(std::memory_order_acquire);
        return true;
      }
    }
    if (should_stop() || now_ns() > deadline) return false;
    relax(spins);
  }
}

int64_t Board::token_depth(int rank) const { return lay_->tokens[rank].load(std::memory_order_acquire); }
void Board::set_local_step(int rank, int64_t s) { lay_->local_step[rank].store(s, std::memory_order_release); }
int64_t Board::local_step(int rank) const { return lay_->local_step[rank].load(std::memory_order_acquire); }

// ---- heartbeats ---------------------------------------------------------------------------------------
void Board::heartbeat(int rank) { lay_->heartbeat_ns[rank].store(now_ns(), std::memory_order_release); }
int64_t Board::last_heartbeat_ns(int rank) const { return lay_->heartbeat_ns[rank].load(std::memory_order_acquire); }
void Board::depart(int rank) { lay_->heartbeat_ns[rank].store(kDeparted, std::memory_order_release); }

// ---- async shards -------------------------------------------------------------------------------------
int64_t Board::shard_version(int owner) const { return lay_->shard_version[owner].load(std::memory_order_acquire); }
void Board::set_shard_version(int owner, int64_t version) { lay_->shard_version[owner].store(version, std::memory_order_release); }
int64_t Board::bump_shard_version(int owner) {
  return lay_->shard_version[owner].fetch_add(1, std::memory_order_acq_rel) + 1;
}
bool Board::ring_doorbell(int owner, int sender, int64_t local_step) {
  int64_t expected = 0;
  return lay_->doorbell[owner][sender].compare_exchange_strong(expected, local_step + 1, std::memory_order_acq_rel);
}
int64_t Board::poll_doorbell(int owner, int sender) const {
  return lay_->doorbell[owner][sender].load(std::memory_order_acquire) - 1;
}
void Board::clear_doorbell(int owner, int sender) { lay_->doorbell[owner][sender].store(0, std::memory_order_release); }

// ---- checkpoint rendezvous --------------------------------------------------------------------------
void Board::request_save(int64_t iteration) { lay_->save_step.store(iteration, std::memory_order_release); }
int64_t Board::save_step() const { return lay_->save_step.load(std::memory_order_acquire); }

// ---- update log ----------------------------------------------------------------------------------------
namespace {
constexpr uint64_t kFrozenBit = uint64_t(1) << 31;
inline int64_t open_u(uint64_t w) { return int64_t(w >> 32); }
inline int open_count(uint64_t w) { return int((w >> 16) & 0x7fff); }
inline uint32_t open_mask(uint64_t w) { return uint32_t(w & 0xffff); }
inline uint64_t make_open(int64_t u, int count, uint32_t mask, bool frozen) {
  return (uint64_t(u) << 32) | (frozen ? kFrozenBit : 0) | (uint64_t(count & 0x7fff) << 16) | (mask & 0xffff);
}
inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
}  // namespace

void Board::agg_reset(int64_t u0) {
  for (auto& e : lay_->agg_hist) {
    e.tag.store(0, std::memory_order_relaxed);
    for (auto& r : e.ready) r.store(0, std::memory_order_relaxed);
  }
  for (auto& a : lay_->agg_applied) a.store(u0, std::memory_order_relaxed);
  lay_->agg_abort.store(0, std::memory_order_relaxed);
  lay_->agg_dead.store(0, std::memory_order_relaxed);
  lay_->agg_open.store(make_open(u0, 0, 0, false), std::memory_order_release);
}

// Publishes the record of update u (contributors, owners, scale) and opens u+1. Called by exactly
// one thread: the one whose compare-exchange set the frozen bit.
static void publish_closed(BoardLayout* lay, int64_t u, int count, uint32_t mask, uint32_t owners, float scale) {
  BoardAggEntry& h = lay->agg_hist[u % kAggRing];
  h.tag.store(0, std::memory_order_release);
  for (auto& r : h.ready) r.store(0, std::memory_order_relaxed);
  h.owners.store(owners, std::memory_order_relaxed);
  h.scale_bits.store(f2u(scale), std::memory_order_relaxed);
  h.tag.store((uint64_t(u + 1) << 32) | (uint64_t(count & 0xffff) << 16) | (mask & 0xffff), std::memory_order_release);
  lay->agg_open.store(make_open(u + 1, 0, 0, false), std::memory_order_release);
}

Board::AggArrival Board::agg_arrive(int64_t e, int rank, int add, int quorum, int abort_at, float scale, bool any_update) {
  if (rank < 0 || rank >= kAggMaxRanks) throw std::runtime_error("Board: the update log serves at most 16 replicas");
  AggArrival r{0, 0, 0, 0};
  uint64_t w = lay_->agg_open.load(std::memory_order_acquire);
  for (int spins = 0;; ++spins) {
    const int64_t u = open_u(w);
    r.open_u = u;
    if (!any_update && u != e) {
      r.status = 1;   // an update other than e is open: the gradient was computed on parameters that are no longer current
      return r;
    }
    if (w & kFrozenBit) {
      if (!any_update) {   // quorum complete, the closer is publishing: too late for e
        r.status = 2;
        return r;
      }
      relax(spins);        // interval mode: wait for the next window to open
      w = lay_->agg_open.load(std::memory_order_acquire);
      continue;
    }
    const int c = open_count(w), c2 = c + add;
    const uint32_t m2 = open_mask(w) | (1u << rank);
    const bool closes = quorum > 0 && c2 >= quorum;
    if (lay_->agg_open.compare_exchange_weak(w, make_open(u, c2, m2, closes), std::memory_order_acq_rel)) {
      r.order = c;
      if (abort_at > 0 && c2 >= abort_at) agg_raise_abort(u + 1);
      if (closes) {
        r.closed = 1;
        publish_closed(lay_, u, c2, m2, m2, scale > 0.f ? scale : 1.f / float(c2));
        r.open_u = u + 1;
      }
      return r;
    }
  }
}

bool Board::agg_take_back(int64_t e, int rank, int count) {
  uint64_t w = lay_->agg_open.load(std::memory_order_acquire);
  for (;;) {
    if (open_u(w) != e || (w & kFrozenBit) || !(open_mask(w) & (1u << rank)) || open_count(w) < count) return false;
    if (lay_->agg_open.compare_exchange_weak(w, make_open(e, open_count(w) - count, open_mask(w) & ~(1u << rank), false),
                                             std::memory_order_acq_rel))
      return true;
  }
}

int64_t Board::agg_close(int owner_rank) {
  uint64_t w = lay_->agg_open.load(std::memory_order_acquire);
  for (;;) {
    if ((w & kFrozenBit) || open_count(w) < 1) return -1;
    if (lay_->agg_open.compare_exchange_weak(w, w | kFrozenBit, std::memory_order_acq_rel)) break;
  }
  const int64_t u = open_u(w);
  publish_closed(lay_, u, open_count(w), open_mask(w), 1u << owner_rank, 1.f / float(open_count(w)));
  return u;
}

int64_t Board::agg_open_update() const { return open_u(lay_->agg_open.load(std::memory_order_acquire)); }
int Board::agg_open_count() const { return open_count(lay_->agg_open.load(std::memory_order_acquire)); }
uint32_t Board::agg_open_mask() const { return open_mask(lay_->agg_open.load(std::memory_order_acquire)); }

bool Board::agg_entry(int64_t u, int* count, uint32_t* mask, uint32_t* owners, float* scale, bool* recycled) const {
  const BoardAggEntry& h = lay_->agg_hist[u % kAggRing];
  const uint64_t t = h.tag.load(std::memory_order_acquire);
  const int64_t tu = int64_t(t >> 32) - 1;
  if (recycled) *recycled = tu > u;
  if (tu != u) return false;
  const uint32_t ow = h.owners.load(std::memory_order_relaxed);
  const uint32_t sb = h.scale_bits.load(std::memory_order_relaxed);
  if (h.tag.load(std::memory_order_acquire) != t) {   // recycled while reading
    if (recycled) *recycled = true;
    return false;
  }
  if (count) *count = int((t >> 16) & 0xffff);
  if (mask) *mask = uint32_t(t & 0xffff);
  if (owners) *owners = ow;
  if (scale) *scale = u2f(sb);
  return true;
}

bool Board::agg_wait_closed(int64_t u, double timeout_s) const {
  const int64_t deadline = now_ns() + int64_t(timeout_s * 1e9);
  for (int spins = 0;; ++spins) {
    bool rec = false;
    if (agg_entry(u, nullptr, nullptr, nullptr, nullptr, &rec) || rec) return true;
    if (should_stop() || now_ns() > deadline) return false;
    relax(spins);
  }
}

void Board::agg_publish_ready(int64_t u, int owner) {
  lay_->agg_hist[u % kAggRing].ready[owner].store(u + 1, std::memory_order_release);
}
bool Board::agg_is_ready(int64_t u, int owner) const {
  return lay_->agg_hist[u % kAggRing].ready[owner].load(std::memory_order_acquire) >= u + 1;
}
bool Board::agg_wait_ready(int64_t u, uint32_t owners, double timeout_s) const {
  const int64_t deadline = now_ns() + int64_t(timeout_s * 1e9);
  for (int spins = 0;; ++spins) {
    bool all = true;
    for (int p = 0; p < kAggMaxRanks; ++p)
      if ((owners >> p) & 1u) all = all && agg_is_ready(u, p);
    if (all) return true;
    if (should_stop() || now_ns() > deadline) return false;
    relax(spins);
  }
}

void Board::agg_set_applied(int rank, int64_t version) { lay_->agg_applied[rank].store(version, std::memory_order_release); }
int64_t Board::agg_applied(int rank) const { return lay_->agg_applied[rank].load(std::memory_order_acquire); }
int64_t Board::agg_min_applied() const {
  const uint32_t dead = lay_->agg_dead.load(std::memory_order_acquire);
  int64_t m = std::numeric_limits<int64_t>::max();
  for (int r = 0; r < lay_->world && r < kAggMaxRanks; ++r)
    if (!((dead >> r) & 1u)) m = std::min(m, agg_applied(r));
  return m;
}
bool Board::agg_wait_min_applied(int64_t version, double timeout_s) const {
  const int64_t deadline = now_ns() + int64_t(timeout_s * 1e9);
  for (int spins = 0; agg_min_applied() < version; ++spins) {
    if (should_stop() || now_ns() > deadline) return false;
    relax(spins);
  }
  return true;
}
int64_t Board::agg_abort() const { return lay_->agg_abort.load(std::memory_order_acquire); }
void Board::agg_raise_abort(int64_t value) {
  int64_t cur = lay_->agg_abort.load(std::memory_order_relaxed);
  while (cur < value && !lay_->agg_abort.compare_exchange_weak(cur, value, std::memory_order_acq_rel)) {
  }
}
void Board::agg_mark_dead(int rank) { lay_->agg_dead.fetch_or(1u << rank, std::memory_order_acq_rel); }
uint32_t Board::agg_dead() const { return lay_->agg_dead.load(std::memory_order_acquire); }

// ---- telemetry -------------------------------------------------------------------------------------------
void Board::post_times(int rank, int64_t iter, int64_t t_dequeue_ns, int64_t t_finish_ns) {
  BoardTimeRec& r = lay_->times[rank][iter % kBoardTimeRing];
  r.iter.store(-1, std::memory_order_release);
  r.t_dequeue_ns.store(t_dequeue_ns, std::memory_order_relaxed);
  r.t_finish_ns.store(t_finish_ns, std::memory_order_relaxed);
  r.iter.store(iter, std::memory_order_release);
}

bool Board::get_times(int rank, int64_t iter, int64_t* t_dequeue_ns, int64_t* t_finish_ns) const {
  const BoardTimeRec& r = lay_->times[rank][iter % kBoardTimeRing];
  if (r.iter.load(std::memory_order_acquire) != iter) return false;
  *t_dequeue_ns = r.t_dequeue_ns.load(std::memory_order_relaxed);
  *t_finish_ns = r.t_finish_ns.load(std::memory_order_relaxed);
  return r.iter.load(std::memory_order_acquire) == iter;
}

}  // namespace dtf
