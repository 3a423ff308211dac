#include "board.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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

Board::Board(const std::string& name, int world, bool create) : name_(name), owner_(create) {
  if (world < 1 || world > kBoardMaxRanks) throw std::runtime_error("Board: world size out of range");
  const std::string path = name[0] == '/' ? name : "/" + name;
  name_ = path;
  if (create) {
    ::shm_unlink(path.c_str());
    fd_ = ::shm_open(path.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd_ < 0) throw std::runtime_error("Board: shm_open(create) failed for " + path);
    if (::ftruncate(fd_, sizeof(BoardLayout)) != 0) throw std::runtime_error("Board: ftruncate failed");
  } else {
    // The creator may still be setting up: retry briefly.
    for (int i = 0; i < 2000 && fd_ < 0; ++i) {
      fd_ = ::shm_open(path.c_str(), O_RDWR, 0600);
      if (fd_ < 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    if (fd_ < 0) throw std::runtime_error("Board: shm_open(attach) failed for " + path);
    struct stat st;
    for (int i = 0; i < 2000; ++i) {
      if (::fstat(fd_, &st) == 0 && (size_t)st.st_size >= sizeof(BoardLayout)) break;
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
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
    std::atomic_thread_fence(std::memory_order_release);
    lay_->magic = kMagic;
  } else {
    for (int i = 0; i < 4000 && lay_->magic != kMagic; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    if (lay_->magic != kMagic) throw std::runtime_error("Board: segment was never initialised");
    if (lay_->world != world) throw std::runtime_error("Board: world size mismatch with the creator");
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

// ---- async shards -------------------------------------------------------------------------------------
int64_t Board::shard_version(int owner) const { return lay_->shard_version[owner].load(std::memory_order_acquire); }
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
