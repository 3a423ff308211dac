// Launch dispatcher and recorded step programs.
//
// Every kernel in csrc/kernels takes exactly one POD argument struct by value. Host launchers
// call dtf::launch(); when a Program is recording on the calling thread the launch is appended
// to that program instead of being issued, so a whole training step (forward, backward,
// optimizer) becomes a flat list of launches that can be (a) replayed from C++ with no Python
// in the loop, (b) split into segments with collectives issued between them, or (c) captured
// once into a hipGraph and replayed with a single host call.
//
// This plays the role of the TensorFlow graph executor the reference relies on
// (reference: DistributedLeNet/src/distributed_train.py:312 `sess.run([train_op, global_step])`).
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace dtf {

#define DTF_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(_e)); \
  } while (0)

// Short-circuit reference carried by the kernels of a training step (kernel argument structs
// with a member `abort` get it from the launch dispatcher): a kernel whose `stop` word has risen
// above the replica's `version` word returns at once, so once the update a replica works for is
// complete, the rest of its forward / backward pass costs launch latency only and leaves zeros in
// the gradients it did not reach (reference: every backward op wrapped in a stop check,
// DistributedLeNet/src/short_circuit_gradients.py:487-545). Null pointers (the default) disable it
// at no cost: the pointers are kernel arguments.
struct AbortRef {
  const unsigned long long* stop = nullptr;     // e + 1 once update e has all the gradients it will take
  const unsigned long long* version = nullptr;  // updates this replica has applied
};
// The reference given to kernels launched or recorded from this thread (default: disabled).
AbortRef& current_abort_ref();

namespace detail {
template <typename A>
inline auto set_abort(A& a, int) -> decltype(a.abort, void()) { a.abort = current_abort_ref(); }
template <typename A>
inline void set_abort(A&, long) {}
}  // namespace detail

struct LaunchRecord {
  enum Kind { KERNEL = 0, MEMSET = 1, MEMCPY = 2, SYNC = 3 } kind = KERNEL;
  std::string name;
  // Lane the record is issued on: lane 0 is the stream handed to run()/replay(), higher lanes
  // are side streams owned by the program. SYNC makes `lane` wait for everything recorded so
  // far on `src_lane` (an event recorded on one stream and waited for on the other).
  int lane = 0;
  int src_lane = 0;
  int event = -1;
  const void* fn = nullptr;
  dim3 grid, block;
  size_t shmem = 0;
  std::vector<char> args;  // the kernel's POD argument struct
  // memset / memcpy
  void* dst = nullptr;
  const void* src = nullptr;
  int value = 0;
  size_t bytes = 0;
};

class Program {
 public:
  ~Program();
  std::vector<LaunchRecord> records;
  // Segment boundaries: segment i covers records [seg_begin[i], seg_begin[i+1]).
  std::vector<size_t> seg_begin{0};

  void mark_segment() { seg_begin.push_back(records.size()); }
  // Phases ("forward", "backward", "optimizer", ...) for the step timeline: a phase covers the records
  // from this call to the next mark_phase (or the end). They do not affect execution.
  std::vector<std::pair<size_t, std::string>> phases;
  void mark_phase(const std::string& name) { phases.emplace_back(records.size(), name); }
  // One timed launch of a traced run. ts/dur in microseconds; ts is relative to `base` (or to the
  // first launch of the run when base is null). phase = index into `phases`, -1 before the first mark.
  struct TraceEvent {
    std::string name;
    int lane;
    int phase;
    double ts_us, dur_us;
    unsigned blocks, threads;
  };
  // Issues the launches of one segment (or all) like run(), each bracketed by a pair of timing
  // events on its lane's stream, waits for them and returns one event per launch: the per-op step
  // timeline (reference: RunOptions.FULL_TRACE + timeline.Timeline,
  // DistributedLeNet/src/distributed_train.py:300-305,324-328). Slower than run() - the launches are
  // not graph-replayed and the call blocks - so it is for the steps that are being traced only.
  std::vector<TraceEvent> run_traced(hipStream_t stream, int seg, hipEvent_t base) const;
  // Lane that subsequently recorded launches go to (recording thread only).
  void set_lane(int lane);
  int lane() const { return lane_; }
  // Records: `dst_lane` waits for the work recorded so far on `src_lane`.
  void sync_lanes(int src_lane, int dst_lane);
  void push(LaunchRecord&& r) {
    r.lane = lane_;
    records.push_back(std::move(r));
  }
  size_t num_segments() const { return seg_begin.size(); }
  // Issue the launches of one segment (or all with seg < 0) on `stream`.
  void run(hipStream_t stream, int seg = -1) const;
  // Capture segment `seg` (or everything) into a hipGraph and keep the executable.
  void instantiate(hipStream_t stream, int seg = -1);
  // Replay a previously instantiated graph.
  void replay(hipStream_t stream, int seg = -1) const;
  bool instantiated(int seg = -1) const;

 private:
  struct GraphSlot {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
  };
  // slot 0 = whole program, slot i+1 = segment i
  std::vector<GraphSlot> graphs_;
  int lane_ = 0;
  mutable std::vector<hipStream_t> side_;   // side_[i] serves lane i+1
  mutable std::vector<hipEvent_t> events_;
  hipStream_t lane_stream(hipStream_t main, int lane) const;
};

// The program currently recording on this thread (nullptr = launch immediately).
Program*& current_recorder();

void ensure_dynamic_lds(const void* fn, size_t shmem);
void issue(const LaunchRecord& r, hipStream_t stream);

template <typename Args>
inline void launch(const char* name, void (*kernel)(Args), dim3 grid, dim3 block, size_t shmem,
                   hipStream_t stream, const Args& args) {
  static_assert(std::is_trivially_copyable<Args>::value, "kernel args must be POD");
  LaunchRecord r;
  r.kind = LaunchRecord::KERNEL;
  r.name = name;
  r.fn = reinterpret_cast<const void*>(kernel);
  r.grid = grid;
  r.block = block;
  r.shmem = shmem;
  r.args.resize(sizeof(Args));
  Args with_abort = args;
  detail::set_abort(with_abort, 0);
  std::memcpy(r.args.data(), &with_abort, sizeof(Args));
  ensure_dynamic_lds(r.fn, shmem);
  if (Program* p = current_recorder()) {
    p->push(std::move(r));
  } else {
    issue(r, stream);
  }
}

void memset_async(void* dst, int value, size_t bytes, hipStream_t stream);
void memcpy_async(void* dst, const void* src, size_t bytes, hipStream_t stream);

}  // namespace dtf
