#include "program.h"

#include <mutex>
#include <unordered_map>

namespace dtf {

Program*& current_recorder() {
  static thread_local Program* rec = nullptr;
  return rec;
}

AbortRef& current_abort_ref() {
  static thread_local AbortRef ref;
  return ref;
}

void ensure_dynamic_lds(const void* fn, size_t shmem) {
  // Kernels asking for more than the default dynamic LDS window must raise the function's
  // limit once. Remember the largest request per kernel.
  if (shmem <= 48 * 1024) return;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> seen;
  std::lock_guard<std::mutex> lock(mu);
  auto it = seen.find(fn);
  if (it != seen.end() && it->second >= shmem) return;
  DTF_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  seen[fn] = shmem;
}

void issue(const LaunchRecord& r, hipStream_t stream) {
  switch (r.kind) {
    case LaunchRecord::KERNEL: {
      void* argv[1] = {const_cast<char*>(r.args.data())};
      DTF_HIP_CHECK(hipLaunchKernel(r.fn, r.grid, r.block, argv, r.shmem, stream));
      break;
    }
    case LaunchRecord::MEMSET:
      DTF_HIP_CHECK(hipMemsetAsync(r.dst, r.value, r.bytes, stream));
      break;
    case LaunchRecord::MEMCPY:
      DTF_HIP_CHECK(hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyDeviceToDevice, stream));
      break;
    case LaunchRecord::SYNC:
      throw std::runtime_error("lane synchronisation exists only inside a recorded program");
  }
}

void memset_async(void* dst, int value, size_t bytes, hipStream_t stream) {
  LaunchRecord r;
  r.kind = LaunchRecord::MEMSET;
  r.name = "memset";
  r.dst = dst;
  r.value = value;
  r.bytes = bytes;
  if (Program* p = current_recorder())
    p->push(std::move(r));
  else
    issue(r, stream);
}

void memcpy_async(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  LaunchRecord r;
  r.kind = LaunchRecord::MEMCPY;
  r.name = "memcpy";
  r.dst = dst;
  r.src = src;
  r.bytes = bytes;
  if (Program* p = current_recorder())
    p->push(std::move(r));
  else
    issue(r, stream);
}

Program::~Program() {
  for (hipEvent_t e : events_) (void)hipEventDestroy(e);
  for (hipStream_t s : side_) (void)hipStreamDestroy(s);
  for (auto& g : graphs_) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
}

void Program::set_lane(int lane) {
  if (lane < 0 || lane > 7) throw std::runtime_error("Program::set_lane: lane must be 0..7");
  lane_ = lane;
}

void Program::sync_lanes(int src_lane, int dst_lane) {
  if (src_lane == dst_lane) return;
  LaunchRecord r;
  r.kind = LaunchRecord::SYNC;
  r.name = "sync";
  r.lane = dst_lane;
  r.src_lane = src_lane;
  r.event = (int)events_.size();
  events_.push_back(nullptr);
  records.push_back(std::move(r));
}

hipStream_t Program::lane_stream(hipStream_t main, int lane) const {
  if (lane == 0) return main;
  while ((int)side_.size() < lane) {
    hipStream_t s = nullptr;
    DTF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    side_.push_back(s);
  }
  return side_[lane - 1];
}

void Program::run(hipStream_t stream, int seg) const {
  size_t b = 0, e = records.size();
  if (seg >= 0) {
    if ((size_t)seg >= seg_begin.size()) throw std::runtime_error("Program::run: bad segment");
    b = seg_begin[seg];
    e = ((size_t)seg + 1 < seg_begin.size()) ? seg_begin[seg + 1] : records.size();
  }
  for (size_t i = b; i < e; ++i) {
    const LaunchRecord& r = records[i];
    if (r.kind == LaunchRecord::SYNC) {
      hipEvent_t& ev = events_[r.event];
      if (!ev) DTF_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      DTF_HIP_CHECK(hipEventRecord(ev, lane_stream(stream, r.src_lane)));
      DTF_HIP_CHECK(hipStreamWaitEvent(lane_stream(stream, r.lane), ev, 0));
    } else {
      issue(r, lane_stream(stream, r.lane));
    }
  }
}

std::vector<Program::TraceEvent> Program::run_traced(hipStream_t stream, int seg, hipEvent_t base) const {
  size_t b = 0, e = records.size();
  if (seg >= 0) {
    if ((size_t)seg >= seg_begin.size()) throw std::runtime_error("Program::run_traced: bad segment");
    b = seg_begin[seg];
    e = ((size_t)seg + 1 < seg_begin.size()) ? seg_begin[seg + 1] : records.size();
  }
  struct Pair {
    size_t rec;
    hipEvent_t t0, t1;
  };
  std::vector<Pair> pairs;
  pairs.reserve(e - b);
  auto cleanup = [&pairs]() {
    for (auto& p : pairs) {
      if (p.t0) (void)hipEventDestroy(p.t0);
      if (p.t1) (void)hipEventDestroy(p.t1);
    }
  };
  std::vector<TraceEvent> out;
  try {
    for (size_t i = b; i < e; ++i) {
      const LaunchRecord& r = records[i];
      if (r.kind == LaunchRecord::SYNC) {
        hipEvent_t& ev = events_[r.event];
        if (!ev) DTF_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        DTF_HIP_CHECK(hipEventRecord(ev, lane_stream(stream, r.src_lane)));
        DTF_HIP_CHECK(hipStreamWaitEvent(lane_stream(stream, r.lane), ev, 0));
        continue;
      }
      hipStream_t s = lane_stream(stream, r.lane);
      Pair p{i, nullptr, nullptr};
      DTF_HIP_CHECK(hipEventCreate(&p.t0));
      pairs.push_back(p);
      DTF_HIP_CHECK(hipEventCreate(&pairs.back().t1));
      DTF_HIP_CHECK(hipEventRecord(pairs.back().t0, s));
      issue(r, s);
      DTF_HIP_CHECK(hipEventRecord(pairs.back().t1, s));
    }
    // side lanes are joined by the program's own SYNC records; waiting for each end event covers all
    for (auto& p : pairs) DTF_HIP_CHECK(hipEventSynchronize(p.t1));
    hipEvent_t origin = base ? base : (pairs.empty() ? nullptr : pairs.front().t0);
    for (auto& p : pairs) {
      const LaunchRecord& r = records[p.rec];
      float ts = 0.f, dur = 0.f;
      DTF_HIP_CHECK(hipEventElapsedTime(&ts, origin, p.t0));
      DTF_HIP_CHECK(hipEventElapsedTime(&dur, p.t0, p.t1));
      int phase = -1;
      for (size_t k = 0; k < phases.size(); ++k)
        if (phases[k].first <= p.rec) phase = (int)k;
      TraceEvent ev;
      ev.name = r.name;
      ev.lane = r.lane;
      ev.phase = phase;
      ev.ts_us = (double)ts * 1e3;
      ev.dur_us = (double)dur * 1e3;
      ev.blocks = r.kind == LaunchRecord::KERNEL ? r.grid.x * r.grid.y * r.grid.z : 0;
      ev.threads = r.kind == LaunchRecord::KERNEL ? r.block.x * r.block.y * r.block.z : 0;
      out.push_back(std::move(ev));
    }
  } catch (...) {
    cleanup();
    throw;
  }
  cleanup();
  return out;
}

void Program::instantiate(hipStream_t stream, int seg) {
  size_t slot = (size_t)(seg + 1);
  if (graphs_.size() <= slot) graphs_.resize(slot + 1);
  GraphSlot& g = graphs_[slot];
  if (g.exec) {
    (void)hipGraphExecDestroy(g.exec);
    g.exec = nullptr;
  }
  if (g.graph) {
    (void)hipGraphDestroy(g.graph);
    g.graph = nullptr;
  }
  // The capture always runs on a stream of its own, never on the caller's: other threads may be
  // polling events recorded on the caller's stream (the RCCL process group's watchdog does, for
  // every collective issued there), and querying an event whose stream is capturing is an error
  // that ends the process. A captured graph is not tied to the stream it was captured on.
  (void)stream;
  hipStream_t cap = nullptr;
  const bool own = true;
  DTF_HIP_CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  DTF_HIP_CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  try {
    run(cap, seg);
  } catch (...) {
    hipGraph_t tmp = nullptr;
    (void)hipStreamEndCapture(cap, &tmp);
    if (tmp) (void)hipGraphDestroy(tmp);
    if (own) (void)hipStreamDestroy(cap);
    throw;
  }
  DTF_HIP_CHECK(hipStreamEndCapture(cap, &g.graph));
  DTF_HIP_CHECK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
  if (own) DTF_HIP_CHECK(hipStreamDestroy(cap));
}

bool Program::instantiated(int seg) const {
  size_t slot = (size_t)(seg + 1);
  return slot < graphs_.size() && graphs_[slot].exec != nullptr;
}

void Program::replay(hipStream_t stream, int seg) const {
  size_t slot = (size_t)(seg + 1);
  if (slot >= graphs_.size() || !graphs_[slot].exec)
    throw std::runtime_error("Program::replay: segment not instantiated");
  DTF_HIP_CHECK(hipGraphLaunch(graphs_[slot].exec, stream));
}

}  // namespace dtf
