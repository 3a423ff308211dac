// Device-side synchronisation primitives shared by the hand-written communication kernels
// (update_log.hip, async_shards.hip): control words in fine-grained, peer-mapped memory - one
// 64-bit word per 128-byte line - read and written with system-scope atomics, and the "leader
// step", a grid rendezvous that carries a verdict.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dtf {
namespace dsync {

using u64 = unsigned long long;

constexpr int kWordStride = 16;   // 64-bit words per control word: one 128-byte line each
constexpr u64 kVerdictError = ~0ull;

__device__ __forceinline__ u64* W(u64* ctl, int word) { return ctl + (size_t)word * kWordStride; }
__device__ __forceinline__ u64 ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_acq(const u64* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_rel(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool cas(u64* p, u64& expected, u64 desired) {
  return __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void acquire_fence() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(8); }

// Where one launch keeps its rendezvous state (all in the launching replica's own control block).
struct LeaderCtx {
  u64* counter;            // reset to 0 before the launch; every block adds 1 per step
  u64* dec_tag;            // 4 tag words (kWordStride apart): seq<<16 | step number
  u64* dec_val;            // 4 verdict words
  u64 seq;                 // launch sequence number: tags of earlier launches never match
  long long timeout_ticks; // 100 MHz wall-clock ticks the leader waits for the blocks
  int nsync = 0;
};

// Grid rendezvous with a verdict: every block reports in, block 0 waits for all of them, its thread 0
// runs `fn` (the single-thread protocol action) and publishes what it returns; every thread of every
// block returns that value, which is what keeps the blocks' control flow identical. The data a block
// wrote before the step is visible system-wide before anything `fn` publishes. Returns kVerdictError
// when the blocks did not all report in within the timeout (they must be co-resident) - `fn` is then
// not run - or when a block gave up waiting for the leader.
template <typename F>
__device__ __forceinline__ u64 leader_step(LeaderCtx& c, u64* s_payload, F&& fn) {
  __threadfence_system();
  __syncthreads();
  ++c.nsync;
  if (threadIdx.x == 0) {
    const u64 tag = (c.seq << 16) | (u64)c.nsync;
    __hip_atomic_fetch_add(c.counter, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    u64 payload = kVerdictError;
    u64* tg = c.dec_tag + (size_t)(c.nsync & 3) * kWordStride;
    u64* vl = c.dec_val + (size_t)(c.nsync & 3) * kWordStride;
    if (blockIdx.x == 0) {
      const u64 want = (u64)c.nsync * gridDim.x;
      bool ok = true;
      while (__hip_atomic_load(c.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (wall_clock64() - t0 > c.timeout_ticks) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (ok) payload = fn();
      st(vl, payload);
      __hip_atomic_store(tg, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      bool ok = true;
      // the leader's own waits may take several timeouts before it gives up
      while (__hip_atomic_load(tg, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != tag) {
        if (wall_clock64() - t0 > 4 * c.timeout_ticks) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (ok) payload = ld(vl);
    }
    *s_payload = payload;
  }
  __syncthreads();
  return *s_payload;
}

}  // namespace dsync
}  // namespace dtf
