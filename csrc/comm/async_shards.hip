// See async_shards.h.
#include "async_shards.h"

#include <stdexcept>

#include "../kernels/optim_math.h"
#include "device_sync.h"

namespace dtf {
namespace {

using namespace dsync;

constexpr u64 kExit = 1000, kNone = 1001;

__device__ __forceinline__ void bump(u64* own, int which, u64 by = 1) { st(W(own, AW_STATS + which), ld(W(own, AW_STATS + which)) + by); }

// ---- pull ---------------------------------------------------------------------------------------------
// The version words are read by a one-thread-per-owner kernel launched BEFORE the copy kernel:
// kernels of one stream run in order, so every data load of every block of the copy comes after
// the version reads, and the gradient computed on these parameters is tagged with a version that
// is at most as new as any value that was copied: staleness is never under-estimated. (Read inside
// the copy kernel by block 0 only, a block scheduled ahead of block 0 could copy older values than
// the recorded version.) A pull does not exclude the owner's service sweep: see async_shards.h.
__global__ void async_pull_begin_kernel(AsyncArgs a) {
  if (blockIdx.x != 0 || (int)threadIdx.x >= a.world) return;
  u64* own = a.ctl[a.rank];
  const u64 v = ld_acq(W(a.ctl[threadIdx.x], AW_VERSION));
  st(W(own, AW_PULLED + threadIdx.x), v);
  if (a.status) st(a.status + threadIdx.x, v);
}

__global__ __launch_bounds__(256) void async_pull_kernel(AsyncArgs a) {
  const long long n4 = a.shard_floats >> 2;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  acquire_fence();
  floatx4* dst = reinterpret_cast<floatx4*>(a.params);
  for (int k = 0; k < a.world; ++k) {
    const int o = (a.rank + k) % a.world;   // every replica starts at a different owner
    const floatx4* src = reinterpret_cast<const floatx4*>(a.shard[o]);
    for (long long i = gtid; i < n4; i += gstride) dst[(long long)o * n4 + i] = __builtin_nontemporal_load(src + i);
  }
}

// ---- push ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void async_push_kernel(AsyncArgs a) {
  __shared__ u64 s_payload;
  u64* own = a.ctl[a.rank];
  LeaderCtx lc{W(own, AW_WRK_COUNTER), W(own, AW_WRK_TAG), W(own, AW_WRK_VAL), ld(W(own, AW_WRK_SEQ)), a.timeout_ticks};
  const long long n4 = a.shard_floats >> 2;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  // which owners have consumed this replica's previous push (bounded wait for the others)
  const u64 free_mask = leader_step(lc, &s_payload, [&]() -> u64 {
    u64 mask = 0;
    const long long t0 = wall_clock64();
    for (;;) {
      for (int o = 0; o < a.world; ++o)
        if (!((mask >> o) & 1) && ld(W(a.ctl[o], AW_DOORBELL + a.rank)) == 0) mask |= 1ull << o;
      if (mask == (1ull << a.world) - 1 || wall_clock64() - t0 > a.push_wait_ticks) break;
      nap();
    }
    acquire_fence();
    return mask;
  });
  if (free_mask == kVerdictError) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st(W(own, AW_ERROR), 2);
    return;
  }
  const floatx4* g4 = reinterpret_cast<const floatx4*>(a.grads);
  for (int k = 0; k < a.world; ++k) {
    const int o = (a.rank + k) % a.world;
    if (!((free_mask >> o) & 1)) continue;
    floatx4* dst = reinterpret_cast<floatx4*>(a.inbox[o]) + (long long)a.rank * n4;
    for (long long i = gtid; i < n4; i += gstride) dst[i] = g4[(long long)o * n4 + i];
  }
  leader_step(lc, &s_payload, [&]() -> u64 {
    for (int o = 0; o < a.world; ++o) {
      if ((free_mask >> o) & 1) {
        st_rel(W(a.ctl[o], AW_DOORBELL + a.rank), ld(W(own, AW_PULLED + o)) + 1);
        bump(own, 2);
      } else {
        bump(own, 3);   // the owner never consumed the previous push: this slice is dropped
      }
    }
    if (a.status)
      for (int k = 0; k < 4 + kAsyncHistBins; ++k) st(a.status + 8 + k, ld(W(own, AW_STATS + k)));
    return 0;
  });
}

__global__ void async_begin_kernel(AsyncArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u64* own = a.ctl[a.rank];
  st(W(own, AW_WRK_SEQ), ld(W(own, AW_WRK_SEQ)) + 1);
  st(W(own, AW_WRK_COUNTER), 0);
}

// ---- service --------------------------------------------------------------------------------------------
__global__ void async_service_begin_kernel(AsyncArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u64* own = a.ctl[a.rank];
  st(W(own, AW_SVC_SEQ), ld(W(own, AW_SVC_SEQ)) + 1);
  st(W(own, AW_SVC_COUNTER), 0);
}

__global__ __launch_bounds__(256) void async_service_kernel(AsyncArgs a) {
  __shared__ u64 s_payload;
  u64* own = a.ctl[a.rank];
  LeaderCtx lc{W(own, AW_SVC_COUNTER), W(own, AW_SVC_TAG), W(own, AW_SVC_VAL), ld(W(own, AW_SVC_SEQ)), a.timeout_ticks};
  const long long n4 = a.shard_floats >> 2;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  int next_sender = 0;          // leader only: round-robin over the senders
  for (;;) {
    // verdict: sender << 32 | (pushed version + 1), or kExit
    const u64 verdict = leader_step(lc, &s_payload, [&]() -> u64 {
      for (;;) {
        for (int k = 0; k < a.world; ++k) {
          const int s = (next_sender + k) % a.world;
          const u64 bell = ld(W(own, AW_DOORBELL + s));
          if (bell) {
            acquire_fence();
            next_sender = (s + 1) % a.world;
            const long long version = (long long)ld(W(own, AW_VERSION));
            const long long stale = version - (long long)(bell - 1);
            if (stale <= a.tau) return ((u64)s << 32) | bell;
            bump(own, 1);
            st_rel(W(own, AW_DOORBELL + s), 0);   // dropped: the inbox slot is free again
          }
        }
        return kExit;   // nothing (left) to apply: a sweep never stays resident
      }
    });
    if (verdict == kExit || verdict == kVerdictError) break;
    const int s = (int)(verdict >> 32);
    const long long version = (long long)ld(W(own, AW_VERSION));   // only this kernel writes it, behind a rendezvous
    acquire_fence();
    const floatx4* g4 = reinterpret_cast<const floatx4*>(a.inbox[a.rank]) + (long long)s * n4;
    floatx4* p4 = reinterpret_cast<floatx4*>(a.shard[a.rank]);
    floatx4* m4 = reinterpret_cast<floatx4*>(a.m);   // the own shard's slots (peer-mapped, read by the chief at a save)
    const long long t = version + 1;
    const double lr = (double)a.lr0 * pow((double)a.decay, (double)(version / (a.decay_steps > 0 ? a.decay_steps : 1)));
    if (a.opt == 0) {
      const float lr_t = (float)(lr * sqrt(1.0 - pow((double)a.beta2, (double)t)) / (1.0 - pow((double)a.beta1, (double)t)));
      floatx4* v4 = reinterpret_cast<floatx4*>(a.v);
      for (long long i = gtid; i < n4; i += gstride) {
        floatx4 p = p4[i], m = m4[i], v = v4[i];
        adam_update4(p, m, v, g4[i], 1.f, lr_t, a.beta1, a.beta2, a.eps);
        p4[i] = p;
        m4[i] = m;
        v4[i] = v;
      }
    } else {
      for (long long i = gtid; i < n4; i += gstride) {
        floatx4 p = p4[i], m = m4[i];
        momentum_update4(p, m, g4[i], 1.f, (float)lr, a.momentum, a.weight_decay);
        p4[i] = p;
        m4[i] = m;
      }
    }
    const u64 done = leader_step(lc, &s_payload, [&]() -> u64 {
      const long long stale = version - (long long)((verdict & 0xffffffffull) - 1);
      st_rel(W(own, AW_VERSION), (u64)(version + 1));
      bump(own, 0);
      bump(own, 4 + (int)(stale < kAsyncHistBins - 1 ? (stale < 0 ? 0 : stale) : kAsyncHistBins - 1));
      st_rel(W(own, AW_DOORBELL + s), 0);
      return 0;
    });
    if (done == kVerdictError) break;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.status) {
    for (int k = 0; k < 4 + kAsyncHistBins; ++k) st(a.status + 8 + k, ld(W(own, AW_STATS + k)));
    st_rel(a.status + 8 + 4 + kAsyncHistBins, ld(W(own, AW_VERSION)));
  }
}

void validate(const AsyncArgs& a, int blocks) {
  if (a.world < 1 || a.world > kAsyncMaxPeers || a.rank < 0 || a.rank >= a.world) throw std::runtime_error("async shards: bad rank/world");
  if (a.shard_floats <= 0 || (a.shard_floats & 3)) throw std::runtime_error("async shards: shard size must be a positive multiple of 4 floats");
  if (blocks < 1 || blocks > kAsyncMaxBlocks) throw std::runtime_error("async shards: 1..64 blocks (all must be co-resident)");
  for (int p = 0; p < a.world; ++p)
    if (!a.ctl[p] || !a.shard[p] || !a.inbox[p]) throw std::runtime_error("async shards: unmapped peer buffer");
  if (a.timeout_ticks <= 0) throw std::runtime_error("async shards: timeout must be positive");
}

}  // namespace

void async_pull(const AsyncArgs& a, int blocks, hipStream_t stream) {
  validate(a, blocks);
  if (!a.params) throw std::runtime_error("async shards: null parameter arena");
  launch("async_pull_begin", async_pull_begin_kernel, dim3(1), dim3(64), 0, stream, a);
  launch("async_pull", async_pull_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

void async_push(const AsyncArgs& a, int blocks, hipStream_t stream) {
  validate(a, blocks);
  if (!a.grads) throw std::runtime_error("async shards: null gradient");
  launch("async_begin", async_begin_kernel, dim3(1), dim3(64), 0, stream, a);
  launch("async_push", async_push_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

void async_service(const AsyncArgs& a, int blocks, hipStream_t stream) {
  validate(a, blocks);
  if (!a.m || (a.opt == 0 && !a.v)) throw std::runtime_error("async shards: null optimizer slot");
  launch("async_service_begin", async_service_begin_kernel, dim3(1), dim3(64), 0, stream, a);
  launch("async_service", async_service_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

}  // namespace dtf
