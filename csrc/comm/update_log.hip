// See update_log.h for the protocol. Structure of one ulog_step launch (G blocks):
//
//   stage            all blocks copy (or add) the gradient into the own staging buffer
//   leader step 1    block 0, thread 0: arrival compare-exchange on the coordinator's word, closes the
//                    update if it completed the quorum, decides what happens next
//   while APPLY:     (update u = this replica's version)
//     reduce         an owner sums its shard over the contributors' staging buffers
//     leader step    publish the shard, wait for all owners' shards
//     gather         every block reads the owners' shards and applies the optimizer to its slice
//     leader step    version = u + 1, pushed to every peer; decide what happens next
//
// A "leader step" is a grid rendezvous with a verdict: every block reports in on a counter, block 0
// waits for all of them, runs the single-thread protocol action and publishes a tagged verdict that
// the other blocks wait for; verdicts are what keeps the blocks' control flow identical. All blocks
// of the launch must be co-resident (at most kUlogMaxBlocks of 256 threads).
#include "update_log.h"

#include <stdexcept>

#include "../kernels/optim_math.h"
#include "device_sync.h"

namespace dtf {
namespace {

using namespace dsync;
static_assert(kWordStride == kUlogWordStride, "control blocks use one word per 128-byte line");

constexpr u64 kFrozen = 1ull << 31;
constexpr u64 kActDone = 0, kActApply = 1, kActError = kVerdictError, kActOk = 3;

enum UlogError : int {
  UE_BACKPRESSURE = 1,   // the slowest live replica did not catch up in time
  UE_GRID = 2,           // the blocks of a launch did not all report in (not co-resident?)
  UE_RECYCLED = 3,       // this replica fell more than kUlogRing updates behind
  UE_CLOSE = 4,          // the update this replica contributed to never closed
  UE_READY = 5,          // an owner never published its reduced shard
  UE_FROZEN = 6,         // a window stayed frozen (its closer never opened the next one)
};

__device__ __forceinline__ long long open_u(u64 w) { return (long long)(w >> 32); }
__device__ __forceinline__ int open_count(u64 w) { return (int)((w >> 16) & 0x7fff); }
__device__ __forceinline__ unsigned open_mask(u64 w) { return (unsigned)(w & 0xffff); }
__device__ __forceinline__ u64 make_open(long long u, int count, unsigned mask, bool frozen) {
  return ((u64)u << 32) | (frozen ? kFrozen : 0ull) | ((u64)(count & 0x7fff) << 16) | (u64)(mask & 0xffff);
}

struct Rec {
  int count;
  unsigned mask, owners;
  float scale;
  bool closer_reduces;
};

// 1: update u is closed (record in r); 0: not closed yet; -1: the slot already serves a later update
__device__ __forceinline__ int read_hist(u64* own, long long u, Rec& r) {
  const int slot = (int)(u % kUlogRing);
  // relaxed first (this is what waits poll): an acquire at system scope invalidates cache lines
  const u64 t = ld(W(own, UW_HIST_TAG + slot));
  const long long tu = (long long)(t >> 32) - 1;
  if (tu > u) return -1;
  if (tu != u) return 0;
  acquire_fence();
  const u64 aux = ld(W(own, UW_HIST_AUX + slot));
  if (ld_acq(W(own, UW_HIST_TAG + slot)) != t) return -1;
  r.count = (int)((t >> 16) & 0xffff);
  r.mask = (unsigned)(t & 0xffff);
  r.owners = (unsigned)((aux >> 32) & 0xffff);
  r.scale = __uint_as_float((unsigned)(aux & 0xffffffffull));
  r.closer_reduces = (aux >> 63) != 0;
  return 1;
}

// A ready word is (u + 1) in its low bits; a stand-in for a dead owner adds its own rank + 1 above
// kReadyHolderShift: that is whose ring holds the shard.
constexpr int kReadyHolderShift = 48;
constexpr u64 kReadyUpdateMask = (1ull << kReadyHolderShift) - 1;

// Owners of update u whose reduced shard is not published yet.
__device__ __forceinline__ unsigned unready_owners(u64* own, long long u, unsigned owners) {
  const int slot = (int)(u % kUlogRing);
  unsigned missing = 0;
  for (int p = 0; p < kUlogMaxPeers; ++p)
    if (((owners >> p) & 1u) && (ld(W(own, UW_READY + slot * kUlogMaxPeers + p)) & kReadyUpdateMask) < (u64)(u + 1)) missing |= 1u << p;
  return missing;
}

__device__ __forceinline__ bool all_ready(u64* own, long long u, unsigned owners) {
  if (unready_owners(own, u, owners)) return false;
  acquire_fence();
  return true;
}

__device__ __forceinline__ long long min_applied(const UlogArgs& a) {
  u64* own = a.ctl[a.rank];
  const unsigned dead = (unsigned)ld(W(own, UW_DEAD));
  long long m = 0x7fffffffffffffffll;
  for (int p = 0; p < a.world; ++p)
    if (!((dead >> p) & 1u)) {
      const long long v = (long long)ld(W(own, UW_APPLIED + p));
      m = v < m ? v : m;
    }
  return m;
}

__device__ __forceinline__ void raise_error(u64* own, int code) {
  if (ld(W(own, UW_ERROR)) == 0) st(W(own, UW_ERROR), (u64)code);
}

// Publishes the record of update u on every replica's block and opens u + 1. Called by the thread
// whose compare-exchange set the frozen bit - and, should that replica die between the two, by the
// contributor that stands in for it (republish_if_closer_died): both write the same record, and the
// next update is opened by a compare-exchange from the frozen word, so it is opened exactly once
// (a plain store by a closer that was only slow could wipe out arrivals for u + 1).
__device__ void publish_closed(const UlogArgs& a, long long u, int count, unsigned mask, unsigned owners, float scale,
                               bool closer_reduces) {
  const int slot = (int)(u % kUlogRing);
  const u64 aux = ((u64)(closer_reduces ? 1 : 0) << 63) | ((u64)(owners & 0xffff) << 32) | (u64)__float_as_uint(scale);
  const u64 tag = ((u64)(u + 1) << 32) | ((u64)(count & 0xffff) << 16) | (u64)(mask & 0xffff);
  const unsigned dead = (unsigned)ld(W(a.ctl[a.rank], UW_DEAD));   // a dead replica's block may be gone
  for (int p = 0; p < a.world; ++p)
    if (!((dead >> p) & 1u)) st(W(a.ctl[p], UW_HIST_AUX + slot), aux);
  for (int p = 0; p < a.world; ++p)
    if (!((dead >> p) & 1u)) st_rel(W(a.ctl[p], UW_HIST_TAG + slot), tag);
  u64 frozen_word = make_open(u, count, mask, true);
  cas(W(a.ctl[0], UW_OPEN), frozen_word, make_open(u + 1, 0, 0, false));
}

// The replica whose arrival completed the quorum of update u froze the open word and died before it
// published the closed record: the word still says who contributed and how many gradients there
// are, which is all the record holds. Of the contributors that are waiting for that record, the
// lowest-ranked live one publishes it in the closer's place. Only contributors do: they carry the
// scale of the update in their launch arguments. Conditions: the word is frozen at u, a contributor
// of u is marked dead (the watchdog takes seconds to say so; a live closer needs microseconds), the
// coordinator is alive (its block holds the word), and the state has lasted `grace` ticks.
__device__ bool republish_if_closer_died(const UlogArgs& a, u64* own, long long u, long long& frozen_since) {
  if (a.mode != ULOG_CONTRIBUTE) return false;
  const unsigned dead = (unsigned)ld(W(own, UW_DEAD));
  if (!dead || (dead & 1u)) return false;
  const u64 w = ld_acq(W(a.ctl[0], UW_OPEN));
  if (!(w & kFrozen) || open_u(w) != u) {
    frozen_since = 0;
    return false;
  }
  const unsigned mask = open_mask(w), live = mask & ~dead;
  if (!(mask & dead) || !live || (live & (0u - live)) != (1u << a.rank)) return false;
  const long long now = wall_clock64();
  if (!frozen_since) frozen_since = now;
  if (now - frozen_since < 100000) return false;   // 1 ms at 100 MHz
  const int count = open_count(w);
  publish_closed(a, u, count, mask, mask, a.scale > 0.f ? a.scale : 1.f / (float)count, false);
  return true;
}

__device__ void raise_abort_everywhere(const UlogArgs& a, long long value) {
  for (int p = 0; p < a.world; ++p)
    __hip_atomic_fetch_max(W(a.ctl[p], UW_ABORT), (u64)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Registers `add` gradients of this replica for update e (any_update: for whatever update is open).
// 0 accepted, 1 stale (another update is open), 2 late (quorum already complete), 3 timed out.
__device__ int arrive(const UlogArgs& a, long long e, int add, int quorum, bool any_update, int& order, bool& closed,
                      long long& open_now) {
  u64* openw = W(a.ctl[0], UW_OPEN);
  u64 w = ld_acq(openw);
  const long long t0 = wall_clock64();
  order = 0;
  closed = false;
  for (;;) {
    const long long u = open_u(w);
    open_now = u;
    if (!any_update && u != e) return 1;
    if (w & kFrozen) {
      if (!any_update) return 2;
      if (wall_clock64() - t0 > a.timeout_ticks) return 3;
      nap();   // interval mode: the closer opens the next window in a moment
      w = ld_acq(openw);
      continue;
    }
    const int c = open_count(w), c2 = c + add;
    const unsigned m2 = open_mask(w) | (1u << a.rank);
    const bool closes = quorum > 0 && c2 >= quorum;
    if (cas(openw, w, make_open(u, c2, m2, closes))) {
      order = c;
      if (a.abort_at > 0 && c2 >= a.abort_at) raise_abort_everywhere(a, u + 1);
      if (closes) {
        closed = true;
        publish_closed(a, u, c2, m2, m2, a.scale > 0.f ? a.scale : 1.f / (float)c2, false);
        open_now = u + 1;
      }
      return 0;
    }
  }
}

// Interval mode: withdraws this replica's `count` staged gradients from the still-open update e.
__device__ bool take_back(const UlogArgs& a, long long e, int count) {
  u64* openw = W(a.ctl[0], UW_OPEN);
  u64 w = ld_acq(openw);
  for (;;) {
    if (open_u(w) != e || (w & kFrozen) || !(open_mask(w) & (1u << a.rank)) || open_count(w) < count) return false;
    if (cas(openw, w, make_open(e, open_count(w) - count, open_mask(w) & ~(1u << a.rank), false))) return true;
  }
}

__device__ __forceinline__ void bump_stat(u64* own, int which, u64 by) { st(W(own, UW_STATS + which), ld(W(own, UW_STATS + which)) + by); }

// ---------------------------------------------------------------------------------------------------
// ulog_begin: one thread. D0 = seq<<32 | count<<8 | withdrawn<<3 | dropped<<2 | stage op (0 none, 1 copy, 2 add).
__global__ void ulog_begin_kernel(UlogArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u64* own = a.ctl[a.rank];
  const u64 seq = ld(W(own, UW_SEQ)) + 1;
  st(W(own, UW_SEQ), seq);
  st(W(own, UW_COUNTER), 0);
  const long long e = (long long)ld(W(own, UW_VERSION));
  u64 op = 0, count = 1, dropped = 0, withdrawn = 0;
  if (ld(W(own, UW_ERROR)) == 0) {
    const long long t0 = wall_clock64();
    if (a.mode == ULOG_CONTRIBUTE) {
      if (e < a.max_version) {
        // the ring slot of update e last served e - ring: every live replica must be past that
        op = 1;
        while (min_applied(a) < e - kUlogRing + 1) {
          if (wall_clock64() - t0 > a.timeout_ticks) {
            raise_error(own, UE_BACKPRESSURE);
            op = 0;
            break;
          }
          nap();
        }
      }
    } else if (a.mode == ULOG_OPEN) {
      op = 1;
      const u64 pend = ld(W(own, UW_PENDING));
      bool keep_pending = false;
      if (pend) {
        const long long pu = (long long)(pend >> 16) - 1;
        const int pc = (int)(pend & 0xffff);
        if (take_back(a, pu, pc)) {
          op = 2;   // window still open and the staged sum withdrawn: nobody reads it, add to it
          count = (u64)(pc + 1);
        } else {
          // the window closed with the staged sum in it: its owner may still be reading. Nobody waits
          // in this mode: a gradient that cannot be staged yet is dropped.
          Rec r;
          const int c = read_hist(own, pu, r);
          if (c == 0 || (c == 1 && !all_ready(own, pu, r.owners))) {
            op = 0;
            dropped = 1;
            keep_pending = true;
          }
        }
      }
      if (op != 0) {
        const long long open_now = open_u(ld_acq(W(a.ctl[0], UW_OPEN)));
        if ((a.max_staleness >= 0 && open_now - e > a.max_staleness) || open_now >= a.max_version ||
            min_applied(a) < open_now - kUlogRing + 1) {
          withdrawn = op == 2 ? 1 : 0;
          op = 0;
          dropped = (open_now < a.max_version || withdrawn) ? 1 : 0;
        }
      }
      if (!keep_pending) st(W(own, UW_PENDING), 0);
    }
  }
  st(W(own, UW_D0), (seq << 32) | (count << 8) | (withdrawn << 3) | (dropped << 2) | op);
}

// ---------------------------------------------------------------------------------------------------
// Leader only: what to do about update u. wait_u: the update this launch must see closed.
__device__ u64 next_action(const UlogArgs& a, u64* own, long long u, long long wait_u) {
  if (ld(W(own, UW_ERROR)) != 0) return kActError;
  const bool patient = a.mode == ULOG_CONTRIBUTE || a.mode == ULOG_SKIP;
  const long long t0 = wall_clock64();
  long long frozen_since = 0;
  for (;;) {
    Rec r;
    const int c = read_hist(own, u, r);
    if (c == 1) {
      // a replica that waits for nobody applies only what is already reduced
      if (!patient && !all_ready(own, u, r.owners)) return kActDone;
      return kActApply;
    }
    if (c < 0) {
      raise_error(own, UE_RECYCLED);
      return kActError;
    }
    if (u != wait_u) return kActDone;
    if (republish_if_closer_died(a, own, u, frozen_since)) continue;
    if (wall_clock64() - t0 > a.timeout_ticks) {
      raise_error(own, UE_CLOSE);
      return kActError;
    }
    nap();
  }
}

__global__ __launch_bounds__(256) void ulog_step_kernel(UlogArgs a) {
  __shared__ u64 s_payload;
  u64* own = a.ctl[a.rank];
  const u64 seq = ld(W(own, UW_SEQ));
  const u64 d0 = ld(W(own, UW_D0));
  const int op = (int)(d0 & 3);
  const int add = (int)((d0 >> 8) & 0xffff);
  const long long n4 = a.n >> 2;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  long long version = (long long)ld(W(own, UW_VERSION));
  LeaderCtx lc{W(own, UW_COUNTER), W(own, UW_DEC_TAG), W(own, UW_DEC_VAL), seq, a.timeout_ticks};

  // ---- stage ------------------------------------------------------------------------------------------
  if (op != 0) {
    const floatx4* g4 = reinterpret_cast<const floatx4*>(a.grads);
    floatx4* s4 = reinterpret_cast<floatx4*>(a.stage[a.rank]);
    if (op == 1)
      for (long long i = gtid; i < n4; i += gstride) s4[i] = g4[i] * a.weight;
    else
      for (long long i = gtid; i < n4; i += gstride) s4[i] += g4[i] * a.weight;
  }

  // ---- arrive -----------------------------------------------------------------------------------------
  // leader-only state (block 0, thread 0)
  long long wait_u = -1;
  u64 last_flags = 0, applied_now = 0, contributors = 0;
  u64 act = leader_step(lc, &s_payload, [&]() -> u64 {
    if (op != 0) {
      int order = 0;
      bool closed = false;
      long long open_now = 0;
      const bool any = a.mode == ULOG_OPEN;
      const int status = arrive(a, version, add, any ? 0 : a.quorum, any, order, closed, open_now);
      if (status == 3) {
        raise_error(own, UE_FROZEN);
        return kActError;
      }
      if (status == 0) {
        // a staged sum that was withdrawn and re-added with one more gradient was counted before
        bump_stat(own, US_ACCEPTED, op == 2 ? 1ull : (u64)add);
        last_flags = 1ull | (closed ? 2ull : 0ull) | ((u64)order << 8);
        if (any)
          st(W(own, UW_PENDING), ((u64)(open_now + 1) << 16) | (u64)add);
        else
          wait_u = version;
      } else {
        bump_stat(own, US_DROPPED, (u64)add);
        if (status == 2) wait_u = version;   // the closer is publishing this very update: a short wait
      }
    } else {
      if (d0 & 4) {
        if (d0 & 8) {   // the withdrawn sum went down with the new gradient: they no longer count as accepted
          st(W(own, UW_STATS + US_ACCEPTED), ld(W(own, UW_STATS + US_ACCEPTED)) - (u64)(add - 1));
          bump_stat(own, US_DROPPED, (u64)add);
        } else {
          bump_stat(own, US_DROPPED, 1);
        }
      }
      if (a.mode == ULOG_SKIP) wait_u = version;
    }
    return next_action(a, own, version, wait_u);
  });

  // ---- apply every closed update, oldest first --------------------------------------------------------
  while (act == kActApply) {
    Rec r;
    read_hist(own, version, r);   // stable: the slot is not recycled before this replica has applied it
    const int slot = (int)(version % kUlogRing);
    const int nown = __popc(r.owners);
    const long long quanta = (a.n + 63) / 64;
    const long long chunk4 = (quanta + nown - 1) / nown * 16;   // float4s per owner
    const bool owner = !r.closer_reduces && ((r.owners >> a.rank) & 1u);
    if (owner) {
      const int idx = __popc(r.owners & ((1u << a.rank) - 1u));
      const long long lo = idx * chunk4;
      const long long hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
      floatx4* out = reinterpret_cast<floatx4*>(a.red[a.rank]) + (long long)slot * n4;
      for (long long i = lo + gtid; i < hi; i += gstride) {
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < a.world; ++q)
          if ((r.mask >> q) & 1u) acc += __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(a.stage[q]) + i);
        out[i] = acc;
      }
    }
    // publish the own shard, then wait for every owner's. An owner the watchdog has given up on will
    // never publish: the lowest-ranked live contributor stands in for it - it sums the dead owner's
    // shard into its own ring (peer memory outlives its process for as long as it is mapped here)
    // and publishes the ready word in the dead owner's name, tagged with where the shard lives.
    bool published = !owner;
    for (;;) {
      const u64 verdict = leader_step(lc, &s_payload, [&]() -> u64 {
        if (!published)
          for (int p = 0; p < a.world; ++p) st_rel(W(a.ctl[p], UW_READY + slot * kUlogMaxPeers + a.rank), (u64)(version + 1));
        const long long t0 = wall_clock64();
        for (;;) {
          const unsigned missing = unready_owners(own, version, r.owners);
          if (!missing) break;
          const unsigned dead = (unsigned)ld(W(own, UW_DEAD));
          const unsigned live_contrib = r.mask & ~dead;
          if ((missing & dead) && live_contrib && (live_contrib & (0u - live_contrib)) == (1u << a.rank))
            return kActOk + 1 + (u64)(missing & dead);   // this replica stands in for them
          if (wall_clock64() - t0 > a.timeout_ticks) {
            raise_error(own, UE_READY);
            return kActError;
          }
          nap();
        }
        acquire_fence();
        return kActOk;
      });
      published = true;
      if (verdict == kActOk) break;
      if (verdict == kActError || verdict < kActOk) {
        act = kActError;
        break;
      }
      const unsigned stand_in = (unsigned)(verdict - kActOk - 1);
      floatx4* out = reinterpret_cast<floatx4*>(a.red[a.rank]) + (long long)slot * n4;
      for (int d = 0; d < kUlogMaxPeers; ++d) {
        if (!((stand_in >> d) & 1u)) continue;
        const int idx = __popc(r.owners & ((1u << d) - 1u));
        const long long lo = idx * chunk4;
        const long long hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
        for (long long i = lo + gtid; i < hi; i += gstride) {
          floatx4 acc = {0.f, 0.f, 0.f, 0.f};
          for (int q = 0; q < a.world; ++q)
            if ((r.mask >> q) & 1u) acc += __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(a.stage[q]) + i);
          out[i] = acc;
        }
      }
      const u64 pub = leader_step(lc, &s_payload, [&]() -> u64 {
        for (int d = 0; d < kUlogMaxPeers; ++d)
          if ((stand_in >> d) & 1u)
            for (int p = 0; p < a.world; ++p)
              if (!(((unsigned)ld(W(own, UW_DEAD)) >> p) & 1u))
                st_rel(W(a.ctl[p], UW_READY + slot * kUlogMaxPeers + d), (u64)(version + 1) | ((u64)(a.rank + 1) << kReadyHolderShift));
        return kActOk;
      });
      if (pub != kActOk) {
        act = kActError;
        break;
      }
    }
    if (act == kActError) break;
    acquire_fence();

    // gather the owners' shards, optimizer fused. holder_of: the rank whose ring holds the k-th
    // owner's shard (the owner itself, or its stand-in), four bits each - an indexed local array
    // would live in scratch memory
    unsigned holder_of = 0;
    {
      int k = 0;
      for (int p = 0; p < kUlogMaxPeers; ++p)
        if ((r.owners >> p) & 1u) {
          const u64 word = ld(W(own, UW_READY + slot * kUlogMaxPeers + p));
          const unsigned holder = (word >> kReadyHolderShift) ? (unsigned)(word >> kReadyHolderShift) - 1u : (unsigned)p;
          holder_of |= holder << (4 * k++);
        }
    }
    auto reduced = [&](long long i) -> floatx4 {
      const unsigned o = (holder_of >> (4 * (int)(i / chunk4))) & 15u;
      return __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(a.red[o]) + (long long)slot * n4 + i);
    };
    const long long t = version + 1;
    const double lr = (double)a.lr0 * pow((double)a.decay, (double)(version / (a.decay_steps > 0 ? a.decay_steps : 1)));
    floatx4* p4 = reinterpret_cast<floatx4*>(a.p);
    floatx4* m4 = reinterpret_cast<floatx4*>(a.m);
    if (a.opt == ULOG_ADAM) {
      const float lr_t = (float)(lr * sqrt(1.0 - pow((double)a.beta2, (double)t)) / (1.0 - pow((double)a.beta1, (double)t)));
      floatx4* v4 = reinterpret_cast<floatx4*>(a.v);
      for (long long i = gtid; i < n4; i += gstride) {
        const floatx4 g = reduced(i);
        floatx4 p = p4[i], m = m4[i], v = v4[i];
        adam_update4(p, m, v, g, r.scale, lr_t, a.beta1, a.beta2, a.eps);
        p4[i] = p;
        m4[i] = m;
        v4[i] = v;
      }
    } else {
      for (long long i = gtid; i < n4; i += gstride) {
        const floatx4 g = reduced(i);
        floatx4 p = p4[i], m = m4[i];
        momentum_update4(p, m, g, r.scale, (float)lr, a.momentum, a.weight_decay);
        p4[i] = p;
        m4[i] = m;
      }
    }
    act = leader_step(lc, &s_payload, [&]() -> u64 {
      st(W(own, UW_VERSION), (u64)(version + 1));
      for (int p = 0; p < a.world; ++p) st_rel(W(a.ctl[p], UW_APPLIED + a.rank), (u64)(version + 1));
      ++applied_now;
      contributors = (u64)r.count;
      return next_action(a, own, version + 1, wait_u);
    });
    ++version;
  }

  // ---- bookkeeping (leader) ---------------------------------------------------------------------------
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (act == kActError) raise_error(own, UE_GRID);   // keeps an earlier, more specific code if there is one
    st(W(own, UW_STATS + US_VERSION), ld(W(own, UW_VERSION)));
    st(W(own, UW_STATS + US_APPLIED), applied_now);
    st(W(own, UW_STATS + US_LAST), last_flags);
    st(W(own, UW_STATS + US_ERROR), ld(W(own, UW_ERROR)));
    if (applied_now) st(W(own, UW_STATS + US_CONTRIB), contributors);
    st(W(own, UW_STATS + US_SEQ), seq);
    if (a.status) {
      for (int k = 1; k < 8; ++k) st(a.status + k, ld(W(own, UW_STATS + k)));
      st_rel(a.status + US_SEQ, seq);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Chief's interval timer. status (host-visible, nullable): [0] = closed update + 1, or 0.
__global__ void ulog_close_begin_kernel(UlogArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u64* own = a.ctl[a.rank];
  st(W(own, UW_COUNTER2), 0);
  st(W(own, UW_CLOSE_DEC), 0);
  if (a.status) st(a.status, 0);
  if (ld(W(own, UW_ERROR)) != 0) return;
  u64* openw = W(a.ctl[0], UW_OPEN);
  u64 w = ld_acq(openw);
  for (;;) {
    if ((w & kFrozen) || open_count(w) < 1) return;
    // a live replica still needs the ring slot: skip this tick
    if (min_applied(a) < open_u(w) - kUlogRing + 1) return;
    if (cas(openw, w, w | kFrozen)) break;
  }
  const long long u = open_u(w);
  st(W(own, UW_CLOSE_DEC), (u64)(u + 1));
  publish_closed(a, u, open_count(w), open_mask(w), 1u << a.rank, 1.f / (float)open_count(w), true);
}

__global__ __launch_bounds__(256) void ulog_close_reduce_kernel(UlogArgs a) {
  __shared__ int s_last;
  u64* own = a.ctl[a.rank];
  const u64 dec = ld(W(own, UW_CLOSE_DEC));
  if (!dec) return;
  const long long u = (long long)dec - 1;
  Rec r;
  if (read_hist(own, u, r) != 1) return;
  const int slot = (int)(u % kUlogRing);
  const long long n4 = a.n >> 2;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  floatx4* out = reinterpret_cast<floatx4*>(a.red[a.rank]) + (long long)slot * n4;
  for (long long i = gtid; i < n4; i += gstride) {
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < a.world; ++q)
      if ((r.mask >> q) & 1u) acc += __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(a.stage[q]) + i);
    out[i] = acc;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const u64 prev = __hip_atomic_fetch_add(W(own, UW_COUNTER2), 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == gridDim.x - 1;
    if (s_last) {
      __threadfence_system();
      for (int p = 0; p < a.world; ++p) st_rel(W(a.ctl[p], UW_READY + slot * kUlogMaxPeers + a.rank), (u64)(u + 1));
      if (a.status) st_rel(a.status, (u64)(u + 1));
    }
  }
}

__global__ void ulog_init_kernel(u64* ctl, long long u0) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st(W(ctl, UW_OPEN), make_open(u0, 0, 0, false));
  st(W(ctl, UW_VERSION), (u64)u0);
  for (int p = 0; p < kUlogMaxPeers; ++p) st(W(ctl, UW_APPLIED + p), (u64)u0);
  st(W(ctl, UW_STATS + US_VERSION), (u64)u0);
}

__global__ void ulog_mark_dead_kernel(u64* ctl, unsigned mask) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  __hip_atomic_fetch_or(W(ctl, UW_DEAD), (u64)mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void ulog_poke_kernel(u64* ctl, int word, u64 value, long long delay_ticks) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < delay_ticks) nap();
  st_rel(W(ctl, word), value);
}

void validate(const UlogArgs& a, int blocks, bool needs_optimizer) {
  if (a.world < 1 || a.world > kUlogMaxPeers || a.rank < 0 || a.rank >= a.world) throw std::runtime_error("update log: bad rank/world");
  if (a.n <= 0 || (a.n & 3)) throw std::runtime_error("update log: n must be a positive multiple of 4");
  if (blocks < 1 || blocks > kUlogMaxBlocks) throw std::runtime_error("update log: 1..128 blocks (all must be co-resident)");
  for (int p = 0; p < a.world; ++p)
    if (!a.ctl[p] || !a.stage[p] || !a.red[p]) throw std::runtime_error("update log: unmapped peer buffer");
  if (a.timeout_ticks <= 0) throw std::runtime_error("update log: timeout must be positive");
  if (needs_optimizer) {
    if (!a.grads || !a.p || !a.m || (a.opt == ULOG_ADAM && !a.v)) throw std::runtime_error("update log: null gradient / parameter / slot pointer");
    if (a.mode < ULOG_CONTRIBUTE || a.mode > ULOG_SKIP) throw std::runtime_error("update log: bad mode");
    if (a.mode == ULOG_CONTRIBUTE && (a.quorum < 1 || a.quorum > a.world)) throw std::runtime_error("update log: quorum must be 1..world");
  }
}

}  // namespace

void ulog_step(const UlogArgs& a, int blocks, hipStream_t stream) {
  validate(a, blocks, true);
  launch("ulog_begin", ulog_begin_kernel, dim3(1), dim3(64), 0, stream, a);
  launch("ulog_step", ulog_step_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

void ulog_close(const UlogArgs& a, int blocks, hipStream_t stream) {
  validate(a, blocks, false);
  launch("ulog_close_begin", ulog_close_begin_kernel, dim3(1), dim3(64), 0, stream, a);
  launch("ulog_close_reduce", ulog_close_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

void ulog_init_ctl(unsigned long long* ctl, long long u0, hipStream_t stream) {
  if (!ctl) throw std::runtime_error("update log: null control block");
  hipLaunchKernelGGL(ulog_init_kernel, dim3(1), dim3(64), 0, stream, ctl, u0);
  DTF_HIP_CHECK(hipGetLastError());
}

void ulog_mark_dead(unsigned long long* ctl, unsigned mask, hipStream_t stream) {
  if (!ctl) throw std::runtime_error("update log: null control block");
  hipLaunchKernelGGL(ulog_mark_dead_kernel, dim3(1), dim3(64), 0, stream, ctl, mask);
  DTF_HIP_CHECK(hipGetLastError());
}

void ulog_peek(const unsigned long long* ctl, int first, int count, unsigned long long* out) {
  if (!ctl || first < 0 || count < 1 || first + count > UW_COUNT) throw std::runtime_error("update log: bad peek range");
  for (int k = 0; k < count; ++k)
    DTF_HIP_CHECK(hipMemcpy(out + k, ctl + (size_t)(first + k) * kUlogWordStride, sizeof(unsigned long long), hipMemcpyDeviceToHost));
}

void ulog_poke(unsigned long long* ctl, int word, unsigned long long value) {
  if (!ctl || word < 0 || word >= UW_COUNT) throw std::runtime_error("update log: bad poke word");
  DTF_HIP_CHECK(hipMemcpy(ctl + (size_t)word * kUlogWordStride, &value, sizeof(value), hipMemcpyHostToDevice));
}

void ulog_poke_async(unsigned long long* ctl, int word, unsigned long long value, long long delay_ticks, hipStream_t stream) {
  if (!ctl || word < 0 || word >= UW_COUNT) throw std::runtime_error("update log: bad poke word");
  if (delay_ticks < 0 || delay_ticks > 100000000) throw std::runtime_error("update log: poke delay must be within a second");
  hipLaunchKernelGGL(ulog_poke_kernel, dim3(1), dim3(64), 0, stream, ctl, word, value, delay_ticks);
  DTF_HIP_CHECK(hipGetLastError());
}

}  // namespace dtf
