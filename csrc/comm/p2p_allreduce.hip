// See p2p_allreduce.h. Synchronisation protocol of call number `seq` (flags only ever grow):
//
//   staged[r]   on every peer: rank r has copied its gradient into its exchange buffer
//   reduced[r]  on every peer: rank r has written the reduced shard r into its exchange buffer
//   done[r]     on every peer: rank r has finished reading other ranks' exchange buffers
//
// A rank writes shard s of its exchange buffer only after it is the owner of s and everybody has
// staged (phase 1), and starts the next call's staging copy only after every peer reported done,
// so no buffer is overwritten while a peer may still read it. Every wait has a wall-clock bound;
// a rank that times out raises its error word and leaves, it never spins forever.
#include "p2p_allreduce.h"

#include <stdexcept>

#include "../runtime/program.h"

namespace dtf {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

enum Phase : int { PH_STAGED = 0, PH_REDUCED = 1, PH_DONE = 2, PH_COUNTER = 3 };

__device__ __forceinline__ uint32_t* flag_of(uint32_t* block, int phase, int peer) {
  return block + ((size_t)phase * kMaxPeers + peer) * kFlagStride;
}

__device__ __forceinline__ void signal_all(const P2PAllReduceArgs& a, int phase) {
  // one lane per peer; the data written before is made visible system-wide first
  __threadfence_system();
  if ((int)threadIdx.x < a.world)
    __hip_atomic_store(flag_of(a.flags[threadIdx.x], phase, a.rank), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Every thread of the block returns the same verdict: true when all peers reached `seq` in `phase`.
__device__ __forceinline__ bool wait_all(const P2PAllReduceArgs& a, int phase, int* s_ok) {
  if (threadIdx.x == 0) *s_ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < a.world) {
    const uint32_t* f = flag_of(a.flags[a.rank], phase, threadIdx.x);
    const long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
      if (wall_clock64() - t0 > a.timeout_ticks || __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(a.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  return *s_ok != 0;
}

// Grid-wide rendezvous inside one rank: returns true in the last block to arrive.
__device__ __forceinline__ bool last_block(const P2PAllReduceArgs& a, int which, int* s_last) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t* c = flag_of(a.flags[a.rank], PH_COUNTER, which);
    const uint32_t prev = __hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = (prev == gridDim.x - 1);
    if (*s_last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return *s_last != 0;
}

__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PAllReduceArgs a) {
  __shared__ int s_ok, s_last;
  const long long n4 = a.n >> 2;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  // shard s covers float4 indices [s*chunk, min(n4, (s+1)*chunk))
  const long long chunk = (n4 + a.world - 1) / a.world;

  // ---- phase 0: stage the own contribution (weighted) and tell everybody ------------------------
  {
    const floatx4* src = reinterpret_cast<const floatx4*>(a.data);
    floatx4* dst = reinterpret_cast<floatx4*>(a.xbuf[a.rank]);
    const float w = a.weight[a.rank];
    for (long long i = gtid; i < n4; i += gstride) dst[i] = src[i] * w;
  }
  if (last_block(a, 0, &s_last)) signal_all(a, PH_STAGED);

  // ---- phase 1: reduce the own shard from all peers ----------------------------------------------
  if (!wait_all(a, PH_STAGED, &s_ok)) return;
  {
    const long long lo = chunk * a.rank;
    const long long hi = lo + chunk < n4 ? lo + chunk : n4;
    floatx4* own = reinterpret_cast<floatx4*>(a.xbuf[a.rank]);
    floatx4* out = reinterpret_cast<floatx4*>(a.data);
    for (long long i = lo + gtid; i < hi; i += gstride) {
      floatx4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int p = 0; p < a.world; ++p) {
        if (a.weight[p] == 0.f) continue;   // masked-out replica: its buffer is not even read
        acc += __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(a.xbuf[p]) + i);
      }
      own[i] = acc;
      out[i] = acc;
    }
  }
  if (last_block(a, 1, &s_last)) signal_all(a, PH_REDUCED);

  // ---- phase 2: gather the other reduced shards ----------------------------------------------------
  if (!wait_all(a, PH_REDUCED, &s_ok)) return;
  {
    floatx4* out = reinterpret_cast<floatx4*>(a.data);
    for (int k = 1; k < a.world; ++k) {
      const int p = (a.rank + k) % a.world;   // every rank starts at a different peer
      const long long lo = chunk * p;
      const long long hi = lo + chunk < n4 ? lo + chunk : n4;
      const floatx4* src = reinterpret_cast<const floatx4*>(a.xbuf[p]);
      for (long long i = lo + gtid; i < hi; i += gstride) out[i] = __builtin_nontemporal_load(src + i);
    }
  }
  // ---- end: nobody may restage before every peer has finished reading -----------------------------
  if (last_block(a, 2, &s_last)) {
    signal_all(a, PH_DONE);
    wait_all(a, PH_DONE, &s_ok);
  }
}

}  // namespace

void p2p_allreduce(const P2PAllReduceArgs& a, int blocks, hipStream_t stream) {
  if (a.world < 1 || a.world > kMaxPeers || a.rank < 0 || a.rank >= a.world)
    throw std::runtime_error("p2p_allreduce: bad rank/world");
  if (a.n & 3) throw std::runtime_error("p2p_allreduce: n must be a multiple of 4");
  if (blocks < 1 || blocks > 256) throw std::runtime_error("p2p_allreduce: 1..256 blocks (all must be co-resident)");
  for (int p = 0; p < a.world; ++p)
    if (!a.xbuf[p] || !a.flags[p]) throw std::runtime_error("p2p_allreduce: unmapped peer buffer");
  if (!a.data || !a.error) throw std::runtime_error("p2p_allreduce: null data/error pointer");
  launch("p2p_allreduce", p2p_allreduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

void* ipc_alloc(size_t bytes, bool fine_grained) {
  void* p = nullptr;
  if (fine_grained)
    DTF_HIP_CHECK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
  else
    DTF_HIP_CHECK(hipMalloc(&p, bytes));
  DTF_HIP_CHECK(hipMemset(p, 0, bytes));
  DTF_HIP_CHECK(hipDeviceSynchronize());
  return p;
}

void ipc_free(void* p) {
  if (p) DTF_HIP_CHECK(hipFree(p));
}

std::string ipc_export(void* p) {
  hipIpcMemHandle_t h;
  DTF_HIP_CHECK(hipIpcGetMemHandle(&h, p));
  return std::string(reinterpret_cast<const char*>(&h), sizeof(h));
}

void* ipc_import(const std::string& handle) {
  if (handle.size() != sizeof(hipIpcMemHandle_t)) throw std::runtime_error("ipc_import: bad handle size");
  hipIpcMemHandle_t h;
  memcpy(&h, handle.data(), sizeof(h));
  void* p = nullptr;
  DTF_HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  return p;
}

void ipc_release(void* p) {
  if (p) DTF_HIP_CHECK(hipIpcCloseMemHandle(p));
}

}  // namespace dtf
