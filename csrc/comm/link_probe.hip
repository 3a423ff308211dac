// See link_probe.h.
#include "link_probe.h"

#include <stdexcept>

#include "../kernels/common.h"
#include "device_sync.h"

namespace dtf {
namespace {

using namespace dsync;

// Waits until *word == want (acquire); false on timeout.
__device__ bool wait_for(const u64* word, u64 want, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  while (ld(word) != want) {
    if (wall_clock64() - t0 > timeout_ticks) return false;
    nap();
  }
  return ld_acq(word) == want;
}

// A round trip: role 0 writes the payload (value = round number) into the peer's buffer, releases
// the peer's flag, and waits for its own flag; role 1 waits, checks the payload, answers the same way.
__device__ bool round_trips(const LinkProbeArgs& a, long long floats, u64 base, u64& bad, long long& ticks) {
  __shared__ int s_ok;
  const long long n4 = floats >> 2;
  floatx4* dst = reinterpret_cast<floatx4*>(a.peer_buf);
  const floatx4* src = reinterpret_cast<const floatx4*>(a.own_buf);
  u64 wrong = 0;
  const long long t_begin = wall_clock64();
  for (int it = 1; it <= a.iters; ++it) {
    const u64 tag = base + (u64)it;
    const float val = (float)(tag & 0xffff);
    if (a.role == 1) {
      if (threadIdx.x == 0) s_ok = wait_for(W(a.own_ctl, LP_FLAG), tag, a.timeout_ticks);
      __syncthreads();
      if (!s_ok) return false;
      acquire_fence();
      for (long long i = threadIdx.x; i < n4; i += blockDim.x) {
        const floatx4 v = __builtin_nontemporal_load(src + i);
        wrong += (v[0] != val) + (v[1] != val) + (v[2] != val) + (v[3] != val);
      }
    }
    for (long long i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = floatx4{val, val, val, val};
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) st_rel(W(a.peer_ctl, LP_FLAG), tag);
    if (a.role == 0) {
      if (threadIdx.x == 0) s_ok = wait_for(W(a.own_ctl, LP_FLAG), tag, a.timeout_ticks);
      __syncthreads();
      if (!s_ok) return false;
      acquire_fence();
      for (long long i = threadIdx.x; i < n4; i += blockDim.x) {
        const floatx4 v = __builtin_nontemporal_load(src + i);
        wrong += (v[0] != val) + (v[1] != val) + (v[2] != val) + (v[3] != val);
      }
    }
    __syncthreads();
  }
  ticks = wall_clock64() - t_begin;
  bad += wrong;
  return true;
}

__global__ __launch_bounds__(256) void link_probe_kernel(LinkProbeArgs a) {
  __shared__ unsigned long long s_bad[256];
  u64 bad = 0;
  long long t_flag = 0, t_payload = 0;
  bool ok = round_trips(a, 0, 0, bad, t_flag);
  if (ok && a.payload_floats > 0) ok = round_trips(a, a.payload_floats, 1u << 20, bad, t_payload);
  s_bad[threadIdx.x] = bad;
  __syncthreads();
  if (threadIdx.x != 0) return;
  u64 total_bad = 0;
  for (int i = 0; i < 256; ++i) total_bad += s_bad[i];
  // both ranks increment one counter with compare-exchange: no increment may be lost
  u64 retries = 0;
  for (int k = 0; ok && k < a.cas_iters; ++k) {
    u64 seen = ld_acq(a.counter);
    while (!cas(a.counter, seen, seen + 1)) ++retries;   // `seen` is refreshed by a failed exchange
  }
  st(a.result + 0, (u64)t_flag);
  st(a.result + 1, (u64)t_payload);
  st(a.result + 2, total_bad);
  st(a.result + 3, ok ? 0ull : 1ull);
  st_rel(a.result + 4, retries);
}

}  // namespace

void link_probe(const LinkProbeArgs& a, hipStream_t stream) {
  if (!a.own_ctl || !a.peer_ctl || !a.counter || !a.result) throw std::runtime_error("link_probe: null control word");
  if (a.payload_floats < 0 || (a.payload_floats & 3) || (a.payload_floats > 0 && (!a.own_buf || !a.peer_buf)))
    throw std::runtime_error("link_probe: payload must be a multiple of 4 floats with both buffers mapped");
  if (a.iters < 1 || a.iters > 100000 || a.cas_iters < 0 || a.role < 0 || a.role > 1 || a.timeout_ticks <= 0)
    throw std::runtime_error("link_probe: bad iteration count / role / timeout");
  hipLaunchKernelGGL(link_probe_kernel, dim3(1), dim3(256), 0, stream, a);
  DTF_HIP_CHECK(hipGetLastError());
}

}  // namespace dtf
