// pnode_amd -- device-side helpers shared by the kernel files (pn_kernels.hip, pn_krylov.hip): register vectors,
// cache-policy loads/stores, the wave/block reduction tree and the in-launch finish of a grid-wide reduction.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

constexpr int kBlock = 256;     // 4 waves of 64
constexpr int kWave = 64;

// VW elements of T as one register-resident vector (16 bytes on the vector path)
template <typename T, int VW>
struct VecOf {
  typedef T type __attribute__((ext_vector_type(VW)));
};
template <typename T, int VW>
using Vec = typename VecOf<T, VW>::type;

// cache policy of the vector accesses (LD/ST template parameters):
//   loads : 0 plain, 1 non-temporal, 2 non-temporal for operand 0 only (the state vector coming
//           from its trajectory slot is cold, the stage derivatives are still cache-resident)
//   stores: 0 plain, 1 non-temporal, 2 write-through (sc0 sc1: the line does not stay dirty in
//           the XCD's L2, so the end-of-kernel write-back has nothing left to do)
template <int LD, typename V>
__device__ __forceinline__ V pn_load(const V *p) {
  if (LD == 1) return __builtin_nontemporal_load(p);
  return *p;
}
template <int ST, typename V>
__device__ __forceinline__ void pn_store(V *p, const V &v) {
  if (ST == 1) {
    __builtin_nontemporal_store(v, p);
  } else if (ST == 2 && sizeof(V) == 16) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  } else {
    *p = v;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

__device__ __forceinline__ double block_sum(double v) {
  __shared__ double lds[kBlock / kWave];
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) s += lds[w];
  }
  return s;   // valid in thread 0
}

// ---------------------------------------------------------------------------------------
// In-launch finish of a grid-wide reduction (no second kernel): every block publishes its partial(s) and draws a
// ticket; the block that draws the last one adds all partials IN INDEX ORDER (bit-reproducible, no float atomics)
// and writes the result.  gfx950 has 8 XCDs with private, mutually non-coherent L2s and per-CU L1s that other CUs'
// stores never refresh, so the hand-off uses the form cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md
// (visibility, valid forms) measure as sound without fences: the payload is stored write-through (8-byte agent-scope
// atomic store = global_store_dwordx2 sc1) by ONE lane, that lane drains its stores (s_waitcnt vmcnt(0)) and then adds
// to ONE agent-scope counter; the block whose add came last reads the payload with agent-scope (sc1) loads after a
// workgroup barrier.  The counters are put back to zero by the blocks that complete them, so the work area stays ready
// for the next launch (and for hipGraph replays); it must be zero-filled once before its first use.
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ void publish_partial(double *slot, double v) {
  __hip_atomic_store((gu64 *)slot, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double read_partial(const double *slot) {
  return __longlong_as_double((long long)__hip_atomic_load((gu64 *)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// Arrival counters.  Atomic adds to ONE word serialise at 11-13 ns each (MI355X_MICROARCH.md, "fanin"): a thousand
// blocks arriving on one counter cost more than the kernel they finish (measured: +9 us at 1024 blocks, +20 us at 2048 on
// pn_combine_wrms).  So the arrivals are spread over kTicketShards counters, each on a 128-byte line of its own; the
// block that completes a shard arrives on the top counter, and the block that completes the top counter is the last
// of the grid.  The area is kTicketDoubles doubles at the start of the caller's work block.
constexpr int kTicketShards = 32;
constexpr int kTicketStride = 16;                                   // doubles per counter line (128 bytes)
constexpr int kTicketDoubles = (kTicketShards + 1) * kTicketStride;

// thread 0 has published this block's partials; returns (to every thread of the block) whether this block is the
// last of the `total` blocks of the grid to get here.  `bid`: this block's linear index.
__device__ __forceinline__ bool draw_ticket(double *area, unsigned total, unsigned bid) {
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partials have left this wave before the ticket is drawn
    const unsigned ns = total < (unsigned)kTicketShards ? total : (unsigned)kTicketShards;
    const unsigned shard = bid % ns;
    const unsigned share = total / ns + (shard < total % ns ? 1u : 0u);
    gu32 *sc = (gu32 *)(area + (1 + shard) * kTicketStride);
    gu32 *top = (gu32 *)area;
    int last = 0;
    const unsigned t = __hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == share - 1u) {
      __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned t2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t2 == ns - 1u) {
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}
// partial[start], partial[start + stride], ... (< n) added in that order.  Eight loads are issued before the first is
// used: an agent-scope load takes ~1 us and the running sum would otherwise wait for each in turn (found in round 3:
// 64 dependent loads per lane made the last block of the Krylov products the longest part of the kernel).
__device__ __forceinline__ double strided_sum(const double *partial, int start, int stride, int n) {
  double s = 0;
  int i = start;
  for (; i + 7 * stride < n; i += 8 * stride) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = read_partial(partial + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < n; i += stride) s += read_partial(partial + i);
  return s;
}
// sum of partial[0..nblocks) in the order the two-kernel version used: thread-strided, then the block tree
__device__ __forceinline__ double ordered_sum(const double *partial, int nblocks) {
  const double s = strided_sum(partial, threadIdx.x, kBlock, nblocks);
  __syncthreads();            // block_sum's LDS words may still be read by thread 0 of the previous use
  return block_sum(s);
}


}  // namespace
