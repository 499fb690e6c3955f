// pnode_amd -- hand-written gfx950 (MI355X, CDNA4) kernels for the explicit-RK forward sweep
// and the discrete-adjoint reverse sweep.  See include/pnode_amd.h for the ABI and the
// reference call sites (pnode/petsc_adjoint.py) each entry point replaces.
//
// Every kernel here is a pure HBM stream: arithmetic intensity <= 0.5 flop/byte, no reuse
// between lanes, so there is nothing to stage in LDS and nothing for MFMA.  What matters
// (guides: cdna_hip_programming.md Appendix B "Element-wise", Guidelines 11/13):
//   * 16-byte accesses per lane (global_load_dwordx4 / global_store_dwordx4), lane-contiguous,
//     so one wave instruction moves 1 KiB;
//   * ALL loads of a thread issued before the first use (the compiler then emits one
//     s_waitcnt per consumer, the loads overlap) -- several KiB in flight per wave;
//   * a grid of ~8 blocks of 256 threads per CU for the 8 MiB state vectors of the target
//     configuration, i.e. one wave of blocks over the 256 CUs, no grid-stride loop tail;
//   * one launch per RK stage instead of PETSc's VecCopy + VecMAXPY (+ VecScale + VecAXPY)
//     sequence, and results written straight into the trajectory slot the adjoint reads.
// LDS is used only for the cross-wave step of the WRMS error-norm reduction; the in-wave
// step is a 64-lane shuffle tree.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pnode_amd.h"
#include "pn_internal.h"
#include "pn_device.h"

namespace {

template <typename T, int NIN>
struct LinArgs {
  const T *x[NIN];
  T c[NIN];
};

// ---------------------------------------------------------------------------------------
// out = sum_j c[j]*x[j]   (+ out2 = c2*out)
//   rk_stage:   x0 = u (c0 = 1), x_j = K_j, c_j = h*a_ij
//   adj_theta:  x0 = lambda (c0 = H*b_i) or the first dlam, x_j = dlam_j, c_j = H*a_ji
//   adj_accum:  x0 = lambda, x_j = dlam_j / forcing, all c = 1; out2 = next cotangent
// Summation order is fixed: ((c0*x0) + c1*x1) + c2*x2 ... with fused multiply-adds.
// ---------------------------------------------------------------------------------------
template <typename T, int NIN, int VW, int VPT, bool OUT2, int BLOCK, int LD = 0, int ST = 0>
__global__ __launch_bounds__(BLOCK) void pn_lincomb_kernel(LinArgs<T, NIN> a, T *out, T *out2, T c2, int64_t nvec,
                                                           int64_t n, int64_t grid_stride, int xcd_remap) {
  // `out` / `out2` carry no __restrict__: the ABI lets them alias an input (lambda is updated in place,
  // pn_lincomb documents it); every thread loads all its elements before it stores any of them
  using V = Vec<T, VW>;
  constexpr int kBlock = BLOCK;
  // grid_stride == 0: one tile of BLOCK*VPT vectors per block; otherwise the grid is capped and
  // every block walks the vector with that stride (cdna_hip_programming.md Guideline 11)
  // Workgroups are handed to the 8 XCDs round-robin (block b -> XCD b % 8).  xcd_remap = 1 (PN_TUNE "xcd=1", an
  // experiment: nothing is re-used between workgroups, so it buys nothing -- profiles/r02_ab_xcd.txt) gives every XCD
  // one contiguous eighth of the vector instead of every eighth tile.
  int64_t bid = blockIdx.x;
  if (xcd_remap) {
    const int64_t per = gridDim.x / 8;
    if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
  }
  int64_t base = bid * (kBlock * VPT) + threadIdx.x;
  do {
    V r[VPT][NIN];
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
      const int64_t i = base + (int64_t)p * kBlock;
      if (i < nvec) {
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
          if (LD == 1 || (LD == 2 && j == 0)) r[p][j] = pn_load<1>(reinterpret_cast<const V *>(a.x[j]) + i);
          else r[p][j] = pn_load<0>(reinterpret_cast<const V *>(a.x[j]) + i);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < VPT; ++p) {
      const int64_t i = base + (int64_t)p * kBlock;
      if (i < nvec) {
        V o, o2;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          T acc = a.c[0] * r[p][0][e];
#pragma unroll
          for (int j = 1; j < NIN; ++j) acc = fma(a.c[j], r[p][j][e], acc);
          o[e] = acc;
          if (OUT2) o2[e] = c2 * acc;
        }
        pn_store<ST>(reinterpret_cast<V *>(out) + i, o);
        if (OUT2) pn_store<ST>(reinterpret_cast<V *>(out2) + i, o2);
      }
    }
    base += grid_stride;
  } while (grid_stride > 0 && base < nvec);
  // ragged tail (n not a multiple of the vector width): the first lanes of block 0
  if (VW > 1 && blockIdx.x == 0) {
    const int64_t i = nvec * VW + threadIdx.x;
    if (i < n) {
      T acc = a.c[0] * a.x[0][i];
#pragma unroll
      for (int j = 1; j < NIN; ++j) acc = fma(a.c[j], a.x[j][i], acc);
      out[i] = acc;
      if (OUT2) out2[i] = c2 * acc;
    }
  }
  if (ST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // inline-asm stores are ours to wait for
}

// ---------------------------------------------------------------------------------------
// Embedded error estimate fused with the solution update (TSEvaluateStep + WRMS norm).
//   unew = x0 + sum_j cb[j]*K_j   (WRITE: stored; !WRITE: x0 already is unew, cb ignored)
//   err  = sum_j ce[j]*K_j ;  uhat = unew + err
//   block partial of  (err/(atol + rtol*max(|unew|,|uhat|)))^2  in double
// Reduction: per-thread double -> 64-lane shuffle tree -> LDS across the 4 waves -> one
// double per block; a second one-block kernel adds the block partials in index order, so
// the norm is bit-reproducible run to run (no float atomics).
// ---------------------------------------------------------------------------------------
template <typename T, int NK>
struct ErrArgs {
  const T *k[NK];
  T cb[NK];
  T ce[NK];
};

template <typename T>
__device__ __forceinline__ double wrms_term(T unew, T err, double atol, double rtol) {
  // the reference measures |u - uhat| between the two STORED solutions (TSErrorWeightedNorm
  // takes the vectors, not the increment), so uhat is rounded to the storage type first
  const double un = (double)unew;
  const double uh = (double)(T)(unew + err);
  const double tol = atol + rtol * fmax(fabs(un), fabs(uh));
  const double q = (un - uh) / tol;
  return q * q;
}
// fp32 states: the ratio in fp32 (as a single-precision PETSc computes all of TSErrorWeightedNorm), the sum of squares
// in double.  u - uhat is exact in fp32 (the two are within a factor of two of each other); tol and the quotient carry
// one fp32 rounding each.  The double-precision division of the generic form was a visible share of the kernel at 4096 x 512.
template <>
__device__ __forceinline__ double wrms_term<float>(float unew, float err, double atol, double rtol) {
  const float uh = unew + err;
  const float tol = (float)atol + (float)rtol * fmaxf(fabsf(unew), fabsf(uh));
  const float q = (unew - uh) / tol;
  return (double)q * (double)q;
}

// FIN: where the norm is finished.  0 (default): on the HOST -- every block stores its partial into the caller's pinned
// block (result[1 + block]; result[0] = number of blocks) and pn_stream_wait_wrms, which has to wait for the stream anyway,
// adds them in index order: the kernel ends with its last load.  1: in the launch (arrival counters, pn_device.h): the
// last-arriving block adds the partials; its hand-off (write-through store, drained, one or two returning atomics) is
// ~3-4 us of pure latency at the end of EVERY block, more than the finishing kernel it replaces (PN_TUNE "wfin=1").
template <typename T, int NK, int VW, int VPT, bool WRITE, int ST, int FIN>
__global__ __launch_bounds__(kBlock) void pn_combine_wrms_kernel(const T *x0, ErrArgs<T, NK> a, T *unew_out, double atol,
                                                                 double rtol, double *__restrict__ work, int64_t nvec,
                                                                 int64_t n, double inv_n, double *result) {
  // x0 / unew_out carry no __restrict__ (the caller may update a state in place); every thread loads all its
  // elements before it stores any.  Loads are plain (the stage derivatives are recent), the new state is stored
  // non-temporally (ST = 1: it is not read again before func has run), as in pn_lincomb_kernel.
  using V = Vec<T, VW>;
  const int64_t base = (int64_t)blockIdx.x * (kBlock * VPT) + threadIdx.x;
  V ru[VPT], rk[VPT][NK];
#pragma unroll
  for (int p = 0; p < VPT; ++p) {
    const int64_t i = base + (int64_t)p * kBlock;
    if (i < nvec) {
      ru[p] = reinterpret_cast<const V *>(x0)[i];
#pragma unroll
      for (int j = 0; j < NK; ++j) rk[p][j] = reinterpret_cast<const V *>(a.k[j])[i];
    }
  }
  double sum = 0;
#pragma unroll
  for (int p = 0; p < VPT; ++p) {
    const int64_t i = base + (int64_t)p * kBlock;
    if (i < nvec) {
      V o;
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        T un = ru[p][e], er = (T)0;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          if (WRITE) un = fma(a.cb[j], rk[p][j][e], un);
          er = fma(a.ce[j], rk[p][j][e], er);
        }
        o[e] = un;
        sum += wrms_term<T>(un, er, atol, rtol);
      }
      if (WRITE) pn_store<ST>(reinterpret_cast<V *>(unew_out) + i, o);
    }
  }
  if (VW > 1 && blockIdx.x == 0) {
    const int64_t i = nvec * VW + threadIdx.x;
    if (i < n) {
      T un = x0[i], er = (T)0;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        if (WRITE) un = fma(a.cb[j], a.k[j][i], un);
        er = fma(a.ce[j], a.k[j][i], er);
      }
      if (WRITE) unew_out[i] = un;
      sum += wrms_term<T>(un, er, atol, rtol);
    }
  }
  const double s = block_sum(sum);
  if (FIN == 0) {
    if (threadIdx.x == 0) {
      result[1 + blockIdx.x] = s;
      if (blockIdx.x == 0) result[0] = (double)gridDim.x;
    }
    return;
  }
  // work: [arrival counters][one partial per block]
  double *partial = work + kTicketDoubles;
  if (threadIdx.x == 0) publish_partial(partial + blockIdx.x, s);
  if (draw_ticket(work, gridDim.x, blockIdx.x)) {
    const double tot = ordered_sum(partial, (int)gridDim.x);
    if (threadIdx.x == 0) {
      result[0] = 0.0;                         // "finished on the device": result[1] is the norm
      result[1] = sqrt(tot * inv_n);
    }
  }
}

// ---------------------------------------------------------------------------------------
// result[j] = <x, y_j>, j < NK: per-thread double partials -> wave shuffle -> LDS -> one double
// per (block, j); a one-block kernel sums the block partials in index order.
// ---------------------------------------------------------------------------------------
template <typename T, int NK>
struct DotArgs {
  const T *y[NK];
};

template <typename T, int NK, int VW>
__global__ __launch_bounds__(kBlock) void pn_dots_kernel(const T *__restrict__ x, DotArgs<T, NK> a,
                                                         double *__restrict__ work, int64_t nvec, int64_t n,
                                                         double *result) {
  double *partial = work + kTicketDoubles;           // work: [arrival counters][NK x gridDim.x partials]
  using V = Vec<T, VW>;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  double s[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) s[j] = 0;
  if (i < nvec) {
    const V xv = reinterpret_cast<const V *>(x)[i];
    V yv[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) yv[j] = reinterpret_cast<const V *>(a.y[j])[i];
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
      for (int e = 0; e < VW; ++e) s[j] += (double)xv[e] * (double)yv[j][e];
  }
  if (VW > 1 && blockIdx.x == 0) {
    const int64_t q = nvec * VW + threadIdx.x;
    if (q < n) {
#pragma unroll
      for (int j = 0; j < NK; ++j) s[j] += (double)x[q] * (double)a.y[j][q];
    }
  }
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const double b = block_sum(s[j]);
    if (threadIdx.x == 0) publish_partial(partial + (int64_t)j * gridDim.x + blockIdx.x, b);
    __syncthreads();
  }
  // the block that draws the last ticket adds the block partials of every product in index order
  if (draw_ticket(work, gridDim.x, blockIdx.x)) {
    for (int j = 0; j < NK; ++j) {
      const double tot = ordered_sum(partial + (int64_t)j * gridDim.x, (int)gridDim.x);
      if (threadIdx.x == 0) result[j] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------------
// mu[off_k + i] += g_k[i] for up to kMaxSeg parameter tensors in one launch.
// ---------------------------------------------------------------------------------------
constexpr int kMaxSeg = 48;
template <typename T>
struct SegArgs {
  const T *g[kMaxSeg];
  int64_t off[kMaxSeg];
  int64_t len[kMaxSeg];
  int first_block[kMaxSeg + 1];
  int nseg;
};

template <typename T, int VW>
__global__ __launch_bounds__(kBlock) void pn_param_accum_kernel(SegArgs<T> a, T *__restrict__ mu, T alpha) {
  using V = Vec<T, VW>;
  int k = 0;
  while (k + 1 < a.nseg && (int)blockIdx.x >= a.first_block[k + 1]) ++k;   // block-uniform
  const T *__restrict__ g = a.g[k];
  T *__restrict__ m = mu + a.off[k];
  const int64_t len = a.len[k];
  const int64_t b = (int64_t)(blockIdx.x - a.first_block[k]);
  constexpr int64_t kElemsPerBlock = (int64_t)kBlock * VW * 2;
  const int64_t lo = b * kElemsPerBlock;
  const bool aligned = ((((uintptr_t)g) | ((uintptr_t)m)) & (sizeof(V) - 1)) == 0;
  if (aligned) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int64_t i = lo + ((int64_t)p * kBlock + threadIdx.x) * VW;
      if (i + VW <= len) {
        V gv = *reinterpret_cast<const V *>(g + i);
        V mv = *reinterpret_cast<V *>(m + i);
#pragma unroll
        for (int e = 0; e < VW; ++e) mv[e] = fma(alpha, gv[e], mv[e]);
        *reinterpret_cast<V *>(m + i) = mv;
      } else {
        for (int64_t q = i; q < len && q < i + VW; ++q) m[q] = fma(alpha, g[q], m[q]);
      }
    }
  } else {
    const int64_t hi = lo + kElemsPerBlock < len ? lo + kElemsPerBlock : len;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) m[i] = fma(alpha, g[i], m[i]);
  }
}

// Many gradient sets at once (the stages of one or of several time steps):
//   mu[off_k + i] = fma(alpha_{S-1}, g_{S-1,k}[i], ... fma(alpha_0, g_{0,k}[i], mu[off_k + i]))
// -- the rounding of S successive single-set launches in the order 0..S-1, with ONE read and ONE
// write of mu.  The host hands every parameter tensor k the compacted list of its live sources
// (g[pbase_k + c], src[pbase_k + c], c < cnt_k); the kernel walks that list in chunks of 16 with all
// loads of a chunk issued before the first fma, so a thread keeps up to 256 bytes in flight and the
// chain of dependent fmas costs no memory latency; one 16-byte vector per thread, so that a parameter
// vector of a few MiB still spreads over every CU with several waves each.
constexpr int kMaxSrc = 32;
constexpr int kMaxSegM = 32;
constexpr int kMaxPtr = 256;
constexpr int kAccChunk = 16;
template <typename T>
struct MultiSegArgs {
  const T *g[kMaxPtr];
  int64_t off[kMaxSegM];
  int64_t len[kMaxSegM];
  int first_block[kMaxSegM + 1];
  T alpha[kMaxSrc];
  short pbase[kMaxSegM];
  short cnt[kMaxSegM];
  unsigned char src[kMaxPtr];
  int nseg;
};

template <typename T, int VW, int C, int P, bool NT>
__device__ __forceinline__ void pn_accum_chunk(const T *const *gp, const T *al, const int64_t *i, const bool *f, Vec<T, VW> *m) {
  using V = Vec<T, VW>;
  V g[P][C];
#pragma unroll
  for (int u = 0; u < C; ++u) {
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (P == 1 || f[p]) {
        // the gradient tensors are read exactly once, long after they were written: stream them past the caches
        if (NT) g[p][u] = __builtin_nontemporal_load(reinterpret_cast<const V *>(gp[u] + i[p]));
        else g[p][u] = *reinterpret_cast<const V *>(gp[u] + i[p]);
      }
  }
#pragma unroll
  for (int u = 0; u < C; ++u) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int e = 0; e < VW; ++e) m[p][e] = fma(al[u], g[p][u][e], m[p][e]);
  }
}

// NB: the argument block is indexed with run-time subscripts only HERE, in the kernel body, and copied into
// locals that are indexed statically: handing `a` (or a reference to it) to a helper makes the compiler keep a
// private copy of the whole 3 KiB block in scratch memory (seen: 40x slower); tests/test_abi.py checks the
// code object for scratch use.
template <typename T, int VW, int P, bool NT>
__global__ __launch_bounds__(kBlock) void pn_param_accum_multi_kernel(const MultiSegArgs<T> a, T *mu) {
  using V = Vec<T, VW>;
  int k = 0;
  while (k + 1 < a.nseg && (int)blockIdx.x >= a.first_block[k + 1]) ++k;   // block-uniform
  T *m = mu + a.off[k];
  const int64_t len = a.len[k];
  const int q0 = a.pbase[k], cnt = a.cnt[k];
  const int64_t b = (int64_t)(blockIdx.x - a.first_block[k]);
  constexpr int64_t kElemsPerBlock = (int64_t)kBlock * VW * P;
  const int64_t lo = b * kElemsPerBlock;
  uintptr_t bits = (uintptr_t)m;
  for (int c = 0; c < cnt; ++c) bits |= (uintptr_t)a.g[q0 + c];
  if ((bits & (sizeof(V) - 1)) == 0) {
    int64_t i[P];
    bool f[P];
    V mv[P];
    bool any = false;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      i[p] = lo + ((int64_t)p * kBlock + threadIdx.x) * VW;
      f[p] = i[p] + VW <= len;
      any = any || f[p];
      if (f[p]) mv[p] = *reinterpret_cast<V *>(m + i[p]);
    }
    if (any) {
      for (int c0 = 0; c0 < cnt; c0 += kAccChunk) {
        const T *gp[kAccChunk];
        T al[kAccChunk];
        const int nn = cnt - c0 < kAccChunk ? cnt - c0 : kAccChunk;
#pragma unroll
        for (int u = 0; u < kAccChunk; ++u) {
          const int q = q0 + c0 + (u < nn ? u : 0);
          gp[u] = a.g[q];
          al[u] = a.alpha[a.src[q]];
        }
        switch (nn) {
#define PN_REM(C) case C: pn_accum_chunk<T, VW, C, P, NT>(gp, al, i, f, mv); break;
          PN_REM(1) PN_REM(2) PN_REM(3) PN_REM(4) PN_REM(5) PN_REM(6) PN_REM(7) PN_REM(8)
          PN_REM(9) PN_REM(10) PN_REM(11) PN_REM(12) PN_REM(13) PN_REM(14) PN_REM(15) PN_REM(16)
#undef PN_REM
          default: break;
        }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (f[p]) {
        *reinterpret_cast<V *>(m + i[p]) = mv[p];
      } else if (i[p] < len) {                  // the ragged end of the tensor: one thread, scalar
        for (int64_t e = i[p]; e < len; ++e) {
          T v = m[e];
          for (int c = 0; c < cnt; ++c) v = fma(a.alpha[a.src[q0 + c]], a.g[q0 + c][e], v);
          m[e] = v;
        }
      }
    }
  } else {
    const int64_t hi = lo + kElemsPerBlock < len ? lo + kElemsPerBlock : len;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
      T v = m[i];
      for (int c = 0; c < cnt; ++c) v = fma(a.alpha[a.src[q0 + c]], a.g[q0 + c][i], v);
      m[i] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------
// Parameter sensitivities of biases: for up to 32 sources j,  mu_j[c] += alpha_j * sum_r G_j[r][c]  (G_j row-major,
// rows_j x cols_j; several sources may name the same mu: they are added in source order) -- the column sums autograd takes
// for d(loss)/d(bias) of Linear layers, fused with the accumulation into mu (RHSJacPShell.multTranspose + the VecAXPY on mu,
// pa.py:341-363), for the stages of one or of several time steps in ONE pass.  HBM-bound: every G is read once.
//   pass 1: one block per (source, 64*VW-column tile, row chunk); a wave reads 64*VW consecutive columns of a row with one
//           16-byte load per lane (coalesced), the 4 waves of a block take every fourth row of the chunk; per-lane sums in
//           double; the waves are combined through LDS in a fixed order; one double per (source, chunk, column) goes to `work`.
//   pass 2: one block per (target mu, 64 columns): for the target's sources in order, the chunk partials are added in a fixed
//           association (4 waves x every fourth chunk, then wave order) and mu is updated by one fma per source:
//           bit-reproducible, no atomics, and independent of how the sources were grouped into launches.
// ---------------------------------------------------------------------------------------
constexpr int kMaxColSrc = 32;

template <typename T>
struct ColsumArgs {
  const T *g[kMaxColSrc];
  int64_t rows[kMaxColSrc], cols[kMaxColSrc], poff[kMaxColSrc];   // poff: offset of the source's partials in `work` (doubles)
  int first_block[kMaxColSrc + 1];
  int ctiles[kMaxColSrc], rpb[kMaxColSrc];
  int nsrc;
};

template <typename T>
struct ColfinArgs {
  T *mu[kMaxColSrc];               // per target
  int64_t cols[kMaxColSrc];        // per target
  int first_block[kMaxColSrc + 1]; // per target
  int src_begin[kMaxColSrc + 1];   // per target: its sources are [src_begin[t], src_begin[t+1]) of the arrays below
  int64_t poff[kMaxColSrc];        // per source, grouped by target, original order within a target
  int chunks[kMaxColSrc];
  T alpha[kMaxColSrc];
  int nt;
};

template <typename T, int VW>
__global__ __launch_bounds__(kBlock) void pn_colsum_partial_kernel(const ColsumArgs<T> a, double *work) {
  using V = Vec<T, VW>;
  constexpr int kWaves = kBlock / kWave;
  int k = 0;
  while (k + 1 < a.nsrc && (int)blockIdx.x >= a.first_block[k + 1]) ++k;   // block-uniform
  const int b = (int)blockIdx.x - a.first_block[k];
  const int tile = b % a.ctiles[k], chunk = b / a.ctiles[k];
  const T *g = a.g[k];
  const int64_t rows = a.rows[k], cols = a.cols[k];
  const int lane = threadIdx.x & (kWave - 1), wy = threadIdx.x / kWave;
  const int64_t c0 = ((int64_t)tile * kWave + lane) * VW;                  // first of this lane's VW columns
  const int64_t r0 = (int64_t)chunk * a.rpb[k];
  const int64_t r1 = r0 + a.rpb[k] < rows ? r0 + a.rpb[k] : rows;
  double acc[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) acc[e] = 0.0;
  const bool vec = VW > 1 && c0 + VW <= cols && ((((uintptr_t)g) | ((uintptr_t)cols * sizeof(T))) & (sizeof(V) - 1)) == 0;
  if (vec) {
    int64_t r = r0 + wy;
    // eight rows in flight per wave (their loads are issued before the first is added): 8 KiB per wave, enough to cover the
    // memory latency with the two or three blocks a CU holds
    for (; r + 7 * kWaves < r1; r += 8 * kWaves) {
      V v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const V *>(g + (r + u * kWaves) * cols + c0));
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[e] += (double)v[u][e];
    }
    for (; r < r1; r += kWaves) {
      const V v = __builtin_nontemporal_load(reinterpret_cast<const V *>(g + r * cols + c0));
#pragma unroll
      for (int e = 0; e < VW; ++e) acc[e] += (double)v[e];
    }
  } else if (c0 < cols) {                       // ragged last columns / unaligned source: scalar loads
    for (int64_t r = r0 + wy; r < r1; r += kWaves)
      for (int e = 0; e < VW && c0 + e < cols; ++e) acc[e] += (double)g[r * cols + c0 + e];
  }
  __shared__ double lds[kWaves][kWave * VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) lds[wy][lane * VW + e] = acc[e];
  __syncthreads();
  if (wy == 0) {
    double *partial = work + a.poff[k] + (int64_t)chunk * cols;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      double s = lds[0][lane * VW + e];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) s += lds[w][lane * VW + e];
      if (c0 + e < cols) partial[c0 + e] = s;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void pn_colsum_finish_kernel(const ColfinArgs<T> a, const double *work) {
  constexpr int kWaves = kBlock / kWave;
  int t = 0;
  while (t + 1 < a.nt && (int)blockIdx.x >= a.first_block[t + 1]) ++t;     // block-uniform
  const int lane = threadIdx.x & (kWave - 1), wy = threadIdx.x / kWave;
  const int64_t cols = a.cols[t];
  const int64_t c = (int64_t)((int)blockIdx.x - a.first_block[t]) * kWave + lane;
  __shared__ double lds[kWaves][kWave];
  T m = (T)0;
  if (wy == 0 && c < cols) m = a.mu[t][c];
  for (int j = a.src_begin[t]; j < a.src_begin[t + 1]; ++j) {
    const double *partial = work + a.poff[j];
    const int chunks = a.chunks[j];
    double s = 0.0;
    if (c < cols) {
      int k = wy;
      for (; k + 7 * kWaves < chunks; k += 8 * kWaves) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(k + u * kWaves) * cols + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; k < chunks; k += kWaves) s += partial[(int64_t)k * cols + c];
    }
    __syncthreads();                 // the previous source's LDS words have been read
    lds[wy][lane] = s;
    __syncthreads();
    if (wy == 0 && c < cols) {
      double tot = lds[0][lane];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) tot += lds[w][lane];
      m = fma(a.alpha[j], (T)tot, m);
    }
  }
  if (wy == 0 && c < cols) a.mu[t][c] = m;
}

// ---------------------------------------------------------------------------------------
// host side: profiling events, launch helpers
// ---------------------------------------------------------------------------------------
struct ProfRec {
  int kid;
  double bytes;
  hipEvent_t e0, e1;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_event_pool;
int64_t g_prof_launches[PN_K_COUNT];
double g_prof_usec[PN_K_COUNT], g_prof_bytes[PN_K_COUNT];

hipEvent_t take_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

int prof_drain_locked() {
  for (auto &r : g_prof_recs) {
    float ms = 0;
    hipError_t err = hipEventSynchronize(r.e1);
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, r.e0, r.e1);
    if (err != hipSuccess) return pn::fail(std::string("prof: ") + hipGetErrorString(err));
    g_prof_launches[r.kid] += 1;
    g_prof_usec[r.kid] += (double)ms * 1e3;
    g_prof_bytes[r.kid] += r.bytes;
    g_event_pool.push_back(r.e0);
    g_event_pool.push_back(r.e1);
  }
  g_prof_recs.clear();
  return 0;
}

template <typename Kern, typename... Args>
int launch_b(int kid, double bytes, Kern kern, dim3 grid, dim3 block, hipStream_t st, Args... args) {
  if (g_prof_on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_recs.size() >= 8192 && prof_drain_locked()) return 1;
    hipEvent_t e0 = take_event(), e1 = take_event();
    if (!e0 || !e1) return pn::fail("prof: hipEventCreate failed");
    hipExtLaunchKernelGGL(kern, grid, block, 0, st, e0, e1, 0, args...);
    g_prof_recs.push_back({kid, bytes, e0, e1});
  } else {
    hipLaunchKernelGGL(kern, grid, block, 0, st, args...);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pn::fail(std::string(pn_kernel_name(kid)) + ": " + hipGetErrorString(err));
  return 0;
}

template <typename Kern, typename... Args>
int launch(int kid, double bytes, Kern kern, dim3 grid, hipStream_t st, Args... args) {
  return launch_b(kid, bytes, kern, grid, dim3(kBlock), st, args...);
}

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
template <typename T> constexpr int vec_width() { return 16 / sizeof(T); }


// Launch geometry of the streaming kernels: BLOCK threads per workgroup, VPT 16-byte vectors
// per thread (all loaded before the first use).  Defaults were picked by timing the target
// configuration in place (bench.py, profiles/); PN_TUNE="vpt=..,block=.." overrides them.
constexpr int kDefaultGridCap = 0;

struct Tune {
  int vpt = 0;
  int block = kBlock;  // threads per workgroup of the streaming kernel: 256 (default), 512 or 1024
  int cap = 0;       // > 0: at most `cap` blocks, grid-stride loop over the rest
  int xcd = 0;       // 1: contiguous eighth of the vector per XCD instead of round-robin tiles (experiment)
  int pvec = 1;      // vectors per thread of pn_param_accum_multi (experiment)
  int pnt = 1;       // non-temporal loads of the gradient tensors in pn_param_accum_multi (read once, cold: -7 % per launch)
  int wvpt = 0;      // vectors per thread of pn_combine_wrms (0: by size)
  int wfin = 0;      // 1: pn_combine_wrms finishes its norm in the launch instead of on the host (experiment)
  int wspin = 0;     // 1: pn_stream_wait_wrms polls hipStreamQuery instead of blocking in hipStreamSynchronize (experiment)
  int ld[PN_K_COUNT], st[PN_K_COUNT];
  Tune() { parse(std::getenv("PN_TUNE")); }
  void parse(const char *e) {
    // defaults chosen by timing the target configuration in place (profiles/)
    vpt = 0;
    block = kBlock;
    cap = kDefaultGridCap;
    xcd = 0;
    pvec = 1;
    pnt = 1;
    wvpt = 0;
    wfin = 0;
    wspin = 0;
    // non-temporal stores: +3..5 % on the solver kernels in place at 8-32 MiB vectors, end-to-end
    // neutral (tools/ab_configs.py, profiles/r01_ab_policy.txt); loads stay plain (operands are hot)
    for (int k = 0; k < PN_K_COUNT; ++k) { ld[k] = 0; st[k] = 1; }
    if (!e) return;
    const char *p;
    if ((p = std::strstr(e, "wvpt="))) wvpt = std::atoi(p + 5);
    if ((p = std::strstr(e, "wfin="))) wfin = std::atoi(p + 5);
    if ((p = std::strstr(e, "wspin="))) wspin = std::atoi(p + 6);
    for (p = e; (p = std::strstr(p, "vpt=")); p += 4)
      if (p == e || (p[-1] != 'w')) { vpt = std::atoi(p + 4); break; }
    if ((p = std::strstr(e, "block="))) block = std::atoi(p + 6);
    if ((p = std::strstr(e, "cap="))) cap = std::atoi(p + 4);
    if ((p = std::strstr(e, "xcd="))) xcd = std::atoi(p + 4);
    if ((p = std::strstr(e, "pvec="))) pvec = std::atoi(p + 5);
    if ((p = std::strstr(e, "pnt="))) pnt = std::atoi(p + 4);
    if ((p = std::strstr(e, "ld="))) for (int k = 0; k < PN_K_COUNT; ++k) ld[k] = std::atoi(p + 3);
    if ((p = std::strstr(e, "st="))) for (int k = 0; k < PN_K_COUNT; ++k) st[k] = std::atoi(p + 3);
    for (int k = 0; k < PN_K_COUNT; ++k) {
      char key[8];
      std::snprintf(key, sizeof key, "ld%d=", k);
      if ((p = std::strstr(e, key))) ld[k] = std::atoi(p + 4);
      std::snprintf(key, sizeof key, "st%d=", k);
      if ((p = std::strstr(e, key))) st[k] = std::atoi(p + 4);
    }
  }
};
Tune &tune() {
  static Tune t;
  return t;
}

inline int pick_vpt(int64_t nvec) {
  const Tune &t = tune();
  if (t.vpt) return t.vpt;
  return nvec < (int64_t)256 * 256 * 8 ? 1 : 2;   // small vectors: spread over as many CUs as possible
}

template <typename T, int NIN, bool OUT2, int VPT, int LD, int ST, int BLOCK = kBlock>
int launch_lincomb_geo(int kid, hipStream_t st, double bytes, const LinArgs<T, NIN> &a, void *out, void *out2,
                       double c2, int64_t nvec, int64_t n) {
  constexpr int VW = vec_width<T>();
  const int64_t per = (int64_t)BLOCK * VPT;
  int64_t nb = (nvec + per - 1) / per, stride = 0;
  if (nb < 1) nb = 1;
  const int cap = tune().cap;
  if (cap > 0 && nb > cap) {
    nb = cap;
    stride = (int64_t)cap * per;
  }
  return launch_b(kid, bytes, pn_lincomb_kernel<T, NIN, VW, VPT, OUT2, BLOCK, LD, ST>, dim3((unsigned)nb), dim3(BLOCK), st,
                  a, (T *)out, (T *)out2, (T)c2, nvec, n, stride, stride == 0 ? tune().xcd : 0);
}

template <typename T, int NIN, bool OUT2>
int launch_lincomb_n(int kid, hipStream_t st, int64_t n, const void *const *x, const double *c, void *out,
                     void *out2, double c2) {
  LinArgs<T, NIN> a;
  bool al = aligned16(out) && (!OUT2 || aligned16(out2));
  for (int j = 0; j < NIN; ++j) {
    a.x[j] = (const T *)x[j];
    a.c[j] = (T)c[j];
    al = al && aligned16(x[j]);
  }
  const double bytes = (double)n * sizeof(T) * (NIN + 1 + (OUT2 ? 1 : 0));
  if (al) {
    constexpr int VW = vec_width<T>();
    const int64_t nvec = n / VW;
    const int vpt = pick_vpt(nvec), ld = tune().ld[kid], stp = tune().st[kid], blk = tune().block;
    if (blk != kBlock) {            // wider workgroups: default cache policy only
#define PN_BLK(B, V) \
  if (blk == B && vpt == V && ld == 0 && stp == 1) \
    return launch_lincomb_geo<T, NIN, OUT2, V, 0, 1, B>(kid, st, bytes, a, out, out2, c2, nvec, n);
      PN_BLK(512, 1) PN_BLK(512, 2) PN_BLK(1024, 1) PN_BLK(1024, 2)
#undef PN_BLK
      return pn::fail("PN_TUNE: block= 512|1024 needs vpt 1|2 and the default cache policy (ld=0,st=1)");
    }
#define PN_GEO(V, L, S) \
  if (vpt == V && ld == L && stp == S) return launch_lincomb_geo<T, NIN, OUT2, V, L, S>(kid, st, bytes, a, out, out2, c2, nvec, n);
    PN_GEO(1, 0, 0) PN_GEO(1, 0, 1) PN_GEO(2, 0, 0) PN_GEO(2, 0, 1) PN_GEO(2, 0, 2) PN_GEO(2, 1, 0) PN_GEO(2, 1, 1)
    PN_GEO(2, 1, 2) PN_GEO(2, 2, 0) PN_GEO(2, 2, 1) PN_GEO(1, 1, 1) PN_GEO(1, 2, 1) PN_GEO(1, 2, 0) PN_GEO(1, 1, 0)
    PN_GEO(4, 0, 1)
#undef PN_GEO
    return pn::fail("PN_TUNE: unsupported (vpt, ld, st) combination");
  }
  dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
  return launch(kid, bytes, pn_lincomb_kernel<T, NIN, 1, 1, OUT2, kBlock>, grid, st, a, (T *)out, (T *)out2, (T)c2, n, n,
                (int64_t)0, 0);
}

template <typename T, bool OUT2>
int launch_lincomb(int kid, hipStream_t st, int64_t n, int nin, const void *const *x, const double *c, void *out,
                   void *out2, double c2) {
  switch (nin) {
#define PN_CASE(N) \
  case N:          \
    return launch_lincomb_n<T, N, OUT2>(kid, st, n, x, c, out, out2, c2);
    PN_CASE(1) PN_CASE(2) PN_CASE(3) PN_CASE(4) PN_CASE(5) PN_CASE(6) PN_CASE(7) PN_CASE(8)
#undef PN_CASE
    default:
      return pn::fail("lincomb: between 1 and 8 input vectors supported");
  }
}

int lincomb(int kid, void *stream, int dtype, int64_t n, int nin, const void *const *x, const double *c, void *out,
            void *out2, double c2) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PN_F32)
    return out2 ? launch_lincomb<float, true>(kid, st, n, nin, x, c, out, out2, c2)
                : launch_lincomb<float, false>(kid, st, n, nin, x, c, out, nullptr, 0);
  if (dtype == PN_F64)
    return out2 ? launch_lincomb<double, true>(kid, st, n, nin, x, c, out, out2, c2)
                : launch_lincomb<double, false>(kid, st, n, nin, x, c, out, nullptr, 0);
  return pn::fail("dtype must be PN_F32 or PN_F64");
}

inline int pick_vpt_wrms(int64_t nvec) {
  const Tune &t = tune();
  if (t.wvpt) return t.wvpt;
  // small vectors spread over as many CUs as possible; at 8 MiB and beyond four vectors per thread (28 loads of 16 bytes in
  // flight per thread at C3b) measured 3-4 % faster than two on two boxes (profiles/r03_c3b_*_trace_stats.csv)
  return nvec < (int64_t)256 * 256 * 8 ? 1 : 4;
}

template <typename T, int NK, bool WRITE>
int launch_wrms_n(hipStream_t st, int64_t n, void *unew, const void *u, const void *const *K, const double *cb,
                  const double *ce, double atol, double rtol, double *work, double *result) {
  ErrArgs<T, NK> a;
  bool al = aligned16(u) && (!WRITE || aligned16(unew));
  for (int j = 0; j < NK; ++j) {
    a.k[j] = (const T *)K[j];
    a.cb[j] = (T)(cb ? cb[j] : 0.0);
    a.ce[j] = (T)ce[j];
    al = al && aligned16(K[j]);
  }
  const double bytes = (double)n * sizeof(T) * (NK + 1 + (WRITE ? 1 : 0));
  const double inv_n = 1.0 / (double)n;
  if (al) {
    constexpr int VW = vec_width<T>();
    const int64_t nvec = n / VW;
    const int vpt = pick_vpt_wrms(nvec), stp = tune().st[PN_K_COMBINE_WRMS] != 0 ? 1 : 0;
    const int64_t per = (int64_t)kBlock * vpt;
    int nblocks = (int)((nvec + per - 1) / per);
    if (nblocks < 1) nblocks = 1;
    const int fin = tune().wfin != 0 ? 1 : 0;
#define PN_WGEO(V, S, F)                                                                                              \
  if (vpt == V && stp == S && fin == F)                                                                               \
    return launch(PN_K_COMBINE_WRMS, bytes, pn_combine_wrms_kernel<T, NK, VW, V, WRITE, S, F>, dim3(nblocks), st, (const T *)u, \
                  a, (T *)unew, atol, rtol, work, nvec, n, inv_n, result);
    PN_WGEO(1, 0, 0) PN_WGEO(1, 1, 0) PN_WGEO(2, 0, 0) PN_WGEO(2, 1, 0) PN_WGEO(4, 0, 0) PN_WGEO(4, 1, 0)
    PN_WGEO(1, 1, 1) PN_WGEO(2, 1, 1) PN_WGEO(4, 1, 1)
#undef PN_WGEO
    return pn::fail("PN_TUNE: wvpt must be 1, 2 or 4 (wfin=1 needs the default store policy)");
  }
  const int nblocks = (int)((n + kBlock - 1) / kBlock);
  return launch(PN_K_COMBINE_WRMS, bytes, pn_combine_wrms_kernel<T, NK, 1, 1, WRITE, 0, 0>, dim3(nblocks), st, (const T *)u, a,
                (T *)unew, atol, rtol, work, n, n, inv_n, result);
}

template <typename T, bool WRITE>
int launch_wrms(hipStream_t st, int64_t n, int nk, void *unew, const void *u, const void *const *K, const double *cb,
                const double *ce, double atol, double rtol, double *partial, double *result) {
  switch (nk) {
#define PN_CASE(N) \
  case N:          \
    return launch_wrms_n<T, N, WRITE>(st, n, unew, u, K, cb, ce, atol, rtol, partial, result);
    PN_CASE(1) PN_CASE(2) PN_CASE(3) PN_CASE(4) PN_CASE(5) PN_CASE(6) PN_CASE(7)
#undef PN_CASE
    default:
      return pn::fail("combine_wrms: between 1 and 7 stage derivatives supported");
  }
}

template <typename T, int NK>
int launch_dots_n(hipStream_t st, int64_t n, const void *x, const void *const *y, double *work, double *result) {
  DotArgs<T, NK> a;
  bool al = aligned16(x);
  for (int j = 0; j < NK; ++j) {
    a.y[j] = (const T *)y[j];
    al = al && aligned16(y[j]);
  }
  const double bytes = (double)n * sizeof(T) * (NK + 1);
  if (al) {
    constexpr int VW = vec_width<T>();
    const int64_t nvec = n / VW;
    int nblocks = (int)((nvec + kBlock - 1) / kBlock);
    if (nblocks < 1) nblocks = 1;
    return launch(PN_K_DOTS, bytes, pn_dots_kernel<T, NK, VW>, dim3(nblocks), st, (const T *)x, a, work, nvec, n, result);
  }
  const int nblocks = (int)((n + kBlock - 1) / kBlock);
  return launch(PN_K_DOTS, bytes, pn_dots_kernel<T, NK, 1>, dim3(nblocks), st, (const T *)x, a, work, n, n, result);
}

template <typename T>
int launch_dots(hipStream_t st, int64_t n, int nk, const void *x, const void *const *y, double *partial, double *result) {
  switch (nk) {
#define PN_CASE(N) \
  case N:          \
    return launch_dots_n<T, N>(st, n, x, y, partial, result);
    PN_CASE(1) PN_CASE(2) PN_CASE(3) PN_CASE(4) PN_CASE(5) PN_CASE(6) PN_CASE(7) PN_CASE(8)
#undef PN_CASE
    default:
      return pn::fail("pn_dots: between 1 and 8 vectors per call");
  }
}

}  // namespace

int pn::prof_events(int kid, double bytes, void **e0, void **e1) {
  if (!g_prof_on) return 0;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_recs.size() >= 8192 && prof_drain_locked()) return -1;
  hipEvent_t a = take_event(), b = take_event();
  if (!a || !b) {
    pn::fail("prof: hipEventCreate failed");
    return -1;
  }
  g_prof_recs.push_back({kid, bytes, a, b});
  *e0 = a;
  *e1 = b;
  return 1;
}

// =========================================================================================
// C ABI
// =========================================================================================
template <typename T, int VW, int P, bool NT>
static int param_accum_multi_t(hipStream_t st, T *mu, int nsrc, const double *alpha, int nseg, const void *const *g,
                               const int64_t *offset, const int64_t *len) {
  const int64_t per_block = (int64_t)kBlock * VW * P;
  int k = 0;
  while (k < nseg) {
    MultiSegArgs<T> a;
    int m = 0, blocks = 0, np = 0;
    double bytes = 0;
    while (k < nseg && m < kMaxSegM) {
      int live = 0;
      for (int j = 0; j < nsrc; ++j) live += g[(size_t)j * nseg + k] != nullptr;
      if (live && len[k] > 0) {
        if (np + live > kMaxPtr) break;          // pointer table full: this tensor goes into the next launch
        a.pbase[m] = (short)np; a.cnt[m] = (short)live;
        for (int j = 0; j < nsrc; ++j)
          if (g[(size_t)j * nseg + k]) { a.g[np] = (const T *)g[(size_t)j * nseg + k]; a.src[np] = (unsigned char)j; ++np; }
        a.off[m] = offset[k]; a.len[m] = len[k]; a.first_block[m] = blocks;
        blocks += (int)((len[k] + per_block - 1) / per_block);
        bytes += (2.0 + live) * (double)len[k] * sizeof(T);
        ++m;
      }
      ++k;
    }
    if (m == 0) {
      if (k < nseg) return pn::fail("pn_param_accum_multi: pointer table overflow");
      break;
    }
    a.first_block[m] = blocks; a.nseg = m;
    for (int j = 0; j < nsrc; ++j) a.alpha[j] = (T)alpha[j];
    int rc = launch(PN_K_PARAM_ACCUM, bytes, pn_param_accum_multi_kernel<T, VW, P, NT>, dim3(blocks), st, a, mu);
    if (rc) return rc;
  }
  return 0;
}

namespace {
static int colsum_chunks(int64_t rows) {
  // Row chunks of a source: 512 rows each (at most 64 chunks).  A function of the source's shape ONLY -- the partial sums,
  // hence every bit of the result, must not depend on how many sources share a launch (the engine groups them differently
  // under -pn_param_accum stage | step | batch).  The batched launches of the default mode (32 sources) fill the chip with
  // 8 chunks x 2 column tiles per 4096 x 512 source; a lone source runs on few blocks -- that is the non-default mode.
  int64_t c = (rows + 511) / 512;
  if (c > 64) c = 64;
  return (int)(c < 1 ? 1 : c);
}

template <typename T, int VW>
static int colsum_multi_t(hipStream_t st, int nsrc, const int64_t *rows, const int64_t *cols, const void *const *g, void *const *mu,
                          const double *alpha, double *work) {
  ColsumArgs<T> a;
  ColfinArgs<T> f;
  a.nsrc = nsrc;
  int blocks = 0;
  int64_t poff = 0;
  double bytes = 0;
  int chunks[kMaxColSrc];
  for (int j = 0; j < nsrc; ++j) {
    a.g[j] = (const T *)g[j]; a.rows[j] = rows[j]; a.cols[j] = cols[j]; a.poff[j] = poff;
    chunks[j] = colsum_chunks(rows[j]);
    a.rpb[j] = (int)((rows[j] + chunks[j] - 1) / chunks[j]);
    a.ctiles[j] = (int)((cols[j] + kWave * VW - 1) / (kWave * VW));
    a.first_block[j] = blocks;
    blocks += a.ctiles[j] * chunks[j];
    poff += (int64_t)chunks[j] * cols[j];
    bytes += (double)rows[j] * (double)cols[j] * sizeof(T);
  }
  a.first_block[nsrc] = blocks;
  // targets in order of first appearance; within a target the sources keep their order
  int nt = 0, fblocks = 0, filled = 0;
  bool done[kMaxColSrc] = {false};
  for (int j = 0; j < nsrc; ++j) {
    if (done[j]) continue;
    f.mu[nt] = (T *)mu[j]; f.cols[nt] = cols[j]; f.first_block[nt] = fblocks; f.src_begin[nt] = filled;
    for (int q = j; q < nsrc; ++q) {
      if (!done[q] && mu[q] == mu[j]) {
        if (cols[q] != cols[j]) return pn::fail("pn_colsum_accum_multi: sources of one mu must have the same number of columns");
        done[q] = true;
        f.poff[filled] = a.poff[q]; f.chunks[filled] = chunks[q]; f.alpha[filled] = (T)alpha[q];
        ++filled;
      }
    }
    fblocks += (int)((cols[j] + kWave - 1) / kWave);
    bytes += 2.0 * (double)cols[j] * sizeof(T);
    ++nt;
  }
  f.first_block[nt] = fblocks; f.src_begin[nt] = filled; f.nt = nt;
  if (blocks == 0) return 0;
  int rc = launch(PN_K_PARAM_ACCUM, bytes, pn_colsum_partial_kernel<T, VW>, dim3((unsigned)blocks), st, a, work);
  if (rc) return rc;
  hipLaunchKernelGGL(pn_colsum_finish_kernel<T>, dim3((unsigned)fblocks), dim3(kBlock), 0, st, f, (const double *)work);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pn::fail(std::string("pn_colsum_accum_multi: ") + hipGetErrorString(err));
  return 0;
}

}  // namespace

extern "C" {

const char *pn_kernel_name(int kid) {
  static const char *names[PN_K_COUNT] = {"pn_rk_stage", "pn_rk_combine_wrms", "pn_adj_theta",
                                          "pn_adj_accum", "pn_param_accum", "pn_copy", "pn_dots", "pn_lincomb",
                                          "pn_linear_wgrad"};
  return kid >= 0 && kid < PN_K_COUNT ? names[kid] : "?";
}

int pn_rk_stage(void *stream, int dtype, int64_t n, void *y, const void *u, int nk, const void *const *K,
                const double *coef) {
  if (nk < 0 || nk > PN_MAX_TERMS - 1) return pn::fail("pn_rk_stage: nk out of range");
  const void *x[PN_MAX_TERMS];
  double c[PN_MAX_TERMS];
  x[0] = u;
  c[0] = 1.0;
  for (int j = 0; j < nk; ++j) {
    x[j + 1] = K[j];
    c[j + 1] = coef[j];
  }
  return lincomb(PN_K_STAGE, stream, dtype, n, nk + 1, x, c, y, nullptr, 0);
}

int64_t pn_wrms_work_bytes(int64_t n) { return (int64_t)sizeof(double) * ((n + kBlock - 1) / kBlock + 4 + kTicketDoubles); }

int pn_rk_combine_wrms(void *stream, int dtype, int64_t n, void *unew, const void *u, int nk, const void *const *K,
                       const double *coef_b, const double *coef_e, double atol, double rtol, void *work,
                       double *result_dev) {
  if (n <= 0) return pn::fail("pn_rk_combine_wrms: empty vector");
  if (!work || !result_dev) return pn::fail("pn_rk_combine_wrms: work/result buffers required");
  hipStream_t st = (hipStream_t)stream;
  double *w = (double *)work;
  if (dtype == PN_F32)
    return unew ? launch_wrms<float, true>(st, n, nk, unew, u, K, coef_b, coef_e, atol, rtol, w, result_dev)
                : launch_wrms<float, false>(st, n, nk, nullptr, u, K, nullptr, coef_e, atol, rtol, w, result_dev);
  if (dtype == PN_F64)
    return unew ? launch_wrms<double, true>(st, n, nk, unew, u, K, coef_b, coef_e, atol, rtol, w, result_dev)
                : launch_wrms<double, false>(st, n, nk, nullptr, u, K, nullptr, coef_e, atol, rtol, w, result_dev);
  return pn::fail("dtype must be PN_F32 or PN_F64");
}

int pn_pinned_scalar(double **host_ptr, double **dev_ptr) {
  void *h = nullptr, *d = nullptr;
  hipError_t err = hipHostMalloc(&h, 64, hipHostMallocMapped);
  if (err == hipSuccess) err = hipHostGetDevicePointer(&d, h, 0);
  if (err != hipSuccess) return pn::fail(std::string("pn_pinned_scalar: ") + hipGetErrorString(err));
  *(double *)h = -1.0;
  *host_ptr = (double *)h;
  *dev_ptr = (double *)d;
  return 0;
}

int pn_pinned_block(int64_t nbytes, double **host_ptr, double **dev_ptr) {
  void *h = nullptr, *d = nullptr;
  if (nbytes < 64) nbytes = 64;
  hipError_t err = hipHostMalloc(&h, (size_t)nbytes, hipHostMallocMapped);
  if (err == hipSuccess) err = hipHostGetDevicePointer(&d, h, 0);
  if (err != hipSuccess) return pn::fail(std::string("pn_pinned_block: ") + hipGetErrorString(err));
  std::memset(h, 0, (size_t)nbytes);
  *host_ptr = (double *)h;
  *dev_ptr = (double *)d;
  return 0;
}

int pn_pinned_free(double *host_ptr) {
  if (!host_ptr) return 0;
  hipError_t err = hipHostFree(host_ptr);
  if (err != hipSuccess) return pn::fail(std::string("pn_pinned_free: ") + hipGetErrorString(err));
  return 0;
}

int pn_stream_create(int priority, void **stream) {
  if (!stream) return pn::fail("pn_stream_create: null argument");
  int least = 0, greatest = 0;
  hipError_t err = hipDeviceGetStreamPriorityRange(&least, &greatest);      // numerically: least >= greatest
  if (err != hipSuccess) return pn::fail(std::string("pn_stream_create: ") + hipGetErrorString(err));
  const int prio = priority > 0 ? least : (priority < 0 ? greatest : 0);
  hipStream_t st = nullptr;
  err = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio);
  if (err != hipSuccess) return pn::fail(std::string("pn_stream_create: ") + hipGetErrorString(err));
  *stream = (void *)st;
  return 0;
}

int pn_stream_destroy(void *stream) {
  if (!stream) return 0;
  hipError_t err = hipStreamDestroy((hipStream_t)stream);
  if (err != hipSuccess) return pn::fail(std::string("pn_stream_destroy: ") + hipGetErrorString(err));
  return 0;
}

int64_t pn_wrms_partials(int64_t n) { return (n + kBlock - 1) / kBlock + 2; }

int pn_stream_wait_wrms(void *stream, const double *host_ptr, int64_t n, double *value) {
  hipError_t err;
  if (tune().wspin) {
    // poll: the runtime's blocking wait spins briefly and then sleeps on an interrupt, whose wake-up is part of every
    // step attempt of an adaptive solve (PN_TUNE "wspin=1")
    while ((err = hipStreamQuery((hipStream_t)stream)) == hipErrorNotReady) {}
  } else {
    err = hipStreamSynchronize((hipStream_t)stream);
  }
  if (err != hipSuccess) return pn::fail(std::string("pn_stream_wait_wrms: ") + hipGetErrorString(err));
  const volatile double *p = (const volatile double *)host_ptr;
  const int64_t nb = (int64_t)p[0];
  if (nb <= 0) {                       // finished on the device
    *value = p[1];
    return 0;
  }
  // the block holds pn_wrms_partials(n) doubles: the count and at most that many minus one partials
  if (nb + 1 > pn_wrms_partials(n)) return pn::fail("pn_stream_wait_wrms: corrupt block count");
  double s = 0;
  for (int64_t i = 0; i < nb; ++i) s += p[1 + i];      // index order: bit-reproducible
  *value = std::sqrt(s * (1.0 / (double)n));           // the expression of the in-launch finish (PN_TUNE wfin=1): same bits
  return 0;
}

int pn_stream_wait_scalar(void *stream, const double *host_ptr, double *value) {
  hipError_t err = hipStreamSynchronize((hipStream_t)stream);
  if (err != hipSuccess) return pn::fail(std::string("pn_stream_wait_scalar: ") + hipGetErrorString(err));
  *value = *(const volatile double *)host_ptr;
  return 0;
}

int pn_lincomb(void *stream, int dtype, int64_t n, void *out, int nin, const void *const *x, const double *c) {
  if (nin < 1 || nin > PN_MAX_TERMS) return pn::fail("pn_lincomb: between 1 and 8 input vectors");
  return lincomb(PN_K_LINCOMB, stream, dtype, n, nin, x, c, out, nullptr, 0);
}

int64_t pn_dots_work_bytes(int64_t n) {
  return (int64_t)sizeof(double) * (PN_MAX_TERMS * ((n + kBlock - 1) / kBlock + 1) + 4 + kTicketDoubles);
}

int pn_dots(void *stream, int dtype, int64_t n, const void *x, int nk, const void *const *y, void *work,
            double *result_dev) {
  if (n <= 0) return pn::fail("pn_dots: empty vector");
  if (!work || !result_dev) return pn::fail("pn_dots: work/result buffers required");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PN_F32) return launch_dots<float>(st, n, nk, x, y, (double *)work, result_dev);
  if (dtype == PN_F64) return launch_dots<double>(st, n, nk, x, y, (double *)work, result_dev);
  return pn::fail("dtype must be PN_F32 or PN_F64");
}

int pn_stream_wait_scalars(void *stream, const double *host_ptr, int count, double *values) {
  hipError_t err = hipStreamSynchronize((hipStream_t)stream);
  if (err != hipSuccess) return pn::fail(std::string("pn_stream_wait_scalars: ") + hipGetErrorString(err));
  for (int i = 0; i < count; ++i) values[i] = ((const volatile double *)host_ptr)[i];
  return 0;
}

int pn_adj_theta(void *stream, int dtype, int64_t n, void *w, const void *lambda, double c_lam, int nk,
                 const void *const *dlam, const double *coef) {
  const void *x[PN_MAX_TERMS];
  double c[PN_MAX_TERMS];
  int m = 0;
  if (lambda) {
    x[m] = lambda;
    c[m++] = c_lam;
  }
  if (nk < 0 || m + nk > PN_MAX_TERMS) return pn::fail("pn_adj_theta: nk out of range");
  for (int j = 0; j < nk; ++j) {
    x[m] = dlam[j];
    c[m++] = coef[j];
  }
  if (m == 0) return pn::fail("pn_adj_theta: nothing to combine (structurally zero stage)");
  return lincomb(PN_K_ADJ_THETA, stream, dtype, n, m, x, c, w, nullptr, 0);
}

int pn_adj_accum(void *stream, int dtype, int64_t n, void *lambda_out, const void *lambda, int nk,
                 const void *const *dlam, const double *coef, const void *forcing, void *w_next, double c_next) {
  const void *x[PN_MAX_TERMS];
  double c[PN_MAX_TERMS];
  int m = 0;
  x[m] = lambda;
  c[m++] = 1.0;
  if (nk < 0 || m + nk + (forcing ? 1 : 0) > PN_MAX_TERMS) return pn::fail("pn_adj_accum: nk out of range");
  for (int j = 0; j < nk; ++j) {
    x[m] = dlam[j];
    c[m++] = coef ? coef[j] : 1.0;
  }
  if (forcing) {
    x[m] = forcing;
    c[m++] = 1.0;
  }
  return lincomb(PN_K_ADJ_ACCUM, stream, dtype, n, m, x, c, lambda_out, w_next, c_next);
}

int pn_copy(void *stream, int dtype, int64_t n, void *y, const void *x) {
  const void *xs[1] = {x};
  const double c[1] = {1.0};
  return lincomb(PN_K_COPY, stream, dtype, n, 1, xs, c, y, nullptr, 0);
}

int pn_zero(void *stream, int dtype, int64_t n, void *y) {
  if (n <= 0) return 0;
  hipError_t err = hipMemsetAsync(y, 0, (size_t)n * (dtype == PN_F32 ? 4 : 8), (hipStream_t)stream);
  if (err != hipSuccess) return pn::fail(std::string("pn_zero: ") + hipGetErrorString(err));
  return 0;
}

int pn_param_accum(void *stream, int dtype, void *mu, double alpha, int nseg, const void *const *g,
                   const int64_t *offset, const int64_t *len) {
  hipStream_t st = (hipStream_t)stream;
  const int esize = dtype == PN_F32 ? 4 : 8;
  const int vw = 16 / esize;
  const int64_t per_block = (int64_t)kBlock * vw * 2;
  int k = 0;
  while (k < nseg) {
    SegArgs<float> af;
    SegArgs<double> ad;
    int m = 0, blocks = 0;
    double bytes = 0;
    while (k < nseg && m < kMaxSeg) {
      if (g[k] && len[k] > 0) {
        const int nb = (int)((len[k] + per_block - 1) / per_block);
        if (dtype == PN_F32) {
          af.g[m] = (const float *)g[k]; af.off[m] = offset[k]; af.len[m] = len[k]; af.first_block[m] = blocks;
        } else {
          ad.g[m] = (const double *)g[k]; ad.off[m] = offset[k]; ad.len[m] = len[k]; ad.first_block[m] = blocks;
        }
        blocks += nb;
        bytes += 3.0 * (double)len[k] * esize;
        ++m;
      }
      ++k;
    }
    if (m == 0) break;
    int rc;
    if (dtype == PN_F32) {
      af.first_block[m] = blocks; af.nseg = m;
      rc = launch(PN_K_PARAM_ACCUM, bytes, pn_param_accum_kernel<float, 4>, dim3(blocks), st, af, (float *)mu, (float)alpha);
    } else if (dtype == PN_F64) {
      ad.first_block[m] = blocks; ad.nseg = m;
      rc = launch(PN_K_PARAM_ACCUM, bytes, pn_param_accum_kernel<double, 2>, dim3(blocks), st, ad, (double *)mu, alpha);
    } else {
      return pn::fail("dtype must be PN_F32 or PN_F64");
    }
    if (rc) return rc;
  }
  return 0;
}

int pn_param_accum_multi(void *stream, int dtype, void *mu, int nsrc, const double *alpha, int nseg,
                         const void *const *g, const int64_t *offset, const int64_t *len) {
  if (nsrc < 1 || nsrc > kMaxSrc) return pn::fail("pn_param_accum_multi: nsrc must be in 1..32");
  if (nseg <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool two = tune().pvec == 2;          // experiment: two vectors per thread (PN_TUNE "pvec=2")
  const bool nt = tune().pnt != 0;            // non-temporal loads of the gradient tensors (PN_TUNE "pnt=0|1")
#define PN_ACC(T, VW)                                                                                          \
  return two ? (nt ? param_accum_multi_t<T, VW, 2, true>(st, (T *)mu, nsrc, alpha, nseg, g, offset, len)       \
                   : param_accum_multi_t<T, VW, 2, false>(st, (T *)mu, nsrc, alpha, nseg, g, offset, len))     \
             : (nt ? param_accum_multi_t<T, VW, 1, true>(st, (T *)mu, nsrc, alpha, nseg, g, offset, len)       \
                   : param_accum_multi_t<T, VW, 1, false>(st, (T *)mu, nsrc, alpha, nseg, g, offset, len));
  if (dtype == PN_F32) { PN_ACC(float, 4) }
  if (dtype == PN_F64) { PN_ACC(double, 2) }
#undef PN_ACC
  return pn::fail("dtype must be PN_F32 or PN_F64");
}

int64_t pn_colsum_work_bytes(int nsrc, const int64_t *rows, const int64_t *cols) {
  int64_t n = 0;
  for (int j = 0; j < nsrc; ++j) n += (int64_t)colsum_chunks(rows[j]) * cols[j];
  return n * (int64_t)sizeof(double);
}

int pn_colsum_accum_multi(void *stream, int dtype, int nsrc, const int64_t *rows, const int64_t *cols, const void *const *g,
                          void *const *mu, const double *alpha, void *work) {
  if (nsrc < 1 || nsrc > kMaxColSrc) return pn::fail("pn_colsum_accum_multi: nsrc must be in 1..32");
  for (int j = 0; j < nsrc; ++j)
    if (rows[j] <= 0 || cols[j] <= 0) return pn::fail("pn_colsum_accum_multi: empty source");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PN_F32) return colsum_multi_t<float, 4>(st, nsrc, rows, cols, g, mu, alpha, (double *)work);
  if (dtype == PN_F64) return colsum_multi_t<double, 2>(st, nsrc, rows, cols, g, mu, alpha, (double *)work);
  return pn::fail("dtype must be PN_F32 or PN_F64");
}

int pn_colsum_accum(void *stream, int dtype, int64_t rows, int64_t cols, const void *g, void *mu, double alpha, void *work) {
  if (rows <= 0 || cols <= 0) return 0;
  const void *gs[1] = {g};
  void *mus[1] = {mu};
  return pn_colsum_accum_multi(stream, dtype, 1, &rows, &cols, gs, mus, &alpha, work);
}

int pn_tune_set(const char *spec) {
  tune().parse(spec);
  return 0;
}

int pn_prof_is_enabled(void) { return g_prof_on ? 1 : 0; }

int pn_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (on && !g_prof_on) {
    for (int i = 0; i < PN_K_COUNT; ++i) { g_prof_launches[i] = 0; g_prof_usec[i] = 0; g_prof_bytes[i] = 0; }
  }
  g_prof_on = on != 0;
  return 0;
}

int pn_prof_collect(int count, int64_t *launches, double *usec, double *bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (prof_drain_locked()) return 1;
  for (int i = 0; i < PN_K_COUNT; ++i) {
    if (i < count) { launches[i] = g_prof_launches[i]; usec[i] = g_prof_usec[i]; bytes[i] = g_prof_bytes[i]; }
    g_prof_launches[i] = 0; g_prof_usec[i] = 0; g_prof_bytes[i] = 0;
  }
  return 0;
}

}  // extern "C"
