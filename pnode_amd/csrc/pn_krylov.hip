// pnode_amd -- device-resident restarted GMRES for the matrix-free Newton-Krylov stage solves of the implicit
// steppers (TS types BE / CN / ARKIMEX with the reference's default linear_solver="petsc": KSPGMRES behind
// ts.getSNES().getKSP(), reference pnode/petsc_adjoint.py:547, 581, 651-656, 701-702; the operator is the
// matrix-free shell IJacShell.mult / multTranspose, pa.py:98-197, which stays above this ABI).
//
// Round 2 ran the small dense part of GMRES on the host (pn_gmres_*): one stream synchronisation per Krylov
// iteration to fetch the Gram-Schmidt products.  Here everything GMRES decides lives in a state block in HBM:
// the products of the iteration in flight, the Hessenberg column, the Givens rotations, the residual estimate, the
// iteration counters and a `stop` flag.  The host only enqueues -- operator application (PyTorch, or a replayed
// hipGraph of it) followed by pn_krylov_step -- and looks at the state when it chooses to (once per chunk of
// iterations); once `stop` is set every later launch of the cycle returns at its entry check, so enqueueing past
// convergence is harmless.  The arithmetic and every decision are those of the host path (pnode_amd/theta.py
// _gmres: classical Gram-Schmidt, ||w - sum h_j V_j|| by Pythagoras, a second pass when more than 3/4 of ||w||^2
// cancels, Givens rotations, residual test after every column): iteration counts are the same.
//
// Per iteration k (w = A V_k already in `w`):
//   kr_dots   pass 1   d_j = <w, V_j> (j <= k), ww = <w, w>; the block that arrives last adds the block partials in
//                      index order and DECIDES: normal column (hk1 = sqrt(ww - sum d_j^2)) or second pass needed
//   kr_update          V_{k+1} = (w - sum d_j V_j)/hk1  (also written to `vin`, the operator's input buffer), or
//                      w <- w - sum d_j V_j in place when a second pass follows
//   kr_dots   pass 2   (returns at entry unless asked for) d2_j = <w, V_j>, ww2 = <w, w>; decides the column
//   kr_update          (same) V_{k+1} = (w - sum d2_j V_j)/hk1
// Cycle end: kr_close back-substitutes y on the device, kr_update adds sum y_j V_j to x.
// With `defer` (several ranks: the products must be summed over the ranks first) the decisions run in one-thread
// kernels of their own, after the caller's all-reduce of the product block.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>

#include "pnode_amd.h"
#include "pn_internal.h"
#include "pn_device.h"

namespace {

constexpr int kMaxRestart = 126;      // coefficient table of kr_update lives in LDS (restart + 2 entries)

// ---- layout of the state block, in doubles ----
enum {
  S_TICKET = 0,                                                       // arrival counters of kr_dots (pn_device.h)
  S_STOP = kTicketDoubles, S_KDONE, S_CUR, S_PHASE, S_TOTAL, S_MAXIT, S_APPLY, S_CLOSED,
  S_BETA, S_BNORM, S_TOL, S_RES, S_HK1, S_RTOL, S_ATOL, S_BREAK, S_NT, S_SPARE,
  S_VEC = kTicketDoubles + 20
};
// stop: 0 running, 1 converged (residual estimate <= tol), 2 happy breakdown (hk1 == 0), 3 iteration limit,
//       4 not-a-number in a norm, 5 singular Hessenberg at the back substitution
struct Lay {
  int m, h, d, c, hess, cs, sn, g, y, part;
};
__host__ __device__ inline Lay layout(int m) {
  Lay L;
  L.m = m;
  L.h = S_VEC;                       // products of the pass in flight: h[0..k] and <w,w>
  L.d = L.h + (m + 2);               // first-pass coefficients, kept for the second pass
  L.c = L.d + (m + 2);               // coefficient table of the next kr_update: c[0] for w, c[1+j] for V_j
  L.hess = L.c + (m + 2);            // Hessenberg matrix, column k at hess + k*(m+1)
  L.cs = L.hess + (m + 1) * m;
  L.sn = L.cs + m;
  L.g = L.sn + m;
  L.y = L.g + (m + 1);
  L.part = (L.y + m + 1) / 2 * 2;    // block partials of kr_dots: (m + 2) rows of nblocks
  return L;
}

// ---- the decisions.  One thread decides, but nothing it touches comes from global memory one dependent load after the
// other (a column's worth of rotations read that way cost 30-40 us per iteration, found in round 3): the block first
// brings the header scalars, the rotations and the coefficients of a first pass into LDS in parallel, thread 0 works on
// LDS, and the block writes the results back in parallel.
constexpr int kHdr = 20;                 // S[S_STOP .. S_SPARE] and spares
struct DecideLds {
  double h[kMaxRestart + 2];             // in: the products of the pass (summed over the ranks where there are several)
  double d[kMaxRestart + 2];             // first-pass coefficients (pass 2: in; pass 1 asking for a second pass: out)
  double c[kMaxRestart + 2];             // out: coefficient table of the next kr_update
  double col[kMaxRestart + 2];           // out: rotated Hessenberg column
  double cs[kMaxRestart + 1], sn[kMaxRestart + 1];
  double hdr[kHdr];
  double gk, gk1;
  int column, refine, ncoef;
};
#define HDR(W, F) (W).hdr[(F) - S_STOP]

// KSPGMRES's update of the Hessenberg column k, as pn_gmres_column does it on the host; thread 0, LDS only
__device__ inline void finish_column(DecideLds &W, int k, double hk1, bool refined) {
  double *col = W.col;
  for (int j = 0; j <= k; ++j) col[j] = refined ? W.d[j] + W.h[j] : W.h[j];
  col[k + 1] = hk1;
  for (int i = 0; i < k; ++i) {
    const double t = W.cs[i] * col[i] + W.sn[i] * col[i + 1];
    col[i + 1] = -W.sn[i] * col[i] + W.cs[i] * col[i + 1];
    col[i] = t;
  }
  const double a = col[k], b = col[k + 1];
  const double r = hypot(a, b);
  if (r == 0.0) {
    W.cs[k] = 1.0; W.sn[k] = 0.0;
  } else {
    W.cs[k] = a / r; W.sn[k] = b / r;
  }
  col[k] = r;
  col[k + 1] = 0.0;
  W.gk1 = -W.sn[k] * W.gk;
  W.gk = W.cs[k] * W.gk;
  const double res = fabs(W.gk1);
  HDR(W, S_RES) = res;
  HDR(W, S_HK1) = hk1;
  HDR(W, S_KDONE) = k + 1;
  const double total = HDR(W, S_TOTAL) + 1.0;
  HDR(W, S_TOTAL) = total;
  int stop = 0;
  if (!(res == res)) stop = 4;
  else if (res <= HDR(W, S_TOL)) stop = 1;
  else if (hk1 == 0.0) stop = 2;
  else if (total >= HDR(W, S_MAXIT)) stop = 3;
  HDR(W, S_STOP) = stop;
  W.column = 1;
}

// mode 0: start of a cycle, W.h[0] = <r, r>.  mode 1: after pass 1 of iteration k, W.h[j] = <w, V_j> (j <= k),
// W.h[k+1] = <w, w>.  mode 2: after pass 2, the same of the once-orthogonalised w.  Thread 0, LDS only.
__device__ inline void decide_in_lds(DecideLds &W, int m, int mode, int k, int first, double rtol, double atol, double maxit) {
  W.column = 0; W.refine = 0; W.ncoef = 0;
  if (mode == 0) {
    const double rr = W.h[0];
    const double beta = sqrt(fmax(rr, 0.0));
    int stop;
    HDR(W, S_BETA) = beta;
    if (first) {
      HDR(W, S_RTOL) = rtol; HDR(W, S_ATOL) = atol; HDR(W, S_MAXIT) = maxit;
      HDR(W, S_BNORM) = beta;
      HDR(W, S_TOL) = fmax(rtol * beta, atol);
      HDR(W, S_TOTAL) = 0; HDR(W, S_BREAK) = 0; HDR(W, S_SPARE) = 0;      // (S_SPARE counts the second passes of the solve)
      HDR(W, S_RES) = beta;
      stop = (beta == 0.0 || beta <= atol) ? 1 : 0;
    } else {
      stop = beta <= HDR(W, S_TOL) ? 1 : 0;
      if (stop) HDR(W, S_RES) = beta;
      if (!stop && HDR(W, S_TOTAL) >= HDR(W, S_MAXIT)) stop = 3;
    }
    if (!(rr == rr)) stop = 4;
    HDR(W, S_STOP) = stop; HDR(W, S_KDONE) = 0; HDR(W, S_CUR) = -1; HDR(W, S_PHASE) = 0; HDR(W, S_CLOSED) = 0;
    HDR(W, S_APPLY) = 0; HDR(W, S_NT) = 0;
    if (!stop) {
      W.gk = beta;                              // g[0]; g[1..m] = 0 written by the block
      W.c[0] = 1.0 / beta;                      // V_0 = r / beta
      W.ncoef = 1;
    }
    return;
  }
  const double ww = W.h[k + 1];
  double ssq = 0.0;
  for (int j = 0; j <= k; ++j) ssq += W.h[j] * W.h[j];
  if (mode == 1) {
    const double rest = ww - ssq;
    HDR(W, S_CUR) = k;
    if (rest > 0.25 * ww && rest > 0.0) {
      const double hk1 = sqrt(rest);
      W.c[0] = 1.0 / hk1;
      for (int j = 0; j <= k; ++j) W.c[1 + j] = -W.h[j] / hk1;
      W.ncoef = k + 2;
      HDR(W, S_NT) = k + 1;
      HDR(W, S_PHASE) = 1;
      finish_column(W, k, hk1, false);
    } else if (!(ww == ww)) {
      HDR(W, S_PHASE) = 0;
      HDR(W, S_STOP) = 4;
    } else {
      // strong cancellation (or w in the span already): orthogonalise in place, then a second pass
      for (int j = 0; j <= k; ++j) { W.d[j] = W.h[j]; W.c[1 + j] = -W.h[j]; }
      W.c[0] = 1.0;
      W.ncoef = k + 2;
      W.refine = 1;
      HDR(W, S_NT) = k + 1;
      HDR(W, S_PHASE) = 2;
      HDR(W, S_SPARE) = HDR(W, S_SPARE) + 1.0;
    }
    return;
  }
  const double rest = fmax(ww - ssq, 0.0);
  const double hk1 = sqrt(rest);
  if (hk1 > 0.0) {
    W.c[0] = 1.0 / hk1;
    for (int j = 0; j <= k; ++j) W.c[1 + j] = -W.h[j] / hk1;
    W.ncoef = k + 2;
  }
  HDR(W, S_NT) = k + 1;
  HDR(W, S_PHASE) = 3;
  finish_column(W, k, hk1, true);
  if (!(ww == ww)) HDR(W, S_STOP) = 4;
}

// the whole block: W.h holds the products; load what the decision reads, decide, store what it produced
__device__ inline void decide_block(double *S, const Lay &L, DecideLds &W, int mode, int k, int first, double rtol, double atol,
                                    double maxit, double *status) {
  const int t = threadIdx.x, nth = blockDim.x;
  if (t < kHdr) W.hdr[t] = S[S_STOP + t];
  if (mode != 0) {
    for (int i = t; i < k; i += nth) { W.cs[i] = S[L.cs + i]; W.sn[i] = S[L.sn + i]; }
    if (mode == 2)
      for (int j = t; j <= k; j += nth) W.d[j] = S[L.d + j];
    if (t == kHdr) W.gk = S[L.g + k];
  }
  __syncthreads();
  if (t == 0) decide_in_lds(W, L.m, mode, k, first, rtol, atol, maxit);
  __syncthreads();
  for (int j = t; j < W.ncoef; j += nth) S[L.c + j] = W.c[j];
  if (W.refine)
    for (int j = t; j <= k; j += nth) S[L.d + j] = W.d[j];
  if (W.column) {
    for (int j = t; j <= k + 1; j += nth) S[L.hess + (int64_t)k * (L.m + 1) + j] = W.col[j];
    if (t == 0) { S[L.cs + k] = W.cs[k]; S[L.sn + k] = W.sn[k]; S[L.g + k] = W.gk; S[L.g + k + 1] = W.gk1; }
  }
  if (mode == 0 && HDR(W, S_STOP) == 0.0)
    for (int i = t; i <= L.m; i += nth) S[L.g + i] = i == 0 ? W.gk : 0.0;
  if (t < kHdr) S[S_STOP + t] = W.hdr[t];
  if (t == 0) {
    status[0] = HDR(W, S_STOP); status[1] = HDR(W, S_KDONE); status[2] = HDR(W, S_TOTAL); status[3] = HDR(W, S_RES);
    status[4] = HDR(W, S_BETA); status[5] = HDR(W, S_BNORM); status[6] = HDR(W, S_TOL); status[7] = HDR(W, S_SPARE);
  }
}

// which pass of kr_dots is due?  mode 0: cycle start, 1: pass 1 of iteration k, 2: pass 2 of iteration k
__device__ inline bool dots_active(const double *S, int m, int mode, int k) {
  if (mode == 0) return true;
  if (mode == 1) return S[S_STOP] == 0.0 && S[S_KDONE] == (double)k && S[S_CLOSED] == 0.0 && k < m;
  return k >= 0 && S[S_CUR] == (double)k && S[S_PHASE] == 2.0;
}

constexpr int kDotGroup = 8;

template <typename T, int VW>
__global__ __launch_bounds__(kBlock) void kr_dots_kernel(double *S, int m, int mode, int k, const T *w, const T *V,
                                                         int64_t ldv, int64_t nvec, int64_t n, int defer, int first,
                                                         double rtol, double atol, double maxit, double *status) {
  // k < 0: "the iteration the state block says is due" -- the launch then carries nothing that changes from one
  // iteration to the next and can be replayed from a hipGraph (grid sized for the longest column; the groups a shorter
  // column does not need leave at once)
  // In that form the groups a shorter column does not need must still DRAW A TICKET: the block that completes the count
  // rewrites the header (KDONE = k + 1), and a block of this launch that only starts after that would read the NEXT
  // column index, find itself needed, publish partials and draw a ticket against a different total -- leaving an arrival
  // counter at one for good.  With every block of the grid counted, the header is rewritten only after
  // every block has read it.
  const bool fused = k < 0;
  if (fused) k = mode == 1 ? (int)S[S_KDONE] : (int)S[S_CUR];
  if (!dots_active(S, m, mode, k)) return;              // (nothing in this launch changes the header then: uniform)
  using Vc = Vec<T, VW>;
  const Lay L = layout(m);
  const int nv = mode == 0 ? 1 : k + 2;                 // vectors to multiply w with; the last one is w itself
  const int v0 = (int)blockIdx.y * kDotGroup;
  const bool idle = v0 >= nv;                           // block-uniform
  if (idle && !fused) return;                           // (explicit k: the grid has no such block)
  const unsigned groups = fused ? gridDim.y : (unsigned)((nv + kDotGroup - 1) / kDotGroup);
  const int cnt = idle ? 0 : (nv - v0 < kDotGroup ? nv - v0 : kDotGroup);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  double s[kDotGroup];
#pragma unroll
  for (int u = 0; u < kDotGroup; ++u) s[u] = 0;
  if (i < nvec && !idle) {
    const Vc xv = reinterpret_cast<const Vc *>(w)[i];
    Vc yv[kDotGroup];
#pragma unroll
    for (int u = 0; u < kDotGroup; ++u)
      if (u < cnt) {
        const int v = v0 + u;
        const T *p = (mode == 0 || v == nv - 1) ? w : V + (int64_t)v * ldv;
        yv[u] = reinterpret_cast<const Vc *>(p)[i];
      }
#pragma unroll
    for (int u = 0; u < kDotGroup; ++u)
      if (u < cnt) {
#pragma unroll
        for (int e = 0; e < VW; ++e) s[u] += (double)xv[e] * (double)yv[u][e];
      }
  }
  if (VW > 1 && blockIdx.x == 0 && !idle) {             // ragged tail
    const int64_t q = nvec * VW + threadIdx.x;
    if (q < n) {
#pragma unroll
      for (int u = 0; u < kDotGroup; ++u)
        if (u < cnt) {
          const int v = v0 + u;
          const T *p = (mode == 0 || v == nv - 1) ? w : V + (int64_t)v * ldv;
          s[u] += (double)w[q] * (double)p[q];
        }
    }
  }
  double *partial = S + L.part;
  const int nbx = (int)gridDim.x;
  // the eight products of this block: eight independent wave trees (the shuffles interleave), ONE pass through LDS, then
  // lanes 0..cnt-1 add the four wave sums in wave order (the order block_sum uses) and publish a partial each
  __shared__ double red[kBlock / kWave][kDotGroup];
  if (!idle) {
#pragma unroll
    for (int u = 0; u < kDotGroup; ++u) s[u] = wave_sum(s[u]);
    if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
      for (int u = 0; u < kDotGroup; ++u) red[threadIdx.x / kWave][u] = s[u];
    }
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
      double b = 0;
#pragma unroll
      for (int wv = 0; wv < kBlock / kWave; ++wv) b += red[wv][threadIdx.x];
      publish_partial(partial + (int64_t)(v0 + threadIdx.x) * nbx + blockIdx.x, b);
    }
  }
  if (draw_ticket(S, gridDim.x * groups, blockIdx.y * gridDim.x + blockIdx.x)) {
    // The last block adds the block partials of the products side by side.  (One product after the other cost ~3 us each:
    // 100 us for 32 products, found in round 3.)
    __shared__ DecideLds W;
    // eight lanes per product (32 products at a time), each adding every eighth partial in index order with eight loads
    // in flight, then the eight lane sums in a fixed shuffle order: bit-reproducible
    const int rl = threadIdx.x >> 3, l = threadIdx.x & 7;
    for (int r0 = 0; r0 < nv; r0 += kBlock / 8) {
      const int r = r0 + rl;
      double acc = r < nv ? strided_sum(partial + (int64_t)r * nbx, l, 8, nbx) : 0.0;
      acc += __shfl_down(acc, 4, 8);
      acc += __shfl_down(acc, 2, 8);
      acc += __shfl_down(acc, 1, 8);
      if (l == 0 && r < nv) W.h[r] = acc;
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) S[L.h + threadIdx.x] = W.h[threadIdx.x];      // where a sharded solve all-reduces them
    if (!defer) decide_block(S, L, W, mode, k, first, rtol, atol, maxit, status);
  }
}

// the decisions as a kernel of their own (several ranks: after the all-reduce of the product block)
__global__ __launch_bounds__(kBlock) void kr_decide_kernel(double *S, int m, int mode, int k, int first, double rtol, double atol,
                                                            double maxit, double *status) {
  // the pass this decision belongs to ran iff the same condition held; nothing has changed the flags since
  if (k < 0) k = mode == 1 ? (int)S[S_KDONE] : (int)S[S_CUR];
  if (!dots_active(S, m, mode, k)) return;
  const Lay L = layout(m);
  __shared__ DecideLds W;
  const int nv = mode == 0 ? 1 : k + 2;
  for (int v = threadIdx.x; v < nv; v += blockDim.x) W.h[v] = S[L.h + v];    // summed over the ranks by the caller
  __syncthreads();
  decide_block(S, L, W, mode, k, first, rtol, atol, maxit, status);
}

// out = c[0]*w + sum_{j<nt} c[1+j]*V_j  (fixed order, fused multiply-adds in the storage type: the rounding of
// pn_lincomb applied to the same terms), coefficient table in the state block.
//   mode 0  after the cycle start:  V_0 = w/beta                     -> V_0 and vin
//   mode 1  after pass 1 of iteration k:  normal column -> V_{k+1} and vin;  second pass due -> w in place
//   mode 2  after pass 2:  V_{k+1} and vin
//   mode 3  cycle end:  w (= the solution x) += sum y_j V_j, in place
template <typename T, int VW>
__global__ __launch_bounds__(kBlock) void kr_update_kernel(const double *S, int m, int mode, int k, T *w, T *V, int64_t ldv,
                                                           T *vin, int64_t nvec, int64_t n) {
  const Lay L = layout(m);
  T *out = nullptr, *out2 = nullptr;
  if (k < 0 && (mode == 1 || mode == 2)) k = (int)S[S_CUR];          // see kr_dots_kernel
  if (mode == 0) {
    if (!(S[S_STOP] == 0.0 && S[S_PHASE] == 0.0 && S[S_KDONE] == 0.0 && S[S_CUR] == -1.0 && S[S_CLOSED] == 0.0)) return;
    out = V; out2 = vin;
  } else if (mode == 1) {
    if (k < 0 || S[S_CUR] != (double)k || S[S_CLOSED] != 0.0) return;   // (a closed cycle: replays enqueued past its end
    if (S[S_PHASE] == 1.0) {                                             //  must not rewrite V_m / vin with the close's table)
      if (S[S_STOP] != 0.0) return;                      // the solve has ended: V_{k+1} is not needed
      out = V + (int64_t)(k + 1) * ldv; out2 = vin;
    } else if (S[S_PHASE] == 2.0) {
      out = w;
    } else {
      return;
    }
  } else if (mode == 2) {
    if (!(k >= 0 && S[S_CUR] == (double)k && S[S_PHASE] == 3.0 && S[S_HK1] > 0.0 && S[S_STOP] == 0.0 && S[S_CLOSED] == 0.0)) return;
    out = V + (int64_t)(k + 1) * ldv; out2 = vin;
  } else {
    if (S[S_APPLY] != 1.0) return;
    out = w;
  }
  const int nt = mode == 0 ? 0 : (int)S[S_NT];
  __shared__ T sc[kMaxRestart + 2];
  if ((int)threadIdx.x <= nt) sc[threadIdx.x] = (T)S[L.c + threadIdx.x];
  __syncthreads();
  using Vc = Vec<T, VW>;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    Vc acc = reinterpret_cast<const Vc *>(w)[i];
    const T c0 = sc[0];
#pragma unroll
    for (int e = 0; e < VW; ++e) acc[e] = c0 * acc[e];
    for (int j0 = 0; j0 < nt; j0 += 8) {
      Vc r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + u < nt) r[u] = reinterpret_cast<const Vc *>(V + (int64_t)(j0 + u) * ldv)[i];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + u < nt) {
          const T c = sc[1 + j0 + u];
#pragma unroll
          for (int e = 0; e < VW; ++e) acc[e] = fma(c, r[u][e], acc[e]);
        }
    }
    reinterpret_cast<Vc *>(out)[i] = acc;
    if (out2) reinterpret_cast<Vc *>(out2)[i] = acc;
  }
  if (VW > 1 && blockIdx.x == 0) {
    const int64_t q = nvec * VW + threadIdx.x;
    if (q < n) {
      T acc = sc[0] * w[q];
      for (int j = 0; j < nt; ++j) acc = fma(sc[1 + j], V[(int64_t)j * ldv + q], acc);
      out[q] = acc;
      if (out2) out2[q] = acc;
    }
  }
}

// cycle end (the solve has stopped, or the restart length is reached): back substitution of the rotated Hessenberg
// system, as pn_gmres_solve; the coefficients become the table of the closing kr_update
__global__ __launch_bounds__(kBlock) void kr_close_kernel(double *S, int m, double *status) {
  const Lay L = layout(m);
  const int t = threadIdx.x;
  const int kd = (int)S[S_KDONE];
  const double stop0 = S[S_STOP];
  const bool ended = stop0 != 0.0 || kd >= m;
  if (S[S_CLOSED] != 0.0 || !ended) {                 // block-uniform
    if (t == 0) S[S_APPLY] = 0;
    return;
  }
  // the rotated Hessenberg system in LDS (the back substitution reads it ~kd^2/2 times, one entry after the other)
  __shared__ double Hs[kMaxRestart * (kMaxRestart + 1) / 2 + kMaxRestart];   // upper triangle, column j at j(j+1)/2
  __shared__ double gs[kMaxRestart + 1], ys[kMaxRestart + 1];
  __shared__ int singular;
  for (int e = t; e < kd * (kd + 1) / 2; e += blockDim.x) {
    int j = 0;
    while ((j + 1) * (j + 2) / 2 <= e) ++j;             // column of packed entry e
    const int i = e - j * (j + 1) / 2;
    Hs[e] = S[L.hess + (int64_t)j * (m + 1) + i];
  }
  for (int i = t; i < kd; i += blockDim.x) gs[i] = S[L.g + i];
  __syncthreads();
  if (t == 0) {
    singular = 0;
    if (kd > 0 && stop0 != 4.0) {
      for (int i = kd - 1; i >= 0; --i) {
        double s = gs[i];
        for (int j = i + 1; j < kd; ++j) s -= Hs[j * (j + 1) / 2 + i] * ys[j];
        const double d = Hs[i * (i + 1) / 2 + i];
        if (d == 0.0) { singular = 1; break; }
        ys[i] = s / d;
      }
    }
  }
  __syncthreads();
  const bool apply = kd > 0 && stop0 != 4.0 && !singular;
  if (apply) {
    for (int j = t; j < kd; j += blockDim.x) { S[L.y + j] = ys[j]; S[L.c + 1 + j] = ys[j]; }
  }
  if (t == 0) {
    S[S_CLOSED] = 1;
    S[S_APPLY] = apply ? 1 : 0;
    if (apply) { S[L.c] = 1.0; S[S_NT] = kd; }
    double stop = stop0;
    if (kd > 0 && stop0 != 4.0 && singular) { stop = 5; S[S_BREAK] = 1; S[S_STOP] = 5; }
    status[0] = stop; status[1] = kd; status[2] = S[S_TOTAL]; status[3] = S[S_RES];
    status[4] = S[S_BETA]; status[5] = S[S_BNORM]; status[6] = S[S_TOL]; status[7] = S[S_SPARE];
  }
}

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

inline int check_launch(const char *what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pn::fail(std::string(what) + ": " + hipGetErrorString(err));
  return 0;
}

struct Geo {
  bool vec;
  int64_t nvec;
  unsigned nbx;
};
template <typename T>
Geo geometry(int64_t n, const void *a, const void *b, int64_t ldv, const void *c) {
  Geo g;
  constexpr int VW = 16 / sizeof(T);
  g.vec = aligned16(a) && aligned16(b) && (c == nullptr || aligned16(c)) && (ldv * (int64_t)sizeof(T)) % 16 == 0;
  g.nvec = g.vec ? n / VW : n;
  int64_t nb = (g.nvec + kBlock - 1) / kBlock;
  g.nbx = (unsigned)(nb < 1 ? 1 : nb);
  return g;
}

template <typename T>
int launch_dots(hipStream_t st, double *S, int m, int mode, int k, const void *w, const void *V, int64_t ldv, int64_t n,
                int defer, int first, double rtol, double atol, double maxit, double *status) {
  constexpr int VW = 16 / sizeof(T);
  const Geo g = geometry<T>(n, w, V, ldv, nullptr);
  const int nv = mode == 0 ? 1 : (k < 0 ? m : k) + 2;       // k < 0: any column of the cycle
  const dim3 grid(g.nbx, (unsigned)((nv + kDotGroup - 1) / kDotGroup));
  if (g.vec)
    hipLaunchKernelGGL((kr_dots_kernel<T, VW>), grid, dim3(kBlock), 0, st, S, m, mode, k, (const T *)w, (const T *)V, ldv,
                       g.nvec, n, defer, first, rtol, atol, maxit, status);
  else
    hipLaunchKernelGGL((kr_dots_kernel<T, 1>), grid, dim3(kBlock), 0, st, S, m, mode, k, (const T *)w, (const T *)V, ldv,
                       g.nvec, n, defer, first, rtol, atol, maxit, status);
  return check_launch("pn_krylov (products)");
}

template <typename T>
int launch_update(hipStream_t st, const double *S, int m, int mode, int k, void *w, void *V, int64_t ldv, void *vin, int64_t n) {
  constexpr int VW = 16 / sizeof(T);
  const Geo g = geometry<T>(n, w, V, ldv, vin);
  if (g.vec)
    hipLaunchKernelGGL((kr_update_kernel<T, VW>), dim3(g.nbx), dim3(kBlock), 0, st, S, m, mode, k, (T *)w, (T *)V, ldv, (T *)vin,
                       g.nvec, n);
  else
    hipLaunchKernelGGL((kr_update_kernel<T, 1>), dim3(g.nbx), dim3(kBlock), 0, st, S, m, mode, k, (T *)w, (T *)V, ldv, (T *)vin,
                       g.nvec, n);
  return check_launch("pn_krylov (update)");
}

int launch_decide(hipStream_t st, double *S, int m, int mode, int k, int first, double rtol, double atol, double maxit,
                  double *status) {
  hipLaunchKernelGGL(kr_decide_kernel, dim3(1), dim3(kBlock), 0, st, S, m, mode, k, first, rtol, atol, maxit, status);
  return check_launch("pn_krylov (decision)");
}

int bad_args(int dtype, int64_t n, int restart, const void *state, const void *status) {
  if (dtype != PN_F32 && dtype != PN_F64) return pn::fail("pn_krylov: dtype must be PN_F32 or PN_F64");
  if (n <= 0) return pn::fail("pn_krylov: empty vector");
  if (restart < 1 || restart > kMaxRestart) return pn::fail("pn_krylov: restart length must be in 1..126");
  if (!state || !status) return pn::fail("pn_krylov: state and status blocks required");
  return 0;
}

}  // namespace

#define PN_BY_DTYPE(CALL_F32, CALL_F64) (dtype == PN_F32 ? (CALL_F32) : (CALL_F64))

extern "C" {

int64_t pn_krylov_state_doubles(int64_t n, int restart) {
  if (restart < 1 || restart > kMaxRestart || n <= 0) return 0;
  const Lay L = layout(restart);
  return (int64_t)L.part + (int64_t)(restart + 2) * ((n + kBlock - 1) / kBlock + 1) + 2;
}

int pn_krylov_products_offset(int restart) { return layout(restart < 1 ? 1 : restart).h; }

int pn_krylov_begin(void *stream, int dtype, int64_t n, int restart, double *state, double *status_dev, const void *r,
                    void *V, int64_t ldv, void *vin, double rtol, double atol, int64_t maxit, int first_cycle, int part) {
  if (bad_args(dtype, n, restart, state, status_dev)) return 1;
  hipStream_t st = (hipStream_t)stream;
  const int defer = part != 0;
  int rc = 0;
  if (part == 0 || part == 1)
    rc = PN_BY_DTYPE(launch_dots<float>(st, state, restart, 0, 0, r, V, ldv, n, defer, first_cycle, rtol, atol, (double)maxit, status_dev),
                     launch_dots<double>(st, state, restart, 0, 0, r, V, ldv, n, defer, first_cycle, rtol, atol, (double)maxit, status_dev));
  if (rc) return rc;
  if (part == 2) rc = launch_decide(st, state, restart, 0, 0, first_cycle, rtol, atol, (double)maxit, status_dev);
  if (rc) return rc;
  if (part == 0 || part == 2)
    rc = PN_BY_DTYPE(launch_update<float>(st, state, restart, 0, 0, const_cast<void *>(r), V, ldv, vin, n),
                     launch_update<double>(st, state, restart, 0, 0, const_cast<void *>(r), V, ldv, vin, n));
  return rc;
}

int pn_krylov_step(void *stream, int dtype, int64_t n, int restart, double *state, double *status_dev, int k, void *w,
                   void *V, int64_t ldv, void *vin, int part) {
  if (bad_args(dtype, n, restart, state, status_dev)) return 1;
  if (k < -1 || k >= restart) return pn::fail("pn_krylov_step: iteration index beyond the restart length");
  hipStream_t st = (hipStream_t)stream;
  const int defer = part != 0;
  int rc = 0;
#define PN_DOTS(MODE)                                                                                              \
  PN_BY_DTYPE(launch_dots<float>(st, state, restart, MODE, k, w, V, ldv, n, defer, 0, 0.0, 0.0, 0.0, status_dev),  \
              launch_dots<double>(st, state, restart, MODE, k, w, V, ldv, n, defer, 0, 0.0, 0.0, 0.0, status_dev))
#define PN_UPD(MODE)                                                             \
  PN_BY_DTYPE(launch_update<float>(st, state, restart, MODE, k, w, V, ldv, vin, n), \
              launch_update<double>(st, state, restart, MODE, k, w, V, ldv, vin, n))
  if (part == 0 || part == 1) rc = PN_DOTS(1);                                      // pass 1 (+ decision unless deferred)
  if (rc || part == 1) return rc;
  if (part == 2) rc = launch_decide(st, state, restart, 1, k, 0, 0.0, 0.0, 0.0, status_dev);
  if (rc) return rc;
  if (part == 0 || part == 2) {
    rc = PN_UPD(1);                                                                 // V_{k+1}, or w orthogonalised in place
    if (!rc) rc = PN_DOTS(2);                                                       // pass 2: returns at entry unless due
  }
  if (rc || part == 2) return rc;
  if (part == 3) rc = launch_decide(st, state, restart, 2, k, 0, 0.0, 0.0, 0.0, status_dev);
  if (rc) return rc;
  return PN_UPD(2);
#undef PN_DOTS
#undef PN_UPD
}

int pn_krylov_close(void *stream, int dtype, int64_t n, int restart, double *state, double *status_dev, void *x, void *V,
                    int64_t ldv) {
  if (bad_args(dtype, n, restart, state, status_dev)) return 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(kr_close_kernel, dim3(1), dim3(kBlock), 0, st, state, restart, status_dev);
  if (check_launch("pn_krylov_close")) return 1;
  return PN_BY_DTYPE(launch_update<float>(st, state, restart, 3, 0, x, V, ldv, nullptr, n),
                     launch_update<double>(st, state, restart, 3, 0, x, V, ldv, nullptr, n));
}

}  // extern "C"
