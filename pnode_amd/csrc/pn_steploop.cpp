// pnode_amd -- the step loops of the explicit-RK path in C++.
//
// In the reference `ts.solve` and `ts.adjointSolve` are PETSc's C loops (reference pnode/petsc_adjoint.py:829, 878):
// TSStep_RK forms every stage vector and calls back into Python only for `func` (evalRHSFunction, pa.py:393-412);
// TSAdjointStep_RK forms every stage cotangent and calls back only for the transposed-Jacobian products
// (RHSJacShell.multTranspose, pa.py:52-82).  Until round 3 the product's stage loop was Python with one ctypes call per
// launch.  Here ONE entry point runs a whole step attempt (or a whole reversed step): it owns the tableau walk, forms
// h*a_ij / H*a_ji*scale_j in double exactly as the Python loop did (same expression order: bit-identical launches),
// launches the state-vector kernels, and calls back only where the reference's C loops call back.
//
// The vector operations go through a table of function pointers (pn_vec_ops): NULL selects this library's HIP entry points;
// the CPU-only test container plugs in its stand-in, so that the loops themselves are covered there against the oracle.
#include <cmath>
#include <cstdint>
#include <string>

#include "pnode_amd.h"
#include "pn_internal.h"

namespace {

struct Ops {
  pn_rk_stage_fn rk_stage;
  pn_rk_combine_wrms_fn combine_wrms;
  pn_adj_theta_fn adj_theta;
  pn_adj_accum_fn adj_accum;
};

Ops resolve(const pn_vec_ops *o) {
  Ops r;
  r.rk_stage = (o && o->rk_stage) ? o->rk_stage : pn_rk_stage;
  r.combine_wrms = (o && o->rk_combine_wrms) ? o->rk_combine_wrms : pn_rk_combine_wrms;
  r.adj_theta = (o && o->adj_theta) ? o->adj_theta : pn_adj_theta;
  r.adj_accum = (o && o->adj_accum) ? o->adj_accum : pn_adj_accum;
  return r;
}

}  // namespace

extern "C" {

// TSStep_RK's body for one attempt of size h from the state u at time t.
int pn_rk_attempt(void *stream, int dtype, int64_t n, const pn_ts *ts, const pn_vec_ops *vec_ops, double t, double h,
                  const void *u, void *unew, void *const *ystage, const void *k0, int have_t_first, double t_first,
                  pn_stage_cb cb, void *user, int want_err, void *work, double *result_dev, const void **kout) {
  if (!ts || !u || !unew || !cb || !kout) return pn::fail("pn_rk_attempt: null argument");
  pn_tableau T;
  if (pn_ts_get_tableau(ts, &T)) return 1;
  const Ops ops = resolve(vec_ops);
  const int s = T.s;
  const bool fsal = T.fsal != 0;
  const void *ptrs[PN_MAX_STAGES];
  double coef[PN_MAX_STAGES], coef_e[PN_MAX_STAGES];
  for (int i = 0; i < s; ++i) {
    const void *y = u;
    if (i > 0) {
      void *yi = (fsal && i == s - 1) ? unew : (ystage ? ystage[i] : nullptr);
      if (!yi) return pn::fail("pn_rk_attempt: no buffer for a stage value");
      int nk = 0;
      for (int j = 0; j < i; ++j)
        if (T.A[i][j] != 0.0) { ptrs[nk] = kout[j]; coef[nk] = h * T.A[i][j]; ++nk; }
      if (ops.rk_stage(stream, dtype, n, yi, u, nk, ptrs, coef)) return 1;
      y = yi;
    }
    if (i == 0 && k0) {
      kout[0] = k0;                                       // first-same-as-last: the previous step's last derivative
    } else {
      const double ti = (i == 0 && have_t_first) ? t_first : t + T.c[i] * h;
      const int64_t p = cb(user, i, ti);
      if (p == 0) return pn::fail("pn_rk_attempt: the stage callback failed");   // (the Python side holds the exception)
      kout[i] = (const void *)(intptr_t)p;
    }
    (void)y;
  }
  if (want_err) {
    int nk = 0;
    for (int j = 0; j < s; ++j) {
      const double e = T.bembed[j] - T.b[j];
      if (e != 0.0 || (!fsal && T.b[j] != 0.0)) { ptrs[nk] = kout[j]; coef[nk] = h * T.b[j]; coef_e[nk] = h * e; ++nk; }
    }
    double atol, rtol;
    if (pn_ts_get_tolerances(ts, &atol, &rtol)) return 1;
    return ops.combine_wrms(stream, dtype, n, fsal ? nullptr : unew, fsal ? (const void *)unew : u, nk, ptrs, coef, coef_e,
                            atol, rtol, work, result_dev);
  }
  if (!fsal) {
    int nk = 0;
    for (int j = 0; j < s; ++j)
      if (T.b[j] != 0.0) { ptrs[nk] = kout[j]; coef[nk] = h * T.b[j]; ++nk; }
    return ops.rk_stage(stream, dtype, n, unew, u, nk, ptrs, coef);
  }
  return 0;
}

// TSAdjointStep_RK for one step [t, t+H]: lambda and (through the callback) mu are advanced to the start of the step.
int pn_rk_adjoint_step(void *stream, int dtype, int64_t n, const pn_ts *ts, const pn_vec_ops *vec_ops, double t, double H,
                       void *lambda, void *wbuf, void *wbuf2, pn_vjp_cb cb, void *user, const void *forcing) {
  if (!ts || !lambda || !wbuf || !cb) return pn::fail("pn_rk_adjoint_step: null argument");
  pn_tableau T;
  if (pn_ts_get_tableau(ts, &T)) return 1;
  const Ops ops = resolve(vec_ops);
  const int s_eff = T.fsal ? T.s - 1 : T.s;               // the first-same-as-last stage has a structurally zero cotangent
  const void *dl[PN_MAX_STAGES] = {nullptr};              // raw VJP results; the true dlambda_i is scale[i]*dl[i]
  double scale[PN_MAX_STAGES];
  for (int i = 0; i < PN_MAX_STAGES; ++i) scale[i] = 1.0;
  const void *ptrs[PN_MAX_STAGES];
  double coef[PN_MAX_STAGES];
  int nw = 0;                                             // cotangent buffers written so far in this step
  for (int i = s_eff - 1; i >= 0; --i) {
    int nk = 0;
    for (int j = i + 1; j < s_eff; ++j)
      if (T.A[j][i] != 0.0 && dl[j]) { ptrs[nk] = dl[j]; coef[nk] = H * T.A[j][i] * scale[j]; ++nk; }
    if (T.b[i] == 0.0 && nk == 0) continue;               // structurally zero cotangent
    int use_w = 0;
    if (nk == 0) {
      scale[i] = H * T.b[i];                              // cotangent = lambda itself; the factor goes to the consumers
    } else {
      use_w = (wbuf2 && (nw & 1)) ? 2 : 1;                // the two cotangent buffers in turn
      ++nw;
      if (ops.adj_theta(stream, dtype, n, use_w == 2 ? wbuf2 : wbuf, T.b[i] != 0.0 ? lambda : nullptr, H * T.b[i], nk, ptrs, coef)) return 1;
    }
    const int64_t p = cb(user, i, t + T.c[i] * H, use_w, scale[i]);
    if (p == -1) return pn::fail("pn_rk_adjoint_step: the VJP callback failed");
    dl[i] = (const void *)(intptr_t)p;                    // 0: func does not depend on its state argument
  }
  int nk = 0;
  for (int i = 0; i < s_eff; ++i)
    if (dl[i]) { ptrs[nk] = dl[i]; coef[nk] = scale[i]; ++nk; }
  return ops.adj_accum(stream, dtype, n, lambda, lambda, nk, ptrs, coef, forcing, nullptr, 0.0);
}

}  // extern "C"
