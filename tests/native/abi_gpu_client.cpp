// A client of the C ABI with no PyTorch and no Python in the process: what a non-Python host of the
// reference's path (a C++ or Fortran driver of PETSc's TS, say) would link against.
//   hipcc -O2 -I include tests/native/abi_gpu_client.cpp -L pnode_amd/lib -lpnode_amd -o abi_gpu_client
// One classic rk4 step of u' = a*u (so every stage derivative is a scaled copy, produced here with
// pn_lincomb standing in for the user's f) forward, then its discrete adjoint, on device buffers owned by
// this program, compared with the same arithmetic on the host; plus the host-side stepper, the
// embedded-error kernel of the 3bs tableau, a linear solve by the device-resident GMRES (pn_krylov_*), and ten steps + their
// adjoint through the step loops pn_rk_attempt / pn_rk_adjoint_step with C callbacks for f and its transposed Jacobian.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pnode_amd.h"

#define CHECK(x)                                                                 \
  do {                                                                           \
    int rc_ = (x);                                                               \
    if (rc_) { std::fprintf(stderr, "%s failed: %s\n", #x, pn_last_error()); return 1; } \
  } while (0)
#define HIP(x)                                                                                       \
  do {                                                                                               \
    hipError_t e_ = (x);                                                                             \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
  } while (0)

int main() {
  const int64_t n = 1000003;                 // odd: exercises the unaligned tail
  const double a = -0.7, h = 0.1;
  hipStream_t st;
  HIP(hipStreamCreate(&st));
  std::vector<double> hu(n), hlam(n);
  for (int64_t i = 0; i < n; ++i) { hu[i] = std::sin(0.001 * i) + 1.5; hlam[i] = std::cos(0.002 * i); }
  double *u, *y, *k[4], *unew, *lam, *w, *dl[4];
  auto bytes = (size_t)n * sizeof(double);
  HIP(hipMalloc(&u, bytes)); HIP(hipMalloc(&y, bytes)); HIP(hipMalloc(&unew, bytes));
  HIP(hipMalloc(&lam, bytes)); HIP(hipMalloc(&w, bytes));
  for (int i = 0; i < 4; ++i) { HIP(hipMalloc(&k[i], bytes)); HIP(hipMalloc(&dl[i], bytes)); }
  HIP(hipMemcpyAsync(u, hu.data(), bytes, hipMemcpyHostToDevice, st));
  HIP(hipMemcpyAsync(lam, hlam.data(), bytes, hipMemcpyHostToDevice, st));

  // tableau from the library (what -ts_rk_type 4 selects)
  pn_tableau tab;
  CHECK(pn_tableau_get("4", &tab));
  if (tab.s != 4 || tab.A[1][0] != 0.5 || tab.b[0] != 1.0 / 6.0) { std::fprintf(stderr, "unexpected rk4 tableau\n"); return 1; }

  // ---- forward: K_i = a*Y_i ("f"), Y_i = u + h*a_{i,i-1} K_{i-1}
  const void *x1[1];
  double c1[1];
  const double *stage_in = u;
  for (int i = 0; i < 4; ++i) {
    if (i > 0) {
      const void *Ks[1] = {k[i - 1]};
      const double cf[1] = {h * tab.A[i][i - 1]};
      CHECK(pn_rk_stage(st, PN_F64, n, y, u, 1, Ks, cf));
      stage_in = y;
    }
    x1[0] = stage_in; c1[0] = a;
    CHECK(pn_lincomb(st, PN_F64, n, k[i], 1, x1, c1));
  }
  const void *K4[4] = {k[0], k[1], k[2], k[3]};
  const double cb[4] = {h * tab.b[0], h * tab.b[1], h * tab.b[2], h * tab.b[3]};
  CHECK(pn_rk_stage(st, PN_F64, n, unew, u, 4, K4, cb));

  // ---- adjoint of that step: J = a*I, so dlam_i = a*w_i
  double scale3 = h * tab.b[3];                       // stage 3: pure multiple of lambda, folded
  x1[0] = lam; c1[0] = a;
  CHECK(pn_lincomb(st, PN_F64, n, dl[3], 1, x1, c1));  // raw VJP with lambda as cotangent
  double sc[4] = {1.0, 1.0, 1.0, scale3};
  for (int i = 2; i >= 0; --i) {
    const void *D[1] = {dl[i + 1]};
    const double cf[1] = {h * tab.A[i + 1][i] * sc[i + 1]};
    CHECK(pn_adj_theta(st, PN_F64, n, w, lam, h * tab.b[i], 1, D, cf));
    x1[0] = w; c1[0] = a;
    CHECK(pn_lincomb(st, PN_F64, n, dl[i], 1, x1, c1));
  }
  const void *D4[4] = {dl[0], dl[1], dl[2], dl[3]};
  CHECK(pn_adj_accum(st, PN_F64, n, lam, lam, 4, D4, sc, nullptr, nullptr, 0.0));

  std::vector<double> gun(n), glam(n);
  HIP(hipMemcpyAsync(gun.data(), unew, bytes, hipMemcpyDeviceToHost, st));
  HIP(hipMemcpyAsync(glam.data(), lam, bytes, hipMemcpyDeviceToHost, st));
  HIP(hipStreamSynchronize(st));

  // host: the amplification factor of rk4 on u' = a u, and of its adjoint (the same number)
  const double z = a * h, R = 1 + z + z * z / 2 + z * z * z / 6 + z * z * z * z / 24;
  double eu = 0, el = 0;
  for (int64_t i = 0; i < n; ++i) {
    eu = std::fmax(eu, std::fabs(gun[i] - R * hu[i]));
    el = std::fmax(el, std::fabs(glam[i] - R * hlam[i]));
  }
  std::printf("rk4 step: max |u_new - R u| = %.2e, adjoint: max |lambda_n - R lambda| = %.2e (R = %.15f)\n", eu, el, R);
  if (!(eu < 1e-14 && el < 1e-14)) return 2;

  // ---- embedded error norm of 3bs through the fused kernel and the pinned scalar
  pn_tableau t3;
  CHECK(pn_tableau_get("3bs", &t3));
  double *work, *rh, *rd;
  HIP(hipMalloc(&work, (size_t)pn_wrms_work_bytes(n)));
  HIP(hipMemset(work, 0, (size_t)pn_wrms_work_bytes(n)));      // the arrival counter starts at zero (include/pnode_amd.h)
  CHECK(pn_pinned_block(8 * pn_wrms_partials(n), &rh, &rd));
  double ce[4], cbb[4];
  for (int j = 0; j < 4; ++j) { ce[j] = h * (t3.bembed[j] - t3.b[j]); cbb[j] = h * t3.b[j]; }
  CHECK(pn_rk_combine_wrms(st, PN_F64, n, unew, u, 4, K4, cbb, ce, 1e-4, 1e-4, work, rd));
  double enorm = 0;
  CHECK(pn_stream_wait_wrms(st, rh, n, &enorm));
  HIP(hipMemcpy(gun.data(), unew, bytes, hipMemcpyDeviceToHost));
  std::vector<double> hk[4];
  for (int j = 0; j < 4; ++j) { hk[j].resize(n); HIP(hipMemcpy(hk[j].data(), k[j], bytes, hipMemcpyDeviceToHost)); }
  long double acc = 0;
  for (int64_t i = 0; i < n; ++i) {
    double un = hu[i], er = 0;
    for (int j = 0; j < 4; ++j) { un += cbb[j] * hk[j][i]; er += ce[j] * hk[j][i]; }
    const double tol = 1e-4 + 1e-4 * std::fmax(std::fabs(un), std::fabs(un + er));
    acc += (long double)(er / tol) * (er / tol);
  }
  const double ref = std::sqrt((double)(acc / n));
  std::printf("3bs WRMS error norm: device %.15e host %.15e\n", enorm, ref);
  if (!(std::fabs(enorm - ref) <= 1e-12 * ref)) return 3;

  // ---- the host-side stepper: fixed steps with an exactly matched end time
  pn_ts *ts = pn_ts_create();
  CHECK(pn_ts_set_rk_type(ts, "4"));
  CHECK(pn_ts_set_option(ts, "ts_adapt_type", "none"));
  const double tspan[1] = {0.35};
  CHECK(pn_ts_begin(ts, 0.0, 0.1, 1, tspan));
  int acc_i, hit, done = 0, steps = 0;
  double tt, hh, sum = 0;
  while (!done) {
    CHECK(pn_ts_attempt(ts, &tt, &hh));
    CHECK(pn_ts_judge(ts, -1.0, &acc_i, &hit, &done));
    sum += hh;
    ++steps;
  }
  std::printf("stepper: %d steps to t = %.17g (sum of steps %.17g)\n", steps, pn_ts_time(ts), sum);
  if (!(steps == 4 && pn_ts_time(ts) == 0.35)) return 4;
  pn_ts_destroy(ts);
  pn_pinned_free(rh);

  // ---- device-resident GMRES (header section 3c) from plain C++: A v = d .* v + 0.4 * roll(v, 1) with d = 2 on the first half of
  // the vector and 3 on the second (nonsymmetric, diagonally dominant); the operator is "applied" with the ABI's own pn_lincomb on
  // a rolled copy, standing in for the user's Jacobian product; the host looks at the state once per 6 iterations
  {
    const int m = 30;
    const int64_t ld = (n + 63) / 64 * 64;
    std::vector<double> hd(n), hb(n);
    for (int64_t i = 0; i < n; ++i) { hd[i] = i < n / 2 ? 2.0 : 3.0; hb[i] = std::cos(0.011 * i) + 0.3; }
    double *b, *x, *V, *vin, *wv, *roll, *state, *sth, *std_;
    HIP(hipMalloc(&b, bytes)); HIP(hipMalloc(&x, bytes)); HIP(hipMalloc(&roll, bytes));
    HIP(hipMalloc(&vin, ld * sizeof(double))); HIP(hipMalloc(&wv, ld * sizeof(double)));
    HIP(hipMalloc(&V, (size_t)(m + 1) * ld * sizeof(double)));
    const int64_t nst = pn_krylov_state_doubles(n, m);
    HIP(hipMalloc(&state, (size_t)nst * sizeof(double)));
    HIP(hipMemset(state, 0, (size_t)nst * sizeof(double)));          // arrival counters start at zero
    CHECK(pn_pinned_block(64, &sth, &std_));
    HIP(hipMemcpyAsync(b, hb.data(), bytes, hipMemcpyHostToDevice, st));
    HIP(hipMemsetAsync(x, 0, bytes, st));
    auto apply = [&](double *in, double *out) -> int {
      // roll by one element: two device copies; then out = d*in + 0.4*roll on each half (pn_lincomb on sub-vectors)
      if (hipMemcpyAsync(roll + 1, in, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
      if (hipMemcpyAsync(roll, in + (n - 1), sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
      const int64_t h1 = n / 2, h2 = n - h1;
      const void *xa[2] = {in, roll};
      const double ca[2] = {2.0, 0.4};
      if (pn_lincomb(st, PN_F64, h1, out, 2, xa, ca)) return 1;
      const void *xb[2] = {in + h1, roll + h1};
      const double cb[2] = {3.0, 0.4};
      return pn_lincomb(st, PN_F64, h2, out + h1, 2, xb, cb);
    };
    CHECK(pn_krylov_begin(st, PN_F64, n, m, state, std_, b, V, ld, vin, 1e-10, 1e-50, 10000, 1, 0));
    int k = 0, syncs = 0;
    double status[8] = {0};
    for (;;) {
      for (int c = 0; c < 6 && k < m; ++c, ++k) {
        if (apply(vin, wv)) { std::fprintf(stderr, "operator failed: %s\n", pn_last_error()); return 1; }
        CHECK(pn_krylov_step(st, PN_F64, n, m, state, std_, k, wv, V, ld, vin, 0));
      }
      CHECK(pn_krylov_close(st, PN_F64, n, m, state, std_, x, V, ld));
      CHECK(pn_stream_wait_scalars(st, sth, 8, status));
      ++syncs;
      if (status[0] != 0.0 || k >= m) break;
    }
    std::vector<double> hx(n);
    HIP(hipMemcpy(hx.data(), x, bytes, hipMemcpyDeviceToHost));
    long double rr = 0, bb = 0;
    for (int64_t i = 0; i < n; ++i) {
      const double av = hd[i] * hx[i] + 0.4 * hx[(i + n - 1) % n];
      rr += (long double)(hb[i] - av) * (hb[i] - av);
      bb += (long double)hb[i] * hb[i];
    }
    const double rel = std::sqrt((double)(rr / bb));
    std::printf("device GMRES: stop %d after %d iterations, %d host synchronisations, true relative residual %.3e (estimate %.3e)\n",
                (int)status[0], (int)status[2], syncs, rel, status[3] / status[5]);
    if (!((int)status[0] == 1 && (int)status[2] >= 3 && syncs < (int)status[2] && rel < 1e-9)) return 5;
    pn_pinned_free(sth);
  }
  // ---- the step loops (header section 3a) from plain C++: ten rk4 steps of u' = a u forward and their discrete adjoint with
  // pn_rk_attempt / pn_rk_adjoint_step; the two callbacks (f and its transposed-Jacobian product) are C functions that launch
  // pn_lincomb -- what a C++ or Fortran host of the reference's path would hand to PETSc as RHSFunction / RHSJacobian
  {
    struct Ctx {
      hipStream_t st; int64_t n; double a; const double *u; double *ys[PN_MAX_STAGES]; double *K[PN_MAX_STAGES];
      double *lam, *w, *wb, *dl[PN_MAX_STAGES]; int fails;
    } cx{};
    cx.st = st; cx.n = n; cx.a = a;
    double *ua, *ub, *lam2, *w2, *w3;
    HIP(hipMalloc(&ua, bytes)); HIP(hipMalloc(&ub, bytes)); HIP(hipMalloc(&lam2, bytes)); HIP(hipMalloc(&w2, bytes)); HIP(hipMalloc(&w3, bytes));
    for (int i = 0; i < 4; ++i) { cx.K[i] = k[i]; cx.dl[i] = dl[i]; HIP(hipMalloc(&cx.ys[i], bytes)); }
    HIP(hipMemcpyAsync(ua, hu.data(), bytes, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(lam2, hlam.data(), bytes, hipMemcpyHostToDevice, st));
    cx.lam = lam2; cx.w = w2; cx.wb = w3;
    pn_stage_cb f_cb = [](void *user, int stage, double) -> int64_t {
      Ctx *c = (Ctx *)user;
      const void *x[1] = {stage == 0 ? (const void *)c->u : (const void *)c->ys[stage]};
      const double cf[1] = {c->a};
      if (pn_lincomb(c->st, PN_F64, c->n, c->K[stage], 1, x, cf)) { c->fails++; return 0; }
      return (int64_t)(intptr_t)c->K[stage];
    };
    pn_vjp_cb jt_cb = [](void *user, int stage, double, int in_w, double) -> int64_t {
      Ctx *c = (Ctx *)user;                              // J = a I: J^T cot = a * cot; no parameters
      const void *x[1] = {in_w == 1 ? (const void *)c->w : in_w == 2 ? (const void *)c->wb : (const void *)c->lam};   // (the two buffers in turn)
      const double cf[1] = {c->a};
      if (pn_lincomb(c->st, PN_F64, c->n, c->dl[stage], 1, x, cf)) { c->fails++; return -1; }
      return (int64_t)(intptr_t)c->dl[stage];
    };
    pn_ts *ts2 = pn_ts_create();
    CHECK(pn_ts_set_rk_type(ts2, "4"));
    CHECK(pn_ts_set_option(ts2, "ts_adapt_type", "none"));
    const double tend[1] = {1.0};
    CHECK(pn_ts_begin(ts2, 0.0, h, 1, tend));
    const void *kout[PN_MAX_STAGES];
    double *cur = ua, *nxt = ub;
    int acc2, hit2, done2 = 0, nsteps = 0;
    double t0s[16], hs[16];
    while (!done2) {
      double tt2, hh2;
      CHECK(pn_ts_attempt(ts2, &tt2, &hh2));
      cx.u = cur;
      CHECK(pn_rk_attempt(st, PN_F64, n, ts2, nullptr, tt2, hh2, cur, nxt, (void *const *)cx.ys, nullptr, 0, 0.0, f_cb, &cx, 0,
                          nullptr, nullptr, kout));
      CHECK(pn_ts_judge(ts2, -1.0, &acc2, &hit2, &done2));
      t0s[nsteps] = tt2; hs[nsteps] = hh2; ++nsteps;
      double *tmp = cur; cur = nxt; nxt = tmp;
    }
    // reverse: u' = a u is linear, so the stage values are not needed by the callbacks
    for (int sidx = nsteps - 1; sidx >= 0; --sidx)
      CHECK(pn_rk_adjoint_step(st, PN_F64, n, ts2, nullptr, t0s[sidx], hs[sidx], lam2, w2, (sidx & 1) ? w3 : nullptr, jt_cb, &cx, nullptr));
    HIP(hipMemcpyAsync(gun.data(), cur, bytes, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(glam.data(), lam2, bytes, hipMemcpyDeviceToHost, st));
    HIP(hipStreamSynchronize(st));
    const double RN = std::pow(R, nsteps);
    double e1 = 0, e2 = 0;
    for (int64_t i = 0; i < n; ++i) {
      e1 = std::fmax(e1, std::fabs(gun[i] - RN * hu[i]));
      e2 = std::fmax(e2, std::fabs(glam[i] - RN * hlam[i]));
    }
    std::printf("step loops: %d rk4 steps to t = %.17g, max |u - R^N u0| = %.2e, adjoint max |lambda - R^N lambda_T| = %.2e\n",
                nsteps, pn_ts_time(ts2), e1, e2);
    if (!(nsteps == 10 && pn_ts_time(ts2) == 1.0 && cx.fails == 0 && e1 < 1e-13 && e2 < 1e-13)) return 6;
    pn_ts_destroy(ts2);
  }
  // ---- the parameter sensitivities of a Linear layer (round 5/6 entry points): the fused MFMA kernel, single launches and a grouped
  // one, in both precisions, then pn_linear_wgrad_finish into mu; pn_colsum_accum_multi; all against sums formed here in double.
  // On a stream of this library's own making (pn_stream_create), with the event profile switched on (pn_prof_collect with a count).
  {
    if (pn_abi_version() != PN_ABI_VERSION) { std::fprintf(stderr, "ABI version %d, header %d\n", pn_abi_version(), PN_ABI_VERSION); return 7; }
    void *bg = nullptr;
    CHECK(pn_stream_create(1, &bg));
    const int64_t rows = 512, of = 128, inf = 64;
    std::vector<double> hg((size_t)rows * of), hx((size_t)rows * inf), hmu((size_t)of * inf), hmb(of);
    for (size_t i = 0; i < hg.size(); ++i) hg[i] = std::sin(0.37 * (double)i) * 0.5;
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = std::cos(0.11 * (double)i) - 0.2;
    for (size_t i = 0; i < hmu.size(); ++i) hmu[i] = 0.01 * (double)(i % 17);
    for (size_t i = 0; i < hmb.size(); ++i) hmb[i] = -0.03 * (double)(i % 5);
    const double alphas[3] = {0.5, -0.25, 1.5};               // three accumulating launches: one single, one group of two
    std::vector<double> want_w(hmu), want_b(hmb);
    double asum = 0;
    for (double al : alphas) asum += al;
    for (int64_t m = 0; m < of; ++m) {
      double sb = 0;
      for (int64_t kk = 0; kk < rows; ++kk) sb += hg[(size_t)kk * of + m];
      want_b[m] += asum * sb;
      for (int64_t nn = 0; nn < inf; ++nn) {
        double sw = 0;
        for (int64_t kk = 0; kk < rows; ++kk) sw += hg[(size_t)kk * of + m] * hx[(size_t)kk * inf + nn];
        want_w[(size_t)m * inf + nn] += asum * sw;
      }
    }
    CHECK(pn_prof_enable(1));
    for (int dt = 0; dt < 2; ++dt) {
      const int dtype = dt == 0 ? PN_F32 : PN_F64;
      const size_t es = dt == 0 ? 4 : 8;
      if (!pn_linear_wgrad_supported(dtype, rows, of, inf) || pn_linear_wgrad_supported(dtype, rows, of + 1, inf) || pn_linear_wgrad_supported(dtype, 255, of, inf)) return 7;
      int64_t nb = 0;
      const int64_t nw = pn_linear_wgrad_work_bytes(dtype, of, inf, &nb);
      void *g, *x, *pw, *pb, *pw2, *pb2, *muw, *mub, *muw2, *mub2;
      HIP(hipMalloc(&g, hg.size() * es)); HIP(hipMalloc(&x, hx.size() * es)); HIP(hipMalloc(&pw, nw)); HIP(hipMalloc(&pb, nb));
      HIP(hipMalloc(&pw2, nw)); HIP(hipMalloc(&pb2, nb));
      HIP(hipMalloc(&muw, hmu.size() * es)); HIP(hipMalloc(&mub, hmb.size() * es)); HIP(hipMalloc(&muw2, hmu.size() * es)); HIP(hipMalloc(&mub2, hmb.size() * es));
      HIP(hipMemset(pw, 0, nw)); HIP(hipMemset(pb, 0, nb)); HIP(hipMemset(pw2, 0, nw)); HIP(hipMemset(pb2, 0, nb));
      auto up = [&](void *d, const std::vector<double> &h) -> int {
        if (dt == 1) return hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice) != hipSuccess;
        std::vector<float> f(h.begin(), h.end());
        return hipMemcpy(d, f.data(), f.size() * 4, hipMemcpyHostToDevice) != hipSuccess;
      };
      if (up(g, hg) || up(x, hx) || up(muw, hmu) || up(mub, hmb) || up(muw2, hmu) || up(mub2, hmb)) return 7;
      // layer 1 (pw, pb): a single launch, then, with layer 2 (the same operands, its own partial buffers), a group of two, twice
      CHECK(pn_linear_wgrad(bg, dtype, rows, of, inf, g, x, alphas[0], pw, pb));
      CHECK(pn_linear_wgrad(bg, dtype, rows, of, inf, g, x, alphas[0], pw2, pb2));
      for (int r = 1; r < 3; ++r) {
        pn_wgrad_pair q[2] = {{g, x, pw, pb, alphas[r], of, inf}, {g, x, pw2, pb2, alphas[r], of, inf}};
        CHECK(pn_linear_wgrad_group(bg, dtype, rows, 2, q, 0));
      }
      CHECK(pn_linear_wgrad_finish(bg, dtype, of, inf, pw, pb, muw, mub));
      CHECK(pn_linear_wgrad_finish(bg, dtype, of, inf, pw2, pb2, muw2, mub2));
      HIP(hipStreamSynchronize((hipStream_t)bg));
      auto down = [&](const void *d, size_t cnt, std::vector<double> &h) -> int {
        h.resize(cnt);
        if (dt == 1) return hipMemcpy(h.data(), d, cnt * 8, hipMemcpyDeviceToHost) != hipSuccess;
        std::vector<float> f(cnt);
        if (hipMemcpy(f.data(), d, cnt * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (size_t i = 0; i < cnt; ++i) h[i] = f[i];
        return 0;
      };
      std::vector<double> gw, gb, gw2, gb2, zero;
      if (down(muw, hmu.size(), gw) || down(mub, hmb.size(), gb) || down(muw2, hmu.size(), gw2) || down(mub2, hmb.size(), gb2)) return 7;
      double ew = 0, eb = 0, sw = 0, sb = 0;
      for (size_t i = 0; i < gw.size(); ++i) { ew = std::fmax(ew, std::fabs(gw[i] - want_w[i])); sw = std::fmax(sw, std::fabs(want_w[i])); }
      for (size_t i = 0; i < gb.size(); ++i) { eb = std::fmax(eb, std::fabs(gb[i] - want_b[i])); sb = std::fmax(sb, std::fabs(want_b[i])); }
      const bool same = gw == gw2 && gb == gb2;                // single-then-group and single-then-group: the same bits
      if (down(pw, (size_t)nw / es, zero)) return 7;
      double left = 0;
      for (double v : zero) left = std::fmax(left, std::fabs(v));
      std::printf("pn_linear_wgrad (%s): max |dW - ref| = %.2e of %.2e, max |db - ref| = %.2e of %.2e, partial buffers left at %.1e\n",
                  dt == 0 ? "fp32" : "fp64", ew, sw, eb, sb, left);
      const double tol = dt == 0 ? 2e-5 : 1e-12;
      if (!(ew <= tol * sw && eb <= tol * sb && same && left == 0.0)) return 7;
      // pn_colsum_accum_multi: two sources into one mu slice
      if (up(mub, hmb)) return 7;
      const int64_t nbw = pn_colsum_work_bytes(2, (const int64_t[]){rows, rows}, (const int64_t[]){of, of});
      void *work;
      HIP(hipMalloc(&work, nbw));
      const void *gs[2] = {g, g};
      void *mus[2] = {mub, mub};
      const double al2[2] = {0.5, 1.0};
      const int64_t rr[2] = {rows, rows}, cc[2] = {of, of};
      CHECK(pn_colsum_accum_multi(bg, dtype, 2, rr, cc, gs, mus, al2, work));
      HIP(hipStreamSynchronize((hipStream_t)bg));
      if (down(mub, hmb.size(), gb)) return 7;
      double ec = 0;
      for (int64_t m = 0; m < of; ++m) {
        double sbm = 0;
        for (int64_t kk = 0; kk < rows; ++kk) sbm += hg[(size_t)kk * of + m];
        ec = std::fmax(ec, std::fabs(gb[m] - (hmb[m] + 1.5 * sbm)));
      }
      std::printf("pn_colsum_accum_multi (%s): max |mu - ref| = %.2e\n", dt == 0 ? "fp32" : "fp64", ec);
      if (!(ec <= (dt == 0 ? 2e-5 : 1e-12) * sb)) return 7;
    }
    int64_t launches[PN_K_COUNT];
    double usec[PN_K_COUNT], flops[PN_K_COUNT];
    CHECK(pn_prof_collect(PN_K_COUNT, launches, usec, flops));
    CHECK(pn_prof_enable(0));
    std::printf("profile: %lld launches of %s, %.1f us, %.3g FLOP\n", (long long)launches[PN_K_LINEAR_WGRAD], pn_kernel_name(PN_K_LINEAR_WGRAD),
                usec[PN_K_LINEAR_WGRAD], flops[PN_K_LINEAR_WGRAD]);
    // 2 precisions x (2 single launches + 2 groups of two pairs): 8 launches, 12 pairs of 2 * rows * out * in FLOP
    if (!(launches[PN_K_LINEAR_WGRAD] == 8 && flops[PN_K_LINEAR_WGRAD] == 12 * 2.0 * rows * of * inf && usec[PN_K_LINEAR_WGRAD] > 0)) return 7;
    CHECK(pn_stream_destroy(bg));
  }
  std::printf("ABI-CLIENT-OK\n");
  return 0;
}
