// Self-test of the host engine (pnode_amd/csrc/pn_ts.cpp) meant to run under
// -fsanitize=address,undefined on the CPU: stepper state machine, checkpoint scheduler, GMRES core.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "pnode_amd.h"

#define REQUIRE(c)                                                          \
  do {                                                                      \
    if (!(c)) {                                                             \
      std::fprintf(stderr, "%s:%d: REQUIRE(%s) failed: %s\n", __FILE__, __LINE__, #c, pn_last_error()); \
      std::exit(1);                                                         \
    }                                                                       \
  } while (0)

static void stepper() {
  pn_ts *ts = pn_ts_create();
  REQUIRE(pn_ts_set_rk_type(ts, "5dp") == 0);
  REQUIRE(pn_ts_set_rk_type(ts, "bogus") != 0);
  REQUIRE(pn_ts_set_option(ts, "ts_rtol", "1e-6") == 0);
  REQUIRE(pn_ts_set_option(ts, "ts_adapt_clip", "0.2,5") == 0);
  REQUIRE(pn_ts_set_option(ts, "nonsense", "1") != 0);
  const double span[4] = {0.0, 0.5, 0.7, 2.0};
  REQUIRE(pn_ts_begin(ts, 0.0, 0.3, 4, span) == 0);
  int acc, hit, done = 0, hits = 0;
  unsigned seed = 12345;
  while (!done) {
    double t, h;
    REQUIRE(pn_ts_attempt(ts, &t, &h) == 0);
    seed = seed * 1664525u + 1013904223u;
    const double enorm = (seed >> 8) % 100 < 25 ? 1.7 : 0.3;     // reject about a quarter of the attempts
    REQUIRE(pn_ts_judge(ts, enorm, &acc, &hit, &done) == 0);
    if (hit >= 0) ++hits;
  }
  REQUIRE(hits == 3 && pn_ts_time(ts) == 2.0 && pn_ts_rejections(ts) > 0);
  for (int64_t k = 0; k < pn_ts_steps(ts); ++k) {
    double t0, h;
    REQUIRE(pn_ts_step_log(ts, k, &t0, &h) == 0 && h > 0);
  }
  double t0, h;
  REQUIRE(pn_ts_step_log(ts, pn_ts_steps(ts), &t0, &h) != 0);
  pn_ts_destroy(ts);
}

static void scheduler(int mode, int64_t budget, int64_t nsteps, bool known_total) {
  pn_traj *tj = pn_traj_create();
  REQUIRE(pn_traj_begin(tj, mode, budget) == 0);
  if (known_total) REQUIRE(pn_traj_set_total(tj, nsteps) == 0);
  std::map<int64_t, int64_t> content;
  for (int64_t s = 0; s <= nsteps; ++s) {
    const int64_t slot = pn_traj_fwd_slot(tj, s);
    if (slot >= 0) content[slot] = s;
    if (mode == PN_TRAJ_BUDGET) REQUIRE(pn_traj_slots_in_use(tj) <= budget);
  }
  std::vector<int64_t> ss(64), sl(64);
  for (int64_t s = nsteps - 1; s >= 0; --s) {
    int64_t fs, fl;
    int ns;
    REQUIRE(pn_traj_rev_plan(tj, s, &fs, &fl, &ns, ss.data(), sl.data(), 64) == 0);
    REQUIRE(fs <= s && content[fl] == fs);
    for (int k = 0; k < ns; ++k) content[sl[k]] = ss[k];
    REQUIRE(pn_traj_rev_done(tj, s) == 0);
  }
  if (mode == PN_TRAJ_BUDGET) REQUIRE(pn_traj_high_water(tj) <= budget);
  pn_traj_destroy(tj);
}

static void gmres() {
  const int n = 9;
  std::vector<std::vector<double>> A(n, std::vector<double>(n));
  std::vector<double> b(n);
  unsigned seed = 7;
  auto rnd = [&]() { seed = seed * 1103515245u + 12345u; return ((seed >> 16) % 2000) / 1000.0 - 1.0; };
  for (int i = 0; i < n; ++i) {
    b[i] = rnd();
    for (int j = 0; j < n; ++j) A[i][j] = (i == j ? 4.0 : 0.0) + 0.4 * rnd();
  }
  pn_gmres *g = pn_gmres_create(n);
  double beta = 0;
  for (double v : b) beta += v * v;
  beta = std::sqrt(beta);
  std::vector<std::vector<double>> V(1, b);
  for (double &v : V[0]) v /= beta;
  REQUIRE(pn_gmres_begin(g, beta) == 0);
  double res = beta;
  int k = 0;
  for (; k < n; ++k) {
    std::vector<double> w(n, 0.0), h(k + 2, 0.0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) w[i] += A[i][j] * V[k][j];
    for (int j = 0; j <= k; ++j) {
      for (int i = 0; i < n; ++i) h[j] += w[i] * V[j][i];
      for (int i = 0; i < n; ++i) w[i] -= h[j] * V[j][i];
    }
    for (double v : w) h[k + 1] += v * v;
    h[k + 1] = std::sqrt(h[k + 1]);
    REQUIRE(pn_gmres_column(g, k, h.data(), &res) == 0);
    if (res < 1e-12 || h[k + 1] < 1e-14) break;
    for (double &v : w) v /= h[k + 1];
    V.push_back(w);
  }
  if (k == n) k = n - 1;
  std::vector<double> y(k + 1), x(n, 0.0);
  REQUIRE(pn_gmres_solve(g, k, y.data()) == 0);
  for (int j = 0; j <= k; ++j)
    for (int i = 0; i < n; ++i) x[i] += y[j] * V[j][i];
  double r2 = 0;
  for (int i = 0; i < n; ++i) {
    double r = b[i];
    for (int j = 0; j < n; ++j) r -= A[i][j] * x[j];
    r2 += r * r;
  }
  REQUIRE(std::sqrt(r2) < 1e-9);
  REQUIRE(pn_gmres_column(g, n, y.data(), &res) != 0);       // out-of-range iteration is an error, not UB
  pn_gmres_destroy(g);
}

int main() {
  pn_tableau T;
  const char *names[] = {"1fe", "midpoint", "2a", "2b", "3", "3bs", "4", "5f", "5dp"};
  for (const char *nm : names) REQUIRE(pn_tableau_get(nm, &T) == 0 && T.s >= 1 && T.s <= PN_MAX_STAGES);
  stepper();
  for (int mode = 0; mode < 3; ++mode)
    for (int64_t budget : {1, 2, 3, 7, 50})
      for (int64_t n : {1, 2, 9, 100, 333}) {
        scheduler(mode, budget, n, false);
        scheduler(mode, budget, n, true);
      }
  gmres();
  std::puts("host selftest ok");
  return 0;
}
