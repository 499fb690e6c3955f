// pnode_amd -- host side of the engine: Butcher tableaus, the time-stepper state machine
// (step-size controller, exact-final-time matching, time-span bookkeeping) and the
// checkpoint scheduler.  No device code here; everything is unit-testable on a CPU-only box.
//
// The stage arithmetic lives in pn_kernels.hip and the callback into the user's dynamics
// lives above the C ABI (pnode_amd/petsc_adjoint.py), so a solve is driven as
//     pn_ts_begin -> { pn_ts_attempt -> [stages + func calls] -> pn_ts_judge }* .
// Behavioural source: PETSc TS as driven by the reference (pnode/petsc_adjoint.py:637-656,
// 768-775, 812-829); the published algorithms restated are listed per function below and
// in DESIGN.md section 3.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "pnode_amd.h"
#include "pn_internal.h"

namespace pn {
static thread_local std::string g_last_error;
int fail(const std::string &msg) {
  g_last_error = msg;
  return 1;
}
}  // namespace pn

// ------------------------------------------------------------------------------------------
// tableaus
// ------------------------------------------------------------------------------------------
namespace {

struct Row {
  int i, j;
  double num, den;
};

void set_entries(pn_tableau &T, std::initializer_list<Row> rows) {
  for (const Row &r : rows) T.A[r.i][r.j] = r.num / r.den;
}

void finish(pn_tableau &T) {
  for (int i = 0; i < T.s; ++i) {
    double ci = 0;
    for (int j = 0; j < i; ++j) ci += T.A[i][j];
    T.c[i] = ci;
  }
}

bool build_tableau(const std::string &name, pn_tableau &T) {
  std::memset(&T, 0, sizeof(T));
  if (name == "1fe") {                      // forward Euler
    T.s = 1; T.order = 1; T.b[0] = 1;
  } else if (name == "midpoint") {          // explicit midpoint (extension, not a PETSc name)
    T.s = 2; T.order = 2;
    set_entries(T, {{1, 0, 1, 2}});
    T.b[1] = 1;
  } else if (name == "2a") {                // Heun, embedded Euler
    T.s = 2; T.order = 2; T.has_embed = 1;
    set_entries(T, {{1, 0, 1, 1}});
    T.b[0] = 0.5; T.b[1] = 0.5; T.bembed[0] = 1;
  } else if (name == "2b") {                // Ralston, embedded Euler
    T.s = 2; T.order = 2; T.has_embed = 1;
    set_entries(T, {{1, 0, 2, 3}});
    T.b[0] = 0.25; T.b[1] = 0.75; T.bembed[0] = 1;
  } else if (name == "3") {
    T.s = 3; T.order = 3;
    set_entries(T, {{1, 0, 2, 3}, {2, 0, -1, 3}, {2, 1, 1, 1}});
    T.b[0] = 0.25; T.b[1] = 0.5; T.b[2] = 0.25;
  } else if (name == "3bs") {               // Bogacki-Shampine 3(2), FSAL
    T.s = 4; T.order = 3; T.fsal = 1; T.has_embed = 1;
    set_entries(T, {{1, 0, 1, 2}, {2, 1, 3, 4}, {3, 0, 2, 9}, {3, 1, 1, 3}, {3, 2, 4, 9}});
    for (int j = 0; j < 4; ++j) T.b[j] = T.A[3][j];
    const double be[4] = {7.0 / 24, 1.0 / 4, 1.0 / 3, 1.0 / 8};
    std::copy(be, be + 4, T.bembed);
  } else if (name == "4") {                 // classical RK4
    T.s = 4; T.order = 4;
    set_entries(T, {{1, 0, 1, 2}, {2, 1, 1, 2}, {3, 2, 1, 1}});
    const double b[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
    std::copy(b, b + 4, T.b);
  } else if (name == "5f") {                // Fehlberg 5(4)
    T.s = 6; T.order = 5; T.has_embed = 1;
    set_entries(T, {{1, 0, 1, 4},
                    {2, 0, 3, 32}, {2, 1, 9, 32},
                    {3, 0, 1932, 2197}, {3, 1, -7200, 2197}, {3, 2, 7296, 2197},
                    {4, 0, 439, 216}, {4, 1, -8, 1}, {4, 2, 3680, 513}, {4, 3, -845, 4104},
                    {5, 0, -8, 27}, {5, 1, 2, 1}, {5, 2, -3544, 2565}, {5, 3, 1859, 4104}, {5, 4, -11, 40}});
    const double b[6] = {16.0 / 135, 0, 6656.0 / 12825, 28561.0 / 56430, -9.0 / 50, 2.0 / 55};
    const double be[6] = {25.0 / 216, 0, 1408.0 / 2565, 2197.0 / 4104, -1.0 / 5, 0};
    std::copy(b, b + 6, T.b);
    std::copy(be, be + 6, T.bembed);
  } else if (name == "5dp") {               // Dormand-Prince 5(4), FSAL
    T.s = 7; T.order = 5; T.fsal = 1; T.has_embed = 1;
    set_entries(T, {{1, 0, 1, 5},
                    {2, 0, 3, 40}, {2, 1, 9, 40},
                    {3, 0, 44, 45}, {3, 1, -56, 15}, {3, 2, 32, 9},
                    {4, 0, 19372, 6561}, {4, 1, -25360, 2187}, {4, 2, 64448, 6561}, {4, 3, -212, 729},
                    {5, 0, 9017, 3168}, {5, 1, -355, 33}, {5, 2, 46732, 5247}, {5, 3, 49, 176}, {5, 4, -5103, 18656},
                    {6, 0, 35, 384}, {6, 2, 500, 1113}, {6, 3, 125, 192}, {6, 4, -2187, 6784}, {6, 5, 11, 84}});
    for (int j = 0; j < 7; ++j) T.b[j] = T.A[6][j];
    const double be[7] = {5179.0 / 57600, 0, 7571.0 / 16695, 393.0 / 640, -92097.0 / 339200, 187.0 / 2100, 1.0 / 40};
    std::copy(be, be + 7, T.bembed);
  } else {
    return false;
  }
  finish(T);
  return true;
}

constexpr double kEps = std::numeric_limits<double>::epsilon();

}  // namespace

// ------------------------------------------------------------------------------------------
// time stepper
// ------------------------------------------------------------------------------------------
struct pn_ts {
  pn_tableau tab;
  std::string rk_type = "3bs";          // PETSc's default RK tableau
  // controller (PETSc TSAdapt defaults)
  bool adapt_basic = true;
  double atol = 1e-4, rtol = 1e-4;
  double safety = 0.9, reject_safety = 0.5, clip_lo = 0.1, clip_hi = 10.0;
  double dt_min = 1e-20, dt_max = 1e50;
  double match_stretch = 0.01, match_halve = 2.0;   // matchstepfac
  int max_reject = 10;
  // TSCreate: max_steps = PETSC_MAX_INT (the 5000 of old manual pages is long gone: the reference's own
  // examples-pnode/spiral_unstable.py:93-104 takes 16 000 steps of 0.001 without -ts_max_steps)
  int64_t max_steps = INT64_MAX;
  double span_reltol = 1e-6, span_abstol = 10 * kEps;
  // solve state
  double ptime = 0, time_step = 0.01, max_time = 0;
  std::vector<double> span;
  int spanctr = 0;
  double dt_span_cached = 0;
  int64_t steps = 0, rejections = 0;
  int rejections_this_step = 0;
  bool prev_attempt_rejected = false;
  bool finished = true;
  std::vector<double> log_t, log_h;
  // scheme whose stages are computed above the ABI (ARKIMEX, theta): order and embedded-estimate flag for the controller
  int ext_order = 0;
  bool ext_embed = false;
};

extern "C" {

const char *pn_last_error(void) { return pn::g_last_error.c_str(); }
int pn_abi_version(void) { return PN_ABI_VERSION; }

int pn_tableau_get(const char *rk_type, pn_tableau *out) {
  if (!rk_type || !out) return pn::fail("pn_tableau_get: null argument");
  if (!build_tableau(rk_type, *out)) return pn::fail(std::string("unknown RK type '") + rk_type + "'");
  return 0;
}

const char *pn_method_to_rk_type(const char *method) {
  const std::string m = method ? method : "";
  if (m == "euler") return "1fe";
  if (m == "rk2") return "2b";
  if (m == "bosh3" || m == "fixed_bosh3") return "3bs";
  if (m == "rk4") return "4";
  if (m == "dopri5" || m == "fixed_dopri5") return "5dp";
  if (m == "midpoint") return "midpoint";
  return "3bs";
}

pn_ts *pn_ts_create(void) {
  pn_ts *ts = new pn_ts();
  build_tableau(ts->rk_type, ts->tab);
  return ts;
}
void pn_ts_destroy(pn_ts *ts) { delete ts; }

int pn_ts_set_rk_type(pn_ts *ts, const char *rk_type) {
  pn_tableau T;
  if (!rk_type || !build_tableau(rk_type, T)) return pn::fail(std::string("unknown RK type '") + (rk_type ? rk_type : "") + "'");
  ts->tab = T;
  ts->rk_type = rk_type;
  return 0;
}
int pn_ts_get_tableau(const pn_ts *ts, pn_tableau *out) {
  *out = ts->tab;
  return 0;
}

static bool parse_double(const char *v, double *out) {
  if (!v) return false;
  char *end = nullptr;
  *out = std::strtod(v, &end);
  return end != v;
}

int pn_ts_set_option(pn_ts *ts, const char *key, const char *value) {
  const std::string k = key ? key : "";
  double d = 0;
  if (k == "ts_adapt_type") {
    const std::string v = value ? value : "";
    if (v == "none") ts->adapt_basic = false;
    else if (v == "basic") ts->adapt_basic = true;
    else return pn::fail("ts_adapt_type: only 'none' and 'basic' are implemented");
    return 0;
  }
  if (k == "ts_rk_type") return pn_ts_set_rk_type(ts, value);
  if (k == "ts_type") {
    if (std::string(value ? value : "") != "rk") return pn::fail("ts_type: only 'rk' is implemented on this path");
    return 0;
  }
  if (k == "ts_adapt_clip") {
    double lo, hi;
    if (!value || std::sscanf(value, "%lf,%lf", &lo, &hi) != 2) return pn::fail("ts_adapt_clip wants 'lo,hi'");
    ts->clip_lo = lo; ts->clip_hi = hi;
    return 0;
  }
  if (!parse_double(value, &d)) return pn::fail("option '" + k + "' needs a numeric value");
  if (k == "ts_rtol") ts->rtol = d;
  else if (k == "ts_atol") ts->atol = d;
  else if (k == "ts_max_steps") ts->max_steps = (int64_t)d;
  else if (k == "ts_max_reject") ts->max_reject = (int)d;
  else if (k == "ts_adapt_safety") ts->safety = d;
  else if (k == "ts_adapt_reject_safety") ts->reject_safety = d;
  else if (k == "ts_adapt_dt_min") ts->dt_min = d;
  else if (k == "ts_adapt_dt_max") ts->dt_max = d;
  else return pn::fail("unknown option '" + k + "'");
  return 0;
}

int pn_ts_is_adaptive(const pn_ts *ts) { return ts->adapt_basic && (ts->ext_order > 0 ? ts->ext_embed : ts->tab.has_embed != 0); }
int pn_ts_set_scheme(pn_ts *ts, int order, int has_embed) {
  if (order < 0) return pn::fail("pn_ts_set_scheme: order must not be negative");
  ts->ext_order = order;               // 0: back to the RK tableau
  ts->ext_embed = has_embed != 0;
  return 0;
}
int pn_ts_get_tolerances(const pn_ts *ts, double *atol, double *rtol) {
  *atol = ts->atol; *rtol = ts->rtol;
  return 0;
}

static double next_target(const pn_ts *ts) {
  if (!ts->span.empty() && ts->spanctr < (int)ts->span.size()) return ts->span[ts->spanctr];
  return ts->max_time;
}

static bool close_rel(double a, double b, double rtol) {
  return std::fabs(a - b) <= rtol * std::max(std::fabs(a), std::fabs(b));
}

int pn_ts_begin(pn_ts *ts, double t0, double dt0, int nspan, const double *span) {
  if (nspan < 1 || !span) return pn::fail("pn_ts_begin: need at least one time");
  if (!(dt0 > 0)) return pn::fail("pn_ts_begin: step size must be positive");
  ts->span.clear();
  if (nspan == 1) {
    ts->ptime = t0;
    ts->max_time = span[0];
    ts->spanctr = 0;
  } else {
    for (int i = 1; i < nspan; ++i)
      if (!(span[i] > span[i - 1])) return pn::fail("pn_ts_begin: time span must be strictly increasing");
    ts->span.assign(span, span + nspan);
    ts->ptime = span[0];
    ts->max_time = span[nspan - 1];
    ts->spanctr = 1;                 // span[0] is the initial condition itself
  }
  ts->time_step = dt0;
  ts->dt_span_cached = 0;
  ts->steps = 0;
  ts->rejections = 0;
  ts->rejections_this_step = 0;
  ts->prev_attempt_rejected = false;
  ts->log_t.clear();
  ts->log_h.clear();
  ts->finished = !(ts->ptime < ts->max_time);
  // exact-final-time MATCHSTEP at solve start: clamp the first step to the first target
  const double maxdt = next_target(ts) - ts->ptime;
  if (maxdt > 0 && (dt0 >= maxdt || close_rel(dt0, maxdt, 10 * kEps))) {
    if (!ts->span.empty() && dt0 > maxdt) ts->dt_span_cached = dt0;
    ts->time_step = maxdt;
  }
  return 0;
}

int pn_ts_attempt(const pn_ts *ts, double *t, double *h) {
  if (ts->finished) return pn::fail("pn_ts_attempt: solve already finished");
  *t = ts->ptime;
  *h = ts->time_step;
  return 0;
}

// TSAdaptChoose (none | basic) followed by the MATCHSTEP / time-span adjustment, then the
// bookkeeping TSSolve does after an accepted step.
int pn_ts_judge(pn_ts *ts, double enorm, int *accept_out, int *hit_span, int *done) {
  if (ts->finished) return pn::fail("pn_ts_judge: solve already finished");
  const pn_tableau &T = ts->tab;
  const double h = ts->time_step;
  bool accept = true;
  double hnew = h;
  *hit_span = -1;
  *done = 0;
  if (enorm >= 0 || enorm != enorm) {
    if (!(enorm == enorm) || std::isinf(enorm)) {
      ts->finished = true;
      return pn::fail("Infinite or not-a-number generated in the error norm");
    }
    double safety = ts->safety;
    if (enorm > 1.0) {
      if (ts->prev_attempt_rejected) safety *= ts->reject_safety;
      accept = h < (1 + std::sqrt(kEps)) * ts->dt_min;
    }
    const int order = ts->ext_order > 0 ? ts->ext_order : T.order;
    double hfac = enorm > 0 ? safety * std::pow(enorm, -1.0 / (double)order)
                            : std::numeric_limits<double>::infinity();
    hfac = std::min(std::max(hfac, ts->clip_lo), ts->clip_hi);
    hnew = std::min(std::max(h * hfac, ts->dt_min), ts->dt_max);
  }
  if (!accept) {
    ts->time_step = hnew;
    ts->rejections++;
    ts->prev_attempt_rejected = true;
    *accept_out = 0;
    if (++ts->rejections_this_step > ts->max_reject && ts->max_reject >= 0) {
      ts->finished = true;
      return pn::fail("TS diverged: step rejected more than ts_max_reject times");
    }
    return 0;
  }
  // --- accepted: choose the next step so that every target time is hit exactly
  double t = ts->ptime + h;
  {
    double tend;
    if (!ts->span.empty()) {
      const bool hit = ts->spanctr < (int)ts->span.size() &&
                       std::fabs(t - ts->span[ts->spanctr]) <= ts->span_reltol * std::fabs(h) + ts->span_abstol;
      if (hit) {
        tend = ts->spanctr + 1 < (int)ts->span.size() ? ts->span[ts->spanctr + 1] : ts->max_time;
        if (ts->dt_span_cached > 0) {
          // the steps that approached this point were cut (or stretched) to land on it: go back to the step that was
          // wanted before the first of those adjustments -- unless the controller has chosen a new one meanwhile
          if (hnew == h) hnew = ts->dt_span_cached;
          ts->dt_span_cached = 0;
        }
      } else {
        tend = next_target(ts);
      }
    } else {
      tend = ts->max_time;
    }
    if (t < tend) {
      const double hmax = tend - t, wanted = hnew;
      if (wanted * ts->match_halve > hmax) hnew = hmax / 2;
      if (wanted * (1.0 + ts->match_stretch) > hmax) hnew = hmax;
      // remember the unadjusted step ONCE per approach: a halved step that is later stretched onto the point must
      // not replace the user's step in the cache (it would never come back)
      if (!ts->span.empty() && hnew != wanted && !(ts->dt_span_cached > 0)) ts->dt_span_cached = wanted;
    }
  }
  ts->log_t.push_back(ts->ptime);
  ts->log_h.push_back(h);
  // land exactly on the target when the matched step is within round-off of it
  const double tgt = next_target(ts);
  if (t != tgt && close_rel(t, tgt, 16 * kEps)) t = tgt;
  const double tprev = ts->ptime;
  ts->ptime = t;
  ts->time_step = hnew;
  ts->steps++;
  ts->prev_attempt_rejected = false;
  ts->rejections_this_step = 0;
  if (!ts->span.empty() && ts->spanctr < (int)ts->span.size() &&
      std::fabs(t - ts->span[ts->spanctr]) <= ts->span_reltol * std::fabs(t - tprev) + ts->span_abstol) {
    *hit_span = ts->spanctr;
    ts->spanctr++;
  }
  *accept_out = 1;
  if (ts->ptime >= ts->max_time) {
    ts->finished = true;
    *done = 1;
  } else if (ts->steps >= ts->max_steps) {
    ts->finished = true;
    *done = 2;                         // TS_CONVERGED_ITS: stopped by ts_max_steps
  }
  return 0;
}

int64_t pn_ts_count_fixed_steps(const pn_ts *ts) {
  if (pn_ts_is_adaptive(ts)) return -1;
  pn_ts copy = *ts;
  int acc, hit, done = copy.finished ? 1 : 0;
  while (!done) {
    if (pn_ts_judge(&copy, -1.0, &acc, &hit, &done)) return -1;
  }
  return copy.steps;
}

int pn_ts_override_next_dt(pn_ts *ts, double dt) {
  if (!(dt > 0)) return pn::fail("pn_ts_override_next_dt: step size must be positive");
  ts->time_step = dt;
  return 0;
}
int64_t pn_ts_steps(const pn_ts *ts) { return ts->steps; }
int64_t pn_ts_rejections(const pn_ts *ts) { return ts->rejections; }
double pn_ts_time(const pn_ts *ts) { return ts->ptime; }
int pn_ts_step_log(const pn_ts *ts, int64_t k, double *t_start, double *h) {
  if (k < 0 || k >= (int64_t)ts->log_t.size()) return pn::fail("pn_ts_step_log: step out of range");
  *t_start = ts->log_t[k];
  *h = ts->log_h[k];
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// checkpoint scheduler
// ------------------------------------------------------------------------------------------
// Optimal placement of checkpoints (revolve-type dynamic programme).
//   cost(l, c): fewest re-advanced steps needed to reverse l consecutive steps when the state at
//   their start is kept and c slots (that one included) are available; the state a reversal
//   needs right now lives in a work buffer and takes no slot.
//     cost(1, c) = 0,  cost(l, 1) = l(l-1)/2,
//     cost(l, c) = min_{1<=m<l}  m + cost(l-m, c-1) + cost(m, c)
//   first(l, c): the same for the ORIGINAL forward sweep, whose own advance is not a re-advance:
//     first(l, c) = min_m  first(l-m, c-1) + cost(m, c)
// The minimising m is non-decreasing in l, so each row is filled with a moving pointer.
struct CheckpointDP {
  static constexpr int kMaxSlots = 256;   // budgets above this are planned as if they were this (never worse)
  static constexpr int64_t kMaxSteps = 8192;
  int64_t L = 0;
  int C = 0;
  // priced = CAMS-type cost model for checkpoints that carry the stage values of their step: the unit is one step of
  // stage computation, and reversing a step costs one unit MORE unless its checkpoint already holds its stage values
  // (they are written, for free, whenever a sweep steps on from a checkpoint):
  //     cost(1, c) = 0                        (the segment's first advance has just filled the checkpoint's stages)
  //     open(1, c) = 1,  open(l>1, c) = cost(l, c)      (a segment whose start has not been stepped on from yet;
  //                                                      a single step there is reversed from the work buffer)
  //     cost(l, 1) = l(l-1)/2 + (l-1)
  //     cost(l, c) = min_m  m + open(l-m, c-1) + cost(m, c)
  //     first(l, c) = min_m  first(l-m, c-1) + cost(m, c)   (the original sweep steps on from every state it keeps)
  bool priced = false;
  std::vector<std::vector<int64_t>> cost, first;
  std::vector<std::vector<int32_t>> arg_cost, arg_first;

  bool full = false;                     // priced tables: every split was tried (see wants_full)
  bool covers(int64_t l, int c, bool pr) const { return pr == priced && l <= L && c <= C; }
  static bool feasible(int64_t l, int64_t c) { return l >= 1 && l <= kMaxSteps && c >= 1; }
  // In the priced model the minimiser is not monotone in l (the lone-step term): every split is tried while that is
  // affordable (l^2 c <= 2e8, i.e. a thousand steps with 200 slots), a window around the previous minimiser beyond
  // that (then the plan may be a step or two of work off the optimum).
  static bool wants_full(int64_t l, int c, bool pr) { return pr && (double)l * (double)l * (double)c <= 2e8; }

  void build(int64_t l_max, int c_max, bool pr) {
    c_max = std::min(c_max, kMaxSlots);
    priced = pr;
    L = l_max;
    C = c_max;
    full = wants_full(L, C, pr);
    cost.assign(C + 1, std::vector<int64_t>(L + 1, 0));
    first.assign(C + 1, std::vector<int64_t>(L + 1, 0));
    arg_cost.assign(C + 1, std::vector<int32_t>(L + 1, 0));
    arg_first.assign(C + 1, std::vector<int32_t>(L + 1, 0));
    const int64_t extra = priced ? 1 : 0;
    for (int64_t l = 2; l <= L; ++l) cost[1][l] = first[1][l] = l * (l - 1) / 2 + extra * (l - 1);
    for (int c = 2; c <= C; ++c) {
      for (int which = 0; which < 2; ++which) {
        auto &tab = which == 0 ? cost[c] : first[c];
        auto &arg = which == 0 ? arg_cost[c] : arg_first[c];
        const auto &right = which == 0 ? cost[c - 1] : first[c - 1];
        // right part of the reverse recursion: a lone step is reversed from the work buffer (its stages are computed)
        auto value = [&](int64_t l, int64_t m) {
          const int64_t r = (which == 0 && l - m == 1) ? extra : right[l - m];
          return (which == 0 ? m : 0) + r + cost[c][m];
        };
        int64_t mp = 1;
        for (int64_t l = 2; l <= L; ++l) {
          // cost[c][m] for m < l is already final (row filled left to right; `first` needs cost[c])
          int64_t m = full ? 1 : std::max<int64_t>(1, std::min<int64_t>(mp, l - 1) - (priced ? 2 : 0));
          int64_t best_m = m, best = value(l, m), worse = 0;
          ++m;
          while (m <= l - 1 && (full || worse < (priced ? 8 : 4))) {
            const int64_t v = value(l, m);
            if (v < best) { best = v; best_m = m; worse = 0; }
            else ++worse;
            ++m;
          }
          tab[l] = best;
          arg[l] = (int32_t)best_m;
          mp = best_m;
        }
      }
    }
  }
  int64_t split_cost(int64_t l, int c) const { return arg_cost[std::min(c, C)][l]; }
  int64_t split_first(int64_t l, int c) const { return arg_first[std::min(c, C)][l]; }
};

// The tables depend on (priced, L, C) only, and a training loop asks for the same ones at every solve (a new pn_traj per
// forward sweep): they are kept process-wide.  Building the priced table for 1000 steps x 200 slots takes 0.15-0.2 s,
// which used to be stalled host time in EVERY odeint.  A request is served by any cached table that covers it
// and is at least as exact as a fresh build for that request would be.
namespace {
std::mutex g_dp_mu;
std::vector<std::shared_ptr<const CheckpointDP>> g_dp_cache;
int64_t g_dp_builds = 0;

std::shared_ptr<const CheckpointDP> dp_table(int64_t l, int c, bool priced) {
  c = std::min(c, CheckpointDP::kMaxSlots);
  std::lock_guard<std::mutex> lk(g_dp_mu);
  for (const auto &t : g_dp_cache)
    if (t->covers(l, c, priced) && (t->full || !CheckpointDP::wants_full(l, c, priced))) return t;
  auto t = std::make_shared<CheckpointDP>();
  t->build(l, c, priced);
  ++g_dp_builds;
  // drop what the new table makes redundant, keep the cache small
  g_dp_cache.erase(std::remove_if(g_dp_cache.begin(), g_dp_cache.end(),
                                  [&](const std::shared_ptr<const CheckpointDP> &o) {
                                    return t->covers(o->L, o->C, o->priced) && (t->full || !o->full);
                                  }),
                   g_dp_cache.end());
  if (g_dp_cache.size() >= 8) g_dp_cache.erase(g_dp_cache.begin());
  g_dp_cache.push_back(t);
  return t;
}
}  // namespace

struct pn_traj {
  int mode = PN_TRAJ_ALL;
  int64_t total = -1;                  // number of steps of the forward sweep when known in advance
  std::vector<char> planned;           // BUDGET + known total: states the forward sweep keeps
  std::shared_ptr<const CheckpointDP> dp;   // the table the last plan used (process-wide cache, dp_table)
  int64_t max_slots = 0;               // BUDGET mode only
  std::map<int64_t, int64_t> kept;     // step -> slot
  std::vector<int64_t> free_slots;
  std::vector<int64_t> deferred;       // released, but still being read by the step in flight
  int64_t next_new = 0;                // next never-used slot index
  int64_t stride = 1;                  // BUDGET forward thinning stride
  int64_t high_water = 0;
  bool carry = false;                  // checkpoints carry their step's stage values (priced placement)

  bool bounded() const { return mode == PN_TRAJ_BUDGET; }
  int64_t n_free() const {
    return (int64_t)free_slots.size() + (bounded() ? std::max<int64_t>(0, max_slots - next_new) : (int64_t)1 << 40);
  }
  int64_t take() {
    int64_t s;
    if (!free_slots.empty()) {
      // lowest index first keeps the slab compact
      auto it = std::min_element(free_slots.begin(), free_slots.end());
      s = *it;
      free_slots.erase(it);
    } else {
      s = next_new++;
    }
    high_water = std::max<int64_t>(high_water, (int64_t)kept.size() + 1);
    return s;
  }
};

extern "C" {

pn_traj *pn_traj_create(void) { return new pn_traj(); }
void pn_traj_destroy(pn_traj *tj) { delete tj; }

int pn_traj_begin(pn_traj *tj, int mode, int64_t max_slots) {
  if (mode < PN_TRAJ_ALL || mode > PN_TRAJ_BUDGET) return pn::fail("pn_traj_begin: bad mode");
  if (mode == PN_TRAJ_BUDGET && max_slots < 1) return pn::fail("pn_traj_begin: a budget needs at least one slot");
  tj->mode = mode;
  tj->max_slots = max_slots;
  tj->kept.clear();
  tj->free_slots.clear();
  tj->deferred.clear();
  tj->next_new = 0;
  tj->stride = 1;
  tj->high_water = 0;
  tj->total = -1;
  tj->planned.clear();
  tj->carry = false;
  return 0;
}

int pn_traj_set_carry(pn_traj *tj, int carries_stage_values) {
  tj->carry = carries_stage_values != 0;
  return 0;
}

int pn_traj_set_total(pn_traj *tj, int64_t nsteps) {
  tj->total = nsteps;
  tj->planned.clear();
  if (tj->mode != PN_TRAJ_BUDGET || nsteps < 1 || tj->max_slots < 1) return 0;
  if (tj->max_slots >= nsteps) {          // room for the state at the start of every step: keep them all
    tj->planned.assign((size_t)nsteps + 1, 1);
    tj->planned[(size_t)nsteps] = 0;      // the end state is never restored
    return 0;
  }
  if (!CheckpointDP::feasible(nsteps, tj->max_slots)) return 0;
  const int c = (int)std::min<int64_t>(tj->max_slots, CheckpointDP::kMaxSlots);
  if (!tj->dp || !tj->dp->covers(nsteps, c, tj->carry) || (!tj->dp->full && CheckpointDP::wants_full(nsteps, c, tj->carry)))
    tj->dp = dp_table(nsteps, c, tj->carry);
  // the chain of states the original sweep keeps: 0, then the optimal split of what is left
  tj->planned.assign((size_t)nsteps + 1, 0);
  tj->planned[0] = 1;
  int64_t pos = 0, left = nsteps;
  for (int cc = c; cc >= 2 && left > 1; --cc) {
    const int64_t m = tj->dp->split_first(left, cc);
    pos += m;
    left -= m;
    if (left >= 1) tj->planned[(size_t)pos] = 1;
  }
  return 0;
}

int64_t pn_traj_fwd_slot(pn_traj *tj, int64_t step) {
  if (!tj->bounded()) {
    const int64_t s = tj->take();
    tj->kept[step] = s;
    return s;
  }
  if (!tj->planned.empty()) {            // step count known: optimal placement (pn_traj_set_total)
    if (step < 0 || step >= (int64_t)tj->planned.size() || !tj->planned[(size_t)step]) return -1;
    if (tj->n_free() <= 0) return -1;
    const int64_t sl = tj->take();
    tj->kept[step] = sl;
    return sl;
  }
  // online thinning: keep the states at multiples of `stride`; when the budget is full, double
  // the stride and drop the odd multiples.  Step 0 is always kept.  The state of step-1 is the
  // input of the step being computed (and must survive a rejected attempt), so if it is
  // dropped its slot is only recycled from the next call on.
  tj->free_slots.insert(tj->free_slots.end(), tj->deferred.begin(), tj->deferred.end());
  tj->deferred.clear();
  for (;;) {
    if (step % tj->stride != 0) return -1;
    if (tj->n_free() > 0) {
      const int64_t s = tj->take();
      tj->kept[step] = s;
      return s;
    }
    tj->stride *= 2;
    for (auto it = tj->kept.begin(); it != tj->kept.end();) {
      if (it->first % tj->stride != 0) {
        (it->first == step - 1 ? tj->deferred : tj->free_slots).push_back(it->second);
        it = tj->kept.erase(it);
      } else {
        ++it;
      }
    }
  }
}

int pn_traj_rev_plan(pn_traj *tj, int64_t step, int64_t *from_step, int64_t *from_slot, int *nstore,
                     int64_t *store_step, int64_t *store_slot, int cap) {
  auto it = tj->kept.upper_bound(step);
  if (it == tj->kept.begin()) return pn::fail("pn_traj_rev_plan: no checkpoint at or before the requested step");
  --it;
  *from_step = it->first;
  *from_slot = it->second;
  *nstore = 0;
  const int64_t L = step - it->first;       // steps to re-advance
  if (L <= 1) return 0;
  // Reversal proceeds newest-first, so everything after `step` is already released: the open
  // problem is "reverse the L+1 steps starting at from_step with the free slots".  With the
  // dynamic programme available the states to keep on the way are its optimal splits.
  {
    const int64_t span = L + 1, avail = std::min<int64_t>(tj->n_free(), cap);
    if (tj->bounded() && avail >= 1 && CheckpointDP::feasible(span, avail + 1)) {
      const int c0 = (int)std::min<int64_t>(avail + 1, CheckpointDP::kMaxSlots);
      if (!tj->dp || !tj->dp->covers(span, c0, tj->carry) || (!tj->dp->full && CheckpointDP::wants_full(span, c0, tj->carry)))
        tj->dp = dp_table(span, c0, tj->carry);
      int64_t pos = it->first, left = span;
      for (int cc = c0; cc >= 2 && left > 1; --cc) {
        const int64_t m = tj->dp->split_cost(left, cc);
        pos += m;
        left -= m;
        if (pos >= step) break;              // the target itself stays in the work buffer
        const int64_t sl = tj->take();
        tj->kept[pos] = sl;
        store_step[*nstore] = pos;
        store_slot[*nstore] = sl;
        ++*nstore;
      }
      return 0;
    }
  }
  // fallback (very long sweeps): spread the free slots evenly over the open interval
  int64_t k = std::min<int64_t>(std::min<int64_t>(tj->n_free(), L - 1), cap);
  for (int64_t i = 1; i <= k; ++i) {
    const int64_t st = it->first + (i * L) / (k + 1);
    if (st <= it->first || st >= step) continue;
    if (*nstore > 0 && store_step[*nstore - 1] == st) continue;
    const int64_t sl = tj->take();
    tj->kept[st] = sl;
    store_step[*nstore] = st;
    store_slot[*nstore] = sl;
    ++*nstore;
  }
  return 0;
}

int pn_traj_rev_done(pn_traj *tj, int64_t step) {
  for (auto it = tj->kept.lower_bound(step); it != tj->kept.end();) {
    tj->free_slots.push_back(it->second);
    it = tj->kept.erase(it);
  }
  return 0;
}

int64_t pn_traj_dp_builds(void) {
  std::lock_guard<std::mutex> lk(g_dp_mu);
  return g_dp_builds;
}
int64_t pn_traj_slots_in_use(const pn_traj *tj) { return (int64_t)tj->kept.size(); }
int64_t pn_traj_high_water(const pn_traj *tj) { return tj->high_water; }

}  // extern "C"

// ------------------------------------------------------------------------------------------
// GMRES core (small dense part): Hessenberg + Givens rotations, as KSPGMRES keeps them
// ------------------------------------------------------------------------------------------
struct pn_gmres {
  int m;
  std::vector<double> H;      // (m+1) x m, column-major by iteration: H[k*(m+1) + i]
  std::vector<double> cs, sn, g;
};

extern "C" {

pn_gmres *pn_gmres_create(int restart) {
  if (restart < 1) restart = 30;
  pn_gmres *g = new pn_gmres();
  g->m = restart;
  g->H.assign((size_t)(restart + 1) * restart, 0.0);
  g->cs.assign(restart, 0.0);
  g->sn.assign(restart, 0.0);
  g->g.assign(restart + 1, 0.0);
  return g;
}
void pn_gmres_destroy(pn_gmres *g) { delete g; }

int pn_gmres_begin(pn_gmres *g, double beta) {
  std::fill(g->H.begin(), g->H.end(), 0.0);
  std::fill(g->g.begin(), g->g.end(), 0.0);
  g->g[0] = beta;
  return 0;
}

int pn_gmres_column(pn_gmres *g, int k, const double *h, double *resnorm) {
  if (k < 0 || k >= g->m) return pn::fail("pn_gmres_column: iteration index beyond the restart length");
  double *col = &g->H[(size_t)k * (g->m + 1)];
  for (int i = 0; i <= k + 1; ++i) col[i] = h[i];
  for (int i = 0; i < k; ++i) {                     // previous rotations
    const double t = g->cs[i] * col[i] + g->sn[i] * col[i + 1];
    col[i + 1] = -g->sn[i] * col[i] + g->cs[i] * col[i + 1];
    col[i] = t;
  }
  const double a = col[k], b = col[k + 1];
  const double r = std::hypot(a, b);
  if (r == 0.0) {
    g->cs[k] = 1.0; g->sn[k] = 0.0;
  } else {
    g->cs[k] = a / r; g->sn[k] = b / r;
  }
  col[k] = r;
  col[k + 1] = 0.0;
  g->g[k + 1] = -g->sn[k] * g->g[k];
  g->g[k] = g->cs[k] * g->g[k];
  *resnorm = std::fabs(g->g[k + 1]);
  return 0;
}

int pn_gmres_solve(pn_gmres *g, int k, double *y) {
  if (k < 0 || k >= g->m) return pn::fail("pn_gmres_solve: iteration index beyond the restart length");
  for (int i = k; i >= 0; --i) {
    double s = g->g[i];
    for (int j = i + 1; j <= k; ++j) s -= g->H[(size_t)j * (g->m + 1) + i] * y[j];
    const double d = g->H[(size_t)i * (g->m + 1) + i];
    if (d == 0.0) return pn::fail("pn_gmres_solve: singular Hessenberg (breakdown)");
    y[i] = s / d;
  }
  return 0;
}

}  // extern "C"
