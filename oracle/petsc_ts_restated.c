/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * CPU restatement of the slice of PETSc's TS library that caidao22/pnode drives on
 * its explicit-RK path (reference: pnode/petsc_adjoint.py, "pa.py" below).  PETSc is an
 * un-vendored, un-pinned third-party dependency of the reference
 * (.github/workflows/build.sh:4 clones petsc `main`; setup.py:9 pins nothing) and is
 * not installable in this image, so its published algorithms are restated here in
 * plain C and pinned by
 *   (a) the known-answer constants of the reference's own test
 *       (tests/test_pnode.py:183-201, explicit RK on ROBER), and
 *   (b) fp64 autograd through the unrolled RK steps (oracle/autograd_rk.py).
 * What (a)+(b) cannot pin -- the adaptive controller's accepted-step sequence against
 * a real PETSc build -- is "parity unpinned"; see DESIGN.md section 3.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * What is restated (PETSc object -> here), and the reference call site that selects it:
 *   TSRK tableaus 1fe/2a/2b/3/3bs/4/5f/5dp        pa.py:641-650  (ts.setRKType)
 *   TSStep_RK (stage loop, FSAL, rollback)          pa.py:829      (ts.solve)
 *   TSAdaptChoose none|basic + WRMS norm            pa.py:775      (setFromOptions)
 *   exact-final-time MATCHSTEP + time span          pa.py:640, 822
 *   TSTrajectory "memory" (solution-only | stages)  pa.py:771-772  (setSaveTrajectory)
 *   TSAdjointStep_RK                                pa.py:875-878  (adjointSolve)
 *   VecCopy / VecMAXPY / VecAXPY / VecScale / VecSet  (unfused, sequential -- VecSeq
 *   under COMM_SELF, pa.py:367)
 *
 * The file is compiled twice (REAL=double / REAL=float) into one shared object, the
 * way PETSc itself is built for exactly one scalar width (tests/test_pnode.py:127-130).
 * Time, step size and controller arithmetic are kept in double in both builds.
 */
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
#ifndef SFX
#define SFX _f64
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

#define MAXS 7

/* ------------------------------------------------------------------ tableaus */
typedef struct {
  const char *name;
  int s, order, fsal, has_embed;
  double A[MAXS][MAXS], b[MAXS], be[MAXS], c[MAXS];
} Tableau;

static void tab_finish(Tableau *T) {
  for (int i = 0; i < T->s; i++) {
    double ci = 0;
    for (int j = 0; j < T->s; j++) ci += T->A[i][j];
    T->c[i] = ci;
  }
}

static int tab_lookup(const char *name, Tableau *T) {
  memset(T, 0, sizeof(*T));
  if (!strcmp(name, "1fe")) {
    T->name = "1fe"; T->s = 1; T->order = 1; T->b[0] = 1.0;
  } else if (!strcmp(name, "2a")) {
    T->name = "2a"; T->s = 2; T->order = 2; T->has_embed = 1;
    T->A[1][0] = 1.0; T->b[0] = 0.5; T->b[1] = 0.5; T->be[0] = 1.0;
  } else if (!strcmp(name, "2b")) {
    T->name = "2b"; T->s = 2; T->order = 2; T->has_embed = 1;
    T->A[1][0] = 2.0 / 3.0; T->b[0] = 0.25; T->b[1] = 0.75; T->be[0] = 1.0;
  } else if (!strcmp(name, "midpoint")) {
    /* not a PETSc tableau: explicit midpoint, added because BASELINE.json's north_star names
     * it (the reference's method map has no such key); fixed step, no embedding */
    T->name = "midpoint"; T->s = 2; T->order = 2;
    T->A[1][0] = 0.5; T->b[1] = 1.0;
  } else if (!strcmp(name, "3")) {
    T->name = "3"; T->s = 3; T->order = 3;
    T->A[1][0] = 2.0 / 3.0; T->A[2][0] = -1.0 / 3.0; T->A[2][1] = 1.0;
    T->b[0] = 0.25; T->b[1] = 0.5; T->b[2] = 0.25;
  } else if (!strcmp(name, "3bs")) {
    T->name = "3bs"; T->s = 4; T->order = 3; T->fsal = 1; T->has_embed = 1;
    T->A[1][0] = 0.5; T->A[2][1] = 0.75;
    T->A[3][0] = 2.0 / 9.0; T->A[3][1] = 1.0 / 3.0; T->A[3][2] = 4.0 / 9.0;
    T->b[0] = 2.0 / 9.0; T->b[1] = 1.0 / 3.0; T->b[2] = 4.0 / 9.0;
    T->be[0] = 7.0 / 24.0; T->be[1] = 0.25; T->be[2] = 1.0 / 3.0; T->be[3] = 0.125;
  } else if (!strcmp(name, "4")) {
    T->name = "4"; T->s = 4; T->order = 4;
    T->A[1][0] = 0.5; T->A[2][1] = 0.5; T->A[3][2] = 1.0;
    T->b[0] = 1.0 / 6.0; T->b[1] = 1.0 / 3.0; T->b[2] = 1.0 / 3.0; T->b[3] = 1.0 / 6.0;
  } else if (!strcmp(name, "5f")) {
    T->name = "5f"; T->s = 6; T->order = 5; T->has_embed = 1;
    T->A[1][0] = 0.25;
    T->A[2][0] = 3.0 / 32.0; T->A[2][1] = 9.0 / 32.0;
    T->A[3][0] = 1932.0 / 2197.0; T->A[3][1] = -7200.0 / 2197.0; T->A[3][2] = 7296.0 / 2197.0;
    T->A[4][0] = 439.0 / 216.0; T->A[4][1] = -8.0; T->A[4][2] = 3680.0 / 513.0; T->A[4][3] = -845.0 / 4104.0;
    T->A[5][0] = -8.0 / 27.0; T->A[5][1] = 2.0; T->A[5][2] = -3544.0 / 2565.0; T->A[5][3] = 1859.0 / 4104.0; T->A[5][4] = -11.0 / 40.0;
    T->b[0] = 16.0 / 135.0; T->b[2] = 6656.0 / 12825.0; T->b[3] = 28561.0 / 56430.0; T->b[4] = -9.0 / 50.0; T->b[5] = 2.0 / 55.0;
    T->be[0] = 25.0 / 216.0; T->be[2] = 1408.0 / 2565.0; T->be[3] = 2197.0 / 4104.0; T->be[4] = -1.0 / 5.0;
  } else if (!strcmp(name, "5dp")) {
    T->name = "5dp"; T->s = 7; T->order = 5; T->fsal = 1; T->has_embed = 1;
    T->A[1][0] = 1.0 / 5.0;
    T->A[2][0] = 3.0 / 40.0; T->A[2][1] = 9.0 / 40.0;
    T->A[3][0] = 44.0 / 45.0; T->A[3][1] = -56.0 / 15.0; T->A[3][2] = 32.0 / 9.0;
    T->A[4][0] = 19372.0 / 6561.0; T->A[4][1] = -25360.0 / 2187.0; T->A[4][2] = 64448.0 / 6561.0; T->A[4][3] = -212.0 / 729.0;
    T->A[5][0] = 9017.0 / 3168.0; T->A[5][1] = -355.0 / 33.0; T->A[5][2] = 46732.0 / 5247.0; T->A[5][3] = 49.0 / 176.0; T->A[5][4] = -5103.0 / 18656.0;
    T->A[6][0] = 35.0 / 384.0; T->A[6][2] = 500.0 / 1113.0; T->A[6][3] = 125.0 / 192.0; T->A[6][4] = -2187.0 / 6784.0; T->A[6][5] = 11.0 / 84.0;
    for (int j = 0; j < 7; j++) T->b[j] = T->A[6][j];
    T->be[0] = 5179.0 / 57600.0; T->be[2] = 7571.0 / 16695.0; T->be[3] = 393.0 / 640.0;
    T->be[4] = -92097.0 / 339200.0; T->be[5] = 187.0 / 2100.0; T->be[6] = 1.0 / 40.0;
  } else {
    return 1;
  }
  tab_finish(T);
  return 0;
}

/* ------------------------------------------------------------------ Vec ops (VecSeq) */
static void VecCopy(long n, const REAL *x, REAL *y) { memcpy(y, x, (size_t)n * sizeof(REAL)); }
static void VecSet(long n, REAL *x, REAL a) { for (long i = 0; i < n; i++) x[i] = a; }
static void VecScale(long n, REAL *x, REAL a) { for (long i = 0; i < n; i++) x[i] *= a; }
static void VecAXPY(long n, REAL *y, REAL a, const REAL *x) { for (long i = 0; i < n; i++) y[i] += a * x[i]; }

/* y += sum_j a_j x_j ; the remainder (nv mod 4) first, then groups of four, which is the
 * blocking the sequential PETSc kernel uses.  Zero coefficients are NOT skipped. */
static void VecMAXPY(long n, REAL *y, int nv, const REAL *a, REAL *const *x) {
  int j = 0, rem = nv & 3;
  if (rem == 3) {
    const REAL a0 = a[0], a1 = a[1], a2 = a[2]; const REAL *x0 = x[0], *x1 = x[1], *x2 = x[2];
    for (long i = 0; i < n; i++) y[i] += a0 * x0[i] + a1 * x1[i] + a2 * x2[i];
  } else if (rem == 2) {
    const REAL a0 = a[0], a1 = a[1]; const REAL *x0 = x[0], *x1 = x[1];
    for (long i = 0; i < n; i++) y[i] += a0 * x0[i] + a1 * x1[i];
  } else if (rem == 1) {
    const REAL a0 = a[0]; const REAL *x0 = x[0];
    for (long i = 0; i < n; i++) y[i] += a0 * x0[i];
  }
  for (j = rem; j < nv; j += 4) {
    const REAL a0 = a[j], a1 = a[j + 1], a2 = a[j + 2], a3 = a[j + 3];
    const REAL *x0 = x[j], *x1 = x[j + 1], *x2 = x[j + 2], *x3 = x[j + 3];
    for (long i = 0; i < n; i++) y[i] += a0 * x0[i] + a1 * x1[i] + a2 * x2[i] + a3 * x3[i];
  }
}

/* TSErrorWeightedNorm, NORM_2 flavour: sqrt(mean(((u-y)/(atol+rtol*max(|u|,|y|)))^2)) */
static double WRMSNorm2(long n, const REAL *u, const REAL *y, double atol, double rtol) {
  double sum = 0;
  for (long i = 0; i < n; i++) {
    double au = fabs((double)u[i]), ay = fabs((double)y[i]);
    double tol = atol + rtol * (au > ay ? au : ay);
    double e = fabs((double)u[i] - (double)y[i]) / tol;
    sum += e * e;
  }
  return sqrt(sum / (double)n);
}

/* ------------------------------------------------------------------ TS object */
typedef void (*RHSFunction)(void *ctx, double t, const REAL *u, REAL *f);   /* pa.py:393-412 */
typedef void (*RHSJacobian)(void *ctx, double t, const REAL *u);            /* pa.py:443-457 */
typedef void (*JacTMult)(void *ctx, const REAL *x, REAL *y);                /* pa.py:52-82   */
typedef void (*JacPTMult)(void *ctx, const REAL *x, REAL *y);               /* pa.py:341-363 */
typedef void (*PostStep)(void *ctx);                                        /* pa.py:518-532 */

typedef struct {
  long stepnum;
  double time, timeprev;
  REAL *U;            /* solution at the END of the step (PETSc stores post-step state) */
  REAL *Y[MAXS];      /* stage values of the step that ENDED here (when !solution_only)  */
} TrajEntry;

typedef struct {
  long n, np;
  Tableau tab;
  /* callbacks */
  void *ctx; RHSFunction rhs; RHSJacobian jac; JacTMult jact; JacPTMult jacpt; PostStep poststep;
  /* state */
  REAL *vec_sol;          /* aliases the caller's U during TSSolve */
  REAL *Y[MAXS], *YdotRHS[MAXS];
  double ptime, ptime_prev, time_step, max_time;
  long steps, max_steps, reject, nfe;
  int steprestart, reason;
  /* adapt */
  int adapt_basic;        /* 0 = none, 1 = basic */
  double atol, rtol, safety, reject_safety, clip_lo, clip_hi, dt_min, dt_max;
  double matchstepfac[2];
  int max_reject;
  /* time span */
  int nspan, spanctr; double *span_times; REAL **span_sols; double dt_span_cached;
  /* trajectory */
  int save_traj, solution_only; long ntraj, captraj; TrajEntry *traj;
  /* adjoint */
  REAL *lambda, *mu;      /* alias caller buffers (setCostGradients, pa.py:766) */
  REAL *VecsDeltaLam[MAXS], *VecsSensiTemp, *VecDeltaMu;
  REAL *vec_backup; int rollback_exact;   /* see ots_set_rollback_exact */
  long adjoint_steps;
  int monitor;
} TS;

static REAL *vnew(long n) { REAL *p = (REAL *)calloc((size_t)(n > 0 ? n : 1), sizeof(REAL)); return p; }

TS *FN(ots_create)(long n, long np) {
  TS *ts = (TS *)calloc(1, sizeof(TS));
  ts->n = n; ts->np = np;
  tab_lookup("3bs", &ts->tab);            /* PETSc's default RK tableau */
  for (int i = 0; i < MAXS; i++) { ts->Y[i] = vnew(n); ts->YdotRHS[i] = vnew(n); ts->VecsDeltaLam[i] = vnew(n); }
  ts->VecsSensiTemp = vnew(n); ts->VecDeltaMu = vnew(np); ts->vec_backup = vnew(n);
  /* TSCreate: no step limit by default (the reference's spiral_unstable.py takes 16000 steps without -ts_max_steps);
     max_time is always set by the caller (pa.py:813-822) */
  ts->max_steps = LONG_MAX; ts->max_time = 5.0; ts->time_step = 0.1;
  ts->adapt_basic = 1; ts->atol = 1e-4; ts->rtol = 1e-4;
  ts->safety = 0.9; ts->reject_safety = 0.5; ts->clip_lo = 0.1; ts->clip_hi = 10.0;
  ts->dt_min = 1e-20; ts->dt_max = 1e50; ts->max_reject = 10;
  ts->matchstepfac[0] = 0.01; ts->matchstepfac[1] = 2.0;
  ts->solution_only = 1;
  return ts;
}

static void traj_clear(TS *ts) {
  for (long k = 0; k < ts->ntraj; k++) {
    free(ts->traj[k].U);
    for (int i = 0; i < MAXS; i++) free(ts->traj[k].Y[i]);
  }
  ts->ntraj = 0;
}

static void span_clear(TS *ts) {
  if (ts->span_sols) { for (int i = 0; i < ts->nspan; i++) free(ts->span_sols[i]); free(ts->span_sols); }
  free(ts->span_times); ts->span_sols = NULL; ts->span_times = NULL; ts->nspan = 0;
}

void FN(ots_destroy)(TS *ts) {
  traj_clear(ts); free(ts->traj); span_clear(ts);
  for (int i = 0; i < MAXS; i++) { free(ts->Y[i]); free(ts->YdotRHS[i]); free(ts->VecsDeltaLam[i]); }
  free(ts->VecsSensiTemp); free(ts->VecDeltaMu); free(ts->vec_backup); free(ts);
}

int FN(ots_set_rk_type)(TS *ts, const char *name) { return tab_lookup(name, &ts->tab); }
void FN(ots_set_callbacks)(TS *ts, void *ctx, RHSFunction rhs, RHSJacobian jac, JacTMult jact, JacPTMult jacpt, PostStep ps) {
  ts->ctx = ctx; ts->rhs = rhs; ts->jac = jac; ts->jact = jact; ts->jacpt = jacpt; ts->poststep = ps;
}
void FN(ots_set_poststep)(TS *ts, PostStep ps) { ts->poststep = ps; }
void FN(ots_set_adapt)(TS *ts, int basic) { ts->adapt_basic = basic; }
void FN(ots_set_tolerances)(TS *ts, double atol, double rtol) { ts->atol = atol; ts->rtol = rtol; }
void FN(ots_set_max_steps)(TS *ts, long m) { ts->max_steps = m; }
void FN(ots_set_max_reject)(TS *ts, int m) { ts->max_reject = m; }
void FN(ots_set_monitor)(TS *ts, int on) { ts->monitor = on; }
/* 0 (default): a rejected step is undone the way TSRollBack_RK does it, by subtracting the
 * increment again -- exact only to round-off RELATIVE TO THE INCREMENT, so a rejected attempt
 * that blew up (|increment| >> |u|) corrupts u_n.  1: restore a saved copy of u_n instead;
 * this is what the product does (it never overwrites u_n), see DESIGN.md section 3.3. */
void FN(ots_set_rollback_exact)(TS *ts, int on) { ts->rollback_exact = on; }
void FN(ots_set_time_step)(TS *ts, double dt) { ts->time_step = dt; }
double FN(ots_get_time_step)(TS *ts) { return ts->time_step; }
void FN(ots_set_time)(TS *ts, double t) { ts->ptime = t; }
double FN(ots_get_time)(TS *ts) { return ts->ptime; }
void FN(ots_set_max_time)(TS *ts, double t) { ts->max_time = t; span_clear(ts); }
void FN(ots_set_step_number)(TS *ts, long k) { ts->steps = k; }
long FN(ots_get_step_number)(TS *ts) { return ts->steps; }
long FN(ots_get_rejections)(TS *ts) { return ts->reject; }
long FN(ots_get_nfe)(TS *ts) { return ts->nfe; }
int FN(ots_get_reason)(TS *ts) { return ts->reason; }
void FN(ots_set_save_trajectory)(TS *ts, int on, int solution_only) { ts->save_traj = on; ts->solution_only = solution_only; }
void FN(ots_set_cost_gradients)(TS *ts, REAL *lambda, REAL *mu) { ts->lambda = lambda; ts->mu = mu; }
long FN(ots_get_traj_len)(TS *ts) { return ts->ntraj; }
/* accepted-step log: (time at end of step k, step size of step k) for k>=1 */
void FN(ots_get_traj_times)(TS *ts, double *t_end, double *h) {
  for (long k = 0; k < ts->ntraj; k++) { t_end[k] = ts->traj[k].time; h[k] = ts->traj[k].time - ts->traj[k].timeprev; }
}

/* TSSetTimeSpan: ptime = t[0], max_time = t[last] (pa.py:822) */
void FN(ots_set_time_span)(TS *ts, int n, const double *times) {
  span_clear(ts);
  ts->nspan = n;
  ts->span_times = (double *)malloc((size_t)n * sizeof(double));
  memcpy(ts->span_times, times, (size_t)n * sizeof(double));
  ts->span_sols = (REAL **)calloc((size_t)n, sizeof(REAL *));
  for (int i = 0; i < n; i++) ts->span_sols[i] = vnew(ts->n);
  ts->ptime = times[0]; ts->max_time = times[n - 1];
}
const REAL *FN(ots_get_span_solution)(TS *ts, int i) { return ts->span_sols[i]; }
int FN(ots_get_span_count)(TS *ts) { return ts->spanctr; }

static int is_close(double a, double b, double rtol, double atol) {
  double d = fabs(a - b), m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
  return d <= atol || d <= rtol * m;
}
#define SPAN_RTOL 1e-6
#define SPAN_ATOL (10 * 2.220446049250313e-16)
static int span_hit(const TS *ts, double t, double h) {
  /* PetscIsCloseAtTol(t, span_times[ctr], reltol*h + abstol, 0) */
  return fabs(t - ts->span_times[ts->spanctr]) <= SPAN_RTOL * fabs(h) + SPAN_ATOL;
}

/* TSTrajectorySet (memory type): called for step 0 and after every accepted step */
static void traj_set(TS *ts) {
  if (!ts->save_traj) return;
  if (ts->ntraj == ts->captraj) {
    ts->captraj = ts->captraj ? 2 * ts->captraj : 64;
    ts->traj = (TrajEntry *)realloc(ts->traj, (size_t)ts->captraj * sizeof(TrajEntry));
  }
  TrajEntry *e = &ts->traj[ts->ntraj++];
  memset(e, 0, sizeof(*e));
  e->stepnum = ts->steps; e->time = ts->ptime; e->timeprev = ts->ptime_prev;
  e->U = vnew(ts->n); VecCopy(ts->n, ts->vec_sol, e->U);
  if (!ts->solution_only && ts->steps > 0)
    for (int i = 0; i < ts->tab.s; i++) { e->Y[i] = vnew(ts->n); VecCopy(ts->n, ts->Y[i], e->Y[i]); }
}

/* TSAdaptChoose: type none|basic, then the MATCHSTEP / time-span adjustment.
 * `t_new` is the time reached by the step being judged. */
static void adapt_choose(TS *ts, double h, int *accept_io, double *next_h) {
  const Tableau *T = &ts->tab;
  int accept = 1;
  double hnew = h;
  if (ts->adapt_basic && T->has_embed) {
    /* TSEvaluateStep(order-1): X = vec_sol + h*sum((be-b)_j K_j), status PENDING */
    REAL *X = ts->VecsSensiTemp;   /* scratch of length n, unused during the forward sweep */
    REAL w[MAXS];
    for (int j = 0; j < T->s; j++) w[j] = (REAL)(h * (T->be[j] - T->b[j]));
    VecCopy(ts->n, ts->vec_sol, X);
    VecMAXPY(ts->n, X, T->s, w, ts->YdotRHS);
    double enorm = WRMSNorm2(ts->n, ts->vec_sol, X, ts->atol, ts->rtol);
    if (!(enorm == enorm) || isinf(enorm)) { ts->reason = -99; *accept_io = 0; *next_h = h; return; }
    double safety = ts->safety;
    if (enorm > 1.0) {
      if (!*accept_io) safety *= ts->reject_safety;   /* the previous attempt failed too */
      accept = (h < (1 + 1.4901161193847656e-08) * ts->dt_min) ? 1 : 0;
    }
    double hfac = enorm > 0 ? safety * pow(enorm, -1.0 / (double)T->order) : INFINITY;
    if (hfac < ts->clip_lo) hfac = ts->clip_lo;
    if (hfac > ts->clip_hi) hfac = ts->clip_hi;
    hnew = h * hfac;
    if (hnew < ts->dt_min) hnew = ts->dt_min;
    if (hnew > ts->dt_max) hnew = ts->dt_max;
  }
  if (accept) {
    /* exact final time MATCHSTEP (pa.py:640) + span points (pa.py:822) */
    double t = ts->ptime + ts->time_step, tend;
    double a = 1.0 + ts->matchstepfac[0], b = ts->matchstepfac[1];
    if (ts->nspan) {
      if (span_hit(ts, t, ts->time_step)) {
        tend = (ts->spanctr + 1 < ts->nspan) ? ts->span_times[ts->spanctr + 1] : ts->max_time;
        /* a span point is reached: the step that was wanted before the approach was adjusted comes back,
         * unless TSAdaptChoose_<type> picked a different one for the next step */
        if (ts->dt_span_cached > 0) { if (hnew == h) hnew = ts->dt_span_cached; ts->dt_span_cached = 0; }
      } else {
        tend = ts->span_times[ts->spanctr];
      }
    } else {
      tend = ts->max_time;
    }
    if (t < tend) {
      double hmax = tend - t;
      double h_unadjusted = hnew;
      if (hnew * b > hmax) hnew = hmax / 2;
      if (h_unadjusted * a > hmax) hnew = hmax;
      if (ts->nspan && hnew != h_unadjusted && !(ts->dt_span_cached > 0)) ts->dt_span_cached = h_unadjusted;   /* cached once per approach */
    }
  }
  *accept_io = accept; *next_h = hnew;
}

/* TSStep_RK */
static void step_rk(TS *ts) {
  const Tableau *T = &ts->tab;
  const int s = T->s;
  int rejections = 0, accept = 1;
  REAL w[MAXS];
  /* FSAL: K_0 of this step is K_{s-1} of the previous accepted one, unless the stepper
   * was restarted (first step of a solve). */
  if (T->fsal && !ts->steprestart) VecCopy(ts->n, ts->YdotRHS[s - 1], ts->YdotRHS[0]);
  if (ts->rollback_exact) VecCopy(ts->n, ts->vec_sol, ts->vec_backup);
  for (;;) {
    const double t = ts->ptime, h = ts->time_step;
    /* K_0 = f(t_n, u_n) does not depend on h: an FSAL tableau keeps it across a rejection */
    const int skip0 = T->fsal && (!ts->steprestart || rejections > 0);
    for (int i = 0; i < s; i++) {
      VecCopy(ts->n, ts->vec_sol, ts->Y[i]);
      for (int j = 0; j < i; j++) w[j] = (REAL)(h * T->A[i][j]);
      VecMAXPY(ts->n, ts->Y[i], i, w, ts->YdotRHS);
      if (i == 0 && skip0) continue;
      ts->rhs(ts->ctx, t + h * T->c[i], ts->Y[i], ts->YdotRHS[i]); ts->nfe++;
    }
    /* TSEvaluateStep(order): vec_sol += h sum b_j K_j */
    for (int j = 0; j < s; j++) w[j] = (REAL)(h * T->b[j]);
    VecMAXPY(ts->n, ts->vec_sol, s, w, ts->YdotRHS);
    double next_h;
    adapt_choose(ts, h, &accept, &next_h);
    if (ts->reason) return;
    if (accept) {
      ts->ptime_prev = ts->ptime;
      ts->ptime += ts->time_step;
      ts->time_step = next_h;
      ts->steprestart = 0;
      return;
    }
    /* TSRollBack_RK: subtract the increment again, then retry with the smaller step */
    if (ts->rollback_exact) {
      VecCopy(ts->n, ts->vec_backup, ts->vec_sol);
    } else {
      for (int j = 0; j < s; j++) w[j] = (REAL)(-h * T->b[j]);
      VecMAXPY(ts->n, ts->vec_sol, s, w, ts->YdotRHS);
    }
    ts->time_step = next_h;
    ts->reject++;
    if (++rejections > ts->max_reject && ts->max_reject >= 0) { ts->reason = -3; return; }
  }
}

/* TSSolve (pa.py:829).  U is advanced in place. */
int FN(ots_solve)(TS *ts, REAL *U) {
  ts->vec_sol = U;
  ts->reason = 0; ts->reject = 0; ts->nfe = 0; ts->steprestart = 1;
  ts->spanctr = 0; ts->dt_span_cached = 0; ts->ptime_prev = ts->ptime;
  traj_clear(ts);
  if (ts->nspan) { VecCopy(ts->n, U, ts->span_sols[0]); ts->spanctr = 1; }
  /* MATCHSTEP at solve start: the initial step is clamped to the first target (no 1%
   * stretch, no halving -- those belong to TSAdaptChoose after a step). */
  {
    double tend = ts->nspan ? ts->span_times[ts->spanctr < ts->nspan ? ts->spanctr : ts->nspan - 1] : ts->max_time;
    double maxdt = tend - ts->ptime, dt = ts->time_step;
    if (maxdt > 0 && (dt >= maxdt || is_close(dt, maxdt, 10 * 2.220446049250313e-16, 0))) {
      if (ts->nspan && dt > maxdt) ts->dt_span_cached = dt;
      ts->time_step = maxdt;
    }
  }
  traj_set(ts);
  if (ts->monitor) printf("%ld TS dt %g time %g\n", ts->steps, ts->time_step, ts->ptime);
  while (!ts->reason) {
    if (ts->steps >= ts->max_steps) { ts->reason = 2; break; }       /* CONVERGED_ITS */
    if (ts->ptime >= ts->max_time) { ts->reason = 1; break; }        /* CONVERGED_TIME */
    step_rk(ts);
    if (ts->reason) break;
    ts->steps++;
    /* snap to the target time when the matched step lands within round-off of it */
    {
      double tgt = ts->nspan && ts->spanctr < ts->nspan ? ts->span_times[ts->spanctr] : ts->max_time;
      if (ts->ptime != tgt && is_close(ts->ptime, tgt, 16 * 2.220446049250313e-16, 0)) ts->ptime = tgt;
    }
    if (ts->nspan && ts->spanctr < ts->nspan &&
        fabs(ts->ptime - ts->span_times[ts->spanctr]) <= SPAN_RTOL * fabs(ts->ptime - ts->ptime_prev) + SPAN_ATOL) {
      VecCopy(ts->n, ts->vec_sol, ts->span_sols[ts->spanctr]);
      ts->spanctr++;
    }
    traj_set(ts);
    if (ts->poststep) ts->poststep(ts->ctx);
    if (ts->monitor) printf("%ld TS dt %g time %g\n", ts->steps, ts->time_step, ts->ptime);
  }
  return ts->reason < 0 ? ts->reason : 0;
}

/* ------------------------------------------------------------------ adjoint */
void FN(ots_adjoint_set_steps)(TS *ts, long k) { ts->adjoint_steps = k; }

/* TSAdjointStep_RK for the step [t_n, t_n + H], stages in ts->Y[] */
static void adjoint_step_rk(TS *ts, double tn, double H) {
  const Tableau *T = &ts->tab;
  const int s = T->s;
  const double h = -H;          /* PETSc runs the adjoint with a negative step */
  const double t = tn + H;      /* ts->ptime at the start of the adjoint step  */
  REAL w[MAXS];
  for (int i = s - 1; i >= 0; i--) {
    if (T->fsal && i == s - 1) { VecSet(ts->n, ts->VecsDeltaLam[i], 0); continue; }
    double stage_time = t + h * (1.0 - T->c[i]);
    ts->jac(ts->ctx, stage_time, ts->Y[i]);
    if (T->b[i] != 0.0) {
      for (int j = i + 1; j < s; j++) w[j - i - 1] = (REAL)(T->A[j][i] / T->b[i]);
      VecCopy(ts->n, ts->lambda, ts->VecsSensiTemp);
      VecMAXPY(ts->n, ts->VecsSensiTemp, s - i - 1, w, &ts->VecsDeltaLam[i + 1]);
      ts->jact(ts->ctx, ts->VecsSensiTemp, ts->VecsDeltaLam[i]);
      VecScale(ts->n, ts->VecsDeltaLam[i], (REAL)(-h * T->b[i]));
      if (ts->mu) {
        ts->jacpt(ts->ctx, ts->VecsSensiTemp, ts->VecDeltaMu);
        VecScale(ts->np, ts->VecDeltaMu, (REAL)(-h * T->b[i]));
        VecAXPY(ts->np, ts->mu, (REAL)1.0, ts->VecDeltaMu);
      }
    } else {
      for (int j = i + 1; j < s; j++) w[j - i - 1] = (REAL)T->A[j][i];
      VecSet(ts->n, ts->VecsSensiTemp, 0);
      VecMAXPY(ts->n, ts->VecsSensiTemp, s - i - 1, w, &ts->VecsDeltaLam[i + 1]);
      ts->jact(ts->ctx, ts->VecsSensiTemp, ts->VecsDeltaLam[i]);
      VecScale(ts->n, ts->VecsDeltaLam[i], (REAL)(-h));
      if (ts->mu) {
        /* exact discrete adjoint: the parameter product of THIS stage */
        ts->jacpt(ts->ctx, ts->VecsSensiTemp, ts->VecDeltaMu);
        VecScale(ts->np, ts->VecDeltaMu, (REAL)(-h));
        VecAXPY(ts->np, ts->mu, (REAL)1.0, ts->VecDeltaMu);
      }
    }
  }
  for (int j = 0; j < s; j++) w[j] = 1.0;
  VecMAXPY(ts->n, ts->lambda, s, w, ts->VecsDeltaLam);
}

/* TSAdjointSolve (pa.py:878): reverse `adjoint_steps` steps starting from ts->steps */
int FN(ots_adjoint_solve)(TS *ts) {
  const Tableau *T = &ts->tab;
  for (long k = 0; k < ts->adjoint_steps; k++) {
    long stepnum = ts->steps;               /* the step that ENDED at stepnum */
    if (stepnum < 1 || stepnum >= ts->ntraj) return -5;
    TrajEntry *e1 = &ts->traj[stepnum], *e0 = &ts->traj[stepnum - 1];
    double tn = e0->time, H = e1->time - e0->time;
    if (ts->solution_only) {
      /* recompute the forward step from u_n to regenerate the stage values */
      REAL w[MAXS];
      for (int i = 0; i < T->s; i++) {
        VecCopy(ts->n, e0->U, ts->Y[i]);
        for (int j = 0; j < i; j++) w[j] = (REAL)(H * T->A[i][j]);
        VecMAXPY(ts->n, ts->Y[i], i, w, ts->YdotRHS);
        ts->rhs(ts->ctx, tn + H * T->c[i], ts->Y[i], ts->YdotRHS[i]); ts->nfe++;
      }
    } else {
      for (int i = 0; i < T->s; i++) VecCopy(ts->n, e1->Y[i], ts->Y[i]);
    }
    adjoint_step_rk(ts, tn, H);
    ts->steps--; ts->ptime = tn;
  }
  return 0;
}

/* ------------------------------------------------------------------ introspection for tests */
int FN(ots_tableau_info)(const char *name, int *s, int *order, int *fsal, int *has_embed,
                         double *A /* MAXS*MAXS */, double *b, double *be, double *c) {
  Tableau T;
  if (tab_lookup(name, &T)) return 1;
  *s = T.s; *order = T.order; *fsal = T.fsal; *has_embed = T.has_embed;
  for (int i = 0; i < MAXS; i++) { b[i] = T.b[i]; be[i] = T.be[i]; c[i] = T.c[i]; for (int j = 0; j < MAXS; j++) A[i * MAXS + j] = T.A[i][j]; }
  return 0;
}
double FN(ots_wrms)(long n, const REAL *u, const REAL *y, double atol, double rtol) { return WRMSNorm2(n, u, y, atol, rtol); }
void FN(ots_vec_maxpy)(long n, REAL *y, int nv, const REAL *a, REAL *const *x) { VecMAXPY(n, y, nv, a, x); }
