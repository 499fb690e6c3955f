/*
 * pnode_amd -- C ABI of the MI355X-native explicit-RK time stepper + discrete-adjoint engine.
 *
 * This is the drop-in boundary for the hot path of caidao22/pnode
 * (reference: pnode/petsc_adjoint.py, "pa.py" below).  The reference has no FFI of its own on
 * this path: it binds PETSc's TS / TSAdjoint / TSTrajectory / Vec through petsc4py.  Every
 * entry point below therefore names the petsc4py call (pa.py file:line) or the PETSc routine
 * behind it that it replaces.  Plain pointers, sizes and scalars only -- no torch types.
 *
 * Groups (sections 1-5 below; 5 = the disk tier of the trajectory, added in round 2):
 *   1. device entry points (pn_rk_*, pn_adj_*, pn_param_accum, ...): enqueue ONE hand-written
 *      gfx950 kernel on the caller's HIP stream over raw device pointers.  They replace the
 *      PETSc Vec-op sequences (VecCopy + VecMAXPY + VecScale + VecAXPY + norms) that
 *      TSStep_RK / TSAdjointStep_RK / TSAdaptChoose issue per stage.
 *   2. host entry points (pn_ts_*, pn_traj_*, pn_tableau_*): the time-stepper state machine --
 *      tableau, step-size controller, exact-final-time / time-span matching, checkpoint
 *      scheduler.  No GPU needed; unit-testable on a CPU-only box.
 *
 * The callback into the user's dynamics f(t,u) stays on the host side above this ABI (it is a
 * Python nn.Module, pa.py:393-412), exactly where petsc4py's callback shells sit.
 *
 * All functions return 0 on success, non-zero on failure; pn_last_error() describes the last
 * failure on the calling thread.
 */
#ifndef PNODE_AMD_H
#define PNODE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_MAX_STAGES 7      /* 5dp has 7 stages */
#define PN_MAX_TERMS 8       /* most vectors one kernel combines (lambda + 6 dlambda + forcing) */
#define PN_WGRAD_MAX_PAIRS 8  /* most (cotangent, input) pairs one pn_linear_wgrad_group launch takes */
#define PN_WGRAD_EXACT_FP32 1 /* pn_linear_wgrad_group flag: fp32 states on v_mfma_f32_32x32x2_f32 instead of the split-bf16 form */
#define PN_WGRAD_TILE_64 2    /* pn_linear_wgrad_group flag: always the 64 x 64 workgroup tile (the same bits as the 128 x 128 one a
                                 launch over whole rounds of the chip takes by itself: for comparisons) */
#define PN_ABI_VERSION 4      /* 2 (round 3): pn_rk_combine_wrms writes per-workgroup partials into a pinned BLOCK (pn_wrms_partials)
                                 and pn_stream_wait_wrms finishes the norm; work areas of the reductions are zero-filled once;
                                 pn_krylov_* added.  3 (round 4): the step loops pn_rk_attempt / pn_rk_adjoint_step (section 3a).
                                 4 (round 6): pn_kernel_id grew (PN_K_LINEAR_WGRAD, added in round 5 without a version change) and
                                 pn_prof_collect now takes the caller's array length; pn_colsum_*, pn_linear_wgrad* entered the
                                 ABI; pn_linear_wgrad_work_bytes takes the dtype; pn_linear_wgrad_group added; fp64 form of the
                                 fused kernel.  A client built against version 3 must not call this library: pn_abi_version()
                                 is how it finds out. */

typedef enum { PN_F32 = 0, PN_F64 = 1 } pn_dtype;

const char *pn_last_error(void);
int pn_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * 1. Tableaus.  Replaces ts.setRKType(...) (pa.py:641-650) / -ts_rk_type.
 *    Names: PETSc's "1fe" "2a" "2b" "3" "3bs" "4" "5f" "5dp", plus "midpoint" (extension).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int s;            /* stages */
  int order;        /* order of the propagated solution (exponent of the controller) */
  int fsal;         /* last stage == next step's first stage */
  int has_embed;    /* has an embedded lower-order solution (adaptive-capable) */
  double A[PN_MAX_STAGES][PN_MAX_STAGES];
  double b[PN_MAX_STAGES], bembed[PN_MAX_STAGES], c[PN_MAX_STAGES];
} pn_tableau;

int pn_tableau_get(const char *rk_type, pn_tableau *out);
/* pa.py:641-650: euler->1fe rk2->2b bosh3|fixed_bosh3->3bs rk4->4 dopri5|fixed_dopri5->5dp;
 * any other string -> PETSc's default "3bs".  Returns the rk type name (static storage). */
const char *pn_method_to_rk_type(const char *method);

/* ------------------------------------------------------------------------------------------
 * 2. Device entry points.  `stream` is a hipStream_t.  `dtype` selects f32/f64 storage;
 *    coefficients are passed in double and rounded once to the storage type, as PETSc forms
 *    w[j] = h*A[i][j] in PetscReal before VecMAXPY.  All vectors have `n` elements.
 *    Zero coefficients must be dropped by the caller (nk counts the non-zeros).
 * ---------------------------------------------------------------------------------------- */

/* Y = u + sum_{j<nk} coef[j]*K[j]      (coef[j] = h*a_ij or h*b_j)
 * Replaces TSStep_RK's per-stage VecCopy(vec_sol,Y[i]) + VecMAXPY(Y[i],i,w,YdotRHS) and the
 * final VecMAXPY of TSEvaluateStep_RK, driven by ts.solve (pa.py:829). */
int pn_rk_stage(void *stream, int dtype, int64_t n, void *y, const void *u,
                int nk, const void *const *K, const double *coef);

/* Embedded error estimate, fused with the solution update:
 *   unew = u + sum coef_b[j]*K[j]      (written to `unew` unless unew == NULL; with
 *                                       unew == NULL `u` already IS the new solution -- FSAL)
 *   err  = sum coef_e[j]*K[j]          (coef_e[j] = h*(bembed_j - b_j))
 *   partial sums of (err / (atol + rtol*max(|unew|,|unew+err|)))^2 -> *result_dev =
 *   sqrt(sum/n)   [TSErrorWeightedNorm, NORM_2]   (also NaN/Inf if any element is).
 * Replaces TSEvaluateStep_RK(order-1) + TSErrorWeightedNorm inside TSAdaptChoose_Basic
 * (selected by ts.setFromOptions, pa.py:775).  ONE launch.  The norm is finished on the host, which has to wait for the
 * stream before it can judge the step anyway: every workgroup stores its partial sum into `result_dev` -- the device
 * pointer of a pn_pinned_block() of pn_wrms_partials(n) doubles -- and pn_stream_wait_wrms() adds them in index order
 * (bit-reproducible) and returns sqrt(sum/n).  `work`: pn_wrms_work_bytes(n) bytes of device memory, ZERO-FILLED once
 * before first use (arrival counters of the in-launch finish, PN_TUNE "wfin=1"; untouched otherwise); one work area and
 * one result block per stream. */
int pn_rk_combine_wrms(void *stream, int dtype, int64_t n, void *unew, const void *u,
                       int nk, const void *const *K, const double *coef_b, const double *coef_e,
                       double atol, double rtol, void *work, double *result_dev);
int64_t pn_wrms_work_bytes(int64_t n);
int64_t pn_wrms_partials(int64_t n);
/* Blocks until everything enqueued on `stream` is done, then finishes the norm from the host side of the pinned block
 * pn_rk_combine_wrms wrote.  The one host<->device synchronisation of an adaptive step. */
int pn_stream_wait_wrms(void *stream, const double *result_host, int64_t n, double *value);
/* A stream of the caller's own on the current device: priority > 0 the device's LOWEST priority (background work: the
 * weight-sensitivity products that run beside the reverse sweep's critical chain), < 0 the highest, 0 the default.  Not a
 * replacement of anything in the reference (PETSc's Vec operations run on one stream). */
int pn_stream_create(int priority, void **stream);
int pn_stream_destroy(void *stream);
/* Blocks until everything enqueued on `stream` is done and returns *host_ptr (a pinned, device-visible double). */
int pn_pinned_scalar(double **host_ptr, double **dev_ptr);
int pn_pinned_free(double *host_ptr);
/* the same for a block of `nbytes` (e.g. all Hessenberg entries of one GMRES iteration) */
int pn_pinned_block(int64_t nbytes, double **host_ptr, double **dev_ptr);
int pn_stream_wait_scalar(void *stream, const double *host_ptr, double *value);

/* Adjoint stage cotangent:  w = c_lam*lambda + sum_{j<nk} coef[j]*dlam[j]
 *   with c_lam = H*b_i (lambda == NULL when b_i == 0) and coef[j] = H*a_ji.
 * Replaces TSAdjointStep_RK's VecCopy/VecSet + VecMAXPY into VecsSensiTemp and the VecScale of
 * the transposed-Jacobian product (the scale is folded into the cotangent), pa.py:875-878. */
int pn_adj_theta(void *stream, int dtype, int64_t n, void *w, const void *lambda, double c_lam,
                 int nk, const void *const *dlam, const double *coef);

/* lambda_out = lambda + sum_{j<nk} coef[j]*dlam[j] (+ forcing)     (coef == NULL: all ones)
 * and, optionally fused, a scaled copy  w_next = c_next*lambda_out  (skipped when w_next == NULL).
 * coef[j] carries a scalar that was not applied to a stage cotangent: a stage whose cotangent
 * is a pure multiple of lambda (H*b_i*lambda) is differentiated with lambda itself as the
 * cotangent and the factor H*b_i is folded into the consumers of its result.
 * Replaces the closing VecMAXPY of TSAdjointStep_RK and adj_u_tensor.add_(grad_output[i-1])
 * (pa.py:938). */
int pn_adj_accum(void *stream, int dtype, int64_t n, void *lambda_out, const void *lambda,
                 int nk, const void *const *dlam, const double *coef, const void *forcing,
                 void *w_next, double c_next);

/* mu[off_k : off_k+len_k] += alpha*g_k for every parameter tensor k (g_k == NULL: skipped).
 * Replaces RHSJacPShell.multTranspose's flatten+copy (pa.py:341-363, misc.py:9-14) and
 * TSAdjointStep_RK's VecScale + VecAXPY on the parameter sensitivities. */
int pn_param_accum(void *stream, int dtype, void *mu, double alpha, int nseg, const void *const *g,
                   const int64_t *offset, const int64_t *len);

/* The parameter sensitivities of all stages of one time step in ONE launch:
 *   mu[off_k + e] <- fma(alpha_{S-1}, g_{S-1,k}[e], ... fma(alpha_0, g_{0,k}[e], mu[off_k + e]))
 * for nsrc = S <= 32 gradient sets (the stages of one or of SEVERAL time steps), g[j*nseg + k] = set j's
 * gradient of parameter tensor k (NULL: skipped).  Rounding is that of S successive pn_param_accum calls
 * in the order j = 0..S-1 (results are bit-identical to them); mu is read and written once instead of S
 * times.  Replaces the per-stage VecAXPY on the parameter sensitivities inside TSAdjointStep_RK. */
int pn_param_accum_multi(void *stream, int dtype, void *mu, int nsrc, const double *alpha, int nseg,
                         const void *const *g, const int64_t *offset, const int64_t *len);

/* For up to 32 sources j:  mu[j][c] += alpha[j] * sum_r g[j][r*cols[j] + c]  for c < cols[j] -- the parameter sensitivities of
 * BIASES: the column sum of the cotangent at the output of a Linear layer, added to the bias's slice of mu; the sources are the
 * layers and stages of one or of several time steps (several sources may name the same mu: they are added in source order).
 * Every g is read once, in one pass (per-lane sums in double; partials added in a fixed association: bit-reproducible, no
 * atomics, independent of how sources are grouped into calls).  Replaces, for the bias parameters of func's nn.Linear layers,
 * the `sum` inside autograd's backward plus RHSJacPShell.multTranspose's flatten/copy (pa.py:341-363, misc.py:9-14) plus the
 * VecAXPY on mu inside TSAdjointStep_RK.  `work`: pn_colsum_work_bytes(nsrc, rows, cols) bytes, no initialisation needed.
 * pn_colsum_accum is the one-source form. */
int pn_colsum_accum_multi(void *stream, int dtype, int nsrc, const int64_t *rows, const int64_t *cols, const void *const *g,
                          void *const *mu, const double *alpha, void *work);
int64_t pn_colsum_work_bytes(int nsrc, const int64_t *rows, const int64_t *cols);
int pn_colsum_accum(void *stream, int dtype, int64_t rows, int64_t cols, const void *g, void *mu, double alpha, void *work);

/* The parameter sensitivities of a Linear layer `out = x W^T + b`, fused (csrc/pn_linear.hip; MFMA -- v_mfma_f32_32x32x2_f32, exact
 * fp32, or v_mfma_f64_16x16x4_f64 --, K split over the 8 XCDs, 64 x 64 output tiles):
 *   pw[s][m][n]    += sum_{k in K-range s} (alpha g[k][m]) x[k][n]                                              s = 0..7
 *   pb[s][j][m]    += sum_{k in the slabs of K-range s that tile column j adds up} alpha g[k][m]                j = 0..in_f/64-1
 * for g = the cotangent at the layer's output (rows x out_f) and x = the layer's input (rows x in_f), both row-major and 16-byte
 * aligned.  pw (elements of the state's dtype) / pb (always doubles; the workgroups of a tile row share the column sums of g between
 * them, hence the index j) are partial buffers of pn_linear_wgrad_work_bytes() bytes that the CALLER zero-fills once; they carry the
 * sum over the stages and time steps of a reverse sweep (each launch adds its tile to them: no separate accumulation pass), and
 * pn_linear_wgrad_finish adds them to the parameter's slices of mu (mu_w[m][n] += sum_s pw[s][m][n], s in order; mu_b[m] += sum_{s,j}
 * pb[s][j][m], in index order) and zero-fills them again.  pb / mu_b may be NULL (layer without bias).  Bit-reproducible, and the
 * same bits whether pairs go through pn_linear_wgrad one by one or through pn_linear_wgrad_group.  Replaces, for func's nn.Linear
 * layers, autograd's weight- and bias-gradient kernels, RHSJacPShell.multTranspose's flatten/copy (pa.py:341-363, misc.py:9-14)
 * and the VecAXPY on mu inside TSAdjointStep_RK.  pn_linear_wgrad_supported: fp32 or fp64, rows >= 256 (any number: eight K ranges of whole 32-row slabs, missing rows read as zeros),
 * out_f % 64 == 0, in_f % 64 == 0, out_f * in_f <= 2^22 (other shapes take the general path: a library GEMM accumulating into mu +
 * pn_colsum_accum_multi).
 * pn_linear_wgrad_group: the pairs of SEVERAL layers (1 <= npairs <= PN_WGRAD_MAX_PAIRS; shapes may differ, rows is common) in
 * ONE launch -- all Linear layers of one stage VJP: one launch boundary per stage, and the workgroups of the next pair start while
 * the previous pair drains.  No two pairs of a group may share pw or pb.
 * Arithmetic for fp32 states (flags = 0, also what pn_linear_wgrad uses): every operand is split EXACTLY into three bf16 terms
 * (hi + mid + lo = the fp32 value) and the six largest of the nine cross products -- each exact in fp32; what is dropped is below
 * 2^-23 |g x| -- are accumulated in fp32 on v_mfma_f32_32x32x16_bf16: the bf16 matrix pipe is 16 times as fast per product as the
 * fp32 one.  Error against float64 BELOW that of an fp32 fmaf chain (5.7e-8 against 1.0e-7 of sum |g x|, tools/mb_wgrad_bf16x3.hip);
 * an infinite operand gives NaN.  flags = PN_WGRAD_EXACT_FP32: v_mfma_f32_32x32x2_f32 (a k-ordered fp32 fmaf chain per K range)
 * instead.  Either way bit-reproducible.  fp64 states: v_mfma_f64_16x16x4_f64.
 * Tiling of the split-bf16 form: 128 x 128 workgroup tiles (1024 threads, one workgroup per CU) when every layer's out_f and in_f
 * are multiples of 128 and the launch's tiles fill whole rounds of the chip, else 64 x 64 ones (512 threads); flags =
 * PN_WGRAD_TILE_64 keeps the latter.  The two give the same bits in pw and pb. */
typedef struct {
  const void *g, *x;         /* cotangent at the layer's output (rows x out_f), the layer's input (rows x in_f) */
  void *pw, *pb;             /* the layer's partial buffers (pb may be NULL) */
  double alpha;
  int64_t out_f, in_f;
} pn_wgrad_pair;
int pn_linear_wgrad_supported(int dtype, int64_t rows, int64_t out_f, int64_t in_f);
int64_t pn_linear_wgrad_work_bytes(int dtype, int64_t out_f, int64_t in_f, int64_t *bias_bytes);
int pn_linear_wgrad(void *stream, int dtype, int64_t rows, int64_t out_f, int64_t in_f, const void *g, const void *x, double alpha,
                    void *pw, void *pb);
int pn_linear_wgrad_group(void *stream, int dtype, int64_t rows, int npairs, const pn_wgrad_pair *pairs, int flags);
int pn_linear_wgrad_finish(void *stream, int dtype, int64_t out_f, int64_t in_f, void *pw, void *pb, void *mu_w, void *mu_b);

/* result_dev[j] = <x, y_j> for j < nk <= PN_MAX_TERMS, accumulated in double, reduced in a fixed
 * order (bit-reproducible).  ||x||^2 is the case y_0 == x.  Replaces VecMDot / VecNorm inside
 * the KSP(GMRES) and SNES that PETSc's implicit steppers run (TS type BE/CN, pa.py:651-654;
 * the reference reaches them through ts.getSNES().getKSP(), pa.py:701-702).
 * `work` needs pn_dots_work_bytes(n) bytes, ZERO-FILLED once before its first use (arrival counter, as
 * pn_rk_combine_wrms; one launch); result_dev holds nk doubles (a pn_pinned_scalar() block has room for 8). */
int pn_dots(void *stream, int dtype, int64_t n, const void *x, int nk, const void *const *y,
            void *work, double *result_dev);
int64_t pn_dots_work_bytes(int64_t n);
/* out = sum_{j<nin} c[j]*x[j], 1 <= nin <= PN_MAX_TERMS (out may alias any x[j]).  The general
 * form of the streaming kernel; the Krylov / Newton updates of the implicit steppers use it where
 * PETSc calls VecAXPY / VecAXPBY / VecMAXPY / VecWAXPY / VecScale. */
int pn_lincomb(void *stream, int dtype, int64_t n, void *out, int nin, const void *const *x, const double *c);
/* Waits for the stream, then copies `count` doubles from a pn_pinned_scalar() block. */
int pn_stream_wait_scalars(void *stream, const double *host_ptr, int count, double *values);

/* y = x (device copy on the stream; u0 -> trajectory slot, span solutions -> output). */
int pn_copy(void *stream, int dtype, int64_t n, void *y, const void *x);
/* y = 0 */
int pn_zero(void *stream, int dtype, int64_t n, void *y);

/* Per-kernel timing for bench.py's roofline: when enabled, every device entry point above is
 * launched with a start/stop HIP event pair bound to the dispatch itself; pn_prof_collect()
 * synchronises and returns, per entry point, the number of launches, the summed kernel
 * duration in microseconds and the summed ALGORITHMIC bytes (each distinct input read once +
 * each output written once).  PN_K_LINEAR_WGRAD (pn_linear_wgrad / _group, an MFMA-bound product) is the
 * exception: what it reports as bytes are its FLOPs, 2 * rows * out * in per pair.
 * PN_K_COUNT is versioned with PN_ABI_VERSION (it grows when a kernel is added); pn_prof_collect takes the length `count` of the
 * caller's three arrays and fills min(count, PN_K_COUNT) entries, so a client built against a shorter enum is not overrun. */
typedef enum { PN_K_STAGE = 0, PN_K_COMBINE_WRMS, PN_K_ADJ_THETA, PN_K_ADJ_ACCUM,
               PN_K_PARAM_ACCUM, PN_K_COPY, PN_K_DOTS, PN_K_LINCOMB, PN_K_LINEAR_WGRAD, PN_K_COUNT } pn_kernel_id;
int pn_prof_enable(int on);
int pn_prof_is_enabled(void);
/* Diagnostic: override the launch geometry / cache policy of the streaming kernels at run time
 * ("vpt=2,ld=0,st=1", same grammar as the PN_TUNE environment variable; NULL = defaults).
 * Used by tools/ab_policy.py for interleaved A/B timing; results never depend on it. */
int pn_tune_set(const char *spec);
int pn_prof_collect(int count, int64_t *launches, double *usec, double *bytes);
const char *pn_kernel_name(int kernel_id);

/* ------------------------------------------------------------------------------------------
 * 3. Host time-stepper.  Replaces the PETSc.TS object held by ODEPetsc (pa.py:370) for the
 *    explicit-RK branch: TSSetType(RK)/TSRKSetType (638-650), TSSetExactFinalTime(MATCHSTEP)
 *    (640), TSSetTimeStep (770,813-817), TSSetTime/TSSetMaxTime (819-820), TSSetTimeSpan
 *    (822), TSSetFromOptions (775), the step loop of TSSolve (829) minus the stage
 *    arithmetic, TSAdaptChoose none|basic.
 * ---------------------------------------------------------------------------------------- */
typedef struct pn_ts pn_ts;

pn_ts *pn_ts_create(void);
void pn_ts_destroy(pn_ts *ts);
int pn_ts_set_rk_type(pn_ts *ts, const char *rk_type);
int pn_ts_get_tableau(const pn_ts *ts, pn_tableau *out);
/* Options database subset, PETSc spellings without the leading dash: ts_adapt_type
 * (none|basic), ts_rk_type, ts_rtol, ts_atol, ts_max_steps, ts_max_reject,
 * ts_adapt_safety, ts_adapt_reject_safety, ts_adapt_clip (lo,hi), ts_adapt_dt_min,
 * ts_adapt_dt_max.  Unknown keys return non-zero and are left to the caller. */
int pn_ts_set_option(pn_ts *ts, const char *key, const char *value);
int pn_ts_is_adaptive(const pn_ts *ts);
/* For steppers whose stage arithmetic lives above the ABI (TS types ARKIMEX / THETA, pa.py:651-656): tell the
 * controller the scheme's order (the exponent TSAdaptChoose_Basic uses) and whether it has an embedded
 * solution to estimate the error with.  order == 0 hands the controller back to the RK tableau. */
int pn_ts_set_scheme(pn_ts *ts, int order, int has_embed);
int pn_ts_get_tolerances(const pn_ts *ts, double *atol, double *rtol);

/* Begin a solve.  nspan == 1: integrate [t0, span[0]] (pa.py:818-820, t0 = 0);
 * nspan > 1: time span, t0 = span[0] (pa.py:822).  dt0 = step_size (pa.py:812-817). */
int pn_ts_begin(pn_ts *ts, double t0, double dt0, int nspan, const double *span);
/* Current attempt: time at step start and the step size to use. */
int pn_ts_attempt(const pn_ts *ts, double *t, double *h);
/* Judge the attempt.  enorm < 0: no error estimate (fixed step).  Outputs:
 *   accept      1 = step accepted (time advanced), 0 = rejected (retry with the new h)
 *   hit_span    index of the span point reached by this step, or -1
 *   done        1 = final time reached (or max_steps hit -> status != 0)
 * Restates TSAdaptChoose (+Basic), the MATCHSTEP rule and the span bookkeeping. */
int pn_ts_judge(pn_ts *ts, double enorm, int *accept, int *hit_span, int *done);
/* pa.py:523-525: a per-step list overrides the step size of step `stepno`. */
int pn_ts_override_next_dt(pn_ts *ts, double dt);
int64_t pn_ts_steps(const pn_ts *ts);
int64_t pn_ts_rejections(const pn_ts *ts);
double pn_ts_time(const pn_ts *ts);
/* accepted-step log of the last solve: start time and size of step k, 0 <= k < steps */
int pn_ts_step_log(const pn_ts *ts, int64_t k, double *t_start, double *h);

/* ------------------------------------------------------------------------------------------
 * 3a. The step loops (round 4).  In the reference `ts.solve` and `ts.adjointSolve` are PETSc's C loops
 *     (pa.py:829, 878): TSStep_RK forms every stage vector and calls back into Python only for func
 *     (evalRHSFunction, pa.py:393-412); TSAdjointStep_RK forms every stage cotangent and calls back only for
 *     the transposed-Jacobian products (RHSJacShell.multTranspose / RHSJacPShell.multTranspose, pa.py:52-82,
 *     341-363).  These two entry points are those loops for ONE step attempt / ONE reversed step: the tableau
 *     walk, the coefficients h*a_ij (formed in double, as TSStep_RK forms w[j] = h*A[i][j]) and the launches
 *     are here; the callbacks are the two the reference has.
 *
 *     pn_vec_ops: the vector operations the loops launch.  NULL (or NULL members) = this library's HIP entry
 *     points of section 2; a table of other functions with the same signatures lets the loops run on a
 *     different backend (the CPU-only test container's stand-in).
 * ---------------------------------------------------------------------------------------- */
typedef int (*pn_rk_stage_fn)(void *stream, int dtype, int64_t n, void *y, const void *u, int nk, const void *const *K,
                              const double *coef);
typedef int (*pn_rk_combine_wrms_fn)(void *stream, int dtype, int64_t n, void *unew, const void *u, int nk,
                                     const void *const *K, const double *coef_b, const double *coef_e, double atol,
                                     double rtol, void *work, double *result_dev);
typedef int (*pn_adj_theta_fn)(void *stream, int dtype, int64_t n, void *w, const void *lambda, double c_lam, int nk,
                               const void *const *dlam, const double *coef);
typedef int (*pn_adj_accum_fn)(void *stream, int dtype, int64_t n, void *lambda_out, const void *lambda, int nk,
                               const void *const *dlam, const double *coef, const void *forcing, void *w_next,
                               double c_next);
typedef struct pn_vec_ops {
  pn_rk_stage_fn rk_stage;
  pn_rk_combine_wrms_fn rk_combine_wrms;
  pn_adj_theta_fn adj_theta;
  pn_adj_accum_fn adj_accum;
} pn_vec_ops;

/* evalRHSFunction (pa.py:393-412): evaluate K_stage = f(t, Y_stage) -- the caller knows which buffer holds
 * Y_stage (u itself for stage 0, ystage[stage] otherwise) -- and return the device address of K_stage, which the
 * caller keeps alive until the step is over; 0 = failure. */
typedef int64_t (*pn_stage_cb)(void *user, int stage, double t);
/* One attempt of TSStep_RK with the tableau and tolerances of `ts`, of size h from the state `u` at time t:
 *   for i < s:  Y_i = u + h sum_j a_ij K_j  into ystage[i] (i >= 1; a first-same-as-last tableau writes its last
 *               stage value straight into `unew`);  K_i = cb(user, i, t + c_i h)
 *   k0 != NULL: K_0 is handed in (first-same-as-last, or the retry after a rejection); have_t_first: K_0 is
 *               evaluated at t_first instead of t (re-advancing from a checkpoint, see _first_stage_time);
 *   want_err:   unew and the error norm by pn_rk_combine_wrms (work / result_dev as there), else unew by pn_rk_stage
 *               (nothing to do for a first-same-as-last tableau).
 * kout[0..s) receives the addresses of the stage derivatives. */
int pn_rk_attempt(void *stream, int dtype, int64_t n, const pn_ts *ts, const pn_vec_ops *vec_ops, double t, double h,
                  const void *u, void *unew, void *const *ystage, const void *k0, int have_t_first, double t_first,
                  pn_stage_cb cb, void *user, int want_err, void *work, double *result_dev, const void **kout);

/* RHSJacShell.multTranspose + RHSJacPShell.multTranspose (pa.py:52-82, 341-363) for stage `stage` at time t: the
 * cotangent is lambda itself (cot_in_w == 0), in `wbuf` (1) or in `wbuf2` (2); the callback adds scale * (df/dp)^T cot to mu and
 * returns the device address of J^T cot (kept alive by the caller until the step is over), 0 when f does not depend on its
 * state argument, -1 on failure. */
typedef int64_t (*pn_vjp_cb)(void *user, int stage, double t, int cot_in_w, double scale);
/* TSAdjointStep_RK for the step [t, t + H] (recurrence in SURVEY 8a-6; a stage whose cotangent is a pure multiple of
 * lambda is differentiated with lambda itself, the factor H*b_i folded into the consumers of its result):
 * lambda <- lambda + sum_i dlambda_i (+ forcing, pa.py:938).
 * wbuf2 (may be NULL): a second cotangent buffer; the stages that need one then take wbuf and wbuf2 in turn, so that work the
 * callback left running on another stream (the weight-sensitivity products of the stage before) may still read the previous
 * stage's cotangent while this stage's is written. */
int pn_rk_adjoint_step(void *stream, int dtype, int64_t n, const pn_ts *ts, const pn_vec_ops *vec_ops, double t, double H,
                       void *lambda, void *wbuf, void *wbuf2, pn_vjp_cb cb, void *user, const void *forcing);

/* ------------------------------------------------------------------------------------------
 * 3b. GMRES core for the implicit (theta-method) stage solves: the small dense part of
 *     KSPGMRES -- Hessenberg columns, Givens rotations, residual estimate, back substitution.
 *     The Krylov vectors live in HBM and are orthogonalised with pn_dots + pn_rk_stage-style
 *     linear combinations by the caller; the operator (shift*M - J, or its transpose) is the
 *     matrix-free shell above the ABI (pa.py:98-197 IJacShell.mult / multTranspose).
 * ---------------------------------------------------------------------------------------- */
typedef struct pn_gmres pn_gmres;
pn_gmres *pn_gmres_create(int restart);
void pn_gmres_destroy(pn_gmres *g);
/* start a cycle with initial residual norm beta */
int pn_gmres_begin(pn_gmres *g, double beta);
/* column k of the Hessenberg matrix: h[0..k] = <w, V_j>, h[k+1] = ||w - sum h_j V_j||;
 * applies the rotations and returns the residual-norm estimate after this iteration */
int pn_gmres_column(pn_gmres *g, int k, const double *h, double *resnorm);
/* coefficients y[0..k] of the update x += sum y_j V_j after k+1 iterations */
int pn_gmres_solve(pn_gmres *g, int k, double *y);

/* ------------------------------------------------------------------------------------------
 * 3c. Device-resident GMRES (round 3).  The same KSPGMRES as 3b -- classical Gram-Schmidt, Givens rotations,
 *     residual test after every column, restart -- with everything it decides kept in a state block in HBM, so that
 *     the host synchronises once per chunk of iterations instead of once per iteration (the reference's default
 *     linear_solver="petsc" for TS types BE / CN / ARKIMEX: pa.py:547, 581, 651-656, 701-702; operator = the
 *     matrix-free shell IJacShell.mult / multTranspose above this ABI, pa.py:98-197).
 *       state       pn_krylov_state_doubles(n, restart) doubles of device memory, ZERO-FILLED once before first use
 *       status_dev  device pointer of a pn_pinned_block() of >= 8 doubles; every decision refreshes it:
 *                   [0] stop (0 running, 1 converged, 2 happy breakdown, 3 iteration limit, 4 NaN, 5 singular
 *                   Hessenberg)  [1] iterations done in this cycle  [2] iterations of the solve  [3] residual-norm
 *                   estimate  [4] beta  [5] ||rhs||  [6] tol = max(rtol ||rhs||, atol)  [7] second Gram-Schmidt passes of the solve
 *       V           restart + 1 Krylov vectors of n elements, vector j at V + j*ldv;  vin: the operator's input
 *                   buffer (every new basis vector is also written there);  w: the operator's output A vin
 *     Once `stop` is set, every later pn_krylov_step of the cycle is a no-op on the device: the host may enqueue
 *     iterations ahead and read the status whenever it likes.
 *     `part` = 0: everything in one call (one rank).  Several ranks sum the Gram-Schmidt products over the ranks
 *     first: part 1 leaves them at state + pn_krylov_products_offset(restart) (k + 2 doubles, 1 at a cycle start),
 *     the caller all-reduces that block in stream order, then part 2 (and, for a step, part 3 after a second
 *     all-reduce of the same block) continues.
 * ---------------------------------------------------------------------------------------- */
int64_t pn_krylov_state_doubles(int64_t n, int restart);
int pn_krylov_products_offset(int restart);
/* Start a cycle from the residual r (first_cycle: r = rhs, x = 0): beta = ||r||, tolerances, V_0 = r/beta. */
int pn_krylov_begin(void *stream, int dtype, int64_t n, int restart, double *state, double *status_dev, const void *r,
                    void *V, int64_t ldv, void *vin, double rtol, double atol, int64_t maxit, int first_cycle, int part);
/* Iteration k of the cycle, w = A V_k given: Gram-Schmidt, V_{k+1}, Hessenberg column, residual estimate, stop flag.
 * k = -1: "the iteration the state block says is due" -- the launches then carry nothing that changes from one iteration to
 * the next, so operator application + step can be captured once as a hipGraph and replayed for every iteration. */
int pn_krylov_step(void *stream, int dtype, int64_t n, int restart, double *state, double *status_dev, int k, void *w,
                   void *V, int64_t ldv, void *vin, int part);
/* If the cycle has ended (stopped, or restart length reached) and x has not been updated for it yet:
 * x += sum_j y_j V_j with y from the back substitution; otherwise nothing. */
int pn_krylov_close(void *stream, int dtype, int64_t n, int restart, double *state, double *status_dev, void *x, void *V,
                    int64_t ldv);

/* ------------------------------------------------------------------------------------------
 * 4. Checkpoint scheduler.  Replaces TSTrajectory as enabled by ts.setSaveTrajectory()
 *    (pa.py:771-772) with -ts_trajectory_solution_only / -ts_trajectory_max_cps_ram
 *    (README.md:91-96).  It only plans: slots are indices into HBM slabs owned by the caller.
 *      mode PN_TRAJ_ALL       every step's state AND stage values kept  (solution_only 0)
 *      mode PN_TRAJ_SOLUTION  every step's state kept, stages recomputed (PETSc default)
 *      mode PN_TRAJ_BUDGET    at most max_slots states kept, the rest recomputed
 * ---------------------------------------------------------------------------------------- */
typedef enum { PN_TRAJ_ALL = 0, PN_TRAJ_SOLUTION = 1, PN_TRAJ_BUDGET = 2 } pn_traj_mode;
typedef struct pn_traj pn_traj;

pn_traj *pn_traj_create(void);
void pn_traj_destroy(pn_traj *tj);
int pn_traj_begin(pn_traj *tj, int mode, int64_t max_slots);
/* BUDGET mode, after pn_traj_begin: the checkpoints carry the stage values of their step (-ts_trajectory_solution_only 0:
 * PETSc's checkpoints hold them, README.md:91-96 "optimal checkpointing").  They are written whenever a sweep steps on
 * from a kept state, and reversing such a step then recomputes nothing.  Placement is then optimal for THAT cost
 * (re-advanced steps + stage computations of the reversed steps; the CAMS cost model of PETSc's TSTrajectory memory
 * type) instead of for the re-advanced steps alone. */
int pn_traj_set_carry(pn_traj *tj, int carries_stage_values);
/* Optional, BUDGET mode: the number of steps of the coming forward sweep when it is known in
 * advance (fixed step).  The sweep then keeps the states of the binomial-optimal (revolve-type)
 * schedule instead of thinning online; the reverse sweep places its intermediate checkpoints
 * optimally in either case (dynamic programme, up to 8192 steps x 64 slots). */
int pn_traj_set_total(pn_traj *tj, int64_t nsteps);
/* Number of accepted steps a fixed-step solve started with pn_ts_begin will take (dry run of the
 * state machine on a copy; -1 for an adaptive scheme). */
int64_t pn_ts_count_fixed_steps(const pn_ts *ts);
/* Forward sweep: where does the state at the START of step `step` go?  Returns a slot index,
 * or -1 = not kept (caller uses a work buffer).  May recycle slots (BUDGET mode). */
int64_t pn_traj_fwd_slot(pn_traj *tj, int64_t step);
/* Reverse sweep, to reverse step `step` (needs the state at its start):
 *   from_step, from_slot: nearest kept state at or before `step`;
 *   the caller re-advances from_step -> step, and on the way stores the states at
 *   store_step[k] into store_slot[k] (k < *nstore, at most cap entries). */
int pn_traj_rev_plan(pn_traj *tj, int64_t step, int64_t *from_step, int64_t *from_slot,
                     int *nstore, int64_t *store_step, int64_t *store_slot, int cap);
/* Step `step` has been reversed: the state at the start of step+1.. is no longer needed. */
int pn_traj_rev_done(pn_traj *tj, int64_t step);
int64_t pn_traj_slots_in_use(const pn_traj *tj);
/* Diagnostic: how many checkpoint-placement tables (dynamic programmes) this process has built so far.  The tables are
 * cached process-wide by (cost model, steps, slots): a training loop builds them once, not once per solve. */
int64_t pn_traj_dp_builds(void);
int64_t pn_traj_high_water(const pn_traj *tj);

/* ------------------------------------------------------------------------------------------
 * 5. Disk tier of the trajectory.  Replaces TSTrajectory type "basic" -- PETSc's DEFAULT type, the one the
 *    reference runs with unless `-ts_trajectory_type memory` is given (examples-pnode/ode_demo_petsc.py:26
 *    "By default, disk is used"; README.md:91-96): one file per checkpoint under -ts_trajectory_dirname
 *    (default "SA-data"), removed afterwards unless -ts_trajectory_keep_files.  Here HBM is the default
 *    tier and `-ts_trajectory_type basic` selects this one.  A checkpoint is `slot_bytes` raw bytes (the
 *    state at the start of a step, followed by the step's stage values in store-all mode).
 *    Copies are asynchronous on the caller's stream through `nbuf` pinned staging buffers; file I/O runs on
 *    a thread of the engine.  `device` = 0 makes every pointer a host pointer (plain memcpy): the mode the
 *    CPU-only tests of the host logic use.
 * ---------------------------------------------------------------------------------------- */
typedef struct pn_spill pn_spill;
pn_spill *pn_spill_create(const char *dir, int64_t slot_bytes, int nbuf, int device, int keep_files);
/* waits for outstanding I/O; deletes the files it wrote (and the directory, if it created it and it is
 * empty) unless keep_files was set */
void pn_spill_destroy(pn_spill *sp);
/* forward sweep (TSTrajectorySet): checkpoint `id` <- slot_bytes at `src`.  Returns once the copy is
 * enqueued on `stream`; the caller may recycle `src` with later work on the same stream. */
int pn_spill_put(pn_spill *sp, void *stream, int64_t id, const void *src);
/* reverse sweep: start reading checkpoint `id` in the background (no-op if every staging buffer is busy) */
int pn_spill_prefetch(pn_spill *sp, int64_t id);
/* reverse sweep (TSTrajectoryGet): checkpoint `id` -> `dst`, enqueued on `stream` after the file has been read */
int pn_spill_get(pn_spill *sp, void *stream, int64_t id, void *dst);
/* checkpoint `id` is no longer needed: its file is deleted (unless keep_files) */
int pn_spill_drop(pn_spill *sp, int64_t id);
int pn_spill_stats(pn_spill *sp, int64_t *files, int64_t *bytes_written, int64_t *bytes_read, int64_t *waits);

#ifdef __cplusplus
}
#endif
#endif /* PNODE_AMD_H */
