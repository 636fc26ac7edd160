/*
 * hank_hip.h — C ABI of the MI355X-native household block (the sequence-space JVP hot path).
 *
 * The reference (vasudeva-ram/Julia-NewtonRaphsonHANK) has no FFI; its boundary for this path is a
 * set of Julia call signatures. Each entry point below names the reference interface it replaces;
 * the Julia `ccall` shims that keep those signatures are in INTEGRATION.md / julia/HankHIP.jl.
 *
 * Conventions (all of them the reference's own):
 *   - every array is fp64, column-major; policy / value / distribution matrices are n_a x n_e
 *     with wealth fastest (ForwardIteration.jl:6-10); G = n_a*n_e; P = T-1 transition periods.
 *   - Pi is the row-stochastic productivity transition, column-major: Pi[e + n_e*e2] = P(e -> e2)
 *     (GeneralStructures.jl:500-525; used as V'*Pi' backward, KrusellSmith.jl:59, and D*Pi forward,
 *     ForwardIteration.jl:280-284).
 *   - "xhh" are the per-period household inputs the value function reads from xVals; for the
 *     Krusell-Smith plugin that is (r_t, w_t) (KrusellSmith.jl:53-54): xhh[k + n_hh*t], n_hh = 2
 *     (3 for the one-asset HANK family: (r_t, w_t, tr_t)); hank_n_hh() tells.
 *   - tangent batches are Julia-natural: dxhh is (n_hh, P, N) column-major, dagg is (P, N).
 *   - the caller owns every host buffer; the library copies in/out and retains no pointer past a
 *     call; the context owns all device memory. A context is bound to ONE HIP device (hank_create:
 *     the current one; hank_create_on: the one named), is not thread-safe, and host-pointer calls are synchronous on return.
 *   - every function returns a status (0 = ok). Julia exceptions on this path become codes:
 *     hank_last_error() carries the message the shim turns back into error(...).
 *
 * There is NO CPU fallback: every entry point fails with HANK_ERR_NO_DEVICE when no gfx950 device
 * is usable.
 */
#ifndef HANK_HIP_H
#define HANK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hank_ctx hank_ctx;

enum {
    HANK_OK = 0,
    HANK_ERR_NO_DEVICE = 1,   /* no usable HIP device / HIP runtime error                          */
    HANK_ERR_BAD_ARG = 2,     /* shape / argument errors (error(...) in the Julia callers)         */
    HANK_ERR_KNOTS = 3,       /* Interpolations.jl: knot-vectors must be unique and sorted
                                 (KrusellSmith.jl:69-72; acknowledged at SteadyState.jl:129-131)   */
    HANK_ERR_DOMAIN = 4,      /* Julia DomainError: negative base under a non-integer power
                                 (KrusellSmith.jl:59, :80)                                         */
    HANK_ERR_NOT_READY = 5,   /* boundary / primal not set before a dependent call                 */
    HANK_ERR_NONMONOTONE = 6, /* savings policy not monotone in wealth (cannot happen when the
                                 knots check passes; guards the segmented Young push-forward)      */
    HANK_ERR_NOMEM = 7,
    HANK_ERR_SWEEP = 8,       /* a persistent sweep could not form its workgroup groups or a wait in it timed out (every wait
                                 is bounded in time: 20 ms, HANK_XWAIT_MS at hank_create; the message says how long it
                                 waited); host-pointer entries fall back to the per-period launches by themselves. The
                                 persistent sweeps assume the process has the GPU to itself          */
    HANK_ERR_LAUNCH = 9       /* a kernel launch / graph / copy was refused by the HIP runtime (the message names it)   */
};

/* value-function families resolved from the YAML `function:` name (KrusellSmith.yaml:86,
 * ModelParser.jl:338-342 / _lookup_fn :404-413). */
enum {
    HANK_VF_KRUSELL_SMITH = 0,  /* KrusellSmith.jl:43-83; household inputs (r_t, w_t)                                  */
    HANK_VF_ONE_ASSET_HANK = 1  /* the same EGM step with a lump-sum transfer: inputs (r_t, w_t, tr_t), cash on hand
                                   (1+r) a + w z_e + tr. NOT in the reference: parity unpinned     */
};

/* The model-side constants BackwardIteration/ForwardIteration read from `model::SequenceModel`
 * (GeneralStructures.jl:216-226): heterogeneity grids + params + compspec.T. */
typedef struct {
    int32_t n_a;         /* model.heterogeneity.wealth.n                                          */
    int32_t n_e;         /* model.heterogeneity.productivity.n                                    */
    int32_t T;           /* model.compspec.T ; P = T-1 periods are solved                         */
    int32_t value_fn_id; /* HANK_VF_*                                                             */
    const double *a_grid; /* [n_a] wealth grid, strictly increasing                                */
    const double *z_grid; /* [n_e] productivity grid                                               */
    const double *Pi;     /* [n_e*n_e] column-major row-stochastic                                 */
    double beta, gamma, borrow_cons; /* model.params.β, γ, borrow_cons (KrusellSmith.jl:52)        */
} hank_model;

/* ---- lifetime ------------------------------------------------------------------------------- */
int hank_create(const hank_model *model, hank_ctx **out);   /* on the calling thread's current HIP device */
/* The same on HIP device `device` (0 .. hipGetDeviceCount-1). The reference has no notion of a device: `JVP(func, primal,
 * tangent)` (GeneralStructures.jl:542-550) is linear in `tangent`, so the columns of a tangent batch (the Jacobian
 * columns of SteadyStateJacobian.jl:240-243, ForwardDiff's chunks) shard over one context per GPU of a node; every entry
 * point makes its context's device current for the call and restores the caller's, so ONE host thread can keep all of a
 * node's contexts busy through the *_dev entries ("one process, eight GPUs"). */
int hank_create_on(const hank_model *model, int32_t device, hank_ctx **out);
/* hank_gather_columns: a tangent batch sharded by columns over one context per GPU (hank_create_on), assembled in ONE GPU's memory
 * over xGMI — what a single-process host (the Julia reference is one) needs in place of the RCCL all-gather of the one-process-per-
 * GPU form; `JVP(func, primal, tangent)` (GeneralStructures.jl:542-550) is linear in `tangent`, so the columns of a batch are
 * independent. ctxs[0] receives. d_blocks[k]: device pointer on ctxs[k]'s device of its (P, N_k[k]) column-major block, as written
 * by hank_jvp_dev / hank_primal_jvp_dev on ctxs[k]; d_out: (P, sum N_k) column-major on ctxs[0]'s device, the blocks in order.
 * Asynchronous: every block is copied on its OWN context's stream behind the sweeps that produce it (hipMemcpyPeerAsync from the
 * source device, peer access enabled on first use; a plain device copy where both contexts share a device) and ctxs[0]'s stream
 * waits for all of them: hank_sync(ctxs[0]) — or work enqueued on its stream — sees the assembled matrix. */
int hank_gather_columns(hank_ctx *const *ctxs, int32_t n, const double *const *d_blocks, const int32_t *N_k, double *d_out);
int hank_destroy(hank_ctx *ctx);
const char *hank_last_error(const hank_ctx *ctx); /* never NULL; "" when the last call succeeded  */
int hank_n_hh(const hank_ctx *ctx);               /* household inputs per period (KS: 2)          */
int hank_device_available(void);                  /* 1 when the current HIP device is a usable gfx950, else 0 */

/* Use the caller's HIP stream (a hipStream_t) for everything this context enqueues; NULL restores
 * the context's own stream. */
int hank_set_stream(hank_ctx *ctx, void *hip_stream);
int hank_sync(hank_ctx *ctx);

/* ---- boundary conditions ---------------------------------------------------------------------
 * ss_end_value: terminal marginal value, `ss_end.value` (BackwardIteration.jl:85).
 * ss_init_D:    initial distribution,   `ss_initial.D`  (ForwardIteration.jl:293).             */
int hank_set_boundary(hank_ctx *ctx, const double *ss_end_value, const double *ss_init_D);

/* ---- the fused household block -----------------------------------------------------------------
 * hank_primal  ==  ForwardIteration(BackwardIteration(x, ...), ...) on Float64
 *                  (NewtonRaphson.jl:78-79 inside fullFunction, called at :91).
 *   xhh[n_hh*P] -> agg_out[P] (the aggregate of the single heterogeneous variable, KD for KS:
 *   dot(vec(policy_t), D_t) with the POST-transition D_t, ForwardIteration.jl:301-307).
 *   Also records the linearisation every later hank_jvp uses.
 * hank_jvp     ==  the partials of the same composition under Dual{Tag,Float64,N}, i.e. what
 *                  JVP(fullFunction, x, y) (GeneralStructures.jl:542-550; NewtonRaphson.jl:95)
 *                  pushes through the household block, for N tangent directions at once.
 *   dxhh (n_hh, P, N) -> dagg_out (P, N).                                                        */
int hank_primal(hank_ctx *ctx, const double *xhh, double *agg_out);
int hank_jvp(hank_ctx *ctx, const double *dxhh, int32_t N, double *dagg_out);

/* Same, with DEVICE pointers (inputs already resident in HBM); enqueued on the context's stream,
 * asynchronous: call hank_sync (or synchronise the stream) before reading the outputs.
 * hank_primal_dev reports device-side errors (knots/domain) at the next hank_check. Device pointers must belong to the
 * context's device (hank_create_on); they are not validated. */
int hank_primal_dev(hank_ctx *ctx, const double *d_xhh, double *d_agg_out);
int hank_jvp_dev(hank_ctx *ctx, const double *d_dxhh, int32_t N, double *d_dagg_out);
/* hank_check: sync + fetch the device error word of the last primal. A persistent sweep that could not run (HANK_ERR_SWEEP:
 * its groups did not form, a wait ran into its deadline) is reported here for the asynchronous entries — which cannot re-run
 * a call — ONCE: a context whose schedule was not forced (HANK_SCHEDULE) continues on the per-period launches, so the caller's
 * next call succeeds (hank_stats out[4] counts it). */
int hank_check(hank_ctx *ctx);

/* hank_primal_jvp == JVP(fullFunction, x, y) exactly as the reference evaluates it: the Dual pass
 * recomputes the primal (NewtonRaphson.jl:95; GeneralStructures.jl:546-547), so value and N partials
 * travel together. One call = hank_primal + hank_jvp, but both recurrences advance together: a batch of one pass
 * (N <= 32) as TWO persistent launches that carry value and partials in every workgroup group (k_xdual_back, k_xfwd<D, true>),
 * a batch of 33 to 79 directions as ONE chain of T launches per direction (the tangent sweep runs one period behind the primal
 * sweep inside the same launches), a batch of at least HANK_WIDE_MIN = 80 directions as the Float64 sweeps followed by the on-chip
 * wide sweeps (one workgroup per direction: hank_wide.h). Leaves the context exactly as hank_primal followed by hank_jvp would.
 * PRIMAL MEMO (host-pointer form only): y_Iteration calls JVP(fullFunction, x, y) about 21 times per Newton step at ONE x
 * (NewtonRaphson.jl:91-95). When `xhh` and the boundary are bit-identical to the ones whose linearisation is on record,
 * hank_primal_jvp runs the tangent sweeps alone (= hank_jvp) and returns the recorded value: results equal the un-memoised
 * call's bit for bit under one implementation (HANK_SCHEDULE=launch|xcd) and to the rounding of the aggregate sums (1e-13)
 * in the default schedule, where a recorded primal is served by the persistent tangent sweeps. HANK_PRIMAL_MEMO=0 (read at
 * hank_create) switches it off; hank_stats out[6] counts the skipped primal sweeps. The _dev form never skips work.        */
int hank_primal_jvp(hank_ctx *ctx, const double *xhh, const double *dxhh, int32_t N, double *agg_out,
                    double *dagg_out);
int hank_primal_jvp_dev(hank_ctx *ctx, const double *d_xhh, const double *d_dxhh, int32_t N,
                        double *d_agg_out, double *d_dagg_out);

/* BackwardIteration's return value (BackwardIteration.jl:115): the policy sequence of the last
 * hank_primal, P matrices of n_a x n_e -> out[P*G]; and its partials for the last hank_jvp,
 * out[(G, P, N)] column-major (seqs_data[j][t] as Matrix{Dual}). */
int hank_get_policy_seq(hank_ctx *ctx, double *out);
int hank_get_dpolicy_seq(hank_ctx *ctx, int32_t N, double *out);
/* More than one heterogeneous variable. BackwardIteration keeps one policy sequence per heterogeneous variable
 * (`seqs_data[j][t] = result[varname]`, BackwardIteration.jl:99-112) and ForwardIteration aggregates every one of them with the
 * SAME D_t: `agg_data[j][t] = dot(vec(policy_seqs[varname][t]), D)` (ForwardIteration.jl:303-307). Every forward kernel of the
 * fused sweeps (k_dist_step, k_tan_fwd / k_fused_fwd, k_xfwd, k_wide_fwd) therefore reduces TWO dot products per period with the
 * D_t / dD_t it holds: the policy-weighted one (hank_primal / hank_jvp's output) and the grid-weighted one, sum_pt a(pt) D_t(pt).
 * Any policy that is affine in (a, z_e, 1, a') with coefficients that depend on the period's inputs aggregates as the same
 * affine combination of them; the budget residual c = (1+r_t) a + w_t z_e + tr_t - a' (the c_grid of KrusellSmith.jl:79) is the
 * one built in:
 *   hank_get_het_outputs(ctx, n_het, dxhh, N, agg_out, dagg_out): output 0 = the policy variable of the endogenous dimension
 *   (KD / A: what hank_primal and hank_jvp return), output 1 = consumption; n_het in {1, 2} says how many the model's
 *   `heterogeneous:` section lists. agg_out (P, n_het) column-major of the last primal sweep; dagg_out (P, n_het, N)
 *   column-major of the last tangent sweep — whichever kernel family ran it — whose input `dxhh` (n_hh, P, N) is passed
 *   again (the partials of r_t, w_t, tr_t enter consumption directly). dagg_out NULL or N = 0: values only.
 *   hank_get_grid_aggregates: the raw second reduction, agg2_out[P] = sum a D_t and dagg2_out (P, N) = sum a dD_t, for a host
 *   that assembles other affine outputs itself.
 * The _dev forms take device pointers and are ordered on the context's stream.                                              */
int hank_get_het_outputs(hank_ctx *ctx, int32_t n_het, const double *dxhh, int32_t N, double *agg_out, double *dagg_out);
int hank_get_het_outputs_dev(hank_ctx *ctx, int32_t n_het, const double *d_dxhh, int32_t N, double *d_agg_out, double *d_dagg_out);
int hank_get_grid_aggregates(hank_ctx *ctx, double *agg2_out, int32_t N, double *dagg2_out);
int hank_get_grid_aggregates_dev(hank_ctx *ctx, double *d_agg2_out, int32_t N, double *d_dagg2_out);
/* The distribution path D_1..D_P of the last hank_primal -> out[P*G] (ForwardIteration.jl:297-300). */
int hank_get_dist_seq(hank_ctx *ctx, double *out);

/* ---- granular steps (step-level parity with the reference's functions) -------------------------
 * hank_backward_step[_dual]  ==  model.value_fn(value_next, xVals, model) for the KS plugin
 *   `ValueFunction` (KrusellSmith.jl:43-83): -> (Value, KD). The _dual form carries N partials:
 *   dvalue_next/dvalue_out/dpolicy_out are (G, N) column-major, dxhh_t is (n_hh, N).
 * hank_forward_step[_dual]   ==  transition_step(policy, D_prev, Λ_exog, dim, n_exog)
 *   (ForwardIteration.jl:95-99, Young lottery :37-78) followed by dot(vec(policy), D_new)
 *   (:305-307). Any policy is accepted here (no monotonicity requirement).                       */
int hank_backward_step(hank_ctx *ctx, const double *value_next, const double *xhh_t,
                       double *value_out, double *policy_out);
int hank_backward_step_dual(hank_ctx *ctx, const double *value_next, const double *dvalue_next,
                            const double *xhh_t, const double *dxhh_t, int32_t N,
                            double *value_out, double *dvalue_out, double *policy_out,
                            double *dpolicy_out);
int hank_forward_step(hank_ctx *ctx, const double *policy, const double *D_prev, double *D_out,
                      double *agg_out);
int hank_forward_step_dual(hank_ctx *ctx, const double *policy, const double *dpolicy,
                           const double *D_prev, const double *dD_prev, int32_t N, double *D_out,
                           double *dD_out, double *agg_out, double *dagg_out);

/* ---- steady state ---------------------------------------------------------
 * hank_vfi == the inner fixed point of get_xVals (SteadyState.jl:132-141): value <- model.value_fn(value, xVals, model).Value
 * until max|value_new - value| < tol — checked after EVERY step, like the reference — or max_iter steps (:134 caps at
 * 10 000). The loop is device-resident. Where the grid fits the XCD-local schedule it is ONE persistent launch (k_xvfi:
 * the EGM step of the persistent Float64 sweep with constant prices; every member of the group compares its rows
 * against tol in Float64 and the group barrier carries the vote, so all members stop in the same step: 4.5 us per step
 * at 2000x11); otherwise the EGM step kernels of hank_backward_step iterate on HBM and a stop flag comes back per chunk
 * of steps (23 us per step). Same stopping rule, same step count, same value (tests/test_gpu_steady_state.py).
 * value_io[G]: in = the starting value (ones in the reference, :132), out = the
 * converged value; policy_out[G] = the policy of the step that produced it; *iters_out = steps taken,
 * *supnorm_out = the last max|difference|. The price Newton and invariant_dist stay on the host. */
int hank_vfi(hank_ctx *ctx, const double *xhh_t, double tol, int32_t max_iter, double *value_io, double *policy_out,
             int32_t *iters_out, double *supnorm_out);

/* hank_stationary_dist: the stationary distribution of the steady state by the power method on the device — D <- Lambda D with
 * the forward step of the hot path (Young lottery of `policy`[G] + exogenous transition), until two iterates
 * `check_every` steps apart differ by less than `tol` (max norm) or `max_iter` steps. This is the iteration the host's
 * invariant_dist uses for chains too large for the reference's direct solve (ForwardIteration.jl:436-442); same fixed
 * point. Where the grid fits the XCD-local schedule the whole iteration is ONE persistent launch (k_xstat: 3.0 us per
 * iteration at 2000x11, the verdict of every check rides on the group barrier), otherwise one launch per iteration and a
 * stop flag per chunk of checks (*iters_out then counts the iterations enqueued when the host saw the flag).
 * D_io[G]: in = start (any positive vector), out = the last iterate (NOT normalised). */
int hank_stationary_dist(hank_ctx *ctx, const double *policy, double *D_io, double tol, int32_t max_iter, int32_t check_every,
                         int32_t *iters_out);

/* hank_fake_news: the household block's sequence-space Jacobian AT THE STEADY STATE from its Toeplitz structure — what
 * getSteadyStateJacobian assembles from n_endog backward JVPs seeded at the last period (JBI, SteadyStateJacobian.jl:240-243),
 * the pullbacks through ForwardIteration (JFI, :249-253), `helper[t,s] = JFI[:, t-slice] * JBI[s-slice, :]` (:300-305) and
 * the recursion J̅[s,t] = J̅[s-1,t-1] + helper (:363-371). Here: ONE backward tangent sweep of n_hh directions (unit shock to
 * household input k in the last period: the policy response at every lag), one single-period forward push of all P*n_hh lagged
 * responses (the lottery impulse), P-1 steps of the transposed forward step (the expectation vectors), one product.
 * Requires hank_primal (host-pointer form) at the constant steady-state path with the steady state as both boundaries; anything
 * else is refused with HANK_ERR_NOT_READY (a path that varies over time, a device-pointer primal, or a recorded policy whose
 * first and last period differ by more than 1e-6 of its scale).
 *   F_out  (P, P, n_hh) column-major: F[u, j, k] = effect on the aggregate, u periods after the policy moved, of the policy
 *          response with j periods to go before a unit shock to input k (the reference's helper; "fake news" matrix)
 *   Dv_out (P, n_hh): Dv[j, k] = the direct term, (policy response at lag j) . D_ss
 * d agg_t / d xhh_{k,s} = J_k[t, s] with J_k[t, s] = J_k[t-1, s-1] + F[t, s, k], J_k[0, s] = Dv[s, k] + F[0, s, k],
 * J_k[t, 0] = F[t, 0, k] for t > 0 (0-based; hank_amd.SteadyStateJacobian.household_jacobian does the recursion). */
int hank_fake_news(hank_ctx *ctx, double *F_out, double *Dv_out);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------
 * Device time, in milliseconds, of the sweeps of the most recent hank_primal[_dev]/hank_jvp[_dev],
 * from HIP events recorded on the context's stream around each sweep:
 *   out[0] primal backward, out[1] primal forward, out[2] tangent backward, out[3] tangent forward,
 *   out[4] dual-sweep backward, out[5] dual-sweep forward (hank_primal_jvp); -1 where not applicable.
 * launches[k] = kernel launches inside sweep k. Valid after hank_sync. */
int hank_last_timings(hank_ctx *ctx, double out_ms[6], int32_t launches[6]);

/* Counters of this context (tests and scripts): out[0] sweep kernels launched by the persistent schedule, out[1] tangent
 * workspaces allocated (a change of batch width N re-uses a cached workspace: a small most-recently-used cache, 3 deep),
 * out[2] hipGraphs captured (per-period schedule), out[3] schedule in use (1 = XCD-local persistent sweeps, 0 = one
 * launch per period), out[4] times this context fell back from 1 to 0, out[5] device-resident value-function iterations
 * run by hank_vfi, out[6] primal sweeps the memo of hank_primal_jvp skipped, out[7] primal sweeps run (hank_primal[_dev],
 * hank_primal_jvp[_dev]). */
int hank_stats(hank_ctx *ctx, int64_t out[8]);

/* Which implementation served the last tangent sweep and how the context chooses (bench.py and the tests name the kernel family
 * they measured; the reference has one implementation, ForwardDiff's Dual pass, GeneralStructures.jl:542-550): out[0] the family of
 * the last tangent sweep (0 = one launch per period, 1 = XCD-local persistent sweeps, 2 = on-chip wide sweeps: one workgroup per
 * direction, the loop-carried state in registers), out[1] the wide sweeps' mode (0 off, 1 auto: batches of at least out[2]
 * directions, 2 every batch: HANK_SCHEDULE=wide), out[2] that threshold (HANK_WIDE_MIN at hank_create), out[3] 1 when the grid fits
 * the wide sweeps, out[4] the widest batch the persistent tangent sweeps take at a recorded primal, out[5] 1 when the tangent sweeps
 * rebuild kc and v instead of reading them (record diet), out[6] bytes of the linearisation record, out[7] reserved. */
int hank_info(hank_ctx *ctx, int64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* HANK_HIP_H */
