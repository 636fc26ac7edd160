/*
 * hank_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never on the product path).
 *
 * A plain-C restatement of the sequence-space JVP hot path of
 * vasudeva-ram/Julia-NewtonRaphsonHANK, written from the reference source read as text.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.
 *
 * PARITY PINNING: the reference ships NO golden vectors for this path and no Julia toolchain
 * exists in the build image, so this oracle is "parity unpinned" against reference *outputs*.
 * It is pinned instead by (tests/test_oracle_*.py):
 *   - the reference's own self-consistency checks,
 *   - central finite differences of its own Float64 path (AD-vs-FD, SteadyState.jl:296-356),
 *   - the reference's C++ dual-number classes compiled from /root/reference into oracle/_ref
 *     (ForwardDiff.jl/benchmarks/cpp, +,-,*,sqrt,exp and the rosenbrock known answers).
 * Third-party pieces restated from their published algorithm (source not under /root/reference):
 *   Interpolations.jl 0.16.2 Gridded(Linear()) + Flat() (call site KrusellSmith.jl:69-72) and
 *   DiffRules 1.15.1 `max` rule — both "parity unpinned" in isolation.
 *
 * Numbers are ForwardDiff-style duals: value + NP partials, NP fixed at compile time
 * (this file is compiled once per NP, symbols get the suffix _n<NP>), mirroring
 * Dual{T,Float64,N} / Partials{N,Float64} (ForwardDiff.jl/src/dual.jl:14-21, partials.jl:1-3).
 *
 * All matrices are column-major n_a x n_e with wealth fastest (ForwardIteration.jl:6-10).
 * Pi is column-major, row-stochastic: Pi[e + n_e*e2] = P(e -> e2).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifndef NP
#define NP 1
#endif

#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(CAT(name, _n), NP)

/* status codes shared with the HIP library (include/hank_hip.h) */
#define ORC_OK 0
#define ORC_ERR_KNOTS 3  /* Interpolations: knot-vectors must be unique and sorted        */
#define ORC_ERR_DOMAIN 4 /* Julia DomainError: negative base under a non-integer power    */

typedef struct {
    double v;
    double p[NP];
} dual;

/* ---- Dual arithmetic (ForwardDiff.jl/src/dual.jl, partials.jl) -------------------------- */

/* convert(Dual, x::Real): zero partials (dual.jl:461-462) */
static inline dual d_const(double x) {
    dual r;
    r.v = x;
    for (int k = 0; k < NP; k++) r.p[k] = 0.0;
    return r;
}
/* Dual + Dual, Dual + Real (dual.jl:495-504) */
static inline dual d_add(dual x, dual y) {
    dual r;
    r.v = x.v + y.v;
    for (int k = 0; k < NP; k++) r.p[k] = x.p[k] + y.p[k];
    return r;
}
static inline dual d_add_r(dual x, double y) {
    dual r = x;
    r.v = x.v + y;
    return r;
}
/* Dual - Dual, Dual - Real, Real - Dual (dual.jl:506-515) */
static inline dual d_sub(dual x, dual y) {
    dual r;
    r.v = x.v - y.v;
    for (int k = 0; k < NP; k++) r.p[k] = x.p[k] - y.p[k];
    return r;
}
static inline dual d_sub_r(dual x, double y) {
    dual r = x;
    r.v = x.v - y;
    return r;
}
static inline dual r_sub_d(double x, dual y) {
    dual r;
    r.v = x - y.v;
    for (int k = 0; k < NP; k++) r.p[k] = -y.p[k];
    return r;
}
/* Dual * Dual: value vx*vy, partials mul_tuples(px, py, vy, vx) = vy*px + vx*py
 * (DiffRules-generated, dual.jl:472-486; partials.jl:117-119, :219-221) */
static inline dual d_mul(dual x, dual y) {
    dual r;
    r.v = x.v * y.v;
    for (int k = 0; k < NP; k++) r.p[k] = (y.v * x.p[k]) + (x.v * y.p[k]);
    return r;
}
/* Dual * Real: value*y, partials*y (scale_tuple, partials.jl:199-201) */
static inline dual d_mul_r(dual x, double y) {
    dual r;
    r.v = x.v * y;
    for (int k = 0; k < NP; k++) r.p[k] = x.p[k] * y;
    return r;
}
/* Dual / Real (dual.jl:534): value/y, partials/y (div_tuple_by_scalar, partials.jl:203-205) */
static inline dual d_div_r(dual x, double y) {
    dual r;
    r.v = x.v / y;
    for (int k = 0; k < NP; k++) r.p[k] = x.p[k] / y;
    return r;
}
/* Real / Dual (dual.jl:535-539): divv = x/v ; partials = -(divv/v) * p */
static inline dual r_div_d(double x, dual y) {
    dual r;
    double divv = x / y.v;
    double f = -(divv / y.v);
    r.v = divv;
    for (int k = 0; k < NP; k++) r.p[k] = y.p[k] * f;
    return r;
}
/* Dual / Dual (dual.jl:528-533): _div_partials = mul_tuples(px, py, inv(vy), -(vx/(vy*vy))) */
static inline dual d_div(dual x, dual y) {
    dual r;
    double ia = 1.0 / y.v, ib = -(x.v / (y.v * y.v));
    r.v = x.v / y.v;
    for (int k = 0; k < NP; k++) r.p[k] = (ia * x.p[k]) + (ib * y.p[k]);
    return r;
}
static inline int d_isconstant(dual x) {
    for (int k = 0; k < NP; k++)
        if (x.p[k] != 0.0) return 0;
    return 1;
}
/* Julia's Float64^Float64 throws DomainError for a negative base with a non-integer exponent. */
static inline int pow_domain_error(double v, double y) { return (v < 0.0) && (y != floor(y)) && isfinite(y); }
/* Dual ^ Real (dual.jl:563-572): zero partials if y==0 or partials all zero,
 * else partials * y * v^(y-1) */
static inline dual d_pow_r(dual x, double y, int *status) {
    dual r;
    if (pow_domain_error(x.v, y)) *status = ORC_ERR_DOMAIN;
    r.v = pow(x.v, y);
    if (y == 0.0 || d_isconstant(x)) {
        for (int k = 0; k < NP; k++) r.p[k] = 0.0;
    } else {
        double pw = pow(x.v, y - 1.0);
        for (int k = 0; k < NP; k++) r.p[k] = (x.p[k] * y) * pw;
    }
    return r;
}
/* max(Dual, Real) — DiffRules 1.15.1 rule (restated; source not in /root/reference):
 *   d/dx max(x,y) = ifelse((y > x) | (signbit(y) < signbit(x)), ifelse(isnan(y),1,0),
 *                          ifelse(isnan(x),0,1)) */
static inline dual d_max_r(dual x, double y) {
    dual r;
    double dvx;
    if ((y > x.v) || ((signbit(y) != 0) < (signbit(x.v) != 0)))
        dvx = isnan(y) ? 1.0 : 0.0;
    else
        dvx = isnan(x.v) ? 0.0 : 1.0;
    /* Julia's max propagates NaN; finite inputs on this path */
    r.v = (isnan(x.v) || isnan(y)) ? NAN : ((x.v > y) ? x.v : ((y > x.v) ? y : (signbit(x.v) ? y : x.v)));
    for (int k = 0; k < NP; k++) r.p[k] = x.p[k] * dvx;
    return r;
}

/* ---- exported scalar hooks so tests can pin the dual rules against oracle/_ref ------------ */
/* layout of a dual in memory: (1+NP) contiguous doubles, value first (AoS, like Dual{T,V,N}) */
void FN(orc_dual_binop)(int op, const double *x, const double *y, double *out) {
    dual a, b, r;
    memcpy(&a, x, sizeof(dual));
    memcpy(&b, y, sizeof(dual));
    int st = 0;
    switch (op) {
    case 0: r = d_add(a, b); break;
    case 1: r = d_sub(a, b); break;
    case 2: r = d_mul(a, b); break;
    case 3: r = d_div(a, b); break;
    case 4: r = d_mul_r(a, b.v); break;
    case 5: r = r_sub_d(a.v, b); break;
    case 6: r = r_div_d(a.v, b); break;
    case 7: r = d_pow_r(a, b.v, &st); break;
    case 8: r = d_max_r(a, b.v); break;
    default: r = d_const(NAN);
    }
    memcpy(out, &r, sizeof(dual));
}

typedef struct {
    int32_t n_a, n_e;
    const double *a;  /* wealth grid [n_a]                                  */
    const double *z;  /* productivity grid [n_e]                            */
    const double *Pi; /* [n_e*n_e] column-major, row-stochastic             */
    double beta, gamma, borrow_cons;
} orc_model;

/* ---- Interpolations.jl Gridded(Linear()) + extrapolate(..., Flat()) ----------------------
 * knots: dual, strictly increasing in value; vals: Float64; query x: Float64.
 * Flat(): x is clamped to [knots[1], knots[end]] (the clamped value IS the dual knot).
 * Bracket: i = clamp(searchsortedlast(knots, x), 1, n-1) (comparisons on values only,
 * dual.jl:395-404); f = (x - k_i)/(k_{i+1} - k_i); result = (1-f)*v_i + f*v_{i+1}.        */
static inline dual interp_flat(const dual *knots, const double *vals, int n, double x) {
    dual xc;
    if (x < knots[0].v)
        xc = knots[0];
    else if (x > knots[n - 1].v)
        xc = knots[n - 1];
    else
        xc = d_const(x);
    /* searchsortedlast: last index with knots[i] <= xc */
    int lo = -1, hi = n; /* invariant: knots[lo] <= xc < knots[hi] */
    while (hi - lo > 1) {
        int mid = lo + ((hi - lo) >> 1);
        if (knots[mid].v <= xc.v)
            lo = mid;
        else
            hi = mid;
    }
    int i = lo;
    if (i < 0) i = 0;
    if (i > n - 2) i = n - 2;
    dual f = d_div(d_sub(xc, knots[i]), d_sub(knots[i + 1], knots[i]));
    dual w0 = r_sub_d(1.0, f);
    return d_add(d_mul_r(w0, vals[i]), d_mul_r(f, vals[i + 1]));
}

/* ---- ValueFunction: one EGM step (KrusellSmith.jl:43-83) ---------------------------------
 * value_next, Value, KD: n_a x n_e dual matrices (column-major); r, w duals.               */
static int FN(value_function_impl)(const orc_model *m, const double *value_next_, const double *r_,
                                   const double *w_, const double *tr_, double *Value_, double *KD_) {
    const int n_a = m->n_a, n_e = m->n_e;
    const dual *value_next = (const dual *)value_next_;
    dual *Value = (dual *)Value_, *KD = (dual *)KD_;
    dual r, w, tr = d_const(0.0);
    memcpy(&r, r_, sizeof(dual));
    memcpy(&w, w_, sizeof(dual));
    if (tr_) memcpy(&tr, tr_, sizeof(dual));
    int status = ORC_OK;
    dual *knots = (dual *)malloc(sizeof(dual) * (size_t)n_a);

    dual one_plus_r = d_add_r(r, 1.0);      /* (1 + r)            */
    dual rho = r_div_d(1.0, one_plus_r);    /* 1 / (1 + r)        */

    for (int e = 0; e < n_e; e++) {
        /* :59  cmat = (β .* (value_next * Π')) .^ (-1/γ) ; (V*Π')[a,e] = Σ_e2 V[a,e2]*Π[e,e2] */
        for (int ia = 0; ia < n_a; ia++) {
            dual acc = d_mul_r(value_next[ia], m->Pi[e + n_e * 0]);
            for (int e2 = 1; e2 < n_e; e2++)
                acc = d_add(acc, d_mul_r(value_next[ia + n_a * e2], m->Pi[e + n_e * e2]));
            dual cm = d_pow_r(d_mul_r(acc, m->beta), -1.0 / m->gamma, &status);
            /* :62 impliedstate = (1/(1+r)) .* (cmat .- (w .* labor) .+ policy_a) */
            dual inc = d_mul_r(w, m->z[e]);           /* labour income; + the lump-sum transfer in the HANK family */
            if (tr_) inc = d_add(inc, tr);
            dual t = d_add_r(d_sub(cm, inc), m->a[ia]);
            knots[ia] = d_mul(rho, t);
        }
        /* Interpolations check: knots sorted and unique (values) */
        for (int ia = 1; ia < n_a; ia++)
            if (!(knots[ia].v > knots[ia - 1].v)) status = (status == ORC_OK) ? ORC_ERR_KNOTS : status;
        for (int ia = 0; ia < n_a; ia++) {
            /* :66-73 interpolate vals=grid at knots=impliedstate, evaluated on the grid */
            dual g = interp_flat(knots, m->a, n_a, m->a[ia]);
            /* :76 borrowing constraint */
            g = d_max_r(g, m->borrow_cons);
            /* :79 c_grid = (1+r).*a .+ (w.*z) .- g */
            dual inc = d_mul_r(w, m->z[e]);
            if (tr_) inc = d_add(inc, tr);
            dual c = d_sub(d_add(d_mul_r(one_plus_r, m->a[ia]), inc), g);
            /* :80 value_current = (1+r) .* (c_grid .^ (-γ)) */
            Value[ia + n_a * e] = d_mul(one_plus_r, d_pow_r(c, -m->gamma, &status));
            KD[ia + n_a * e] = g;
        }
    }
    free(knots);
    return status;
}

int FN(orc_value_function)(const orc_model *m, const double *value_next_, const double *r_,
                           const double *w_, double *Value_, double *KD_) {
    return FN(value_function_impl)(m, value_next_, r_, w_, NULL, Value_, KD_);
}
/* The one-asset HANK family (NOT in the reference): the same EGM step with a lump-sum
 * transfer tr in the budget, cash on hand (1+r) a + w z_e + tr. Parity unpinned by construction.          */
int FN(orc_value_function_tr)(const orc_model *m, const double *value_next_, const double *r_,
                              const double *w_, const double *tr_, double *Value_, double *KD_) {
    return FN(value_function_impl)(m, value_next_, r_, w_, tr_, Value_, KD_);
}

/* ---- BackwardIteration (BackwardIteration.jl:46-116) -------------------------------------
 * xr, xw: dual paths of r_t, w_t, t = 1..P (the only xVals entries KS's value_fn reads,
 * KrusellSmith.jl:53-54). Terminal value = ss_end.value with zero partials (:85).
 * policy_seq: P dual matrices, period-major (seqs_data[j][t], :110-112).                    */
static int FN(backward_iteration_impl)(const orc_model *m, int P, const double *xr_, const double *xw_, const double *xt_,
                                       const double *ss_end_value, double *policy_seq_) {
    const int G = m->n_a * m->n_e;
    const dual *xr = (const dual *)xr_, *xw = (const dual *)xw_, *xt = (const dual *)xt_;
    dual *policy_seq = (dual *)policy_seq_;
    dual *value = (dual *)malloc(sizeof(dual) * (size_t)G);
    dual *vnew = (dual *)malloc(sizeof(dual) * (size_t)G);
    for (int i = 0; i < G; i++) value[i] = d_const(ss_end_value[i]);
    int status = ORC_OK;
    for (int i = 1; i <= P; i++) {
        int t = P + 1 - i - 1; /* Julia t = T - i (1-based) -> 0-based */
        int st = FN(value_function_impl)(m, (const double *)value, (const double *)&xr[t],
                                         (const double *)&xw[t], xt ? (const double *)&xt[t] : NULL, (double *)vnew,
                                         (double *)(policy_seq + (size_t)t * G));
        if (st != ORC_OK && status == ORC_OK) status = st;
        dual *tmp = value;
        value = vnew;
        vnew = tmp;
    }
    free(value);
    free(vnew);
    return status;
}

int FN(orc_backward_iteration)(const orc_model *m, int P, const double *xr_, const double *xw_,
                               const double *ss_end_value, double *policy_seq_) {
    return FN(backward_iteration_impl)(m, P, xr_, xw_, NULL, ss_end_value, policy_seq_);
}

int FN(orc_backward_iteration_tr)(const orc_model *m, int P, const double *xr_, const double *xw_, const double *xt_,
                                  const double *ss_end_value, double *policy_seq_) {
    return FN(backward_iteration_impl)(m, P, xr_, xw_, xt_, ss_end_value, policy_seq_);
}

/* ---- transition_step (ForwardIteration.jl:37-99) -----------------------------------------
 * D_new = Λ_exog * (Λ_endog(policy) * D_prev); Λ_endog is Young's lottery, block diagonal in e,
 * searchsortedfirst tie rule (:52); CSC mat-vec accumulation order (column by column).       */
void FN(orc_transition_step)(const orc_model *m, const double *policy_, const double *D_prev_,
                             double *D_new_) {
    const int n_a = m->n_a, n_e = m->n_e, G = n_a * n_e;
    const dual *policy = (const dual *)policy_, *D_prev = (const dual *)D_prev_;
    dual *D_new = (dual *)D_new_;
    const double *grid = m->a;
    dual *mid = (dual *)malloc(sizeof(dual) * (size_t)G);
    for (int i = 0; i < G; i++) mid[i] = d_const(0.0);
    for (int e = 0; e < n_e; e++) {
        for (int ia = 0; ia < n_a; ia++) {
            int col = e * n_a + ia;
            dual p = policy[col];
            /* searchsortedfirst(grid, p): first index with grid[m] >= p (1-based mm) */
            int lo = -1, hi = n_a; /* grid[lo] < p <= grid[hi] */
            while (hi - lo > 1) {
                int mid_i = lo + ((hi - lo) >> 1);
                if (grid[mid_i] < p.v)
                    lo = mid_i;
                else
                    hi = mid_i;
            }
            int mm = hi + 1; /* 1-based */
            if (mm == 1) {
                mid[e * n_a + 0] = d_add(mid[e * n_a + 0], d_mul(d_const(1.0), D_prev[col]));
            } else if (mm > n_a) {
                mid[e * n_a + n_a - 1] = d_add(mid[e * n_a + n_a - 1], d_mul(d_const(1.0), D_prev[col]));
            } else {
                /* w = (p - grid[m-1]) / (grid[m] - grid[m-1]) */
                dual wgt = d_div_r(d_sub_r(p, grid[mm - 2]), grid[mm - 1] - grid[mm - 2]);
                dual omw = d_sub(d_const(1.0), wgt); /* one(eltype) - w (:68) */
                mid[e * n_a + mm - 2] = d_add(mid[e * n_a + mm - 2], d_mul(omw, D_prev[col]));
                mid[e * n_a + mm - 1] = d_add(mid[e * n_a + mm - 1], d_mul(wgt, D_prev[col]));
            }
        }
    }
    /* Λ_exog = kron(sparse(Π'), I): D_new[(e2,a)] += Π[e,e2] * mid[(e,a)] (:280-284, :98) */
    for (int i = 0; i < G; i++) D_new[i] = d_const(0.0);
    for (int e = 0; e < n_e; e++)
        for (int ia = 0; ia < n_a; ia++)
            for (int e2 = 0; e2 < n_e; e2++) {
                double pi = m->Pi[e + n_e * e2];
                if (pi != 0.0) /* sparse(Π') drops structural zeros */
                    D_new[e2 * n_a + ia] = d_add(D_new[e2 * n_a + ia], d_mul_r(mid[e * n_a + ia], pi));
            }
    free(mid);
}

/* ---- ForwardIteration (ForwardIteration.jl:253-311) --------------------------------------
 * D_0 = ss_initial.D (zero partials, :293); agg[t] = dot(vec(policy_t), D_t) with the
 * POST-transition D_t (:301-307).                                                            */
void FN(orc_forward_iteration)(const orc_model *m, int P, const double *policy_seq_,
                               const double *ss_init_D, double *agg_, double *D_seq_out_) {
    const int G = m->n_a * m->n_e;
    const dual *policy_seq = (const dual *)policy_seq_;
    dual *agg = (dual *)agg_;
    dual *D = (dual *)malloc(sizeof(dual) * (size_t)G);
    dual *Dn = (dual *)malloc(sizeof(dual) * (size_t)G);
    for (int i = 0; i < G; i++) D[i] = d_const(ss_init_D[i]);
    for (int t = 0; t < P; t++) {
        const dual *pol = policy_seq + (size_t)t * G;
        FN(orc_transition_step)(m, (const double *)pol, (const double *)D, (double *)Dn);
        dual *tmp = D;
        D = Dn;
        Dn = tmp;
        dual s = d_const(0.0);
        for (int i = 0; i < G; i++) s = d_add(s, d_mul(pol[i], D[i]));
        agg[t] = s;
        if (D_seq_out_) memcpy((dual *)D_seq_out_ + (size_t)t * G, D, sizeof(dual) * (size_t)G);
    }
    free(D);
    free(Dn);
}

/* ---- more than one heterogeneous variable (BackwardIteration.jl:99-112, ForwardIteration.jl:303-307) ------------
 * A value function that returns (Value, KD, C) keeps one policy sequence per variable; ForwardIteration moves D_t with
 * the endogenous dimension's policy variable and aggregates EVERY variable with the same D_t:
 * agg_j[t] = dot(vec(policy_j[t]), D_t). The reference ships no two-output plugin: consumption here is the c_grid its
 * KS plugin already forms at KrusellSmith.jl:79, returned as a second policy. Parity unpinned by construction.       */
/* c_grid of every period from the savings policies: (1 + r) .* policy_a .+ (w .* labor_mat [.+ tr]) .- griddedpolicy */
void FN(orc_consumption_policy)(const orc_model *m, int P, const double *xr_, const double *xw_, const double *xt_,
                                const double *policy_seq_, double *cons_seq_) {
    const int n_a = m->n_a, n_e = m->n_e;
    const size_t G = (size_t)n_a * n_e;
    const dual *xr = (const dual *)xr_, *xw = (const dual *)xw_, *xt = (const dual *)xt_;
    const dual *pol = (const dual *)policy_seq_;
    dual *cons = (dual *)cons_seq_;
    for (int t = 0; t < P; t++) {
        dual one_plus_r = d_add_r(xr[t], 1.0);
        for (int e = 0; e < n_e; e++) {
            dual inc = d_mul_r(xw[t], m->z[e]);
            if (xt) inc = d_add(inc, xt[t]);
            for (int ia = 0; ia < n_a; ia++)
                cons[(size_t)t * G + e * n_a + ia] = d_sub(d_add(d_mul_r(one_plus_r, m->a[ia]), inc), pol[(size_t)t * G + e * n_a + ia]);
        }
    }
}
/* policy_seqs_: n_het sequences of P x G duals, the endogenous dimension's policy variable first; agg_: [n_het][P] duals */
void FN(orc_forward_iteration_het)(const orc_model *m, int P, int n_het, const double *policy_seqs_,
                                   const double *ss_init_D, double *agg_) {
    const int G = m->n_a * m->n_e;
    const dual *seqs = (const dual *)policy_seqs_;
    dual *agg = (dual *)agg_;
    dual *D = (dual *)malloc(sizeof(dual) * (size_t)G);
    dual *Dn = (dual *)malloc(sizeof(dual) * (size_t)G);
    for (int i = 0; i < G; i++) D[i] = d_const(ss_init_D[i]);
    for (int t = 0; t < P; t++) {
        FN(orc_transition_step)(m, (const double *)(seqs + (size_t)t * G), (const double *)D, (double *)Dn);
        dual *tmp = D;
        D = Dn;
        Dn = tmp;
        for (int j = 0; j < n_het; j++) {
            const dual *pol = seqs + ((size_t)j * P + t) * G;
            dual s = d_const(0.0);
            for (int i = 0; i < G; i++) s = d_add(s, d_mul(pol[i], D[i]));
            agg[(size_t)j * P + t] = s;
        }
    }
    free(D);
    free(Dn);
}

/* ---- household block: BackwardIteration -> ForwardIteration (NewtonRaphson.jl:78-79) ------ */
int FN(orc_household_block)(const orc_model *m, int P, const double *xr, const double *xw,
                            const double *ss_end_value, const double *ss_init_D, double *agg,
                            double *policy_seq_out) {
    const size_t G = (size_t)m->n_a * m->n_e;
    double *pol = policy_seq_out ? policy_seq_out : (double *)malloc(sizeof(dual) * G * (size_t)P);
    int st = FN(orc_backward_iteration)(m, P, xr, xw, ss_end_value, pol);
    FN(orc_forward_iteration)(m, P, pol, ss_init_D, agg, NULL);
    if (!policy_seq_out) free(pol);
    return st;
}

/* household block of the one-asset HANK family: dual paths r_t, w_t, tr_t */
int FN(orc_household_block_tr)(const orc_model *m, int P, const double *xr, const double *xw, const double *xt,
                               const double *ss_end_value, const double *ss_init_D, double *agg,
                               double *policy_seq_out) {
    const size_t G = (size_t)m->n_a * m->n_e;
    double *pol = policy_seq_out ? policy_seq_out : (double *)malloc(sizeof(dual) * G * (size_t)P);
    int st = FN(backward_iteration_impl)(m, P, xr, xw, xt, ss_end_value, pol);
    FN(orc_forward_iteration)(m, P, pol, ss_init_D, agg, NULL);
    if (!policy_seq_out) free(pol);
    return st;
}

/* ---- Krusell-Smith fullFunction (NewtonRaphson.jl:77-83) ---------------------------------
 * x: dual n_endog x P column-major, rows (Y, KS, r, w) (KrusellSmith.yaml:67-77);
 * assemble_full_xMat (GeneralStructures.jl:329-377) with max_lag = 1, max_lead = 0 and the
 * four compiled equations (KrusellSmith.yaml:90-94, ModelParser.jl:217-259):
 *   F1 = Y - Z*KS(-1)^α ; F2 = (r+δ) - α*Z*KS(-1)^(α-1) ; F3 = w - (1-α)*Z*KS(-1)^α ; F4 = KS - KD
 * n-ary products are folded left (ModelParser.jl:96-105). KS(-1) at t=1 is ss_start.KS
 * (padded column, GeneralStructures.jl:350-354; shift_lag :441-443).
 * out: dual n_eq x P column-major (all equations at t=1, then t=2, ..., Aggregation.jl:14-15). */
int FN(orc_ks_full_function)(const orc_model *m, int P, double alpha, double delta,
                             const double *x_, const double *Z, double KS_ss_start,
                             const double *ss_end_value, const double *ss_init_D, double *out_,
                             double *agg_out_) {
    const dual *x = (const dual *)x_;
    dual *out = (dual *)out_;
    dual *xr = (dual *)malloc(sizeof(dual) * (size_t)P);
    dual *xw = (dual *)malloc(sizeof(dual) * (size_t)P);
    dual *agg = (dual *)malloc(sizeof(dual) * (size_t)P);
    for (int t = 0; t < P; t++) {
        xr[t] = x[2 + 4 * t];
        xw[t] = x[3 + 4 * t];
    }
    int st = FN(orc_household_block)(m, P, (const double *)xr, (const double *)xw, ss_end_value,
                                     ss_init_D, (double *)agg, NULL);
    int pst = ORC_OK;
    for (int t = 0; t < P; t++) {
        dual Y = x[0 + 4 * t], KS = x[1 + 4 * t], r = x[2 + 4 * t], w = x[3 + 4 * t];
        dual KSlag = (t == 0) ? d_const(KS_ss_start) : x[1 + 4 * (t - 1)];
        dual KD = agg[t];
        double Zt = Z[t];
        /* Y .- (Z .* (KS(-1) .^ α))  with Z Float64: Real*Dual */
        dual ka = d_pow_r(KSlag, alpha, &pst);
        out[0 + 4 * t] = d_sub(Y, d_mul_r(ka, Zt));
        /* (r .+ δ) .- ((α .* Z) .* KS(-1).^(α-1)) : n-ary * folded left */
        dual ka1 = d_pow_r(KSlag, alpha - 1.0, &pst);
        out[1 + 4 * t] = d_sub(d_add_r(r, delta), d_mul_r(ka1, alpha * Zt));
        /* w .- (((1-α) .* Z) .* KS(-1).^α) */
        out[2 + 4 * t] = d_sub(w, d_mul_r(ka, (1.0 - alpha) * Zt));
        /* KS .- KD */
        out[3 + 4 * t] = d_sub(KS, KD);
    }
    if (agg_out_) memcpy(agg_out_, agg, sizeof(dual) * (size_t)P);
    free(xr);
    free(xw);
    free(agg);
    return st != ORC_OK ? st : pst;
}

int FN(orc_np)(void) { return NP; }
