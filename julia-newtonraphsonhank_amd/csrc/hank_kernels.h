// hank_kernels.h — CDNA4 (gfx950) kernels of the household block. fp64 throughout, no MFMA:
// the path is interpolation / 2-nnz SpMV / reductions (HBM-bound), the only contraction is the
// n_e x n_e productivity mixing.
//
// Data layout in HBM (per context, P = T-1 periods, G = n_a*n_e, wealth fastest):
//   primal record  [P][n_e][n_a] : s (EGM knots), kc (d s/d E), ib/A/B (bracket + tangent weights),
//                                  u, v (marginal-value coefficients), pol (savings policy),
//                                  lo/lw/ig (Young lottery), start (segment offsets), Dseq[P+1]
//   tangent state  [n_e][n_a][N] : tangent index FASTEST (the AoS layout of Dual{T,V,N}): one grid
//                                  point's N partials are one contiguous row, so the bracket gather
//                                  and the lottery segment sums move whole coalesced rows.
//   dpol           [P][n_e][n_a][N] : the policy-partials sequence BackwardIteration materialises
//                                  (BackwardIteration.jl:88, :110-112); written once by the
//                                  backward sweep, read once by the forward sweep = the
//                                  algorithmic HBM traffic of a JVP batch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hank {

constexpr int RBP = 32;        // rows (wealth points) per block in the primal kernels

struct Consts {
    int n_a, n_e, G, P;
    int n_hh;   // household inputs per period: 2 = (r, w) Krusell-Smith; 3 = (r, w, tr) one-asset HANK (lump-sum transfer)
    int diet;   // the tangent sweeps rebuild kc from s and v from u instead of reading them (CRRA with gamma = 1 or 2: see diet_kc)
    double beta, gamma, bc;
    const double *a, *z, *Pi;  // device
};

struct Record {
    double *s, *kc, *A, *B, *u, *v, *pol, *lw, *ig, *Dseq;
    int *ib, *lo, *start, *clo;
    int4 *seg;      // per target row: {start[r-1], start[r], start[r+1], -} — ONE read instead of three
    double2 *lwg;   // per source row: {lottery weight w, weight-tangent factor ig * D_{t-1}} — ONE 16-byte read per source in the forward tangent kernel
};

enum { ERR_KNOTS = 3, ERR_DOMAIN = 4, ERR_NONMONO = 6 };

__device__ inline void set_err(int *err, int code, int t, int e, int j) {
    if (atomicCAS(&err[0], 0, code) == 0) {
        err[1] = t;
        err[2] = e;
        err[3] = j;
    }
}

__device__ inline bool pow_domain_error(double v, double y) { return (v < 0.0) && (y != floor(y)); }

// Reciprocal and reciprocal square root of the primal EGM step: the hardware estimate (v_rcp_f64 / v_rsq_f64) and two Newton
// steps, without the div_scale / div_fmas / div_fixup wrapper of an IEEE division (~18 instructions) or the scaling of an IEEE
// square root. The operands — a marginal value, a consumption level, the distance of two sorted knots — are positive, finite and
// far from the denormal range wherever the step does not report ERR_KNOTS / ERR_DOMAIN anyway (zero or negative arguments give
// inf / NaN as the IEEE forms do); the results are within an ulp or two of the correctly rounded ones (oracle tolerance: 1e-10).
// A row of the Dual pass's backward half does four divisions and two roots: ~110 of its ~500 instructions per wave and period,
// on the critical path of a SIMD's three waves (the staircase at the tile barrier). EVERY primal
// kernel takes its quotients and roots from here (egm_X / egm_Y / diet_v are shared), so k_egm_step, k_xprimal_back, k_xdual_back
// and k_xvfi still agree with each other bit for bit.
#ifndef HANK_IEEE_DIV
#define HANK_IEEE_DIV 0     // dev knob: 1 = the IEEE forms (A/B, profiles/r05c_fast_div.log)
#endif
__device__ __forceinline__ double fast_rcp(double x) {
#if HANK_IEEE_DIV
    return 1.0 / x;
#else
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
#endif
}
__device__ __forceinline__ double fast_rsqrt(double x) {
#if HANK_IEEE_DIV
    return 1.0 / sqrt(x);
#else
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(y * 0.5, __builtin_fma(-(x * y), y, 1.0), y);
    y = __builtin_fma(y * 0.5, __builtin_fma(-(x * y), y, 1.0), y);
    return y;
#endif
}
__device__ __forceinline__ double fast_sqrt(double x) {     // x rsqrt(x), one correction of the product
#if HANK_IEEE_DIV
    return sqrt(x);
#else
    const double y = fast_rsqrt(x), g = x * y;
    return __builtin_fma(y * 0.5, __builtin_fma(-g, g, x), g);
#endif
}

// x^y with the exponents CRRA utility produces most often taken through a reciprocal / reciprocal square root
// (gamma = 2: y = -1/2 and y = -2; gamma = 1: y = -1) — a generic fp64 pow is a few
// hundred VALU instructions on the primal sweep's critical path. Same value as pow() to an ulp or two.
__device__ inline double pow_crra(double x, double y) {
    if (y == -0.5) return fast_rsqrt(x);
    if (y == -2.0) return fast_rcp(x * x);
    if (y == -1.0) return fast_rcp(x);
    return pow(x, y);
}

// ---- block reductions (64-wide wavefronts) ------------------------------------------------
__device__ inline double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}
// sum over the first `nthreads` threads of the block (the others must have left the kernel);
// result valid in thread 0; red must hold >= 16 doubles
__device__ inline double block_sum(double x, double *red, int nthreads) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nw = (nthreads + 63) >> 6;
    x = wave_sum(x);
    __syncthreads();
    if (lane == 0) red[wv] = x;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < nw; k++) r += red[k];
    return r;
}

// Workgroup barrier for LDS hand-offs: waits for this wave's LDS operations only. __syncthreads() carries a fence that
// also drains vmcnt — every write-through (sc1) store issued before it would have to reach memory before the barrier
// opens, although nothing behind these barriers reads global memory another thread of the block wrote (dev knob
// HANK_RAW_BARRIER=0 restores __syncthreads()).
#ifndef HANK_RAW_BARRIER
#define HANK_RAW_BARRIER 1
#endif
__device__ __forceinline__ void lds_barrier() {
#if HANK_RAW_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// dev knob: how the big streaming stores leave the CU (0 plain, 1 sc1 write-through, 2 nontemporal)
#ifndef HANK_ST_DPOL
#define HANK_ST_DPOL 1
#endif
#ifndef HANK_ST_STATE
#define HANK_ST_STATE 1
#endif
#ifndef HANK_ST_REC
#define HANK_ST_REC 0
#endif
template <int MODE>
__device__ __forceinline__ void st_mode(double *p, double x) {
    if (MODE == 1) __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) __builtin_nontemporal_store(x, p);
    else *p = x;
}
template <int MODE>
__device__ __forceinline__ void st_mode_i(int *p, int x) {
    if (MODE == 1) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) __builtin_nontemporal_store(x, p);
    else *p = x;
}

// acc += sum_{k=k0}^{n-1} P[k*ps] * V[k*vs], added in index order. The n_e-long mixing sums sit on every launch's
// critical path; as a plain loop each LDS read waits for the one before (n_e round trips of ~100 clocks: the launch
// floor is 4.1 us at n_e = 4 and 6.2 us at n_e = 11). Reads are issued four columns at a time, the order of the
// additions is unchanged.
__device__ __forceinline__ double mix_mul(double p, double v) { return p * v; }
__device__ __forceinline__ double2 mix_mul(double p, double2 v) { return make_double2(p * v.x, p * v.y); }
__device__ __forceinline__ double mix_add(double a, double b) { return a + b; }
__device__ __forceinline__ double2 mix_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
template <typename T>
__device__ __forceinline__ T mix_sum(T acc, const T *V, int vs, const double *P, int ps, int k0, int n) {
    int k = k0;
    for (; k + 3 < n; k += 4) {
        const T v0 = V[k * vs], v1 = V[(k + 1) * vs], v2 = V[(k + 2) * vs], v3 = V[(k + 3) * vs];
        const double p0 = P[k * ps], p1 = P[(k + 1) * ps], p2 = P[(k + 2) * ps], p3 = P[(k + 3) * ps];
        acc = mix_add(acc, mix_mul(p0, v0));
        acc = mix_add(acc, mix_mul(p1, v1));
        acc = mix_add(acc, mix_mul(p2, v2));
        acc = mix_add(acc, mix_mul(p3, v3));
    }
    if (k + 1 < n) {
        const T v0 = V[k * vs], v1 = V[(k + 1) * vs];
        const double p0 = P[k * ps], p1 = P[(k + 1) * ps];
        acc = mix_add(acc, mix_mul(p0, v0));
        acc = mix_add(acc, mix_mul(p1, v1));
        k += 2;
    }
    if (k < n) acc = mix_add(acc, mix_mul(P[k * ps], V[k * vs]));
    return acc;
}

// RECORD DIET. Two of the seven coefficients a backward tangent step reads per point are functions of their neighbours in the
// record: s = rho ((cm - (w z_e + tr)) + a) gives back cm, and kc = rho beta (-1/gamma) cm^(1 + gamma); u = cg^(-gamma) gives back
// cg, and v = (1 + r)(-gamma) u / cg. With gamma = 1 or 2 the powers are products and one square root: 16 of the 52 bytes per
// point and period are not read (by each of the 8 groups of a persistent sweep). Other gammas keep reading kc and v (a pow per
// point, period and group costs more than the 16 bytes). Both implementations use these two functions, so their policy partials
// stay bit-identical: the PRIMAL sweeps record kc and v as these two functions give them, the launch-path tangent kernels (whose
// lane-sparse coefficient loads are off their critical path: rebuilding cost them 1-3 %) read the record, the persistent
// tangent sweeps (8 groups re-read the record) rebuild: the same bits either way. Against the textbook expressions the values
// move by rounding (cm is rebuilt through a cancellation: 1e-13).
__device__ __forceinline__ double diet_kc(const Consts &c, double s, double rho, double opr, double wz_tr, double xa) {
    const double cm = (s * opr + wz_tr) - xa;
    const double k = rho * (c.beta * (-1.0 / c.gamma));
    return c.gamma == 2.0 ? k * (cm * cm * cm) : k * (cm * cm);
}
__device__ __forceinline__ double diet_v(const Consts &c, double u, double opr) {
    return c.gamma == 2.0 ? opr * (-2.0 * (u * fast_sqrt(u))) : opr * (-(u * u));
}

// household inputs of period t: xhh[n_hh*t + k]; the lump-sum transfer is 0 for families without one
__device__ __forceinline__ double hh_tr(const Consts &c, const double *xhh, int t) { return c.n_hh > 2 ? xhh[c.n_hh * t + 2] : 0.0; }

// ---- EGM step, split in the two halves that fuse across the period boundary ------------------
// X half (KrusellSmith.jl:59-62): from V_{t+1} (all e2 of this row, in LDS) to the endogenous
// knot s_t[a,e] and kc = d s / d E  (= rho * d c / d E).
__device__ inline void egm_X(const Consts &c, const double *Vsh, const double *Pish, int row, int a,
                             int e, double r, double w, double tr, double *s_out, double *kc_out, int *err,
                             int t) {
    const double E = mix_sum(Vsh[row] * Pish[e], Vsh + row, RBP, Pish + e, c.n_e, 1, c.n_e);
    const double bE = E * c.beta;
    const double ex = -1.0 / c.gamma;
    if (pow_domain_error(bE, ex)) set_err(err, ERR_DOMAIN, t, e, a);
    const double cm = pow_crra(bE, ex);
    const double rho = 1.0 / (1.0 + r);
    const double s1 = rho * ((cm - (w * c.z[e] + tr)) + c.a[a]);
    st_mode<HANK_ST_REC>(s_out, s1);
    // d cmat/dE = beta*ex*(bE)^(ex-1)  (Dual^Real, ForwardDiff dual.jl:563-572); under the record diet the recorded value IS the
    // one the persistent tangent sweeps rebuild from s (diet_kc): whoever reads it and whoever rebuilds it hold the same bits
    st_mode<HANK_ST_REC>(kc_out, c.diet ? diet_kc(c, s1, rho, 1.0 + r, w * c.z[e] + tr, c.a[a]) : rho * (c.beta * ex * (cm / bE)));
}

// Y half (KrusellSmith.jl:66-80): interpolate the policy on the exogenous grid point a from the
// knots of column e (sc = s_t[:,e]), apply the borrowing constraint, form the marginal value.
// Interpolations.jl Gridded(Linear()) + Flat(): bracket = last knot <= x, flat outside.
struct YOut {
    double g, A, B, u, v, V;
    int ib;
};
// `guess` (>= 0): the bracket of the same point one period later — brackets move by a few knots per
// period, so a probe there plus a short gallop replaces the 11 dependent loads of a cold bisection.
// `sc` is anything indexable that returns the knot s_t[i, e] (a plain pointer; the XCD-local sweep passes a
// loader that reads the L2-resident state with L1-bypassing loads) — the arithmetic is the same for both.
template <typename KNOTS>
__device__ inline YOut egm_Y(const Consts &c, const KNOTS sc, int a, int e, double r, double w, double tr,
                             int *err, int t, int guess) {
    YOut o;
    const int n = c.n_a;
    const double x = c.a[a];
    const double sa = sc[a];
    // Interpolations' check_gridded: knots must be sorted and unique (values)
    if (a > 0 ? !(sa > sc[a - 1]) : !(sa == sa)) set_err(err, ERR_KNOTS, t, e, a);
    const double s0 = sc[0], sN = sc[n - 1];
    double g, A = 0.0, B = 0.0;
    int i;
    if (x < s0) {
        g = c.a[0];
        i = 0;
    } else if (x > sN) {
        g = c.a[n - 1];
        i = n - 2;
    } else {
        int lo = -1, hi = n;  // sc[lo] <= x < sc[hi]
        bool have = false;    // vlo/vhi hold sc[lo], sc[hi] (bracket found by the first probe)
        double vlo = 0.0, vhi = 0.0;
        if (guess >= 0) {
            // probe the guessed bracket AND its two neighbours in one round trip (four independent loads): a bracket
            // that moved by one knot — the common miss — needs no second trip
            const int p = guess < n - 1 ? guess : n - 2;
            const int pm = p > 0 ? p - 1 : 0, pp = p + 2 < n ? p + 2 : n - 1;
            const double sm = sc[pm], sp = sc[p], sp1 = sc[p + 1], sp2 = sc[pp];
            if (sp <= x && x < sp1) {
                lo = p; hi = p + 1; have = true; vlo = sp; vhi = sp1;   // the usual case
            } else if (p + 2 < n && sp1 <= x && x < sp2) {
                lo = p + 1; hi = p + 2; have = true; vlo = sp1; vhi = sp2;
            } else if (p > 0 && sm <= x && x < sp) {
                lo = p - 1; hi = p; have = true; vlo = sm; vhi = sp;
            } else if (sp1 <= x) {          // gallop up
                lo = p + 1;
                for (int step = 1;; step <<= 1) {
                    const int q = lo + step;
                    if (q >= n) break;
                    if (sc[q] <= x) lo = q; else { hi = q; break; }
                }
            } else {                         // gallop down
                hi = p;
                for (int step = 1;; step <<= 1) {
                    const int q = hi - step;
                    if (q < 0) break;
                    if (sc[q] <= x) { lo = q; break; } else hi = q;
                }
            }
        }
        while (hi - lo > 1) {
            const int mid = lo + ((hi - lo) >> 1);
            if (sc[mid] <= x) lo = mid; else hi = mid;
        }
        i = lo < 0 ? 0 : (lo > n - 2 ? n - 2 : lo);
        const double si = have ? vlo : sc[i], sj = have ? vhi : sc[i + 1];
        const double h = sj - si, rh = fast_rcp(h);      // (one reciprocal serves both quotients)
        const double f = (x - si) * rh;
        const double ai = c.a[i], aj = c.a[i + 1];
        g = (1.0 - f) * ai + f * aj;
        const double sl = (aj - ai) * rh;
        A = -sl * (1.0 - f);
        B = -sl * f;
    }
    // max(g, borrow_cons), DiffRules rule: partial passes unless (bc > g) | signbit(bc) < signbit(g)
    const double bc = c.bc;
    if ((bc > g) || ((int)signbit(bc) < (int)signbit(g))) {
        A = 0.0;
        B = 0.0;
    }
    g = (g > bc) ? g : ((bc > g) ? bc : (signbit(g) ? bc : g));
    const double opr = 1.0 + r;
    const double cg = (opr * x + (w * c.z[e] + tr)) - g;
    if (pow_domain_error(cg, -c.gamma)) set_err(err, ERR_DOMAIN, t, e, a);
    const double u = pow_crra(cg, -c.gamma);
    o.g = g;
    o.A = A;
    o.B = B;
    o.ib = i;
    o.u = u;
    o.v = c.diet ? diet_v(c, u, opr) : opr * ((-c.gamma) * (u / cg));
    o.V = opr * u;
    return o;
}

// block = RBP rows x n_e columns; thread (row, e) with row fastest. dynamic LDS:
// Vsh[n_e*RBP] + Pish[n_e*n_e]
// `stop` (may be null): the device-resident value-function iteration freezes its state once it has converged
__global__ void k_egm_X(Consts c, const double *Vin, const double *xt, double *s_out,
                        double *kc_out, int *err, int t, const int *stop) {
    extern __shared__ double sh[];
    if (stop && *stop) return;
    double *Vsh = sh, *Pish = sh + c.n_e * RBP;
    const int row = threadIdx.x % RBP, e = threadIdx.x / RBP;
    const int a = blockIdx.x * RBP + row;
    for (int k = threadIdx.x; k < c.n_e * c.n_e; k += blockDim.x) Pish[k] = c.Pi[k];
    if (a < c.n_a) Vsh[e * RBP + row] = Vin[e * c.n_a + a];
    __syncthreads();
    if (a < c.n_a) egm_X(c, Vsh, Pish, row, a, e, xt[0], xt[1], c.n_hh > 2 ? xt[2] : 0.0, &s_out[e * c.n_a + a], &kc_out[e * c.n_a + a], err, t);
}

// Y only (granular step): record slot pointers are for ONE period
__global__ void k_egm_Y(Consts c, const double *s, double r, double w, double tr, double *pol, int *ib,
                        double *A, double *B, double *u, double *v, double *Vout, int *err, int t, const int *stop) {
    if (stop && *stop) return;
    const int row = threadIdx.x % RBP, e = threadIdx.x / RBP;
    const int a = blockIdx.x * RBP + row;
    if (a >= c.n_a) return;
    const YOut o = egm_Y(c, s + e * c.n_a, a, e, r, w, tr, err, t, -1);
    const int off = e * c.n_a + a;
    pol[off] = o.g; ib[off] = o.ib; A[off] = o.A; B[off] = o.B; u[off] = o.u; v[off] = o.v;
    Vout[off] = o.V;
}

// fused sweep step: Y of period t, then X of period t-1 (value never leaves the CU).
// Body shared by k_egm_step and the dual-sweep kernel k_fused_back; runs on the first RBP*n_e
// threads of block `bid`; sh = Vsh[n_e*RBP] + Pish[n_e*n_e].
__device__ inline void egm_step_body(const Consts &c, const Record &R, const double *xhh, int t, int *err, int bid, double *sh) {
    double *Vsh = sh, *Pish = sh + c.n_e * RBP;
    const int nthr = RBP * c.n_e;
    const int row = threadIdx.x % RBP, e = threadIdx.x / RBP;
    const int a = bid * RBP + row;
    for (int k = threadIdx.x; k < c.n_e * c.n_e; k += nthr) Pish[k] = c.Pi[k];
    const size_t base = (size_t)t * c.G;
    if (a < c.n_a) {
        const double r = xhh[c.n_hh * t], w = xhh[c.n_hh * t + 1], tr = hh_tr(c, xhh, t);
        const int guess = (t + 1 < c.P) ? R.ib[base + c.G + (size_t)e * c.n_a + a] : -1;
        const YOut o = egm_Y(c, R.s + base + (size_t)e * c.n_a, a, e, r, w, tr, err, t, guess);
        const size_t off = base + (size_t)e * c.n_a + a;
        st_mode<HANK_ST_REC>(&R.pol[off], o.g); st_mode_i<HANK_ST_REC>(&R.ib[off], o.ib); st_mode<HANK_ST_REC>(&R.A[off], o.A);
        st_mode<HANK_ST_REC>(&R.B[off], o.B); st_mode<HANK_ST_REC>(&R.u[off], o.u); st_mode<HANK_ST_REC>(&R.v[off], o.v);
        Vsh[e * RBP + row] = o.V;
    }
    lds_barrier();
    if (t > 0 && a < c.n_a) {
        const double r1 = xhh[c.n_hh * (t - 1)], w1 = xhh[c.n_hh * (t - 1) + 1], tr1 = hh_tr(c, xhh, t - 1);
        const size_t off1 = base - c.G + (size_t)e * c.n_a + a;
        egm_X(c, Vsh, Pish, row, a, e, r1, w1, tr1, &R.s[off1], &R.kc[off1], err, t - 1);
    }
}
__global__ void k_egm_step(Consts c, Record R, const double *xhh, int t, int *err) {
    extern __shared__ double sh[];
    egm_step_body(c, R, xhh, t, err, blockIdx.x, sh);
}

// ---- Young lottery for every (period, column) at once (ForwardIteration.jl:37-78) ------------
// one block per column; builds lo / lw / ig and the segment offsets that turn the 2-nnz-per-column
// scatter into a deterministic gather (the policy is monotone in wealth):
//   clo      = number of sources clamped at the FIRST grid point (policy <= grid[1], :54-58): a
//              prefix [0, clo) of the column, all mass to row 0 with weight one and no weight
//              tangent — the borrowing-constraint mass point, summed separately (it is long);
//   start[r] = max(clo, first source j with lo_j >= r): target row r receives w_j from the sources
//              in [start[r-1], start[r]) and 1-w_j from those in [start[r], start[r+1]).
__global__ void k_lottery(Consts c, Record R, int ncols, int *err, int write_seg, int use_ib) {
    extern __shared__ int shlo[];
    __shared__ int sh_clo;
    const int col = blockIdx.x;
    if (col >= ncols) return;
    const int t = col / c.n_e, e = col % c.n_e;
    const size_t base = (size_t)col * c.n_a;  // == t*G + e*n_a
    const int n = c.n_a;
    if (threadIdx.x == 0) sh_clo = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double p = R.pol[base + j];
        int lo = -1, hi = n;  // grid[lo] < p <= grid[hi]  (searchsortedfirst, :52)
        if (use_ib) {
            // the policy is the interpolation's convex combination of grid[ib] and grid[ib + 1] (egm_Y): its lottery bracket is the
            // interpolation's unless it sits on a grid point or was moved by the borrowing constraint — a probe of that bracket and
            // of the one below replaces the 11 dependent loads of the bisection (which still decides whatever the probe does not)
            const int g0 = min(max(R.ib[base + j], 0), n - 2);
            const double a0 = c.a[g0], a1 = c.a[g0 + 1];
            if (a0 < p && p <= a1) { lo = g0; hi = g0 + 1; }
            else if (g0 > 0 && p <= a0 && c.a[g0 - 1] < p) { lo = g0 - 1; hi = g0; }
        }
        while (hi - lo > 1) {
            const int mid = lo + ((hi - lo) >> 1);
            if (c.a[mid] < p) lo = mid; else hi = mid;
        }
        int l;
        double w, ig;
        if (hi == 0) {            // m == 1: all mass on the first point (:54-58)
            l = 0; w = 0.0; ig = 0.0;
            atomicMax(&sh_clo, j + 1);
        } else if (hi >= n) {     // m > n_a: all mass on the last point (:59-63)
            l = n - 2; w = 1.0; ig = 0.0;
        } else {                  // interior (:64-73)
            l = hi - 1;
            const double gap = c.a[hi] - c.a[l];
            w = (p - c.a[l]) / gap;
            ig = 1.0 / gap;
        }
        R.lo[base + j] = l; R.lw[base + j] = w; R.ig[base + j] = ig;
        shlo[j] = (hi == 0) ? -1 : l;   // clamped-low sources sort before every interior bracket
    }
    __syncthreads();
    const int clo = sh_clo;
    if (threadIdx.x == 0) R.clo[col] = clo;
    int *st = R.start + (size_t)col * (n + 1);
    int *shst = shlo + n;          // segment offsets staged in LDS (second half of the dynamic LDS)
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const int prev = j ? shlo[j - 1] : -1, cur = shlo[j];
        if (cur < prev) set_err(err, ERR_NONMONO, t, e, j);
        for (int r = (prev < 0 ? 0 : prev + 1); r <= cur; r++) shst[r] = j;   // j >= clo whenever cur >= 0
        if (j == n - 1)
            for (int r = (cur < 0 ? 0 : cur + 1); r <= n; r++) shst[r] = n;
    }
    __syncthreads();
    for (int r = threadIdx.x; r <= n; r += blockDim.x) st[r] = shst[r];
    // per target row: its three segment bounds in one 16-byte record — what the launch family's gather reads (and k_xstat, and
    // k_fn_impulse); a third of this kernel's bytes, and the persistent Dual pass reads none of them: written there on demand
    // (write_seg = 0, then k_seg_build before the first reader)
    if (write_seg)
        for (int r = threadIdx.x; r < n; r += blockDim.x)
            R.seg[base + r] = make_int4(r > 0 ? shst[r - 1] : shst[r], shst[r], shst[r + 1], 0);
}
// the same records from the segment offsets k_lottery left (one block per column)
__global__ void k_seg_build(Consts c, Record R, int ncols) {
    const int col = blockIdx.x, n = c.n_a;
    if (col >= ncols) return;
    const int *st = R.start + (size_t)col * (n + 1);
    const size_t base = (size_t)col * n;
    for (int r = threadIdx.x; r < n; r += blockDim.x)
        R.seg[base + r] = make_int4(r > 0 ? st[r - 1] : st[r], st[r], st[r + 1], 0);
}

// ---- distribution push-forward, one period (ForwardIteration.jl:95-99, :297-308) -------------
// block = RBP rows x n_e, thread (row, e) with row fastest (a 32-lane half-wave per column);
// dynamic LDS: Dsh[n_e*RBP] + Pish[n_e*n_e] + red[16]
// body shared by k_dist_step and k_fused_fwd; first RBP*n_e threads of block `bid` of `nblocks`
// Din / Dout (may be null): explicit D_{t-1} and D_t buffers instead of the record's Dseq rows — the stationary-distribution
// iteration of the steady state reuses this step with one period's lottery and two ping-pong buffers
__device__ inline void dist_step_body(const Consts &c, const Record &R, int t, double *aggpart, int bid, int nblocks, double *sh,
                                      const double *Din = nullptr, double *Dout = nullptr) {
    double *Dsh = sh, *Pish = sh + c.n_e * RBP, *red = Pish + c.n_e * c.n_e;
    const int nthr = RBP * c.n_e;
    const int row = threadIdx.x % RBP, e = threadIdx.x / RBP;
    const int r = bid * RBP + row;
    const int n = c.n_a;
    for (int k = threadIdx.x; k < c.n_e * c.n_e; k += nthr) Pish[k] = c.Pi[k];
    const double *Dprev = Din ? Din : R.Dseq + (size_t)t * c.G;
    const size_t base = (size_t)t * c.G;
    const double *lw = R.lw + base + (size_t)e * n, *Dp = Dprev + (size_t)e * n;
    double acc = 0.0;
    if (r < n) {
        if (!Din) R.lwg[base + (size_t)e * n + r] = make_double2(lw[r], R.ig[base + (size_t)e * n + r] * Dp[r]);
        const int *st = R.start + ((size_t)t * c.n_e + e) * (n + 1);
        const int st1 = st[r], st2 = st[r + 1], st0 = r > 0 ? st[r - 1] : st1;
        for (int j = st0; j < st1; j++) acc += lw[j] * Dp[j];
        for (int j = st1; j < st2; j++) acc += (1.0 - lw[j]) * Dp[j];
    }
    if (bid == 0) {  // the mass point: sum_{j < clo} D_prev[j] -> row 0, by the column's 32 lanes
        const int clo = R.clo[(size_t)t * c.n_e + e];
        double part = 0.0;
        for (int j = row; j < clo; j += RBP) part += Dp[j];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (row == 0) acc += part;
    }
    if (r < n) Dsh[e * RBP + row] = acc;
    lds_barrier();
    // two aggregates of the POST-transition distribution (ForwardIteration.jl:301-307): the policy-weighted one (the heterogeneous
    // variable itself) and the wealth-GRID-weighted one, sum_pt a(pt) D_t(pt) — with it every heterogeneous output that is affine
    // in the policy, the state and the household inputs (consumption, cash on hand: BackwardIteration.jl:99-112 routes every key
    // of the plugin's NamedTuple) is assembled on the host. aggpart: [t][block][2]
    double part = 0.0, part2 = 0.0;
    if (r < n) {
        const int e2 = e;  // D_new[r,e2] = sum_e D_mid[r,e] * Pi[e,e2]
        double Dn = 0.0;
        Dn = mix_sum(Dn, Dsh + row, RBP, Pish + c.n_e * e2, 1, 0, c.n_e);
        st_mode<HANK_ST_REC>(Dout ? &Dout[(size_t)e2 * n + r] : &R.Dseq[(size_t)(t + 1) * c.G + (size_t)e2 * n + r], Dn);
        part = R.pol[base + (size_t)e2 * n + r] * Dn;
        part2 = c.a[r] * Dn;
    }
    const double tot = block_sum(part, red, nthr);
    const double tot2 = block_sum(part2, red, nthr);
    if (threadIdx.x == 0) { aggpart[((size_t)t * nblocks + bid) * 2] = tot; aggpart[((size_t)t * nblocks + bid) * 2 + 1] = tot2; }
}
__global__ void k_dist_step(Consts c, Record R, int t, double *aggpart) {
    extern __shared__ double sh[];
    dist_step_body(c, R, t, aggpart, blockIdx.x, gridDim.x, sh);
}
// one application of the (fixed) transition of record period 0 to an explicit distribution: D_out = Lambda D_in
__global__ void k_dist_iter(Consts c, Record R, const double *Din, double *Dout, double *aggpart, const int *stop) {
    extern __shared__ double sh[];
    if (stop && *stop) return;
    dist_step_body(c, R, 0, aggpart, blockIdx.x, gridDim.x, sh, Din, Dout);
}

// one convergence check of the steady state's inner fixed point (SteadyState.jl:136-139: max|value_new - value| < tol,
// after EVERY step): one block; state = {stop, steps done}. A NaN norm never stops the loop, as in the reference.
__global__ void k_vfi_check(const double *Vnew, const double *Vold, int G, double tol, int *state, double *norm_out) {
    __shared__ double red[16];
    if (state[0]) return;
    double m = 0.0;
    bool bad = false;
    for (int i = threadIdx.x; i < G; i += blockDim.x) {
        const double d = fabs(Vnew[i] - Vold[i]);
        if (!(d == d)) bad = true;
        m = d > m ? d : m;
    }
    if (bad) m = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(m, off, 64); m = (o > m || !(o == o)) ? o : m; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); k++) m = (red[k] > m || !(red[k] == red[k])) ? red[k] : m;
        state[1] += 1;
        *norm_out = m;
        if (m < tol) state[0] = 1;
    }
}

// zero fills as kernels (no memset nodes inside the captured graphs)
__global__ void k_zero_i32(int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
__global__ void k_zero_f64(double *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

// out[t*N + n] = sum_b parts[(t*nb + b)*N + n]; one block per (t, 64-wide tangent chunk), four
// groups of 64 lanes stride over the row blocks, fixed combination order: bitwise reproducible
__global__ void k_reduce_parts(const double *__restrict__ parts, int nb, int N, double *__restrict__ out) {
    __shared__ double red[256];
    const int t = blockIdx.x, nl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + nl;
    double s = 0.0;
    if (n < N) {
        const double *p = parts + (size_t)t * nb * N + n;
        int b = g;
        for (; b + 28 < nb; b += 32) {      // 8 independent loads in flight per lane
            const double v0 = p[(size_t)b * N], v1 = p[(size_t)(b + 4) * N], v2 = p[(size_t)(b + 8) * N], v3 = p[(size_t)(b + 12) * N];
            const double v4 = p[(size_t)(b + 16) * N], v5 = p[(size_t)(b + 20) * N], v6 = p[(size_t)(b + 24) * N], v7 = p[(size_t)(b + 28) * N];
            s += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
        }
        for (; b < nb; b += 4) s += p[(size_t)b * N];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && n < N) out[(size_t)t * N + n] = (red[nl] + red[64 + nl]) + (red[128 + nl] + red[192 + nl]);
}

// ---- tangent sweeps ------------------------------------------------------------------------
// One wavefront per productivity column e: lane = (tangent nl fastest, wealth row rl), so a wave
// instruction moves RB = 64/NC adjacent rows x NC lanes' worth of tangents = one contiguous 512-byte
// (one direction per lane) or 1-KiB (two per lane) piece of the [e][a][N] state. The n_e x n_e mixing goes through a 64 x n_e LDS tile (conflict-free: the lane
// is the fastest LDS index). Block = 64*n_e threads.
// KV virtual rows per column (rows n_a .. n_a+KV-1 of the dD state) hold partial sums of the
// mass-point row 0 — see k_tan_fwd.

constexpr int KV = 16;
struct TanGeom { int N, NC, lgNC, nbx, ss; };   // ss: source-stationary forward kernel (see tan_fwd_body)   // nbx = regular row blocks = ceil(n_a / (64/NC))

// (n_hh, P, N) column-major  ->  dxr[P][N], dxw[P][N] (, dxt[P][N] when the family has a transfer input)
__global__ void k_tan_in(const double *__restrict__ dxhh, int n_hh, int P, int N, double *__restrict__ dxr, double *__restrict__ dxw,
                         double *__restrict__ dxt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * N) return;
    const int t = idx / N, n = idx - t * N;
    const double *x = dxhh + (size_t)n_hh * ((size_t)t + (size_t)P * n);
    dxr[idx] = x[0];
    dxw[idx] = x[1];
    if (n_hh > 2) dxt[idx] = x[2];
}
// dagg[P][N] -> (P, N) column-major
__global__ void k_tan_out(const double *__restrict__ dagg, int P, int N, double *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * N) return;
    const int n = idx / P, t = idx - n * P;
    out[idx] = dagg[(size_t)t * N + n];
}

// one backward period: Y-tangent of period t (bracket gather -> dpol_t, dV_t), then X-tangent of
// period t-1 (mix over e -> knot tangents ds_{t-1}). `first`: dV_{t+1} = 0 for the last period
// (terminal value has zero partials, BackwardIteration.jl:85) => only the X half, from zeros.
// ds ping-pongs between two [e][a][N] buffers that live in L2 / Infinity Cache.
// RG = row groups per wave: a wave walks RG groups of RB = 64/NC rows with all their loads in
// flight together (the host picks RG per batch width: tan_rg in hank_hip.hip).
// Tangent lanes are templated on VT = double (one direction per lane) or double2 (two adjacent directions per
// lane: every state / dpol access is a 16-byte one, half the vector-memory instructions per byte; the memory
// layout [..][N] is the same, N even). g.N / g.NC count VT elements.
__device__ __forceinline__ void vzero(double &v) { v = 0.0; }
__device__ __forceinline__ void vzero(double2 &v) { v.x = 0.0; v.y = 0.0; }
__device__ __forceinline__ double vshfl_xor(double v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ double2 vshfl_xor(double2 v, int m) { return make_double2(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64)); }
template <int MODE>
__device__ __forceinline__ void st_mode(double2 *p, double2 v) {
    if (MODE == 1) {
        // one 16-byte write-through store (there is no 16-byte atomic store to spell it with); the trailing
        // s_nop keeps the compiler's next instruction off the data registers until the store has read them
        typedef double d2_t __attribute__((ext_vector_type(2)));
        d2_t d = {v.x, v.y};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    } else if (MODE == 2) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    } else {
        *p = v;
    }
}
__device__ __forceinline__ void lds_add(double *p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_add(double2 *p, double2 v) { lds_add(&p->x, v.x); lds_add(&p->y, v.y); }
__device__ __forceinline__ double2 vmul(double a, double2 v) { return make_double2(a * v.x, a * v.y); }
__device__ __forceinline__ double vmul(double a, double v) { return a * v; }
__device__ __forceinline__ double2 vadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double vadd(double a, double b) { return a + b; }
__device__ __forceinline__ double2 vsub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double vsub(double a, double b) { return a - b; }

// XCD-aware block order. The dispatcher deals workgroups to the 8 XCDs round-robin (block p lands on XCD p mod 8), each XCD
// has its own L2, and neighbouring row blocks gather from overlapping rows: with the natural order every XCD touches every
// part of the state and a line is fetched by up to 8 L2s. Here the blocks that share an XCD own CONTIGUOUS row ranges
// (XCD x: logical blocks [start_x, start_x + count_x)), so a state line is fetched once, twice at the 7 seams. A
// bijection on [0, nb): nothing depends on where a block really runs.
// Measured at 2000x11: backward sweep -3 % at N=32, -8 % at N=64 and 128, -3 % at N=256; forward sweep -3 % at N=16, nothing
// at N=32 and +8..12 % (slower) from N=64 on, where one row block's data is many lines already and the natural order
// spreads the XCDs' requests better over the memory channels: the forward sweep keeps the natural order there.
#ifndef HANK_XCDMAP_FWD_MAXN
#define HANK_XCDMAP_FWD_MAXN 32
#endif
__device__ __forceinline__ int xcd_contiguous(int p, int nb) {
    const int x = p & 7, k = p >> 3, q = nb >> 3, rem = nb & 7;
    return x * q + (x < rem ? x : rem) + k;
}

template <int RG, typename VT>
__device__ inline void tan_back_body(const Consts &c, const Record &R, const double *__restrict__ xhh, const VT *__restrict__ dxr,
           const VT *__restrict__ dxw, const VT *__restrict__ dxt, const TanGeom &g, int t, int first, const VT *__restrict__ dsIn,
           VT *__restrict__ dsOut, VT *__restrict__ dpol, int bidx_phys, int bidy, VT (*dVsh)[16 * 64], double *Pish) {
    const int nbr_b = (g.nbx + RG - 1) / RG;
    const int bidx = (bidx_phys < nbr_b) ? xcd_contiguous(bidx_phys, nbr_b) : bidx_phys;
    const int nthr = 64 * c.n_e;
    const int lane = threadIdx.x & 63, e = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the column in a scalar register: its bases become scalar)
    const int nl = lane & (g.NC - 1), rl = lane >> g.lgNC;
    const int RB = 64 >> g.lgNC;
    const int n = bidy * g.NC + nl;
    const size_t N = g.N;
    const int tx = first ? t : t - 1;   // period whose knots are produced
    const int txc = tx < 0 ? 0 : tx;
    int a[RG], bi[RG];
    bool valid[RG];
    double cA[RG], cB[RG], cu[RG], cv[RG], ck[RG], cs[RG], xa[RG];
#pragma unroll
    for (int q = 0; q < RG; q++) {
        a[q] = (bidx * RG + q) * RB + rl;
        valid[q] = (a[q] < c.n_a) && (n < g.N);
    }
    if (RG * RB <= 64) {
        // The record coefficients depend on the wealth row only: a wave touches RG*RB <= 64 distinct rows,
        // so ONE lane-sparse load per array (lane L fetches row L) plus cross-lane broadcasts replaces RG
        // full-width loads per array — the vector memory path is the scarce unit here (a full 64-lane
        // 8-byte load occupies it for >= 8 clocks however few distinct addresses it has).
        const int qa = lane / RB, ra = lane - qa * RB;
        const int al = (bidx * RG + qa) * RB + ra;
        int l_i = 0;
        double l_A = 0, l_B = 0, l_u = 0, l_v = 0, l_k = 0, l_s = 0, l_x = 0;
        if (lane < RG * RB && al < c.n_a) {
            const size_t pt = (size_t)e * c.n_a + al;
            const size_t off = (size_t)t * c.G + pt, off1 = (size_t)txc * c.G + pt;
            l_i = R.ib[off]; l_A = R.A[off]; l_B = R.B[off]; l_u = R.u[off]; l_v = R.v[off];
            l_k = R.kc[off1]; l_s = R.s[off1]; l_x = c.a[al];
        }
#pragma unroll
        for (int q = 0; q < RG; q++) {
            const int src = q * RB + rl;
            bi[q] = __shfl(l_i, src, 64); cA[q] = __shfl(l_A, src, 64); cB[q] = __shfl(l_B, src, 64);
            cu[q] = __shfl(l_u, src, 64); cv[q] = __shfl(l_v, src, 64); ck[q] = __shfl(l_k, src, 64);
            cs[q] = __shfl(l_s, src, 64); xa[q] = __shfl(l_x, src, 64);
        }
    } else {
#pragma unroll
        for (int q = 0; q < RG; q++) {
            bi[q] = 0; cA[q] = cB[q] = cu[q] = cv[q] = ck[q] = cs[q] = xa[q] = 0.0;
            if (valid[q]) {
                const size_t pt = (size_t)e * c.n_a + a[q];
                const size_t off = (size_t)t * c.G + pt, off1 = (size_t)txc * c.G + pt;
                bi[q] = R.ib[off]; cA[q] = R.A[off]; cB[q] = R.B[off]; cu[q] = R.u[off]; cv[q] = R.v[off];
                ck[q] = R.kc[off1]; cs[q] = R.s[off1]; xa[q] = c.a[a[q]];
            }
        }
    }
    const bool nok = n < g.N;
    VT dr, dw, dr1, dw1, dtr, dtr1;     // dtr: tangent of the lump-sum transfer (families with n_hh = 3)
    vzero(dr); vzero(dw); vzero(dr1); vzero(dw1); vzero(dtr); vzero(dtr1);
    if (nok) {
        dr = dxr[(size_t)t * N + n]; dw = dxw[(size_t)t * N + n];
        dr1 = dxr[(size_t)txc * N + n]; dw1 = dxw[(size_t)txc * N + n];
        if (c.n_hh > 2) { dtr = dxt[(size_t)t * N + n]; dtr1 = dxt[(size_t)txc * N + n]; }
    }
    const double ze = c.z[e], rho1 = 1.0 / (1.0 + xhh[c.n_hh * txc]);

    VT d0[RG], d1[RG];
#pragma unroll
    for (int q = 0; q < RG; q++) {
        vzero(d0[q]); vzero(d1[q]);
        if (valid[q] && !first) {
            const VT *col = dsIn + ((size_t)e * c.n_a) * N + n;
            d0[q] = col[(size_t)bi[q] * N]; d1[q] = col[(size_t)(bi[q] + 1) * N];
        }
    }
    for (int k = threadIdx.x; k < c.n_e * c.n_e; k += nthr) Pish[k] = c.Pi[k];
#pragma unroll
    for (int q = 0; q < RG; q++) {
        VT dV;
        vzero(dV);
        if (valid[q] && !first) {
            const VT dg = vadd(vmul(cA[q], d0[q]), vmul(cB[q], d1[q]));
            st_mode<HANK_ST_DPOL>(&dpol[((size_t)t * c.G + (size_t)e * c.n_a + a[q]) * N + n], dg);
            dV = vadd(vmul(cu[q], dr), vmul(cv[q], vsub(vadd(vmul(xa[q], dr), vadd(vmul(ze, dw), dtr)), dg)));
        }
        dVsh[q][e * 64 + lane] = dV;
    }
    lds_barrier();
    if (tx < 0) return;
#pragma unroll
    for (int q = 0; q < RG; q++) {
        if (valid[q]) {
            const VT dE = mix_sum(vmul(Pish[e], dVsh[q][lane]), &dVsh[q][lane], 64, Pish + e, c.n_e, 1, c.n_e);
            st_mode<HANK_ST_STATE>(&dsOut[((size_t)e * c.n_a + a[q]) * N + n],
                                   vsub(vmul(ck[q], dE), vmul(rho1, vadd(vadd(vmul(ze, dw1), dtr1), vmul(cs[q], dr1)))));
        }
    }
}

template <int RG, typename VT>
__global__ void __launch_bounds__(1024)
k_tan_back(Consts c, Record R, const double *__restrict__ xhh, const VT *__restrict__ dxr,
           const VT *__restrict__ dxw, const VT *__restrict__ dxt, TanGeom g, int t, int first, const VT *__restrict__ dsIn,
           VT *__restrict__ dsOut, VT *__restrict__ dpol) {
    __shared__ VT dVsh[RG][16 * 64];
    __shared__ double Pish[256];
    tan_back_body<RG, VT>(c, R, xhh, dxr, dxw, dxt, g, t, first, dsIn, dsOut, dpol, blockIdx.x, blockIdx.y, dVsh, Pish);
}

// the dual-sweep backward launch: blocks [0, nbp) of grid row 0 run the PRIMAL EGM step of period tp,
// the others the tangent step of period tt = tp + 1 (whose record the previous launch wrote) — both
// recurrences advance in one chain of T launches instead of two.
template <int RG, typename VT>
__global__ void __launch_bounds__(1024)
k_fused_back(Consts c, Record R, const double *__restrict__ xhh, int *err, int tp, int nbp,
             const VT *__restrict__ dxr, const VT *__restrict__ dxw, const VT *__restrict__ dxt, TanGeom g, int tt, int first,
             const VT *__restrict__ dsIn, VT *__restrict__ dsOut, VT *__restrict__ dpol) {
    __shared__ VT dVsh[RG][16 * 64];
    __shared__ double Pish[256];
    if ((int)blockIdx.x < nbp) {
        if (blockIdx.y != 0 || tp < 0 || (int)threadIdx.x >= RBP * c.n_e) return;
        egm_step_body(c, R, xhh, tp, err, blockIdx.x, reinterpret_cast<double *>(&dVsh[0][0]));   // needs n_e*RBP + n_e^2 <= 768 doubles
        return;
    }
    if (tt < 0) return;
    tan_back_body<RG, VT>(c, R, xhh, dxr, dxw, dxt, g, tt, first, dsIn, dsOut, dpol, blockIdx.x - nbp, blockIdx.y, dVsh, Pish);
}

// one forward period: segment gather of the lottery tangent (ForwardIteration.jl:37-99 under
// Dual), mix over e, aggregate dagg_t = sum(dpol_t * D_t + pol_t * dD_t) with the POST-transition
// D_t (:301-307). dD state: [e][n_a + KV][N].
//   blocks [0, nbx):      regular target rows; sources j >= clo only.
//   blocks [nbx, nbx+KV): the mass point. Block p sums its share of the clamped prefix [0, clo_e)
//     of every column (weight one, no weight tangent) plus, when row 0 itself is clamped, its share
//     of last period's virtual rows; after the mix the result is stored as VIRTUAL ROW n_a+p: row
//     0's tangent is (real row 0) + sum_p (virtual row p). Everything downstream is linear, so the
//     parts are never combined: a virtual row is a source with row 0's lottery (no own policy
//     tangent), and its aggregate term uses pol[0, e].
template <int RG, typename VT, bool SS>
__device__ inline void tan_fwd_body(const Consts &c, const Record &R, const TanGeom &g, int t, const VT *__restrict__ dDin, VT *__restrict__ dDout,
          const VT *__restrict__ dpol, VT *__restrict__ aggpart, int bidx_phys, int bidy, int nbx_total, VT (*sh)[16 * 64], double *Pish, VT *red, VT *red2) {
    const int nbr_f = (g.nbx + RG - 1) / RG;            // regular blocks; the mass-point blocks behind them keep their place
    const int bidx = (g.N * (int)(sizeof(VT) / 8) <= HANK_XCDMAP_FWD_MAXN && bidx_phys < nbr_f) ? xcd_contiguous(bidx_phys, nbr_f) : bidx_phys;
    const int nthr = 64 * c.n_e;
    const int lane = threadIdx.x & 63, e = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the column in a scalar register: its bases become scalar)
    const int nl = lane & (g.NC - 1), rl = lane >> g.lgNC;
    const int RB = 64 >> g.lgNC;
    const int n = bidy * g.NC + nl;
    const size_t N = g.N;
    const int na = c.n_a, nav = c.n_a + KV;
    const size_t base = (size_t)t * c.G, cb = base + (size_t)e * na;
    const double *Dnew = R.Dseq + base + c.G + (size_t)e * na;
    const VT *dDc = dDin + ((size_t)e * nav) * N + n;
    const VT *dpc = dpol + cb * N + n;
    const int clo = R.clo[(size_t)t * c.n_e + e];
    const int nbr = (g.nbx + RG - 1) / RG;            // regular blocks
    const bool virt_block = bidx >= nbr;
    const bool nok = n < g.N;
    int r[RG];
    bool valid[RG];
    VT acc[RG];
    double cp[RG];
    VT pagg;              // sum of dpol_j * D_t[j] over the sources this thread owns
    vzero(pagg);
    bool in_lds = false;   // the gathered tile already sits in sh (source-stationary path)
    if (SS && !virt_block) {
        // SOURCE-STATIONARY form: the block owns the RG*RB target rows [r0, r0+rows); column e's wave walks the
        // contiguous source range that feeds them, RB source rows per instruction, and adds each source's two lottery
        // parts into the LDS tile of its target rows (ds_add_f64; a wave only touches its own column's tile).
        // Every source row is loaded once per block (the gather form loads it for both of its target rows, 35 %
        // more fabric reads at N=32) with half the vector-memory instructions per row.
        const int rows = RG * RB, r0 = bidx * rows, lgRB = 6 - g.lgNC;
#pragma unroll
        for (int q = 0; q < RG; q++) {
            r[q] = r0 + q * RB + rl;
            valid[q] = (r[q] < na) && nok;
            vzero(acc[q]);
            cp[q] = valid[q] ? R.pol[cb + r[q]] : 0.0;
            sh[q][e * 64 + lane] = acc[q];
        }
        const int *st = R.start + ((size_t)t * c.n_e + e) * (na + 1);
        const int rend = min(r0 + rows, na);
        const int jlo = st[r0 > 0 ? r0 - 1 : 0], jhi = st[rend];
        for (int jb = jlo; jb < jhi; jb += 2 * RB) {
            // two chunks per trip, all their loads first
            int l[2];
            double2 wg[2];
            double dn[2];
            VT dd[2], dp[2];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int j = jb + u * RB + rl;
                ok[u] = (j < jhi) && nok;
                l[u] = 0; wg[u] = make_double2(0.0, 0.0); dn[u] = 0.0; vzero(dd[u]); vzero(dp[u]);
                if (ok[u]) {
                    l[u] = R.lo[cb + j];
                    wg[u] = R.lwg[cb + j];
                    dn[u] = Dnew[j];
                    dd[u] = dDc[(size_t)j * N];
                    dp[u] = dpc[(size_t)j * N];
                    if (j == 0)   // (then clo == 0) row 0 not clamped: its virtual rows follow row 0's interior lottery
                        for (int k = 0; k < KV; k++) dd[u] = vadd(dd[u], dDc[(size_t)(na + k) * N]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (ok[u]) {
                    const VT gt = vmul(wg[u].y, dp[u]);
                    const int tl = l[u] - r0, th = tl + 1;
                    if (tl >= 0 && tl < rows)
                        lds_add(&sh[tl >> lgRB][e * 64 + ((tl & (RB - 1)) << g.lgNC) + nl], vsub(vmul(1.0 - wg[u].x, dd[u]), gt));
                    if (th >= 0 && th < rows) {   // the source's FIRST-segment target: its aggregate term is taken here, once
                        lds_add(&sh[th >> lgRB][e * 64 + ((th & (RB - 1)) << g.lgNC) + nl], vadd(vmul(wg[u].x, dd[u]), gt));
                        pagg = vadd(pagg, vmul(dn[u], dp[u]));
                    }
                }
            }
        }
        in_lds = true;
    } else if (!SS && !virt_block) {
        int s0[RG], s1[RG], s2[RG];
#pragma unroll
        for (int q = 0; q < RG; q++) {
            r[q] = (bidx * RG + q) * RB + rl;
            valid[q] = (r[q] < na) && nok;
            s0[q] = s1[q] = s2[q] = 0; cp[q] = 0.0;
            if (valid[q]) {
                const int4 sg = R.seg[cb + r[q]];
                s0[q] = sg.x; s1[q] = sg.y; s2[q] = sg.z;
                cp[q] = R.pol[cb + r[q]];
            }
        }
#pragma unroll
        for (int q = 0; q < RG; q++) {
            VT s;
            vzero(s);
            if (valid[q]) {
                if (RG == 1 && g.lgNC >= 4) {
                // every unclamped source sits in exactly one FIRST segment [s0, s1): its aggregate term
                // dpol_j * D_t[j] is taken there, so dpol is not read a third time by its own row.
                // Small batches (one row group per wave): the first FS sources are loaded under lane predicates
                // with no wait between them — a counted loop serialises one memory round trip per source
                // (forward sweep 3.68 -> 3.40 ms at N=32; it costs 25 % at N=128 with two row groups per wave,
                // and 14 % at N=16 where a wave spans 8 rows).
                {
                    constexpr int FS = 2;
                    VT pd[FS], pp[FS];
                    double pn[FS];
                    double2 pw[FS];
#pragma unroll
                    for (int k = 0; k < FS; k++) {
                        const int j = s0[q] + k;
                        vzero(pd[k]); vzero(pp[k]); pn[k] = 0.0; pw[k] = make_double2(0.0, 0.0);
                        if (j < s2[q]) {
                            pp[k] = dpc[(size_t)j * N];
                            pw[k] = R.lwg[cb + j];
                            pd[k] = dDc[(size_t)j * N];
                            if (j < s1[q]) pn[k] = Dnew[j];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < FS; k++) {
                        const int j = s0[q] + k;
                        if (j < s1[q]) {
                            s = vadd(s, vadd(vmul(pw[k].x, pd[k]), vmul(pw[k].y, pp[k])));
                            pagg = vadd(pagg, vmul(pn[k], pp[k]));
                        } else if (j < s2[q]) {
                            s = vadd(s, vsub(vmul(1.0 - pw[k].x, pd[k]), vmul(pw[k].y, pp[k])));
                        }
                    }
                    for (int j = s0[q] + FS; j < s2[q]; j++) {
                        const VT dpj = dpc[(size_t)j * N];
                        const double2 wg = R.lwg[cb + j];
                        if (j < s1[q]) {
                            s = vadd(s, vadd(vmul(wg.x, dDc[(size_t)j * N]), vmul(wg.y, dpj)));
                            pagg = vadd(pagg, vmul(Dnew[j], dpj));
                        } else {
                            s = vadd(s, vsub(vmul(1.0 - wg.x, dDc[(size_t)j * N]), vmul(wg.y, dpj)));
                        }
                    }
                }
                } else {
                    for (int j = s0[q]; j < s1[q]; j++) {
                        // every unclamped source sits in exactly one FIRST segment: its aggregate term
                        // dpol_j * D_t[j] is taken here, so dpol is not read a third time by its own row
                        const VT dpj = dpc[(size_t)j * N];
                        const double2 wg = R.lwg[cb + j];
                        s = vadd(s, vadd(vmul(wg.x, dDc[(size_t)j * N]), vmul(wg.y, dpj)));
                        pagg = vadd(pagg, vmul(Dnew[j], dpj));
                    }
                    for (int j = s1[q]; j < s2[q]; j++) {
                        const double2 wg = R.lwg[cb + j];
                        s = vadd(s, vsub(vmul(1.0 - wg.x, dDc[(size_t)j * N]), vmul(wg.y, dpc[(size_t)j * N])));
                    }
                }
                // row 0 not clamped: its virtual rows follow row 0's (interior) lottery
                if (clo == 0 && s2[q] > 0 && (s0[q] == 0 || s1[q] == 0)) {
                    const double w0 = (s0[q] == 0 && s1[q] > 0) ? R.lw[cb] : 1.0 - R.lw[cb];
                    VT v;
                    vzero(v);
                    for (int k = 0; k < KV; k++) v = vadd(v, dDc[(size_t)(na + k) * N]);
                    s = vadd(s, vmul(w0, v));
                }
            }
            acc[q] = s;
        }
    } else {
        const int p = bidx - nbr;
#pragma unroll
        for (int q = 0; q < RG; q++) { r[q] = na + p; valid[q] = (q == 0) && nok && (rl == 0); vzero(acc[q]); cp[q] = 0.0; }
        cp[0] = R.pol[cb];       // a virtual row carries row 0's policy and no policy tangent of its own
        VT s;
        vzero(s);
        if (nok && clo > 0) {
            const int M = clo + KV;                         // clamped sources, then the virtual rows
            const int lo = (int)(((long long)M * p) / KV), hi = (int)(((long long)M * (p + 1)) / KV);
            for (int i = lo + rl; i < hi; i += RB) {
                s = vadd(s, dDc[(size_t)(i < clo ? i : na + (i - clo)) * N]);
                if (i < clo) pagg = vadd(pagg, vmul(Dnew[i], dpc[(size_t)i * N]));   // clamped sources: policy partial is 0 except on a knot tie
            }
        }
        for (int off = 32; off >= g.NC; off >>= 1) { s = vadd(s, vshfl_xor(s, off)); pagg = vadd(pagg, vshfl_xor(pagg, off)); }
        if (rl != 0) vzero(pagg);
        acc[0] = s;
    }
    for (int k = threadIdx.x; k < c.n_e * c.n_e; k += nthr) Pish[k] = c.Pi[k];
    if (!in_lds) {
#pragma unroll
        for (int q = 0; q < RG; q++) sh[q][e * 64 + lane] = acc[q];
    }
    lds_barrier();
    VT part, part2;       // the policy-weighted aggregate's partial and the wealth-grid-weighted one's (see dist_step_body)
    vzero(part); vzero(part2);
#pragma unroll
    for (int q = 0; q < RG; q++) {
        if (valid[q]) {
            // dD_t[r,e] = sum_k dD_mid[r,k] * Pi[k,e]
            const VT dDn = mix_sum(vmul(Pish[c.n_e * e], sh[q][lane]), &sh[q][lane], 64, Pish + c.n_e * e, 1, 1, c.n_e);
            st_mode<HANK_ST_STATE>(&dDout[((size_t)e * nav + r[q]) * N + n], dDn);
            part = vadd(part, vmul(cp[q], dDn));
            part2 = vadd(part2, vmul(c.a[r[q] < na ? r[q] : 0], dDn));       // (a virtual row is a part of row 0)
        }
    }
    part = vadd(part, pagg);
    // aggregate of this block: the columns' partials meet in `red` (its own LDS array: no barrier is needed before
    // writing it), ONE barrier, then wave 0 sums over columns and over the RB row lanes of each tangent
    red[e * 64 + lane] = part;
    red2[e * 64 + lane] = part2;
    lds_barrier();
    if (e == 0) {
        VT s = red[lane], s2 = red2[lane];
        for (int k = 1; k < c.n_e; k++) { s = vadd(s, red[k * 64 + lane]); s2 = vadd(s2, red2[k * 64 + lane]); }
        for (int off = 32; off >= g.NC; off >>= 1) { s = vadd(s, vshfl_xor(s, off)); s2 = vadd(s2, vshfl_xor(s2, off)); }
        if (rl == 0 && nok) {       // aggpart: [t][block][2 N]: the first aggregate's N partials, then the second's
            aggpart[((size_t)t * nbx_total + bidx) * 2 * N + n] = s;
            aggpart[((size_t)t * nbx_total + bidx) * 2 * N + N + n] = s2;
        }
    }
}

template <int RG, typename VT, bool SS>
__global__ void __launch_bounds__(1024)
k_tan_fwd(Consts c, Record R, TanGeom g, int t, const VT *__restrict__ dDin, VT *__restrict__ dDout,
          const VT *__restrict__ dpol, VT *__restrict__ aggpart) {
    __shared__ VT sh[RG][16 * 64];
    __shared__ VT red[16 * 64], red2[16 * 64];
    __shared__ double Pish[256];
    tan_fwd_body<RG, VT, SS>(c, R, g, t, dDin, dDout, dpol, aggpart, blockIdx.x, blockIdx.y, gridDim.x, sh, Pish, red, red2);
}

// the dual-sweep forward launch: blocks [0, nbp) of grid row 0 run the PRIMAL distribution step of
// period tp, the others the tangent step of period tt = tp - 1 (D_{tt+1} was written by the previous launch).
template <int RG, typename VT, bool SS>
__global__ void __launch_bounds__(1024)
k_fused_fwd(Consts c, Record R, int tp, int nbp, double *__restrict__ paggpart, TanGeom g, int tt,
            const VT *__restrict__ dDin, VT *__restrict__ dDout, const VT *__restrict__ dpol,
            VT *__restrict__ aggpart) {
    __shared__ VT sh[RG][16 * 64];
    __shared__ VT red[16 * 64], red2[16 * 64];
    __shared__ double Pish[256];
    if ((int)blockIdx.x < nbp) {
        if (blockIdx.y != 0 || tp < 0 || (int)threadIdx.x >= RBP * c.n_e) return;
        dist_step_body(c, R, tp, paggpart, blockIdx.x, nbp, reinterpret_cast<double *>(&sh[0][0]));
        return;
    }
    if (tt < 0) return;
    tan_fwd_body<RG, VT, SS>(c, R, g, tt, dDin, dDout, dpol, aggpart, blockIdx.x - nbp, blockIdx.y, gridDim.x - nbp, sh, Pish, red, red2);
}

// ---- granular tangent halves (hank_backward_step_dual) ---------------------------------------
// X-tangent alone: dV' (G,N col-major) -> ds[e][a][N]
__global__ void k_tan_X(Consts c, const double *kc, const double *s, double r, const double *dr,
                        const double *dw, const double *dtr, int N, const double *dVin_colmajor, double *dsOut) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= c.G * N) return;
    const int n = idx % N, pt = idx / N, e = pt / c.n_a, a = pt - e * c.n_a;
    double dE = 0.0;
    for (int e2 = 0; e2 < c.n_e; e2++) dE += dVin_colmajor[(size_t)n * c.G + (size_t)e2 * c.n_a + a] * c.Pi[e + c.n_e * e2];
    const double rho = 1.0 / (1.0 + r);
    dsOut[(size_t)pt * N + n] = kc[pt] * dE - rho * ((c.z[e] * dw[n] + (dtr ? dtr[n] : 0.0)) + s[pt] * dr[n]);
}
// Y-tangent alone: ds -> dpol, dV (both (G,N) col-major for the caller)
__global__ void k_tan_Y(Consts c, const int *ib, const double *A, const double *B, const double *u,
                        const double *v, const double *dr, const double *dw, const double *dtr, int N,
                        const double *ds, double *dpol_cm, double *dV_cm) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= c.G * N) return;
    const int n = idx % N, pt = idx / N, e = pt / c.n_a, a = pt - e * c.n_a;
    const int i = ib[pt];
    const double *col = ds + ((size_t)e * c.n_a) * N + n;
    const double dg = A[pt] * col[(size_t)i * N] + B[pt] * col[(size_t)(i + 1) * N];
    dpol_cm[(size_t)n * c.G + pt] = dg;
    dV_cm[(size_t)n * c.G + pt] = u[pt] * dr[n] + v[pt] * ((c.a[a] * dr[n] + (c.z[e] * dw[n] + (dtr ? dtr[n] : 0.0))) - dg);
}

// ---- granular forward step for ARBITRARY policies (atomic scatter; parity tests of a6/a7) -----
// Dmid [G*(1+N)]: slot 0 = value, slots 1..N = partials, layout [pt][1+N]
__global__ void k_scatter_general(Consts c, const double *policy, const double *dpolicy_cm,
                                  const double *Dprev, const double *dDprev_cm, int N, double *Dmid) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = 1 + N;
    if (idx >= c.G * W) return;
    const int k = idx % W, pt = idx / W, e = pt / c.n_a;
    const int n = c.n_a;
    const double p = policy[pt];
    int lo = -1, hi = n;
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (c.a[mid] < p) lo = mid; else hi = mid;
    }
    const double Dv = Dprev[pt];
    const double dD = k ? dDprev_cm[(size_t)(k - 1) * c.G + pt] : 0.0;
    const double val = k ? dD : Dv;  // the slot's "D"
    if (hi == 0) {
        atomicAdd(&Dmid[((size_t)e * n + 0) * W + k], val);
    } else if (hi >= n) {
        atomicAdd(&Dmid[((size_t)e * n + n - 1) * W + k], val);
    } else {
        const int l = hi - 1;
        const double gap = c.a[hi] - c.a[l];
        const double w = (p - c.a[l]) / gap;
        if (k == 0) {
            atomicAdd(&Dmid[((size_t)e * n + l) * W], (1.0 - w) * Dv);
            atomicAdd(&Dmid[((size_t)e * n + hi) * W], w * Dv);
        } else {
            const double dwt = dpolicy_cm[(size_t)(k - 1) * c.G + pt] / gap;
            atomicAdd(&Dmid[((size_t)e * n + l) * W + k], (1.0 - w) * dD - dwt * Dv);
            atomicAdd(&Dmid[((size_t)e * n + hi) * W + k], w * dD + dwt * Dv);
        }
    }
}
// mix over e + per-point aggregate terms; outputs col-major (G) and (G,N); aggterm[pt*(1+N)+k]
__global__ void k_mix_general(Consts c, const double *Dmid, const double *policy,
                              const double *dpolicy_cm, int N, double *Dout, double *dDout_cm,
                              double *aggterm) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = 1 + N;
    if (idx >= c.G * W) return;
    const int k = idx % W, pt = idx / W, e2 = pt / c.n_a, a = pt - e2 * c.n_a;
    double s = 0.0;
    for (int e = 0; e < c.n_e; e++) s += Dmid[((size_t)e * c.n_a + a) * W + k] * c.Pi[e + c.n_e * e2];
    if (k == 0) {
        Dout[pt] = s;
        aggterm[(size_t)pt * W] = policy[pt] * s;
    } else {
        dDout_cm[(size_t)(k - 1) * c.G + pt] = s;
        double Dn = 0.0;
        for (int e = 0; e < c.n_e; e++) Dn += Dmid[((size_t)e * c.n_a + a) * W] * c.Pi[e + c.n_e * e2];
        aggterm[(size_t)pt * W + k] = policy[pt] * s + dpolicy_cm[(size_t)(k - 1) * c.G + pt] * Dn;
    }
}
// out[k] = sum_pt aggterm[pt*W + k], one block per k (fixed order within the block tree)
__global__ void k_colsum(const double *aggterm, int G, int W, double *out) {
    __shared__ double red[16];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int pt = threadIdx.x; pt < G; pt += blockDim.x) s += aggterm[(size_t)pt * W + k];
    const double tot = block_sum(s, red, blockDim.x);
    if (threadIdx.x == 0) out[k] = tot;
}

// (G,P,N) col-major export of dpol[P][G][N]
__global__ void k_export_dpol(const double *dpol, int G, int P, int N, double *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)G * P * N;
    if (idx >= total) return;
    const size_t n = idx / ((size_t)G * P), rem = idx - n * (size_t)G * P, t = rem / G, pt = rem - t * G;
    out[idx] = dpol[(t * G + pt) * N + n];
}

}  // namespace hank
