// hank_small.h — EXPERIMENT, not compiled into the library: small-grid sweeps, ONE launch per sweep, one
// workgroup per tangent direction.
//
// When the whole asset x income grid fits a workgroup's LDS (G <= SMALL_G_MAX points) the per-period
// all-to-all (bracket gather across wealth, mixing across productivity) never has to leave the CU: the
// loop-carried state lives in LDS for all T-1 periods, the only synchronisation is the workgroup barrier
// (two per period) and tangent directions are independent workgroups on different CUs.
//
// Measured on MI355X, 500x4, T=300 (parity-green through the whole GPU suite when wired into run_jvp):
//   tangent backward / forward sweep   0.65 / 1.08 ms  (2.2 / 3.6 us per period) at N = 1 and N = 32, 0.95 / 1.36 ms at N = 256
//   per-period launches, same grid     1.11 / 1.27 ms  at N = 1, 2.41 / 2.67 ms at N = 256
// In-kernel stamps (s_memtime, backward kernel, cycles per period of ~6400): waiting for + issuing the 14 record
// loads per thread 2600, the X half's LDS mixing 1750, Y half 700, two barriers 680, loop overhead 630.
// One CU pulls the 52 B/point record through its 64 B/clk vector-memory path (114 KB per period at G = 2000 =
// 1800 clocks at best) and its 128 B/clk LDS: the sweep is bounded by ONE CU's throughput, not by latency — deeper
// run-ahead rings (SMALL_K = 2, 3) and 512 x 4 / 256 x 8 thread geometries were slower. A 1.3-1.7x gain at N = 1 did
// not justify a second code path for the reference's small configurations; sharing one record fetch between several
// directions per workgroup is the next step if this is revisited.
//
// Same arithmetic as tan_back_body / tan_fwd_body (hank_kernels.h) and the same record; dpol here is
// [n][t][e][a] — the column-major (G, P, N) array BackwardIteration returns.
#pragma once
#include "hank_kernels.h"
#include <type_traits>

namespace hank {

// workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight
// (__syncthreads() drains vmcnt and would put every run-ahead fetch back on the critical path)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef HANK_SMALL_STAMPS
__device__ unsigned long long g_small_dbg[16];
#define SSTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); g_small_dbg[k] += now_ - last_; last_ = now_; } } while (0)
#else
#define SSTAMP(k) do {} while (0)
#endif

constexpr int SMALL_NT = 1024;          // threads per workgroup
constexpr int SMALL_PPT = 2;             // grid points per thread
constexpr int SMALL_G_MAX = SMALL_NT * SMALL_PPT;

// LDS doubles: backward dV[G] ds[G] Pi[n_e^2]; forward dD[G] mid[G] dpS[G] wS[G] gS[G] Pi[n_e^2] wpart[2*16]
static inline size_t small_back_lds(const Consts &c) { return sizeof(double) * (2 * (size_t)c.G + (size_t)c.n_e * c.n_e); }
static inline size_t small_fwd_lds(const Consts &c) { return sizeof(double) * (5 * (size_t)c.G + (size_t)c.n_e * c.n_e + 32); }

// The record coefficients of a period are fetched into registers SMALL_K periods before they are used: one
// period of the loop takes ~0.3 us of LDS work, a fetch from L2/HBM ~2 us, so a one-period run-ahead leaves the
// loop waiting on memory (measured 2.5 us per period); the ring below is statically indexed (t loop unrolled by K).
constexpr int SMALL_K = 1;

struct BackCoef { int ib; double A, B, u, v, kc, s; };
struct BackPer { double dr, dw, r; };

// backward tangent sweep of direction n = blockIdx.x: the partials of BackwardIteration.jl:90-113 under
// Dual{T,Float64,N}; dxhh is the caller's (n_hh, P, N) column-major tangent array.
__global__ void __launch_bounds__(SMALL_NT)
k_small_tan_back(Consts c, Record R, const double *__restrict__ xhh, const double *__restrict__ dxhh, double *__restrict__ dpol) {
    extern __shared__ double lds[];
    const int G = c.G, P = c.P, na = c.n_a, ne = c.n_e;
    double *dV = lds, *ds = lds + G, *Pish = ds + G;
    const int n = blockIdx.x;
    const double *dxn = dxhh + 2 * (size_t)P * n;
    double *dpn = dpol + (size_t)n * P * G;
    for (int k = threadIdx.x; k < ne * ne; k += SMALL_NT) Pish[k] = c.Pi[k];
    int p[SMALL_PPT], e[SMALL_PPT], a[SMALL_PPT];
    bool ok[SMALL_PPT];
    double ze[SMALL_PPT], xa[SMALL_PPT];
    BackCoef ring[SMALL_K][SMALL_PPT];
    BackPer per[SMALL_K];
    auto fetch = [&](int k, int t) {      // stage k <- period t (nothing when t < 0)
        per[k].dr = per[k].dw = per[k].r = 0.0;      // (r is only divided by when the period is processed: no wait here)
        if (t >= 0) { per[k].dr = dxn[2 * (size_t)t]; per[k].dw = dxn[2 * (size_t)t + 1]; per[k].r = xhh[2 * t]; }
#pragma unroll
        for (int q = 0; q < SMALL_PPT; q++) {
            BackCoef &f = ring[k][q];
            if (ok[q] && t >= 0) {
                const size_t o = (size_t)t * G + p[q];
                f.ib = R.ib[o]; f.A = R.A[o]; f.B = R.B[o]; f.u = R.u[o]; f.v = R.v[o]; f.kc = R.kc[o]; f.s = R.s[o];
            }
        }
    };
#pragma unroll
    for (int q = 0; q < SMALL_PPT; q++) {
        p[q] = threadIdx.x + q * SMALL_NT;
        ok[q] = p[q] < G;
        e[q] = ok[q] ? p[q] / na : 0;
        a[q] = ok[q] ? p[q] - e[q] * na : 0;
        ze[q] = c.z[e[q]]; xa[q] = c.a[a[q]];
        if (ok[q]) dV[p[q]] = 0.0;                                   // dV_T = 0 (BackwardIteration.jl:85)
#pragma unroll
        for (int k = 0; k < SMALL_K; k++) ring[k][q] = BackCoef{0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int k = 0; k < SMALL_K; k++) fetch(k, P - 1 - k);
    __syncthreads();
#ifdef HANK_SMALL_STAMPS
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    // one period; the ring stage k is a compile-time constant so the ring stays in registers
    auto stage = [&](auto kc, int t) {
        constexpr int k = decltype(kc)::value;
        SSTAMP(0);
        BackCoef f[SMALL_PPT];
        const BackPer pr = per[k];
        const double rho = 1.0 / (1.0 + pr.r);
#pragma unroll
        for (int q = 0; q < SMALL_PPT; q++) f[q] = ring[k][q];
        fetch(k, t - SMALL_K);
        SSTAMP(1);
        // X half of period t: knot tangents from dV_{t+1}. The loop is a chain of LDS round trips (~100 clocks each
        // with 4 waves per SIMD to hide them): the points of a thread advance together and the column loop is
        // unrolled so that 8 reads are in flight at a time; the summation order is unchanged.
        {
            double dE[SMALL_PPT];
#pragma unroll
            for (int q = 0; q < SMALL_PPT; q++) dE[q] = dV[a[q]] * Pish[e[q]];
#pragma unroll 4
            for (int e2 = 1; e2 < ne; e2++)
#pragma unroll
                for (int q = 0; q < SMALL_PPT; q++) dE[q] += dV[e2 * na + a[q]] * Pish[e[q] + ne * e2];
#pragma unroll
            for (int q = 0; q < SMALL_PPT; q++)
                if (ok[q]) ds[p[q]] = f[q].kc * dE[q] - rho * (ze[q] * pr.dw + f[q].s * pr.dr);
        }
        SSTAMP(2);
        lds_barrier();
        SSTAMP(3);
        // Y half of period t: bracket gather -> policy tangent, marginal-value tangent
#pragma unroll
        for (int q = 0; q < SMALL_PPT; q++)
            if (ok[q]) {
                const double *col = ds + e[q] * na;
                const double dg = f[q].A * col[f[q].ib] + f[q].B * col[f[q].ib + 1];
                dpn[(size_t)t * G + p[q]] = dg;
                dV[p[q]] = f[q].u * pr.dr + f[q].v * ((xa[q] * pr.dr + ze[q] * pr.dw) - dg);   // every read of dV_{t+1} happened before the barrier above
            }
        SSTAMP(4);
        lds_barrier();
        SSTAMP(5);
    };
    int t = P - 1;
    for (; t >= SMALL_K - 1; t -= SMALL_K) {          // whole groups of SMALL_K periods, no exits inside
        stage(std::integral_constant<int, 0>{}, t);
        if (SMALL_K > 1) stage(std::integral_constant<int, 1 % SMALL_K>{}, t - 1);
        if (SMALL_K > 2) stage(std::integral_constant<int, 2 % SMALL_K>{}, t - 2);
    }
    if (t >= 0) stage(std::integral_constant<int, 0>{}, t);
    if (SMALL_K > 2 && t >= 1) stage(std::integral_constant<int, 1 % SMALL_K>{}, t - 1);
}

struct FwdCoef { double dp, w, g, pol, D; int4 seg; int clo; };

// forward tangent sweep of direction n = blockIdx.x: the partials of ForwardIteration.jl:297-308
// (Young lottery push-forward :37-99, mixing, post-transition dot); dagg_cm is (P, N) column-major.
__global__ void __launch_bounds__(SMALL_NT)
k_small_tan_fwd(Consts c, Record R, const double *__restrict__ dpol, double *__restrict__ dagg_cm) {
    extern __shared__ double lds[];
    const int G = c.G, P = c.P, na = c.n_a, ne = c.n_e;
    double *dD = lds, *mid = dD + G, *dpS = mid + G, *wS = dpS + G, *gS = wS + G, *Pish = gS + G, *wpart = Pish + ne * ne;
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *dpn = dpol + (size_t)n * P * G;
    for (int k = threadIdx.x; k < ne * ne; k += SMALL_NT) Pish[k] = c.Pi[k];
    int p[SMALL_PPT], e[SMALL_PPT], r[SMALL_PPT];
    bool ok[SMALL_PPT];
    // per period, in the point's role as a SOURCE: policy tangent, lottery weight, ig*D_{t-1}; as a TARGET:
    // segment bounds, policy, D_t
    FwdCoef ring[SMALL_K][SMALL_PPT];
    auto fetch = [&](int k, int t) {
#pragma unroll
        for (int q = 0; q < SMALL_PPT; q++)
            if (ok[q] && t < P) {
                FwdCoef &f = ring[k][q];
                const size_t o = (size_t)t * G + p[q];
                f.dp = dpn[o];
                const double2 wg = R.lwg[o];
                f.w = wg.x; f.g = wg.y;
                f.seg = R.seg[o]; f.pol = R.pol[o]; f.D = R.Dseq[o + G];
                f.clo = (r[q] == 0) ? R.clo[(size_t)t * ne + e[q]] : 0;
            }
    };
#pragma unroll
    for (int q = 0; q < SMALL_PPT; q++) {
        p[q] = threadIdx.x + q * SMALL_NT;
        ok[q] = p[q] < G;
        e[q] = ok[q] ? p[q] / na : 0;
        r[q] = ok[q] ? p[q] - e[q] * na : 0;
        if (ok[q]) dD[p[q]] = 0.0;                            // dD_0 = 0 (ForwardIteration.jl:293)
#pragma unroll
        for (int k = 0; k < SMALL_K; k++) ring[k][q] = FwdCoef{0.0, 0.0, 0.0, 0.0, 0.0, make_int4(0, 0, 0, 0), 0};
    }
#pragma unroll
    for (int k = 0; k < SMALL_K; k++) fetch(k, k);
    auto stage = [&](auto kc, int t) {
        constexpr int k = decltype(kc)::value;
        FwdCoef f[SMALL_PPT];
#pragma unroll
        for (int q = 0; q < SMALL_PPT; q++) {
            f[q] = ring[k][q];
            if (ok[q]) { dpS[p[q]] = f[q].dp; wS[p[q]] = f[q].w; gS[p[q]] = f[q].g; }
        }
        fetch(k, t + SMALL_K);
        lds_barrier();      // also orders last period's dD writes before this period's gather
        if (threadIdx.x < 64 && t > 0) {      // last period's aggregate: 16 wave partials, fixed order
            double s = lane < SMALL_NT / 64 ? wpart[((t - 1) & 1) * 16 + lane] : 0.0;
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) dagg_cm[(size_t)(t - 1) + (size_t)P * n] = s;
        }
        // gather: the first two sources of each segment are read speculatively (index clamped into the column: an LDS
        // read costs no traffic) so that all 16 reads of a point are in flight together; longer segments finish in loops
#pragma unroll
        for (int q = 0; q < SMALL_PPT; q++)
            if (ok[q]) {
                const int cb = e[q] * na;
                const int s0 = f[q].seg.x, s1 = f[q].seg.y, s2 = f[q].seg.z;
                int jj[4] = {s0, s0 + 1, s1, s1 + 1};
                double vw[4], vd[4], vp[4], vg[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = cb + min(jj[k], na - 1);
                    vw[k] = wS[j]; vd[k] = dD[j]; vp[k] = dpS[j]; vg[k] = gS[j];
                }
                double s = 0.0;
                if (s0 < s1) s += vw[0] * vd[0] + vp[0] * vg[0];
                if (s0 + 1 < s1) s += vw[1] * vd[1] + vp[1] * vg[1];
                for (int j = s0 + 2; j < s1; j++) s += wS[cb + j] * dD[cb + j] + dpS[cb + j] * gS[cb + j];
                if (s1 < s2) s += (1.0 - vw[2]) * vd[2] - vp[2] * vg[2];
                if (s1 + 1 < s2) s += (1.0 - vw[3]) * vd[3] - vp[3] * vg[3];
                for (int j = s1 + 2; j < s2; j++) s += (1.0 - wS[cb + j]) * dD[cb + j] - dpS[cb + j] * gS[cb + j];
                for (int j = 0; j < f[q].clo; j++) s += dD[cb + j];   // the mass point: weight one, no weight tangent (:54-58)
                mid[p[q]] = s;
            }
        lds_barrier();
        double part = 0.0;
        {
            double dDn[SMALL_PPT];      // dD_t[r,e] = sum_k dD_mid[r,k] * Pi[k,e]
#pragma unroll
            for (int q = 0; q < SMALL_PPT; q++) dDn[q] = mid[r[q]] * Pish[ne * e[q]];
#pragma unroll 4
            for (int kk = 1; kk < ne; kk++)
#pragma unroll
                for (int q = 0; q < SMALL_PPT; q++) dDn[q] += mid[kk * na + r[q]] * Pish[kk + ne * e[q]];
#pragma unroll
            for (int q = 0; q < SMALL_PPT; q++)
                if (ok[q]) {
                    dD[p[q]] = dDn[q];
                    part += f[q].pol * dDn[q] + f[q].dp * f[q].D;      // dot(vec(policy_t), D_t) under duals (:305-307)
                }
        }
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) wpart[(t & 1) * 16 + wave] = part;
    };
    int t = 0;
    for (; t + SMALL_K <= P; t += SMALL_K) {
        stage(std::integral_constant<int, 0>{}, t);
        if (SMALL_K > 1) stage(std::integral_constant<int, 1 % SMALL_K>{}, t + 1);
        if (SMALL_K > 2) stage(std::integral_constant<int, 2 % SMALL_K>{}, t + 2);
    }
    if (t < P) stage(std::integral_constant<int, 0>{}, t);
    if (SMALL_K > 2 && t + 1 < P) stage(std::integral_constant<int, 1 % SMALL_K>{}, t + 1);
    __syncthreads();
    if (threadIdx.x < 64) {
        double s = lane < SMALL_NT / 64 ? wpart[((P - 1) & 1) * 16 + lane] : 0.0;
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) dagg_cm[(size_t)(P - 1) + (size_t)P * n] = s;
    }
}

}  // namespace hank
