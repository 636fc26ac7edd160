// hank_xcd.h — EXPERIMENT, not compiled into the library: persistent backward tangent sweep, ONE launch, one group
// of workgroups per XCD, SEVERAL tangent directions per group (the record is read once per workgroup and period).
//
// Measured on MI355X against the per-period launches (tangent backward sweep, ms; dpol bit-identical):
//   2000x11, T=300:  N=4  2.46 vs 1.69   N=32  3.30 vs 2.02   N=64  4.57 vs 3.62   N=128  7.90 vs 6.29
//   500x4,   T=300:  N=4  0.74 vs 1.08   N=32  1.42 vs 1.14   N=128 5.55 vs 1.40
// i.e. 8-11 us per period at 2000x11 where a launch takes 6.7: the group barrier (drain of the sc1 stores, workgroup
// barrier of n_e waves, arrival, 32 x n_e pollers on one L2 word) plus the sc1 gather cost more than the kernel
// boundary they replace, and the per-direction work is serial inside a workgroup. Third persistent design that
// loses to hipGraph-replayed launches on this path (see hank_cluster.h, hank_small.h).
//
// 256 workgroups (one per CU); the 32 workgroups with the same blockIdx % 8 form a group (same XCD under
// round-robin dispatch — a speed assumption only, every hand-off below is agent-scope correct on any placement).
// A group owns the tangent directions n = x, x+8, x+16, ...; each of its workgroups owns a slab of wealth rows
// (all n_e columns: wave = column, lane = row). The loop-carried dV lives in LDS for the whole sweep; once per
// period the knot tangents ds of the group's directions are published with write-through (sc1) stores into a
// double-buffered exchange array (L2-resident), the group meets at ONE barrier (a monotonic counter in L2),
// and the bracket gather reads them back with sc1 loads. No kernel boundary, no cross-XCD traffic except dpol.
// Every spin is bounded: on timeout a global word is set and all waits fall through (the grid always drains).
#pragma once
#include "hank_kernels.h"

namespace hank {

typedef unsigned long long xu64;
constexpr int XCD_GROUPS = 8, XCD_MEMBERS = 32;
constexpr int XCD_TMAX = 16;                 // directions per group and launch (N <= 128)
constexpr unsigned XCD_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ void xst_sc1(double *p, double x) {
    __hip_atomic_store(reinterpret_cast<xu64 *>(p), (xu64)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double xld_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const xu64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void xlds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// group barrier: every storing wave has drained (vmcnt(0)), workgroup barrier, one lane arrives on the group's
// counter, every wave polls it (one lane, broadcast). `target` = members * (episodes so far).
__device__ __forceinline__ void xcd_group_barrier(unsigned *counter, unsigned target, unsigned *timeout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    for (;;) {
        unsigned v = 0;
        if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = __builtin_amdgcn_readfirstlane(v);
        if ((int)(v - target) >= 0) break;
        if (++spins > XCD_SPIN_LIMIT || ((spins & 1023u) == 0u && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// dynamic LDS: dV[TPX][n_e][64] + Pi[n_e^2]
// dxhh: (n_hh, P, N) column-major; dpol: [N][P][G] (= (G, P, N) column-major); xbuf: [2][8][XCD_TMAX][G]
template <int TPX>
__global__ void __launch_bounds__(1024)
k_xcd_tan_back(Consts c, Record R, const double *__restrict__ xhh, const double *__restrict__ dxhh, int N, int RW,
               double *__restrict__ xbuf, unsigned *counters, unsigned *timeout, double *__restrict__ dpol) {
    extern __shared__ double lds[];
    const int G = c.G, P = c.P, na = c.n_a, ne = c.n_e;
    double *dV = lds, *Pish = lds + (size_t)TPX * ne * 64;
    const int lane = threadIdx.x & 63, e = threadIdx.x >> 6;
    const int grp = blockIdx.x & 7, mem = blockIdx.x >> 3;
    const int a = mem * RW + lane;
    const bool ok = lane < RW && a < na;
    const int pt = e * na + (ok ? a : 0);
    for (int k = threadIdx.x; k < ne * ne; k += blockDim.x) Pish[k] = c.Pi[k];
#pragma unroll
    for (int k = 0; k < TPX; k++) dV[(k * ne + e) * 64 + lane] = 0.0;     // dV_T = 0 (BackwardIteration.jl:85)
    const double ze = c.z[e], xa = c.a[ok ? a : 0];
    unsigned *counter = counters + grp * 32;     // one counter per group, 128 bytes apart
    double *xb = xbuf + (size_t)grp * XCD_TMAX * G;
    const size_t xstride = (size_t)XCD_GROUPS * XCD_TMAX * G;   // between the two exchange buffers
    // coefficients of the period being processed, fetched one period ahead
    int ib = 0, ibN = 0;
    double cA = 0, cB = 0, cu = 0, cv = 0, ck = 0, cs = 0, nA = 0, nB = 0, nu = 0, nv = 0, nk = 0, ns = 0;
    if (ok) {
        const size_t o = (size_t)(P - 1) * G + pt;
        ib = R.ib[o]; cA = R.A[o]; cB = R.B[o]; cu = R.u[o]; cv = R.v[o]; ck = R.kc[o]; cs = R.s[o];
    }
    __syncthreads();
    int cur = 0;
    unsigned episode = 0;
    for (int t = P - 1; t >= 0; t--) {
        if (t > 0 && ok) {
            const size_t o = (size_t)(t - 1) * G + pt;
            ibN = R.ib[o]; nA = R.A[o]; nB = R.B[o]; nu = R.u[o]; nv = R.v[o]; nk = R.kc[o]; ns = R.s[o];
        }
        const double rho = 1.0 / (1.0 + xhh[c.n_hh * t]);
        double dr[TPX], dw[TPX];
#pragma unroll
        for (int k = 0; k < TPX; k++) {
            const int n = grp + 8 * k;
            dr[k] = dw[k] = 0.0;
            if (n < N) { const double *x = dxhh + (size_t)c.n_hh * ((size_t)t + (size_t)P * n); dr[k] = x[0]; dw[k] = x[1]; }
        }
        // X half of period t for every direction of the group: knot tangents from dV_{t+1} -> exchange buffer
        double *xo = xb + (size_t)cur * xstride;
#pragma unroll
        for (int k = 0; k < TPX; k++) {
            const double *col = dV + (size_t)k * ne * 64 + lane;
            double dE = col[0] * Pish[e];
            for (int e2 = 1; e2 < ne; e2++) dE += col[e2 * 64] * Pish[e + ne * e2];
            if (ok && grp + 8 * k < N) xst_sc1(&xo[(size_t)k * G + pt], ck * dE - rho * (ze * dw[k] + cs * dr[k]));
        }
        episode++;
        xcd_group_barrier(counter, episode * XCD_MEMBERS, timeout);
        // Y half: bracket gather from the exchange buffer -> policy tangent, marginal-value tangent
#pragma unroll
        for (int k = 0; k < TPX; k++) {
            const int n = grp + 8 * k;
            if (ok && n < N) {
                const double *colx = xo + (size_t)k * G + (size_t)e * na;
                const double dg = cA * xld_sc1(&colx[ib]) + cB * xld_sc1(&colx[ib + 1]);
                dpol[((size_t)n * P + t) * G + pt] = dg;
                dV[(k * ne + e) * 64 + lane] = cu * dr[k] + cv * ((xa * dr[k] + ze * dw[k]) - dg);
            }
        }
        xlds_barrier();
        cur ^= 1;
        ib = ibN; cA = nA; cB = nB; cu = nu; cv = nv; ck = nk; cs = ns;
    }
}

}  // namespace hank
