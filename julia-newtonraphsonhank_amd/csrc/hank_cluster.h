// hank_cluster.h — persistent "cluster" tangent sweeps: ONE launch per sweep.
//
// The tangent recurrences are strict in t, and between two periods every wealth row of a
// tangent needs rows owned by others (bracket gather backward, lottery gather forward). Per-period
// launches pay a ~3 us launch floor plus two dependent DRAM round trips 2(T-1) times. Here a
// CLUSTER of CS workgroups (one per CU, same XCD) owns one tangent direction for the whole sweep:
//   - each member keeps its slab of wealth rows (all n_e columns) of the loop-carried state
//     (dV backward, dD forward) in LDS; the n_e x n_e mixing is LDS-local;
//   - once per period the members publish their slab of the gathered quantity (knot tangents ds /
//     distribution tangents dD) with write-through `sc1` stores into a double-buffered exchange
//     tile that lives in the XCD's L2, drain, raise a per-member epoch flag, poll the CS flags, and
//     gather with `sc1` loads (L1-bypassing) — the placement-independent hand-off of
//     cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms" row 1
//     (one workgroup per CU, every hand-off byte stored and loaded sc1, every storing wave drains
//     before the workgroup barrier, one lane signals);
//   - clusters are independent (tangent directions are independent), so there is no grid-wide
//     synchronisation anywhere; a cluster loops over tangents n = c, c + nclusters, ...
// Every spin is bounded: on timeout a global word is set and all waits fall through, so the grid
// always drains (the host then reports an internal error).
//
// dpol layout here: [n][t][e][a] (tangent slowest, wealth fastest: lanes run along wealth), which is
// exactly the column-major (G, P, N) array BackwardIteration returns.
#pragma once
#include "hank_kernels.h"

namespace hank {

typedef unsigned long long u64_t;
constexpr int CL_MAXPASS = 8;            // 64-row passes per member slab (slab <= 512 rows)
constexpr unsigned CL_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ void st_sc1(double *p, double x) {
    __hip_atomic_store(reinterpret_cast<u64_t *>(p), (u64_t)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

struct ClGeom {
    int CS;         // workgroups per cluster (power of two)
    int nclusters;  // clusters in the grid
    int RM;         // wealth rows per member slab
    int xcd_map;    // 1: members of a cluster share blockIdx % 8 (same XCD under round-robin dispatch)
};

__device__ __forceinline__ void cl_ids(const ClGeom &g, int &cluster, int &member) {
    const int b = blockIdx.x;
    if (g.xcd_map) {
        const int xcd = b & 7, slot = b >> 3;
        cluster = xcd + 8 * (slot / g.CS);
        member = slot % g.CS;
    } else {
        cluster = b / g.CS;
        member = b % g.CS;
    }
}

// publish (every wave has issued its sc1 stores) + wait for all CS members of the cluster.
__device__ __forceinline__ void cl_barrier(unsigned *flags, const ClGeom &g, int cluster, int member,
                                           unsigned epoch, unsigned *timeout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its sc1 stores
    __syncthreads();
    if (g.CS > 1) {
        if (threadIdx.x == 0)
            __hip_atomic_store(&flags[cluster * g.CS + member], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            unsigned spins = 0;
            for (;;) {
                unsigned v = epoch;
                if (lane < g.CS)
                    v = __hip_atomic_load(&flags[cluster * g.CS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((int)(v - epoch) >= 0)) break;
                if (++spins > CL_SPIN_LIMIT ||
                    __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    if (lane == 0) __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

// ---- backward tangent sweep ----------------------------------------------------------------------
// block = 64*n_e threads (wave e <-> productivity column e, lanes <-> wealth rows of the slab);
// dynamic LDS: dVsh[n_e][NPASS*64]
template <int NPASS>
__global__ void __launch_bounds__(1024)
k_tanc_back(int n_a, int n_e, int G, int P, int N, ClGeom g, const double *__restrict__ agrid,
            const double *__restrict__ zg, const double *__restrict__ Pi, TAN_REC_PARAMS,
            const double *__restrict__ dxr, const double *__restrict__ dxw, double *xbuf,
            unsigned *flags, unsigned *timeout, double *__restrict__ dpol) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RP = NPASS * 64;
    double *dVsh = lds;
    const int lane = threadIdx.x & 63;
    const int e = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int cluster, member;
    cl_ids(g, cluster, member);
    const int r_lo = member * g.RM;
    const int r_hi = min(n_a, r_lo + g.RM);
    const double ze = zg[e];
    double *xb = xbuf + (size_t)cluster * 2 * G;
    unsigned epoch = 0;
    for (int n = cluster; n < N; n += g.nclusters) {
#pragma unroll
        for (int p = 0; p < NPASS; p++) dVsh[e * RP + p * 64 + lane] = 0.0;   // dV_T = 0 (BackwardIteration.jl:85)
        __syncthreads();
        double *dpn = dpol + (size_t)n * P * G;
        for (int t = P - 1; t >= 0; t--) {
            const size_t tb = (size_t)t * G + (size_t)e * n_a;
            const double dr = dxr[(size_t)t * N + n], dw = dxw[(size_t)t * N + n], rh = rho[t];
            double *xe = xb + (size_t)(t & 1) * G + (size_t)e * n_a;
            // X half of period t: mix dV_{t+1} over e -> knot tangents of my slab, published sc1
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, a = r_lo + rl;
                if (a < r_hi) {
                    double dE = dVsh[rl] * Pi[e];
                    for (int e2 = 1; e2 < n_e; e2++) dE += dVsh[e2 * RP + rl] * Pi[e + n_e * e2];
                    st_sc1(&xe[a], rkc[tb + a] * dE - rh * (ze * dw + rs[tb + a] * dr));
                }
            }
            // Y-half coefficients do not depend on the peers: fetch them while the hand-off settles
            int ci[NPASS];
            double cA[NPASS], cB[NPASS], cu[NPASS], cv[NPASS], cx[NPASS];
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int a = r_lo + p * 64 + lane;
                if (a < r_hi) {
                    ci[p] = ib[tb + a]; cA[p] = rA[tb + a]; cB[p] = rB[tb + a];
                    cu[p] = ru[tb + a]; cv[p] = rv[tb + a]; cx[p] = agrid[a];
                }
            }
            cl_barrier(flags, g, cluster, member, ++epoch, timeout);
            // Y half: bracket gather from the cluster's exchange tile
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, a = r_lo + rl;
                if (a < r_hi) {
                    const double ds0 = ld_sc1(&xe[ci[p]]), ds1 = ld_sc1(&xe[ci[p] + 1]);
                    const double dg = cA[p] * ds0 + cB[p] * ds1;
                    dpn[tb + a] = dg;
                    dVsh[e * RP + rl] = cu[p] * dr + cv[p] * ((cx[p] * dr + ze * dw) - dg);
                }
            }
            __syncthreads();
        }
    }
}

// ---- forward tangent sweep -----------------------------------------------------------------------
// dynamic LDS: dDsh[n_e][RP] + midsh[n_e][RP] + red[16]
template <int NPASS>
__global__ void __launch_bounds__(1024)
k_tanc_fwd(int n_a, int n_e, int G, int P, int N, ClGeom g, const double *__restrict__ Pi,
           const double *__restrict__ lw, const double *__restrict__ ig, const double *__restrict__ Dseq,
           const double *__restrict__ pol, const int *__restrict__ start, const int *__restrict__ clo_,
           double *xbuf, unsigned *flags, unsigned *timeout, const double *__restrict__ dpol,
           double *__restrict__ aggpart /* [N][P][CS] */) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RP = NPASS * 64;
    double *dDsh = lds, *midsh = lds + (size_t)n_e * RP, *red = midsh + (size_t)n_e * RP;
    const int lane = threadIdx.x & 63;
    const int e = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int cluster, member;
    cl_ids(g, cluster, member);
    const int r_lo = member * g.RM;
    const int r_hi = min(n_a, r_lo + g.RM);
    double *xb = xbuf + (size_t)cluster * 2 * G;
    unsigned epoch = 0;
    for (int n = cluster; n < N; n += g.nclusters) {
#pragma unroll
        for (int p = 0; p < NPASS; p++) dDsh[e * RP + p * 64 + lane] = 0.0;   // dD_0 = 0 (ForwardIteration.jl:293)
        __syncthreads();
        const double *dpn = dpol + (size_t)n * P * G;
        for (int t = 0; t < P; t++) {
            const size_t tb = (size_t)t * G + (size_t)e * n_a;
            double *xe = xb + (size_t)(t & 1) * G + (size_t)e * n_a;
            const double *Dprev = Dseq + tb, *Dnew = Dseq + tb + G;
            const int *st = start + ((size_t)t * n_e + e) * (n_a + 1);
            // publish dD_{t-1} of my slab
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, a = r_lo + rl;
                if (a < r_hi) st_sc1(&xe[a], dDsh[e * RP + rl]);
            }
            // peer-independent loads first
            int s0[NPASS], s1[NPASS], s2[NPASS];
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int r = r_lo + p * 64 + lane;
                if (r < r_hi) {
                    s1[p] = st[r]; s2[p] = st[r + 1]; s0[p] = r > 0 ? st[r - 1] : s1[p];
                }
            }
            const int clo = clo_[(size_t)t * n_e + e];
            cl_barrier(flags, g, cluster, member, ++epoch, timeout);
            // lottery-segment gather of my target rows
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, r = r_lo + rl;
                if (r < r_hi) {
                    double acc = 0.0;
                    for (int j = s0[p]; j < s1[p]; j++)
                        acc += lw[tb + j] * ld_sc1(&xe[j]) + (dpn[tb + j] * ig[tb + j]) * Dprev[j];
                    for (int j = s1[p]; j < s2[p]; j++)
                        acc += (1.0 - lw[tb + j]) * ld_sc1(&xe[j]) - (dpn[tb + j] * ig[tb + j]) * Dprev[j];
                    midsh[e * RP + rl] = acc;
                }
            }
            if (member == 0 && clo > 0) {   // the mass point: sum_{j<clo} dD_{t-1}[j] -> row 0, by the column's wave
                double s = 0.0;
                for (int j = lane; j < clo; j += 64) s += ld_sc1(&xe[j]);
                s = wave_sum(s);
                if (lane == 0) midsh[e * RP] += s;
            }
            __syncthreads();
            double part = 0.0;
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, r = r_lo + rl;
                if (r < r_hi) {
                    double dDn = midsh[rl] * Pi[n_e * e];      // dD_t[r,e] = sum_k dD_mid[r,k] * Pi[k,e]
                    for (int k = 1; k < n_e; k++) dDn += midsh[k * RP + rl] * Pi[k + n_e * e];
                    dDsh[e * RP + rl] = dDn;
                    part += pol[tb + r] * dDn + dpn[tb + r] * Dnew[r];
                }
            }
            const double tot = block_sum(part, red);     // (contains the barriers that fence midsh/dDsh reuse)
            if (threadIdx.x == 0) aggpart[((size_t)n * P + t) * g.CS + member] = tot;
        }
    }
}

// dagg[t*N + n] = sum_m aggpart[(n*P + t)*CS + m]
__global__ void k_tanc_sum(const double *__restrict__ aggpart, int P, int N, int CS, double *__restrict__ dagg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * N) return;
    const int t = idx / N, n = idx - t * N;
    double s = 0.0;
    for (int m = 0; m < CS; m++) s += aggpart[((size_t)n * P + t) * CS + m];
    dagg[idx] = s;
}

}  // namespace hank
