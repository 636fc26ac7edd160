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
//   - the per-period linearisation record (~52 B per grid point) is cold in HBM when a period
//     starts; a run-ahead HELPER wave per workgroup touches the slab's record lines two periods
//     early with register-free LDS-DMA loads, so the workers' just-in-time coefficient reads are L2
//     hits and DRAM latency leaves the per-period critical path.
// Every spin is bounded: on timeout a global word is set and all waits fall through, so the grid
// always drains (the host then reports an internal error).
//
// block = 64*(n_e+1) threads: wave e < n_e <-> productivity column e (lanes <-> wealth rows of the
// slab, NPASS passes of 64), wave n_e = helper.
// dpol layout here: [n][t][e][a] (tangent slowest, wealth fastest: lanes run along wealth), which is
// exactly the column-major (G, P, N) array BackwardIteration returns.
#pragma once
#include "hank_kernels.h"

namespace hank {

typedef unsigned long long u64_t;
constexpr int CL_MAXPASS = 8;            // 64-row passes per member slab (slab <= 512 rows)
constexpr unsigned CL_SPIN_LIMIT = 1u << 22;
constexpr int CL_AHEAD = 2;              // periods the helper wave runs ahead

__device__ __forceinline__ void st_sc1(double *p, double x) {
    __hip_atomic_store(reinterpret_cast<u64_t *>(p), (u64_t)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight
// (__syncthreads() would drain vmcnt and put every prefetch on the critical path)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// register-free touch of the cache lines [p, p + bytes): one dword per 128-byte line lands in a
// scratch LDS row through LDS-DMA; the point is the L2 fill. Whole wave participates.
__device__ __forceinline__ void touch_lines(const void *p, size_t bytes, int lane, void *lds_scratch) {
    const char *base = reinterpret_cast<const char *>(reinterpret_cast<size_t>(p) & ~(size_t)127);
    const size_t span = (reinterpret_cast<size_t>(p) - reinterpret_cast<size_t>(base)) + bytes;
    const int nlines = (int)((span + 127) >> 7);
    for (int l0 = 0; l0 < nlines; l0 += 64) {
        const int l = min(l0 + lane, nlines - 1);   // every lane issues (LDS-DMA wants a full wave)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + ((size_t)l << 7)),
                                         (__attribute__((address_space(3))) void *)lds_scratch, 4, 0, 0);
    }
}

#ifdef HANK_STAMPS
#define STAMP(k) do { if (dbg && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_[k] += now_ - last_; last_ = now_; } } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

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

// ---- the hand-off: data-tagged granules (Guideline 16, R2 form) -----------------------------------
// A published double travels as TWO naturally aligned 8-byte words {tag:32 | hi:32}, {tag:32 | lo:32},
// each written by one sc1 (write-through) store and read by one sc1 (L1-bypassing) load: every word
// is single-copy atomic and carries its own tag, so the data IS the flag — no drain, no workgroup
// barrier, no flag round trip on the per-period critical path. tag = running period count + 1 (never
// 0; the exchange ring is zeroed before every launch). A consumer re-reads until both tags match.
// The exchange tile is a ring of CL_RING period slots; before a member overwrites a slot it checks
// (off the critical path: the loads are issued a period early) that every member has finished
// reading the period that lived there (per-member `done` counters).
constexpr int CL_RING = 4;

__device__ __forceinline__ void st_gran(u64_t *g, unsigned tag, double x) {
    const u64_t b = (u64_t)__double_as_longlong(x), tg = (u64_t)tag << 32;
    __hip_atomic_store(g, tg | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, tg | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ld_gran(const u64_t *g, unsigned tag, double &x) {
    const u64_t h = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64_t l = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = __longlong_as_double((long long)(((h & 0xffffffffull) << 32) | (l & 0xffffffffull)));
    return ((unsigned)(h >> 32) == tag) & ((unsigned)(l >> 32) == tag);
}
__device__ __forceinline__ bool cl_timed_out(unsigned &spins, unsigned *timeout) {
    if (++spins > CL_SPIN_LIMIT ||
        ((spins & 1023u) == 0u && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}
// back-pressure: every member of the cluster has completed `need` periods (wave-level check)
__device__ __forceinline__ void cl_ring_ready(const unsigned *done, const ClGeom &g, int cluster, int lane, int need,
                                              unsigned *timeout) {
    if (g.CS == 1 || need <= 0) return;
    unsigned spins = 0;
    for (;;) {
        int v = need;
        if (lane < g.CS) v = (int)__hip_atomic_load(&done[cluster * g.CS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v - need >= 0)) return;
        if (cl_timed_out(spins, timeout)) return;
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---- backward tangent sweep ----------------------------------------------------------------------
// dynamic LDS: dVsh[n_e][NPASS*64] + 64-dword scratch row for the helper's LDS-DMA touches
template <int NPASS, int MAXT>
__global__ void __launch_bounds__(MAXT)
k_tanc_back(int n_a, int n_e, int G, int P, int N, ClGeom g, const double *__restrict__ agrid,
            const double *__restrict__ zg, const double *__restrict__ Pi, TAN_REC_PARAMS,
            const double *__restrict__ dxr, const double *__restrict__ dxw, u64_t *xbuf,
            unsigned *done, unsigned *timeout, double *__restrict__ dpol, unsigned long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RP = NPASS * 64;
    double *dVsh = lds;
    void *scratch = lds + (size_t)n_e * RP;
#ifdef HANK_STAMPS
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63;
    const int e = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool helper = (e == n_e);
    int cluster, member;
    cl_ids(g, cluster, member);
    const int r_lo = member * g.RM;
    const int r_hi = min(n_a, r_lo + g.RM);
    const int nrows = max(0, r_hi - r_lo);
    u64_t *xb = xbuf + (size_t)cluster * CL_RING * G * 2;

    if (helper) {
        // run-ahead L2 warmer; mirrors the workers' barrier sequence (2 per period + 1 per tangent)
        for (int n = cluster; n < N; n += g.nclusters) {
            lds_barrier();
            for (int t = P - 1; t >= 0; t--) {
                const int tp = t - CL_AHEAD;
                if (tp >= 0 && nrows > 0) {
                    for (int e2 = 0; e2 < n_e; e2++) {
                        const size_t o = (size_t)tp * G + (size_t)e2 * n_a + r_lo;
                        touch_lines(ib + o, (size_t)nrows * 4, lane, scratch);
                        touch_lines(rA + o, (size_t)nrows * 8, lane, scratch);
                        touch_lines(rB + o, (size_t)nrows * 8, lane, scratch);
                        touch_lines(ru + o, (size_t)nrows * 8, lane, scratch);
                        touch_lines(rv + o, (size_t)nrows * 8, lane, scratch);
                        touch_lines(rkc + o, (size_t)nrows * 8, lane, scratch);
                        touch_lines(rs + o, (size_t)nrows * 8, lane, scratch);
                    }
                }
                lds_barrier();   // X reads of dVsh done
                lds_barrier();   // end of period
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    const double ze = zg[e];
    double pir[16];   // row e of Pi, wave-constant
#pragma unroll
    for (int k = 0; k < 16; k++) pir[k] = k < n_e ? Pi[e + n_e * k] : 0.0;
    int q = 0;        // running period count of this cluster (tag = q + 1)
    for (int n = cluster; n < N; n += g.nclusters) {
#pragma unroll
        for (int p = 0; p < NPASS; p++) dVsh[e * RP + p * 64 + lane] = 0.0;   // dV_T = 0 (BackwardIteration.jl:85)
        double *dpn = dpol + (size_t)n * P * G;
        double xk[NPASS], xs[NPASS];   // knot coefficients of the period being published
        {
            const size_t tb = (size_t)(P - 1) * G + (size_t)e * n_a;
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int a = r_lo + p * 64 + lane;
                xk[p] = 0.0; xs[p] = 0.0;
                if (a < r_hi) { xk[p] = rkc[tb + a]; xs[p] = rs[tb + a]; }
            }
        }
        lds_barrier();
        for (int t = P - 1; t >= 0; t--, q++) {
            const unsigned tag = (unsigned)q + 1u;
            const size_t tb = (size_t)t * G + (size_t)e * n_a;
            const double dr = dxr[(size_t)t * N + n], dw = dxw[(size_t)t * N + n], rh = rho[t];
            u64_t *xe = xb + ((size_t)(q % CL_RING) * G + (size_t)e * n_a) * 2;
            STAMP(0);
            // every peer-independent read of this period, issued up front: Y-half coefficients of period
            // t and the knot coefficients of period t-1 (L2-warm thanks to the helper)
            int ci[NPASS];
            double cA[NPASS], cB[NPASS], cu[NPASS], cv[NPASS], cx[NPASS], nk[NPASS], ns[NPASS];
            const size_t tb1 = t > 0 ? tb - G : tb;
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int a = r_lo + p * 64 + lane;
                ci[p] = 0; nk[p] = 0.0; ns[p] = 0.0;
                if (a < r_hi) {
                    ci[p] = ib[tb + a]; cA[p] = rA[tb + a]; cB[p] = rB[tb + a];
                    cu[p] = ru[tb + a]; cv[p] = rv[tb + a]; cx[p] = agrid[a];
                    nk[p] = rkc[tb1 + a]; ns[p] = rs[tb1 + a];
                }
            }
            // X half of period t: mix dV_{t+1} over e -> knot tangents of my slab
            double dsv[NPASS];
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane;
                double dE = 0.0;
#pragma unroll
                for (int e2 = 0; e2 < 16; e2++)
                    if (e2 < n_e) dE += dVsh[e2 * RP + rl] * pir[e2];
                dsv[p] = xk[p] * dE - rh * (ze * dw + xs[p] * dr);
            }
            STAMP(1);
            cl_ring_ready(done, g, cluster, lane, q - (CL_RING - 1), timeout);
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int a = r_lo + p * 64 + lane;
                if (a < r_hi) st_gran(&xe[2 * (size_t)a], tag, dsv[p]);
            }
            STAMP(2);
            lds_barrier();   // all X reads of dVsh are done before any Y write
            STAMP(3);
            // Y half: bracket gather from the cluster's exchange ring; re-read until the tags are current
            double d0[NPASS], d1[NPASS];
            {
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int p = 0; p < NPASS; p++) {
                        const int a = r_lo + p * 64 + lane;
                        if (a < r_hi) {
                            ok &= ld_gran(&xe[2 * (size_t)ci[p]], tag, d0[p]);
                            ok &= ld_gran(&xe[2 * (size_t)(ci[p] + 1)], tag, d1[p]);
                        }
                    }
                    if (__all(ok)) break;
                    if (cl_timed_out(spins, timeout)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            STAMP(4);
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, a = r_lo + rl;
                if (a < r_hi) {
                    const double dg = cA[p] * d0[p] + cB[p] * d1[p];
                    dpn[tb + a] = dg;
                    dVsh[e * RP + rl] = cu[p] * dr + cv[p] * ((cx[p] * dr + ze * dw) - dg);
                }
                xk[p] = nk[p]; xs[p] = ns[p];
            }
            STAMP(5);
            lds_barrier();
            if (g.CS > 1 && threadIdx.x == 0)   // every wave of this workgroup has finished reading period q
                __hip_atomic_store(&done[cluster * g.CS + member], (unsigned)(q + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            STAMP(6);
        }
    }
#ifdef HANK_STAMPS
    if (dbg && threadIdx.x == 0)
        for (int k = 0; k < 8; k++) dbg[blockIdx.x * 8 + k] = acc_[k];
#endif
}

// ---- forward tangent sweep -----------------------------------------------------------------------
// dynamic LDS: dDsh[n_e][RP] + midsh[n_e][RP] + red[16] + scratch row
template <int NPASS, int MAXT>
__global__ void __launch_bounds__(MAXT)
k_tanc_fwd(int n_a, int n_e, int G, int P, int N, ClGeom g, const double *__restrict__ Pi,
           const double *__restrict__ lw, const double *__restrict__ gD /* ig * D_{t-1} */,
           const double *__restrict__ Dseq, const double *__restrict__ pol, const int *__restrict__ start,
           const int *__restrict__ clo_, u64_t *xbuf, unsigned *done, unsigned *timeout,
           const double *__restrict__ dpol, double *__restrict__ aggpart /* [N][P][CS] */) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RP = NPASS * 64;
    double *dDsh = lds, *midsh = lds + (size_t)n_e * RP, *red = midsh + (size_t)n_e * RP;
    void *scratch = red + 16;
    const int lane = threadIdx.x & 63;
    const int e = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool helper = (e == n_e);
    int cluster, member;
    cl_ids(g, cluster, member);
    const int r_lo = member * g.RM;
    const int r_hi = min(n_a, r_lo + g.RM);
    const int nrows = max(0, r_hi - r_lo);
    u64_t *xb = xbuf + (size_t)cluster * CL_RING * G * 2;

    if (helper) {
        // barrier sequence per period: gather->mix, 2 inside the block reduction
        for (int n = cluster; n < N; n += g.nclusters) {
            const double *dpn = dpol + (size_t)n * P * G;
            lds_barrier();
            for (int t = 0; t < P; t++) {
                const int tp = t + CL_AHEAD;
                if (tp < P && nrows > 0) {
                    int sLo = 0, sHi = 0;
                    if (lane < n_e) {
                        const int *st = start + ((size_t)tp * n_e + lane) * (n_a + 1);
                        sLo = st[r_lo > 0 ? r_lo - 1 : 0];
                        sHi = st[r_hi];
                    }
                    for (int e2 = 0; e2 < n_e; e2++) {
                        const int lo = __shfl(sLo, e2, 64), hi = __shfl(sHi, e2, 64);
                        const size_t oc = (size_t)tp * G + (size_t)e2 * n_a;
                        if (hi > lo) {
                            touch_lines(lw + oc + lo, (size_t)(hi - lo) * 8, lane, scratch);
                            touch_lines(gD + oc + lo, (size_t)(hi - lo) * 8, lane, scratch);
                            touch_lines(dpn + oc + lo, (size_t)(hi - lo) * 8, lane, scratch);
                        }
                        touch_lines(pol + oc + r_lo, (size_t)nrows * 8, lane, scratch);
                        touch_lines(Dseq + oc + G + r_lo, (size_t)nrows * 8, lane, scratch);
                        touch_lines(dpn + oc + r_lo, (size_t)nrows * 8, lane, scratch);
                    }
                }
                lds_barrier();   // midsh complete
                lds_barrier();   // block reduction (1)
                lds_barrier();   // block reduction (2)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    double pic[16];   // column e of Pi: dD_t[r,e] = sum_k dD_mid[r,k] * Pi[k,e]
#pragma unroll
    for (int k = 0; k < 16; k++) pic[k] = k < n_e ? Pi[k + n_e * e] : 0.0;
    int q = 0;
    for (int n = cluster; n < N; n += g.nclusters) {
#pragma unroll
        for (int p = 0; p < NPASS; p++) dDsh[e * RP + p * 64 + lane] = 0.0;   // dD_0 = 0 (ForwardIteration.jl:293)
        const double *dpn = dpol + (size_t)n * P * G;
        lds_barrier();
        for (int t = 0; t < P; t++, q++) {
            const unsigned tag = (unsigned)q + 1u;
            const size_t tb = (size_t)t * G + (size_t)e * n_a;
            u64_t *xe = xb + ((size_t)(q % CL_RING) * G + (size_t)e * n_a) * 2;
            const double *Dnew = Dseq + tb + G;
            const int *st = start + ((size_t)t * n_e + e) * (n_a + 1);
            // peer-independent reads first
            int s0[NPASS], s1[NPASS], s2[NPASS];
            double cp[NPASS], cDn[NPASS], cdp[NPASS];
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int r = r_lo + p * 64 + lane;
                s0[p] = s1[p] = s2[p] = 0;
                if (r < r_hi) {
                    s1[p] = st[r]; s2[p] = st[r + 1]; s0[p] = r > 0 ? st[r - 1] : s1[p];
                    cp[p] = pol[tb + r]; cDn[p] = Dnew[r]; cdp[p] = dpn[tb + r];
                }
            }
            const int clo = clo_[(size_t)t * n_e + e];
            // publish dD_{t-1} of my slab
            cl_ring_ready(done, g, cluster, lane, q - (CL_RING - 1), timeout);
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, a = r_lo + rl;
                if (a < r_hi) st_gran(&xe[2 * (size_t)a], tag, dDsh[e * RP + rl]);
            }
            // lottery-segment gather of my target rows (per-lane re-read until the source row is current)
            unsigned spins = 0;
            bool dead = false;
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, r = r_lo + rl;
                if (r < r_hi) {
                    double acc = 0.0;
                    for (int j = s0[p]; j < s2[p]; j++) {
                        double v;
                        while (!ld_gran(&xe[2 * (size_t)j], tag, v) && !dead) {
                            dead = cl_timed_out(spins, timeout);
                            __builtin_amdgcn_s_sleep(1);
                        }
                        const double w = lw[tb + j], dwD = dpn[tb + j] * gD[tb + j];
                        acc += (j < s1[p]) ? (w * v + dwD) : ((1.0 - w) * v - dwD);
                    }
                    midsh[e * RP + rl] = acc;
                }
            }
            if (member == 0 && clo > 0) {   // the mass point: sum_{j<clo} dD_{t-1}[j] -> row 0, by the column's wave
                double s = 0.0;
                for (int j = lane; j < clo; j += 64) {
                    double v;
                    while (!ld_gran(&xe[2 * (size_t)j], tag, v) && !dead) {
                        dead = cl_timed_out(spins, timeout);
                        __builtin_amdgcn_s_sleep(1);
                    }
                    s += v;
                }
                s = wave_sum(s);
                if (lane == 0) midsh[e * RP] += s;
            }
            lds_barrier();
            double part = 0.0;
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int rl = p * 64 + lane, r = r_lo + rl;
                if (r < r_hi) {
                    double dDn = 0.0;
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        if (k < n_e) dDn += midsh[k * RP + rl] * pic[k];
                    dDsh[e * RP + rl] = dDn;
                    part += cp[p] * dDn + cdp[p] * cDn[p];
                }
            }
            // block reduction of the aggregate term over the n_e worker waves (fixed order)
            part = wave_sum(part);
            lds_barrier();
            if (lane == 0) red[e] = part;
            lds_barrier();
            if (threadIdx.x == 0) {
                double tot = 0.0;
                for (int k = 0; k < n_e; k++) tot += red[k];
                aggpart[((size_t)n * P + t) * g.CS + member] = tot;
                if (g.CS > 1)   // all gathers of period q by this workgroup have completed (they fed midsh)
                    __hip_atomic_store(&done[cluster * g.CS + member], (unsigned)(q + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
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
