// hank_wsweep.h — the slab sweeps: the N partials of BackwardIteration / ForwardIteration as persistent launches in which
// ONE WAVE owns a slab of DW directions, a lane owns a wealth row with ALL its productivity columns, and a workgroup keeps
// the loop-carried state of its own rows in LDS.
//
// Same recurrences, record and group formation as the tangent sweeps of hank_xsweep.h (BackwardIteration.jl:90-113,
// ForwardIteration.jl:297-308 under Dual{Tag,Float64,N}; closed forms in DESIGN.md section 1). What changes is who owns what:
//   * workgroup c of an XCD's group still owns 63 wealth rows, but lane = row and the lane walks the n_e columns itself:
//     the n_e x n_e mixing (dE = dV' Pi^T backward, dD Pi forward) happens in the lane's registers — no LDS tile, no
//     barrier between gather and mixing;
//   * wave k of the workgroup owns slab k = DW directions of the group's NW*DW; 8 groups x NW slabs x DW directions per
//     pass. Slabs never exchange anything: every wave waits on ITS word of the source members' flag lines (one 128-B line
//     per member, word k = wave k's episode), and the CU's scheduler fills one slab's round trips with the other slabs' work;
//   * the loop-carried state of a wave's OWN rows (ds_t backward, dD_t forward) stays in a wave-private LDS strip; only
//     the rows some OTHER member reads — the halo: about a third of the rows at 63-row slabs of the 2000-point grid — are
//     also stored to the ping-pong buffer in the XCD's L2 (plain stores, sc1 loads, as in hank_xsweep.h). With every row
//     exchanged through L2 a pass of 256 directions keeps 2 x 5.6 MB per XCD alive in a 4 MB L2: every state byte went out
//     to the fabric and came back (profiles/r03w1: 4x the algorithmic traffic, 4 TB/s of fabric traffic); the halo of 128
//     directions is 2 x 1 MB per XCD;
//   * the coefficients of a period (the recorded linearisation of the member's 63 x n_e points) are fetched ONCE per
//     workgroup by a loader wave, a period ahead, into a two-slot LDS ring; the slab waves read them from LDS. One workgroup
//     barrier per period hands the slot over;
//   * forward: the sources a wave's 63 target rows draw from are one contiguous range (the policy is monotone); the wave
//     loads that range ONCE per column, 64 rows per instruction (own rows from the strip, halo rows from L2), forms each
//     source's two lottery parts and stages them in its private LDS strip; a target lane then adds its segments' parts in
//     source order. The loads of the next two columns are in flight while a column is summed;
//   * what a period reads that is uniform over the wave (dr_t, dw_t, dtr_t of the slab, rho_t, the clamped prefix lengths,
//     the member ranges to wait for, the publish thresholds) is staged by the wave itself, WCH periods at a time, into a
//     private LDS ring: loaded a chunk ahead into registers, written half a chunk later.
// Arithmetic: the expressions and summation orders of k_xtan_back / k_xtan_fwd, so dpol is bit-identical to every other
// schedule; the aggregate partial sums combine in a different order (rounding only).
#pragma once
#include "hank_xsweep.h"

namespace hank {

#ifndef HANK_WSLEEP
#define HANK_WSLEEP 1      // dev knob: s_sleep units (64 clocks) between two polls of a wave
#endif
constexpr int WCH = 8;            // periods per staged chunk of the wave-uniform inputs (power of two)
constexpr int WSB = 2;            // forward: batches of 64 source rows per round of a column

typedef unsigned int xv2u __attribute__((ext_vector_type(2)));

// rows of DW partials kept as DW/2 planes of 16-byte pairs (DW = 1: 8-byte elements); element index = index of the
// row's first plane, planes `ps` elements apart. Loads are sc1 (they bypass the CU's L1 and are served by the XCD's L2,
// where the plain stores of the other members left the lines).
template <int DW>
struct WRows {
    static constexpr int PL = DW >= 2 ? DW / 2 : 1;
    static constexpr int EB = DW >= 2 ? 16 : 8;      // bytes per element
    __amdgpu_buffer_rsrc_t rs;
    double *base;
    __device__ __forceinline__ void init(double *p, size_t elements) {
        base = p;
        rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(elements * EB), 0x00020000);
    }
    __device__ __forceinline__ void load(unsigned el, unsigned ps, double *v) const {
        if constexpr (DW == 1) {
            const xv2u q = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(el * 8u), 0, 16);      // aux 16 = sc1
            v[0] = __hiloint2double((int)q.y, (int)q.x);
        } else {
#pragma unroll
            for (int k = 0; k < PL; k++) {
                const xv4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((el + k * ps) * 16u), 0, 16);
                v[2 * k] = __hiloint2double((int)q.y, (int)q.x);
                v[2 * k + 1] = __hiloint2double((int)q.w, (int)q.z);
            }
        }
    }
    // the same for the lanes with `on`; the others get zeros without a memory access (their offset is out of the buffer's
    // range: the hardware bounds check returns zero) — no branch around a partly needed load
    __device__ __forceinline__ void load_if(bool on, unsigned el, unsigned ps, double *v) const {
        if constexpr (DW == 1) {
            const xv2u q = __builtin_amdgcn_raw_buffer_load_b64(rs, on ? (int)(el * 8u) : (int)0xfffffff0u, 0, 16);
            v[0] = __hiloint2double((int)q.y, (int)q.x);
        } else {
#pragma unroll
            for (int k = 0; k < PL; k++) {
                const xv4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, on ? (int)((el + k * ps) * 16u) : (int)0xfffffff0u, 0, 16);
                v[2 * k] = __hiloint2double((int)q.y, (int)q.x);
                v[2 * k + 1] = __hiloint2double((int)q.w, (int)q.z);
            }
        }
    }
    __device__ __forceinline__ void store(unsigned el, unsigned ps, const double *v) const {      // plain: the line stays in this L2
        if constexpr (DW == 1) {
            base[el] = v[0];
        } else {
#pragma unroll
            for (int k = 0; k < PL; k++) reinterpret_cast<double2 *>(base)[(size_t)el + (size_t)k * ps] = make_double2(v[2 * k], v[2 * k + 1]);
        }
    }
};
// the policy partials: a pure stream (written once backward, read once forward), element layout as above
template <int DW>
__device__ __forceinline__ void wstream_store(double *p, size_t el, size_t ps, const double *v) {
    if constexpr (DW == 1) {
        p[el] = v[0];
    } else {
#pragma unroll
        for (int k = 0; k < DW / 2; k++) reinterpret_cast<double2 *>(p)[el + k * ps] = make_double2(v[2 * k], v[2 * k + 1]);
    }
}
template <int DW>
__device__ __forceinline__ void wstream_load(const double *p, size_t el, size_t ps, double *v) {
    if constexpr (DW == 1) {
        v[0] = p[el];
    } else {
#pragma unroll
        for (int k = 0; k < DW / 2; k++) { const double2 q = reinterpret_cast<const double2 *>(p)[el + k * ps]; v[2 * k] = q.x; v[2 * k + 1] = q.y; }
    }
}
// a row of the wave's LDS strip ([row][DW], 8*DW-byte rows)
template <int DW>
__device__ __forceinline__ void wstrip_load(const double *q, double *v) {
    if constexpr (DW == 1) {
        v[0] = q[0];
    } else {
#pragma unroll
        for (int k = 0; k < DW / 2; k++) { const double2 w = reinterpret_cast<const double2 *>(q)[k]; v[2 * k] = w.x; v[2 * k + 1] = w.y; }
    }
}
template <int DW>
__device__ __forceinline__ void wstrip_store(double *q, const double *v) {
    if constexpr (DW == 1) {
        q[0] = v[0];
    } else {
#pragma unroll
        for (int k = 0; k < DW / 2; k++) reinterpret_cast<double2 *>(q)[k] = make_double2(v[2 * k], v[2 * k + 1]);
    }
}

// warm a line in the XCD's L2 without holding a register for it: a load the compiler does not count, into one scratch
// register the caller keeps alive (and drains with s_waitcnt vmcnt(0)) — cdna_hip_programming.md section 5.7
__device__ __forceinline__ void wtouch(const void *p, unsigned &scratch) {
    asm volatile("global_load_dword %0, %1, off" : "+v"(scratch) : "v"(p) : "memory");
}

// wave k of member c waits until word k of the flag lines of members [lo, hi] has reached `need` (bounded; false = gave up)
__device__ __forceinline__ bool wpoll(XSync *sy, int x, int k, int lo, int hi, unsigned need) {
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; spins++) {
        const unsigned f = (lane >= lo && lane <= hi) ? xldu(&sy->flag[x][lane][k]) : need;
        if (__all((int)(f - need) >= 0)) return true;
        if (spins > XSPIN_LIMIT || ((spins & 255u) == 255u && xldu(&sy->status[0]) != 0u)) {
            if (lane == 0) xfail(sy, XERR_TIMEOUT, x);
            return false;
        }
        __builtin_amdgcn_s_sleep(HANK_WSLEEP);
    }
}
// this wave's stores have reached L2 -> its episode becomes visible to the waves k of the other members
__device__ __forceinline__ void wpublish(XSync *sy, int x, int c, int k, unsigned episode) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) *reinterpret_cast<volatile unsigned *>(&sy->flag[x][c][k]) = episode;
}

struct WTanBackArgs {
    Consts c;
    Record R;                   // s, kc, ib, A, B, u, v of the recorded primal
    const double *rho;          // [P] 1/(1+r_t)
    const double *dxr, *dxw, *dxt;      // [P][Ntot]
    int Ntot, n0, N;            // this pass: directions [n0, n0+N) of the batch
    int groups, NW;             // slabs = groups*NW, slab = x*NW + wave
    XSync *sy;
    double *st;                 // [2][slabs][planes][G][2]: the halo rows
    double *dpol;               // [P][slabs][planes][G][2]
    const int *src, *rdr;       // [P][members] lo | hi << 8: members whose rows period t's gathers of member c read / that read member c's rows
    const int2 *pub;            // [P][members][16]: member c stores its row a of column e to L2 iff a <= .x or a >= .y (k_wpub_back)
};

// LDS need of the kernels (bytes), the same expressions the kernels carve up
static inline size_t wback_lds(int ne, int NEC, int DW, int NW) {
    return sizeof(double) * ((size_t)NEC * NEC + NEC + 4 + (size_t)NW * (2 * WCH * (6 * DW + 2 + NEC) + (size_t)ne * 64 * DW)) + (size_t)2 * ne * 64 * 36;
}
static inline size_t wfwd_lds(int ne, int NEC, int DW, int NW) {
    return sizeof(double) * ((size_t)NEC * NEC + 4 + (size_t)NW * (2 * WCH * (NEC / 2 + 1) + (size_t)WSB * 64 * 2 * DW + 2 * DW + (size_t)ne * 64 * DW)) + (size_t)2 * ne * 64 * 16;
}

template <int DW, int NEC, int MAXT>
__global__ void __launch_bounds__(MAXT) k_wtan_back(WTanBackArgs A) {
    constexpr int UE = 6 * DW + 2 + NEC;                // doubles per staged period
    constexpr int VPL = (WCH * UE + 63) / 64;
    constexpr int PL = WRows<DW>::PL;
    constexpr int EC = (DW == 4 && MAXT > 256) ? 2 : 4; // columns whose halo loads are in flight together (registers)
    extern __shared__ __attribute__((aligned(16))) double wl[];
    const Consts &c = A.c;
    const Record &R = A.R;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G, NW = A.NW;
    double *PT = wl;                                    // [NEC][NEC]: PT[e][k] = Pi[e, k] (the mixing of column e)
    double *zsh = PT + NEC * NEC;                       // [NEC]
    int *ctl = reinterpret_cast<int *>(zsh + NEC);      // [8]
    double *wave_all = zsh + NEC + 4;                   // per wave: uni [2][WCH][UE], strip [ne][64][DW]
    const size_t WLDS = (size_t)2 * WCH * UE + (size_t)ne * 64 * DW;
    double *rec = wave_all + (size_t)NW * WLDS;         // [2 slots]: A, B, u, v [4][ne][64] doubles, then ib [ne][64] ints
    const size_t RSLOT = (size_t)ne * 64 * 36 / 8;      // doubles per slot
    for (int k = threadIdx.x; k < NEC * NEC; k += blockDim.x) {
        const int e = k / NEC, kk = k - e * NEC;
        PT[k] = (e < ne && kk < ne) ? c.Pi[e + ne * kk] : 0.0;
    }
    for (int k = threadIdx.x; k < NEC; k += blockDim.x) zsh[k] = k < ne ? c.z[k] : 0.0;
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x >= A.groups) return;
    const int Sact = (na + XRW - 1) / XRW;
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }
    if (cW >= Sact) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r0 = cW * XRW, a = r0 + lane;
    const bool own = lane < XRW && a < na;
    const int ao = own ? a : 0;
    if (wv == NW) {
        // ---- the loader wave: the Y coefficients of trip i+1 (period P-1-i) go into slot (i+1) & 1 during trip i. They were
        // loaded into registers a trip earlier (a whole trip of latency budget); the lines the slab waves load themselves
        // a trip later (kc, s) are touched into the XCD's L2 meanwhile.
        double vA[NEC], vB[NEC], vu[NEC], vv[NEC];
        int vi[NEC];
        unsigned scratch = 0;
        auto fetch = [&](int per) {
#pragma unroll
            for (int e = 0; e < NEC; e++) {
                vA[e] = vB[e] = vu[e] = vv[e] = 0.0; vi[e] = 0;
                if (e < ne && own && per >= 0) {
                    const size_t p_ = (size_t)per * G + (size_t)e * na + a;
                    vA[e] = R.A[p_]; vB[e] = R.B[p_]; vu[e] = R.u[p_]; vv[e] = R.v[p_]; vi[e] = R.ib[p_];
                }
            }
        };
        fetch(P - 1);
        const int son = x == 0 ? (cW == 0 ? 0 : (cW == Sact / 3 ? 1 : -1)) : -1;
        (void)son;
        for (int i = 0; i < P; i++) {
            double *sl = rec + (size_t)((i + 1) & 1) * RSLOT;
            int *sli = reinterpret_cast<int *>(sl + (size_t)4 * ne * 64);
#pragma unroll
            for (int e = 0; e < NEC; e++) {
                if (e < ne) {
                    const int o_ = e * 64 + lane;
                    sl[o_] = vA[e]; sl[ne * 64 + o_] = vB[e]; sl[2 * ne * 64 + o_] = vu[e]; sl[3 * ne * 64 + o_] = vv[e]; sli[o_] = vi[e];
                }
            }
            XSTAMPW(0, son, i, 7, NW);
            fetch(P - 2 - i);
            if (own && P - 2 - i >= 0) {
                for (int e = 0; e < ne; e++) {
                    wtouch(R.kc + (size_t)(P - 2 - i) * G + (size_t)e * na + a, scratch);
                    wtouch(R.s + (size_t)(P - 2 - i) * G + (size_t)e * na + a, scratch);
                }
            }
            XSTAMPW(0, son, i, 8, NW);
            xlds_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(scratch) : : "memory");
        return;
    }
    const int slab = x * NW + wv;
    if (slab * DW >= A.N) return;                       // (an idle slab: the workgroup's barrier counts the live waves only)
    double *uni = wave_all + (size_t)wv * WLDS;
    double *strip = uni + 2 * WCH * UE;                 // [ne][64][DW]: ds_t of this wave's own rows
    const int nown = min(XRW, na - r0);
    const double xa = c.a[ao];
    const unsigned nslab = (unsigned)(A.groups * NW), rows = (unsigned)G;
    WRows<DW> st;
    st.init(A.st, (size_t)2 * nslab * PL * rows);
    const unsigned hs = nslab * PL * rows;              // the other half of the ping-pong state
    const unsigned sb0 = (unsigned)slab * PL * rows;
    const size_t dts = (size_t)nslab * PL * G, dsl = (size_t)slab * PL * G;

    double sv[VPL];
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int v = 0; v < VPL; v++) {
            const int idx = lane + 64 * v;
            double val = 0.0;
            const int j = idx / UE, f = idx - j * UE, trip = chunk * WCH + j;
            if (idx < WCH * UE && trip <= P) {
                if (f < 6 * DW) {
                    const int half = f / (3 * DW), ff = f - half * 3 * DW, which = ff / DW, d = ff - which * DW;
                    const int per = half == 0 ? P - trip : P - 1 - trip, n = slab * DW + d;
                    if (per >= 0 && per < P && n < A.N && (which < 2 || c.n_hh > 2)) {
                        const double *sp = which == 0 ? A.dxr : (which == 1 ? A.dxw : A.dxt);
                        val = sp[(size_t)per * A.Ntot + A.n0 + n];
                    }
                } else if (f == 6 * DW) {
                    if (P - 1 - trip >= 0) val = A.rho[P - 1 - trip];
                } else if (f == 6 * DW + 1) {
                    int s = (Sact - 1) << 8, r = (Sact - 1) << 8;
                    const int tY = P - trip;
                    if (A.src && tY >= 0 && tY < P) s = A.src[(size_t)tY * Sact + cW];
                    if (A.rdr && tY + 1 >= 0 && tY + 1 < P) r = A.rdr[(size_t)(tY + 1) * Sact + cW];
                    val = __hiloint2double(r, s);
                } else {                                // who reads my rows of period tx's knots: next trip's gathers
                    const int e = f - (6 * DW + 2), tx = P - 1 - trip;
                    int2 pb = make_int2(na, -1);        // (no thresholds: store every row)
                    if (A.pub && tx >= 0 && e < ne) pb = A.pub[((size_t)tx * Sact + cW) * 16 + e];
                    val = __hiloint2double(pb.y, pb.x);
                }
            }
            sv[v] = val;
        }
    };
    auto stage_write = [&](int chunk) {
#pragma unroll
        for (int v = 0; v < VPL; v++) {
            const int idx = lane + 64 * v;
            if (idx < WCH * UE) uni[(chunk & 1) * WCH * UE + idx] = sv[v];
        }
    };
    stage_load(0);
    stage_write(0);

    double dV[NEC][DW];
#pragma unroll
    for (int e = 0; e < NEC; e++)
#pragma unroll
        for (int d = 0; d < DW; d++) dV[e][d] = 0.0;    // dV_T = 0 (BackwardIteration.jl:85)
    // sequence: X(P-1) | Y(P-1) X(P-2) | ... | Y(1) X(0) | Y(0); the wave publishes episode i+1 when the stores of trip i have drained
    const int son = (x == 0 && wv == 0) ? (cW == 0 ? 0 : (cW == Sact / 3 ? 1 : -1)) : -1;     // dev stamps (make stamp)
    (void)son;
    for (int i = 0; i <= P; i++) {
        XSTAMP(0, son, i, 0);
        const int ci = i / WCH, ji = i - ci * WCH;
        if (ji == 0) stage_load(ci + 1);
        if (ji == WCH / 2) stage_write(ci + 1);
        const double *U = uni + ((size_t)(ci & 1) * WCH + ji) * UE;
        const double rng = U[6 * DW + 1];
        const int rs_ = __builtin_amdgcn_readfirstlane(__double2loint(rng)), rr_ = __builtin_amdgcn_readfirstlane(__double2hiint(rng));
        // the X half's coefficients (period P-1-i): issued now, used after the Y half
        double ck[NEC], cs[NEC];
#pragma unroll
        for (int e = 0; e < NEC; e++) {
            ck[e] = cs[e] = 0.0;
            if (i < P && e < ne && own) { ck[e] = R.kc[(size_t)(P - 1 - i) * G + (size_t)e * na + a]; cs[e] = R.s[(size_t)(P - 1 - i) * G + (size_t)e * na + a]; }
        }
        if (i > 0) {
            // ---- Y-tangent of period t: dg = A ds[ib] + B ds[ib+1]; dV = u dr + v ((a dr + z dw + dtr) - dg)
            const int t = P - i;
            const unsigned cur = (unsigned)((i - 1) & 1);
            if (!wpoll(A.sy, x, wv, rs_ & 255, (rs_ >> 8) & 255, (unsigned)i)) return;
            XSTAMP(0, son, i, 1);
            const double *sl = rec + (size_t)(i & 1) * RSLOT;
            const int *sli = reinterpret_cast<const int *>(sl + (size_t)4 * ne * 64);
            double dr[DW], dw[DW], dt[DW];
#pragma unroll
            for (int d = 0; d < DW; d++) { dr[d] = U[d]; dw[d] = U[DW + d]; dt[d] = U[2 * DW + d]; }
#pragma unroll
            for (int e0 = 0; e0 < NEC; e0 += EC) {
                if (e0 < ne) {
                    double cA[EC], cB[EC], d0[EC][DW], d1[EC][DW];
#pragma unroll
                    for (int u = 0; u < EC; u++) {
                        const int e = e0 + u;
                        cA[u] = cB[u] = 0.0;
#pragma unroll
                        for (int d = 0; d < DW; d++) d0[u][d] = d1[u][d] = 0.0;
                        if (e < ne && own) {
                            const int o_ = e * 64 + lane;
                            cA[u] = sl[o_]; cB[u] = sl[ne * 64 + o_];
                            const int ib = sli[o_], q0 = ib - r0, q1 = q0 + 1;
                            const unsigned el = cur * hs + sb0 + (unsigned)(e * na + ib);
                            // own rows from the strip, the halo from the XCD's L2 (a dead point — flat region, blocked max — reads nothing:
                            // nobody stored those rows for it)
                            if (cA[u] != 0.0 || cB[u] != 0.0) {
                                if (q0 >= 0 && q0 < nown) wstrip_load<DW>(strip + ((size_t)e * 64 + q0) * DW, d0[u]);
                                else st.load(el, rows, d0[u]);
                                if (q1 >= 0 && q1 < nown) wstrip_load<DW>(strip + ((size_t)e * 64 + q1) * DW, d1[u]);
                                else st.load(el + 1, rows, d1[u]);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < EC; u++) {
                        const int e = e0 + u;
                        if (e < ne) {
                            const int o_ = e * 64 + lane;
                            const double ze = zsh[e], cu = sl[2 * ne * 64 + o_], cv = sl[3 * ne * 64 + o_];
                            const bool live = cA[u] != 0.0 || cB[u] != 0.0;
                            double dg[DW];
#pragma unroll
                            for (int d = 0; d < DW; d++) {
                                dg[d] = live ? cA[u] * d0[u][d] + cB[u] * d1[u][d] : 0.0;
                                dV[e][d] = cu * dr[d] + cv * ((xa * dr[d] + (ze * dw[d] + dt[d])) - dg[d]);
                            }
                            if (own) wstream_store<DW>(A.dpol, (size_t)t * dts + dsl + (size_t)e * na + a, (size_t)G, dg);
                        }
                    }
                }
            }
        }
        XSTAMP(0, son, i, 2);
        if (i < P) {
            // ---- X-tangent of period tx: ds = kc dE - rho ((z dw + dtr) + s dr),  dE = dV' Pi^T in this lane's registers
            double dr1[DW], dw1[DW], dt1[DW];
#pragma unroll
            for (int d = 0; d < DW; d++) { dr1[d] = U[3 * DW + d]; dw1[d] = U[4 * DW + d]; dt1[d] = U[5 * DW + d]; }
            const double rho = U[6 * DW];
            // the half this trip overwrites was read by the gathers of trip i-1: every member that reads my rows has published i
            if (i >= 2 && !wpoll(A.sy, x, wv, rr_ & 255, (rr_ >> 8) & 255, (unsigned)i)) return;
            XSTAMP(0, son, i, 3);
#pragma unroll
            for (int e = 0; e < NEC; e++) {
                if (e < ne) {
                    double mx[DW], ds[DW];
#pragma unroll
                    for (int k = 0; k < NEC; k++) {
                        if (k < ne) {
                            const double p = PT[e * NEC + k];
#pragma unroll
                            for (int d = 0; d < DW; d++) mx[d] = k == 0 ? p * dV[k][d] : mx[d] + p * dV[k][d];
                        }
                    }
                    const double ze = zsh[e], pbv = U[6 * DW + 2 + e];
                    const int pbL = __double2loint(pbv), pbH = __double2hiint(pbv);
#pragma unroll
                    for (int d = 0; d < DW; d++) ds[d] = ck[e] * mx[d] - rho * ((ze * dw1[d] + dt1[d]) + cs[e] * dr1[d]);
                    if (own) {
                        wstrip_store<DW>(strip + ((size_t)e * 64 + lane) * DW, ds);
                        if (a <= pbL || a >= pbH) st.store((unsigned)(i & 1) * hs + sb0 + (unsigned)(e * na + a), rows, ds);
                    }
                }
            }
            XSTAMP(0, son, i, 4);
            wpublish(A.sy, x, cW, wv, (unsigned)(i + 1));
            XSTAMP(0, son, i, 5);
            xlds_barrier();                             // the loader has landed the next trip's coefficients
            XSTAMP(0, son, i, 6);
        }
    }
}

// ---- version 3 of the backward slab sweep: the columns are a ROLLED loop -------------------------------------------------
// The fully unrolled column bodies of k_wtan_back are 45-130 KB of code per instance (two CUs share a 64 KB instruction
// cache) and keep dV of every column in registers (96 VGPRs at DW = 4). Here a column's dV takes the place of its ds_t in the
// wave's LDS strip the moment the column's gathers have been read (same wave: LDS operations execute in order), the X half
// loads the lane's own row back (all columns: the mixing is register-blocked per lane) and writes ds_{t-1} in place. The
// record is read straight from global memory, software-pipelined two columns ahead; the halo rows one column ahead (a buffer
// load whose lane offset is out of range returns zero without touching memory: no branch around the in-range lanes).
// No loader wave, no workgroup barrier: every wave is on its own.
static inline size_t vback_lds(int ne, int NEC, int DW, int NW) {
    return sizeof(double) * ((size_t)NEC * NEC + NEC + 4 + (size_t)NW * (2 * WCH * (6 * DW + 2 + NEC) + (size_t)ne * 64 * DW));
}

template <int DW, int NEC, int MAXT>
__global__ void __launch_bounds__(MAXT) k_vtan_back(WTanBackArgs A) {
    constexpr int UE = 6 * DW + 2 + NEC;
    constexpr int VPL = (WCH * UE + 63) / 64;
    constexpr int PL = WRows<DW>::PL;
    extern __shared__ __attribute__((aligned(16))) double wl[];
    const Consts &c = A.c;
    const Record &R = A.R;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G, NW = A.NW;
    double *PT = wl;                                    // [NEC][NEC]: PT[e][k] = Pi[e, k], zero beyond n_e
    double *zsh = PT + NEC * NEC;
    int *ctl = reinterpret_cast<int *>(zsh + NEC);
    double *wave_all = zsh + NEC + 4;
    const size_t WLDS = (size_t)2 * WCH * UE + (size_t)ne * 64 * DW;
    for (int k = threadIdx.x; k < NEC * NEC; k += blockDim.x) {
        const int e = k / NEC, kk = k - e * NEC;
        PT[k] = (e < ne && kk < ne) ? c.Pi[e + ne * kk] : 0.0;
    }
    for (int k = threadIdx.x; k < NEC; k += blockDim.x) zsh[k] = k < ne ? c.z[k] : 0.0;
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x >= A.groups) return;
    const int Sact = (na + XRW - 1) / XRW;
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }
    if (cW >= Sact) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slab = x * NW + wv;
    if (slab * DW >= A.N) return;
    const int r0 = cW * XRW, nown = min(XRW, na - r0);
    const bool own = lane < nown;
    const int a = r0 + (own ? lane : nown - 1);         // (idle lanes shadow the last row: no branches around them, their stores are masked)
    double *uni = wave_all + (size_t)wv * WLDS;
    double *strip = uni + 2 * WCH * UE;                 // [ne][64][DW]
    const double xa = c.a[a];
    const unsigned nslab = (unsigned)(A.groups * NW), rows = (unsigned)G;
    WRows<DW> st;
    st.init(A.st, (size_t)2 * nslab * PL * rows);
    const unsigned hs = nslab * PL * rows, sb0 = (unsigned)slab * PL * rows;
    const size_t dts = (size_t)nslab * PL * G, dsl = (size_t)slab * PL * G;

    double sv[VPL];
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int v = 0; v < VPL; v++) {
            const int idx = lane + 64 * v;
            double val = 0.0;
            const int j = idx / UE, f = idx - j * UE, trip = chunk * WCH + j;
            if (idx < WCH * UE && trip <= P) {
                if (f < 6 * DW) {
                    const int half = f / (3 * DW), ff = f - half * 3 * DW, which = ff / DW, d = ff - which * DW;
                    const int per = half == 0 ? P - trip : P - 1 - trip, n = slab * DW + d;
                    if (per >= 0 && per < P && n < A.N && (which < 2 || c.n_hh > 2)) {
                        const double *sp = which == 0 ? A.dxr : (which == 1 ? A.dxw : A.dxt);
                        val = sp[(size_t)per * A.Ntot + A.n0 + n];
                    }
                } else if (f == 6 * DW) {
                    if (P - 1 - trip >= 0) val = A.rho[P - 1 - trip];
                } else if (f == 6 * DW + 1) {
                    int s = (Sact - 1) << 8, r = (Sact - 1) << 8;
                    const int tY = P - trip;
                    if (A.src && tY >= 0 && tY < P) s = A.src[(size_t)tY * Sact + cW];
                    if (A.rdr && tY + 1 >= 0 && tY + 1 < P) r = A.rdr[(size_t)(tY + 1) * Sact + cW];
                    val = __hiloint2double(r, s);
                } else {
                    const int e = f - (6 * DW + 2), tx = P - 1 - trip;
                    int2 pb = make_int2(na, -1);
                    if (A.pub && tx >= 0 && e < ne) pb = A.pub[((size_t)tx * Sact + cW) * 16 + e];
                    val = __hiloint2double(pb.y, pb.x);
                }
            }
            sv[v] = val;
        }
    };
    auto stage_write = [&](int chunk) {
#pragma unroll
        for (int v = 0; v < VPL; v++) {
            const int idx = lane + 64 * v;
            if (idx < WCH * UE) uni[(chunk & 1) * WCH * UE + idx] = sv[v];
        }
    };
    stage_load(0);
    stage_write(0);
    {   // dV_T = 0 (BackwardIteration.jl:85)
        double z[DW];
#pragma unroll
        for (int d = 0; d < DW; d++) z[d] = 0.0;
        for (int e = 0; e < ne; e++) wstrip_store<DW>(strip + ((size_t)e * 64 + lane) * DW, z);
    }
    struct Co { double A, B, u, v; int ib; };
    for (int i = 0; i <= P; i++) {
        const int ci = i / WCH, ji = i - ci * WCH;
        if (ji == 0) stage_load(ci + 1);
        if (ji == WCH / 2) stage_write(ci + 1);
        const double *U = uni + ((size_t)(ci & 1) * WCH + ji) * UE;
        const double rng = U[6 * DW + 1];
        const int rs_ = __builtin_amdgcn_readfirstlane(__double2loint(rng)), rr_ = __builtin_amdgcn_readfirstlane(__double2hiint(rng));
        if (i > 0) {
            // ---- Y-tangent of period t, column by column: dg = A ds[ib] + B ds[ib+1]; dV = u dr + v ((a dr + z dw + dtr) - dg)
            const int t = P - i;
            const unsigned cur = (unsigned)((i - 1) & 1) * hs + sb0;
            const size_t ro = (size_t)t * G + a;
            auto fetch = [&](int e) {
                Co q;
                const size_t p_ = ro + (size_t)(e < ne ? e : ne - 1) * na;
                q.A = R.A[p_]; q.B = R.B[p_]; q.u = R.u[p_]; q.v = R.v[p_]; q.ib = R.ib[p_];
                return q;
            };
            double h0[DW], h1[DW];                      // the halo rows of the NEXT column (zero where the row is the wave's own)
            auto halo = [&](int e, const Co &q, double *o0, double *o1) {
                const int q0 = q.ib - r0, q1 = q0 + 1;
                const bool live = q.A != 0.0 || q.B != 0.0;
                const unsigned el = cur + (unsigned)(e * na + q.ib);
                st.load_if(live && !(q0 >= 0 && q0 < nown), el, rows, o0);
                st.load_if(live && !(q1 >= 0 && q1 < nown), el + 1, rows, o1);
            };
            if (!wpoll(A.sy, x, wv, rs_ & 255, (rs_ >> 8) & 255, (unsigned)i)) return;
            double dr[DW], dw[DW], dt[DW], xdr[DW];
#pragma unroll
            for (int d = 0; d < DW; d++) { dr[d] = U[d]; dw[d] = U[DW + d]; dt[d] = U[2 * DW + d]; xdr[d] = xa * dr[d]; }
            Co c0 = fetch(0), c1 = fetch(1);
            halo(0, c0, h0, h1);
            for (int e = 0; e < ne; e++) {
                const Co c2 = fetch(e + 2);
                double n0[DW], n1[DW];
                halo(e + 1 < ne ? e + 1 : e, c1, n0, n1);
                // this column: own rows from the strip, the rest from the halo loads issued a column ago
                const int q0 = c0.ib - r0, q1 = q0 + 1;
                const bool in0 = q0 >= 0 && q0 < nown, in1 = q1 >= 0 && q1 < nown;
                double s0[DW], s1[DW], dg[DW], dV[DW];
                wstrip_load<DW>(strip + ((size_t)e * 64 + (in0 ? q0 : 0)) * DW, s0);
                wstrip_load<DW>(strip + ((size_t)e * 64 + (in1 ? q1 : 0)) * DW, s1);
                const bool live = c0.A != 0.0 || c0.B != 0.0;
                const double ze = zsh[e];
#pragma unroll
                for (int d = 0; d < DW; d++) {
                    const double d0 = in0 ? s0[d] : h0[d], d1 = in1 ? s1[d] : h1[d];
                    dg[d] = live ? c0.A * d0 + c0.B * d1 : 0.0;
                    dV[d] = c0.u * dr[d] + c0.v * ((xdr[d] + (ze * dw[d] + dt[d])) - dg[d]);
                }
                wstrip_store<DW>(strip + ((size_t)e * 64 + lane) * DW, dV);      // (after the gathers of this column: same wave, in order)
                if (own) wstream_store<DW>(A.dpol, (size_t)t * dts + dsl + (size_t)e * na + a, (size_t)G, dg);
#pragma unroll
                for (int d = 0; d < DW; d++) { h0[d] = n0[d]; h1[d] = n1[d]; }
                c0 = c1; c1 = c2;
            }
        }
        if (i < P) {
            // ---- X-tangent of period tx: the lane's own row of dV (all columns) back into registers, ds in place
            const int tx = P - 1 - i;
            double v[NEC][DW];
#pragma unroll
            for (int k = 0; k < NEC; k++) {
                if (k < ne) wstrip_load<DW>(strip + ((size_t)k * 64 + lane) * DW, v[k]);
                else {
#pragma unroll
                    for (int d = 0; d < DW; d++) v[k][d] = 0.0;
                }
            }
            double dr1[DW], dw1[DW], dt1[DW];
#pragma unroll
            for (int d = 0; d < DW; d++) { dr1[d] = U[3 * DW + d]; dw1[d] = U[4 * DW + d]; dt1[d] = U[5 * DW + d]; }
            const double rho = U[6 * DW];
            const size_t rx = (size_t)tx * G + a;
            double ck = R.kc[rx], cs = R.s[rx];
            if (i >= 2 && !wpoll(A.sy, x, wv, rr_ & 255, (rr_ >> 8) & 255, (unsigned)i)) return;
            for (int e = 0; e < ne; e++) {
                const size_t pn = rx + (size_t)(e + 1 < ne ? e + 1 : e) * na;
                const double ckn = R.kc[pn], csn = R.s[pn];
                double mx[DW], ds[DW];
                const double *pt = PT + e * NEC;
#pragma unroll
                for (int k = 0; k < NEC; k++) {
                    const double p = pt[k];
#pragma unroll
                    for (int d = 0; d < DW; d++) mx[d] = k == 0 ? p * v[k][d] : mx[d] + p * v[k][d];
                }
                const double ze = zsh[e], pbv = U[6 * DW + 2 + e];
                const int pbL = __double2loint(pbv), pbH = __double2hiint(pbv);
#pragma unroll
                for (int d = 0; d < DW; d++) ds[d] = ck * mx[d] - rho * ((ze * dw1[d] + dt1[d]) + cs * dr1[d]);
                wstrip_store<DW>(strip + ((size_t)e * 64 + lane) * DW, ds);
                if (own && (a <= pbL || a >= pbH)) st.store((unsigned)(i & 1) * hs + sb0 + (unsigned)(e * na + a), rows, ds);
                ck = ckn; cs = csn;
            }
            wpublish(A.sy, x, cW, wv, (unsigned)(i + 1));
        }
    }
}

struct WTanFwdArgs {
    Consts c;
    Record R;                   // pol, seg, clo, lwg, Dseq of the recorded primal
    XSync *sy;
    double *st;                 // [2][slabs][planes][n_e*members*64][2]: the halo rows
    const double *dpol;         // [P][slabs][planes][G][2]
    int groups, NW, N;
    double *daggpart;           // [P][members][W]
    int W;                      // directions the pass's layout holds = slabs*DW
    const int *src, *rdr;       // [P][members] (forward ranges)
    int all_rows;               // dev: 1 = store every row to L2 (no halo selection)
};

template <int DW, int NEC, int MAXT>
__global__ void __launch_bounds__(MAXT) k_wtan_fwd(WTanFwdArgs A) {
    constexpr int UE = NEC / 2 + 1;                     // doubles per staged period: the clamped prefix lengths (ints) + the ranges
    constexpr int VPL = (WCH * UE + 63) / 64;
    constexpr int PL = WRows<DW>::PL;
    constexpr int PF = (DW == 4 && MAXT > 256) ? 1 : 2; // columns whose source loads are in flight ahead of the one being summed (registers)
    extern __shared__ __attribute__((aligned(16))) double wl[];
    const Consts &c = A.c;
    const Record &R = A.R;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G, NW = A.NW;
    double *PT = wl;                                    // [NEC][NEC]: PT[e][k] = Pi[k, e]
    int *ctl = reinterpret_cast<int *>(PT + NEC * NEC); // [8]
    double *wave_all = PT + NEC * NEC + 4;
    const size_t WLDS = (size_t)2 * WCH * UE + (size_t)WSB * 64 * 2 * DW + 2 * DW + (size_t)ne * 64 * DW;
    int4 *rec = reinterpret_cast<int4 *>(wave_all + (size_t)NW * WLDS);       // [2 slots][ne][64]: the target rows' segments
    for (int k = threadIdx.x; k < NEC * NEC; k += blockDim.x) {
        const int e = k / NEC, kk = k - e * NEC;
        PT[k] = (e < ne && kk < ne) ? c.Pi[ne * e + kk] : 0.0;
    }
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x >= A.groups) return;
    const int Sact = (na + XRW - 1) / XRW;
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }
    if (cW >= Sact) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r0 = cW * XRW, r = r0 + lane;
    const bool own = lane < XRW && r < na;
    const bool virt = lane == 63;                       // the member's virtual row (partial sums of row 0)
    if (wv == NW) {
        // ---- the loader wave: the segments of the member's target rows, a period ahead (slot t & 1 holds period t; loaded
        // into registers a period earlier still), and the lines the slab waves read a period later — the sources' lottery
        // records and policy partials, the rows' policy and D_t — touched into the XCD's L2: every row is some member's own
        // row, so the source rows the neighbours gather next period are there when they ask.
        int4 vs[NEC];
        unsigned scratch = 0;
        const int nslab_ = A.groups * NW, PLl = WRows<DW>::PL;
        auto fetch = [&](int t) {
#pragma unroll
            for (int e = 0; e < NEC; e++) {
                vs[e] = make_int4(0, 0, 0, -1);
                if (e < ne && own && t < P) vs[e] = R.seg[(size_t)t * G + (size_t)e * na + r];
            }
        };
        auto touch = [&](int t) {
            if (!own || t >= P) return;
            for (int e = 0; e < ne; e++) {
                const size_t o_ = (size_t)t * G + (size_t)e * na + r;
                wtouch(R.lwg + o_, scratch);
                wtouch(R.pol + o_, scratch);
                wtouch(R.Dseq + o_ + G, scratch);
                for (int k = 0; k < NW; k++) {
                    const int sk = x * NW + k;
                    if (sk * DW >= A.N) break;
                    for (int q = 0; q < PLl; q++)
                        wtouch(A.dpol + ((((size_t)t * nslab_ + sk) * PLl + q) * G + (size_t)e * na + r) * (DW >= 2 ? 2 : 1), scratch);
                }
            }
        };
        fetch(0);
        touch(0);
        const int son = x == 0 ? (cW == 0 ? 0 : (cW == Sact / 3 ? 1 : -1)) : -1;
        (void)son;
        for (int t = 0; t <= P; t++) {
            if (t < P) {
                int4 *sl = rec + (size_t)(t & 1) * ne * 64;
#pragma unroll
                for (int e = 0; e < NEC; e++)
                    if (e < ne) sl[e * 64 + lane] = vs[e];
                fetch(t + 1);
                touch(t + 1);
            }
            XSTAMPW(1, son, t, 8, NW);
            xlds_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(scratch) : : "memory");
        return;
    }
    const int slab = x * NW + wv;
    if (slab * DW >= A.N) return;
    double *uni = wave_all + (size_t)wv * WLDS;         // [2][WCH][UE]
    double *stg = uni + 2 * WCH * UE;                   // [WSB*64][2*DW]: per staged source its two lottery parts
    double *vts = stg + (size_t)WSB * 64 * 2 * DW;      // [2*DW]: the two parts of the virtual rows' sum
    double *strip = vts + 2 * DW;                       // [ne][64][DW]: dD_t of this wave's own rows (slot 63: the virtual row)
    const int nown = min(XRW, na - r0);
    const unsigned GM = (unsigned)(ne * Sact * 64);     // state rows per plane: [n_e][members][64]
    const unsigned nslab = (unsigned)(A.groups * NW);
    WRows<DW> st;
    st.init(A.st, (size_t)2 * nslab * PL * GM);
    const unsigned hs = nslab * PL * GM, sb0 = (unsigned)slab * PL * GM;
    const size_t dts = (size_t)nslab * PL * G, dsl = (size_t)slab * PL * G;

    double sv[VPL];
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int v = 0; v < VPL; v++) {
            const int idx = lane + 64 * v;
            double val = 0.0;
            const int j = idx / UE, f = idx - j * UE, t = chunk * WCH + j;
            if (idx < WCH * UE && t < P) {
                if (f < NEC / 2) {
                    const int e0 = 2 * f;
                    const int c0 = e0 < ne ? R.clo[(size_t)t * ne + e0] : 0, c1 = e0 + 1 < ne ? R.clo[(size_t)t * ne + e0 + 1] : 0;
                    val = __hiloint2double(c1, c0);
                } else {
                    int s = (Sact - 1) << 8, rd = (Sact - 1) << 8;
                    if (A.src) s = A.src[(size_t)t * Sact + cW];
                    if (A.rdr && t > 0) rd = A.rdr[(size_t)(t - 1) * Sact + cW];
                    int vnz = 0;                        // the virtual rows may hold mass: some column was clamped last period
                    if (t > 0)
                        for (int k = 0; k < ne; k++) vnz |= R.clo[(size_t)(t - 1) * ne + k] > 0 ? 1 : 0;
                    val = __hiloint2double(rd, s | (vnz << 16));
                }
            }
            sv[v] = val;
        }
    };
    auto stage_write = [&](int chunk) {
#pragma unroll
        for (int v = 0; v < VPL; v++) {
            const int idx = lane + 64 * v;
            if (idx < WCH * UE) uni[(chunk & 1) * WCH * UE + idx] = sv[v];
        }
    };
    stage_load(0);
    stage_write(0);
    {   // the initial distribution carries no partials (ForwardIteration.jl:293): episode 1
        double z[DW];
#pragma unroll
        for (int d = 0; d < DW; d++) z[d] = 0.0;
        for (int e = 0; e < ne; e++) {
            wstrip_store<DW>(strip + ((size_t)e * 64 + lane) * DW, z);
            if (own || virt) st.store(sb0 + (unsigned)((e * Sact + cW) * 64 + lane), GM, z);
        }
    }
    wpublish(A.sy, x, cW, wv, 1u);
    xlds_barrier();                                     // period 0's segments have landed
    const int son = (x == 0 && wv == 0) ? (cW == 0 ? 0 : (cW == Sact / 3 ? 1 : -1)) : -1;     // dev stamps (make stamp)
    (void)son;
    for (int t = 0; t < P; t++) {
        XSTAMP(1, son, t, 0);
        const int ci = t / WCH, ji = t - ci * WCH;
        if (ji == 0) stage_load(ci + 1);
        if (ji == WCH / 2) stage_write(ci + 1);
        const double *U = uni + ((size_t)(ci & 1) * WCH + ji) * UE;
        const double rng = U[NEC / 2];
        const int rs_ = __builtin_amdgcn_readfirstlane(__double2loint(rng)), rr_ = __builtin_amdgcn_readfirstlane(__double2hiint(rng));
        const bool vnz = ((rs_ >> 16) & 1) != 0;
        const size_t base = (size_t)t * G;
        const unsigned hb = (unsigned)(t & 1) * hs + sb0, hn = (unsigned)((t + 1) & 1) * hs + sb0;
        const int4 *sl = rec + (size_t)(t & 1) * ne * 64;
        if (!wpoll(A.sy, x, wv, rs_ & 255, (rs_ >> 8) & 255, (unsigned)(t + 1))) return;   // the source members have published period t-1
        XSTAMP(1, son, t, 1);
        double acc[NEC][DW], polr[NEC], pd[DW];
        // who reads this lane's rows of dD_t? The members of its two targets under period t+1's lottery (seg.w of period t+1 =
        // the row's bracket as a source there; -1: clamped at the first grid point, summed by this member itself)
        int lon[NEC];
#pragma unroll
        for (int e = 0; e < NEC; e++) {
            lon[e] = -1;
            if (e < ne && own && t + 1 < P) lon[e] = reinterpret_cast<const int *>(R.seg + base + G + (size_t)e * na + r)[3];
        }
#pragma unroll
        for (int d = 0; d < DW; d++) pd[d] = 0.0;
        // what one column's sources need, fetched PF columns ahead of the column being summed
        double2 pwg[PF + 1][WSB];
        double pdD[PF + 1][WSB][DW], pdp[PF + 1][WSB][DW];
        auto issue = [&](int e, int q) {                // column e into pipeline slot q
            const int4 sg = sl[e * 64 + lane];
            const int s0w = __builtin_amdgcn_readlane(max(sg.x, 0), 0), s1w = __builtin_amdgcn_readlane(min(sg.z, na), nown - 1);
            const size_t cb = base + (size_t)e * na;
            const unsigned se = (unsigned)(e * Sact) * 64u;
#pragma unroll
            for (int b = 0; b < WSB; b++) {
                const int j = s0w + b * 64 + lane;
                pwg[q][b] = make_double2(0.0, 0.0);
#pragma unroll
                for (int d = 0; d < DW; d++) pdD[q][b][d] = pdp[q][b][d] = 0.0;
                if (j < s1w) {
                    pwg[q][b] = R.lwg[cb + j];
                    const int jq = j - r0;
                    if (jq >= 0 && jq < nown) wstrip_load<DW>(strip + ((size_t)e * 64 + jq) * DW, pdD[q][b]);
                    else { const int jm = j / XRW; st.load(hb + se + (unsigned)(jm * 64 + (j - jm * XRW)), GM, pdD[q][b]); }
                    wstream_load<DW>(A.dpol, (size_t)t * dts + dsl + (size_t)e * na + j, (size_t)G, pdp[q][b]);
                }
            }
        };
#pragma unroll
        for (int e = 0; e < PF; e++)
            if (e < ne) issue(e, e % (PF + 1));
#pragma unroll
        for (int e = 0; e < NEC; e++) {
            polr[e] = 0.0;
#pragma unroll
            for (int d = 0; d < DW; d++) acc[e][d] = 0.0;
            if (e < ne) {
                if (e + PF < ne) issue(e + PF, (e + PF) % (PF + 1));
                const int qs = e % (PF + 1);
                const double cpair = U[e / 2];
                const int clo_ = __builtin_amdgcn_readfirstlane((e & 1) ? __double2hiint(cpair) : __double2loint(cpair));
                const int clo = min(max(clo_, 0), na);
                const size_t cb = base + (size_t)e * na;
                const unsigned se = (unsigned)(e * Sact) * 64u;      // this column's rows of the state
                // own-row record: the target's segments, its policy, D_t and its policy partials (the aggregate)
                const int4 sg = sl[e * 64 + lane];
                double Dr = 0.0, dpr[DW];
#pragma unroll
                for (int d = 0; d < DW; d++) dpr[d] = 0.0;
                if (own) {
                    polr[e] = R.pol[cb + r];
                    Dr = R.Dseq[cb + G + r];
                    wstream_load<DW>(A.dpol, (size_t)t * dts + dsl + (size_t)e * na + r, (size_t)G, dpr);
                } else if (virt) {
                    polr[e] = R.pol[cb];                             // a virtual row carries row 0's policy
                }
                const int s0 = own ? max(sg.x, 0) : 0, s1 = sg.y, s2 = own ? min(sg.z, na) : 0;
                // the mass point's inputs (this member's clamped rows, its own virtual row) and the virtual rows' sum
                double cT[DW], vT[DW];
#pragma unroll
                for (int d = 0; d < DW; d++) cT[d] = vT[d] = 0.0;
                if ((own && r < clo) || (virt && clo > 0 && vnz)) wstrip_load<DW>(strip + ((size_t)e * 64 + lane) * DW, cT);
                const bool need_vT = vnz && clo == 0 && __any(own && sg.x <= 0 && s2 > 0);
                if (need_vT && lane < Sact) st.load(hb + se + (unsigned)(lane * 64 + 63), GM, vT);
                double sbv[DW];
#pragma unroll
                for (int d = 0; d < DW; d++) sbv[d] = 0.0;
                if (need_vT) {
#pragma unroll
                    for (int d = 0; d < DW; d++) {
                        const double sv_ = xwave_reduce63(vT[d]);       // valid in lane 63
                        sbv[d] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sv_), 63), __builtin_amdgcn_readlane(__double2loint(sv_), 63));
                    }
                }
                // the wave's source range: segments are monotone in the target row
                const int S0 = __builtin_amdgcn_readlane(s0, 0), S1 = __builtin_amdgcn_readlane(s2, nown - 1);
                for (int R0 = S0; R0 < S1; R0 += WSB * 64) {
#pragma unroll
                    for (int b = 0; b < WSB; b++) {
                        const int j = R0 + b * 64 + lane;
                        if (R0 + b * 64 < S1) {
                            double c1[DW], c0[DW];
#pragma unroll
                            for (int d = 0; d < DW; d++) c1[d] = c0[d] = 0.0;
                            if (j < S1) {
                                double2 wg;
                                double dDj[DW], dpj[DW];
                                if (R0 == S0) {         // the first round comes out of the pipeline
                                    wg = pwg[qs][b];
#pragma unroll
                                    for (int d = 0; d < DW; d++) { dDj[d] = pdD[qs][b][d]; dpj[d] = pdp[qs][b][d]; }
                                } else {
                                    wg = R.lwg[cb + j];
                                    const int jq = j - r0;
                                    if (jq >= 0 && jq < nown) wstrip_load<DW>(strip + ((size_t)e * 64 + jq) * DW, dDj);
                                    else { const int jm = j / XRW; st.load(hb + se + (unsigned)(jm * 64 + (j - jm * XRW)), GM, dDj); }
                                    wstream_load<DW>(A.dpol, (size_t)t * dts + dsl + (size_t)e * na + j, (size_t)G, dpj);
                                }
                                const double w1 = wg.x, w0 = 1.0 - wg.x;
#pragma unroll
                                for (int d = 0; d < DW; d++) {
                                    c1[d] = w1 * dDj[d] + wg.y * dpj[d];
                                    c0[d] = w0 * dDj[d] - wg.y * dpj[d];
                                }
                                if (j == 0 && need_vT) {            // source row 0 also moves what sits on the virtual rows
#pragma unroll
                                    for (int d = 0; d < DW; d++) { vts[d] = w1 * sbv[d]; vts[DW + d] = w0 * sbv[d]; }
                                }
                            }
                            double *qd = stg + (size_t)(b * 64 + lane) * 2 * DW;
                            if constexpr (DW == 1) {
                                *reinterpret_cast<double2 *>(qd) = make_double2(c1[0], c0[0]);
                            } else {
#pragma unroll
                                for (int d = 0; d < DW; d += 2) {
                                    *reinterpret_cast<double2 *>(qd + d) = make_double2(c1[d], c1[d + 1]);
                                    *reinterpret_cast<double2 *>(qd + DW + d) = make_double2(c0[d], c0[d + 1]);
                                }
                            }
                        }
                    }
                    const int jlo = max(s0, R0), jhi = min(s2, R0 + WSB * 64);
                    for (int j = jlo; j < jhi; j++) {
                        const double *qd = stg + (size_t)(j - R0) * 2 * DW + (j < s1 ? 0 : DW);
                        if constexpr (DW == 1) {
                            acc[e][0] += qd[0];
                        } else {
#pragma unroll
                            for (int d = 0; d < DW; d += 2) {
                                const double2 v = *reinterpret_cast<const double2 *>(qd + d);
                                acc[e][d] += v.x; acc[e][d + 1] += v.y;
                            }
                        }
                    }
                }
                if (need_vT && own && s0 == 0 && s2 > 0) {
                    const double *qd = vts + (0 < s1 ? 0 : DW);
#pragma unroll
                    for (int d = 0; d < DW; d++) acc[e][d] += qd[d];
                }
                // the mass point (see k_xprimal_fwd): the member's clamped rows go to ITS virtual row
                if (clo > r0) {
#pragma unroll
                    for (int d = 0; d < DW; d++) cT[d] = xwave_reduce63(cT[d]);
                }
                if (virt) {
#pragma unroll
                    for (int d = 0; d < DW; d++) acc[e][d] = cT[d];
                }
                if (own) {
#pragma unroll
                    for (int d = 0; d < DW; d++) pd[d] += dpr[d] * Dr;     // dpol_t D_t of the aggregate (post-transition D_t, :301-307)
                }
                if (e == 0) XSTAMP(1, son, t, 7);
            }
        }
        XSTAMP(1, son, t, 2);
        // every member that read my rows in period t-1 has published it: the other half may be overwritten
        if (t >= 1 && !wpoll(A.sy, x, wv, rr_ & 255, (rr_ >> 8) & 255, (unsigned)(t + 1))) return;
        XSTAMP(1, son, t, 3);
        const bool live = own || virt;
#pragma unroll
        for (int e = 0; e < NEC; e++) {
            if (e < ne) {
                double mx[DW];
#pragma unroll
                for (int k = 0; k < NEC; k++) {
                    if (k < ne) {
                        const double p = PT[e * NEC + k];
#pragma unroll
                        for (int d = 0; d < DW; d++) mx[d] = k == 0 ? p * acc[k][d] : mx[d] + p * acc[k][d];
                    }
                }
                wstrip_store<DW>(strip + ((size_t)e * 64 + lane) * DW, mx);
                if (live) {
                    if (virt || A.all_rows || (lon[e] >= 0 && (lon[e] / XRW != cW || (lon[e] + 1) / XRW != cW)))
                        st.store(hn + (unsigned)((e * Sact + cW) * 64 + lane), GM, mx);
#pragma unroll
                    for (int d = 0; d < DW; d++) pd[d] += polr[e] * mx[d];
                }
            }
        }
#pragma unroll
        for (int d = 0; d < DW; d++) {
            const double s = xwave_reduce63(pd[d]);
            if (lane == 63 && slab * DW + d < A.N) A.daggpart[((size_t)t * Sact + cW) * A.W + slab * DW + d] = s;
        }
        XSTAMP(1, son, t, 4);
        wpublish(A.sy, x, cW, wv, (unsigned)(t + 2));
        XSTAMP(1, son, t, 5);
        xlds_barrier();                                 // the loader has landed the next period's segments
        XSTAMP(1, son, t, 6);
    }
}

// readers of member m's rows in period t = the members whose source range holds m (inverse of k_xsrc_*); one thread per (t, m)
__global__ void k_wrdr(const int *src, int P, int Sact, int *rdr) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * Sact) return;
    const int t = idx / Sact, m = idx - t * Sact;
    int lo = m, hi = m;
    for (int cc = 0; cc < Sact; cc++) {
        const int s = src[(size_t)t * Sact + cc], l = s & 255, h = (s >> 8) & 255;
        if (l <= m && m <= h) { lo = min(lo, cc); hi = max(hi, cc); }
    }
    rdr[idx] = lo | (hi << 8);
}

// backward sweep: which of its rows of period t's knots does member m have to store to L2? The gathers of member m' in
// column e read the rows [gl, gh] (the brackets of its live points, monotone in m'); a row of m is read by some LOWER
// member iff it is <= max_{m' < m} gh, by some HIGHER member iff it is >= min_{m' > m} gl. One block per period.
__global__ void k_wpub_back(Consts c, Record R, int Sact, int2 *pub) {
    __shared__ int gl[16][64], gh[16][64];
    const int t = blockIdx.x, ne = c.n_e, na = c.n_a;
    for (int k = threadIdx.x; k < ne * Sact; k += blockDim.x) {
        const int e = k / Sact, m = k - e * Sact;
        const int r0 = m * XRW, rows = min(XRW, na - r0);
        int lo = 1 << 30, hi = -1;
        const size_t ro = (size_t)t * c.G + (size_t)e * na + r0;
        for (int q = 0; q < rows; q++)
            if (R.A[ro + q] != 0.0 || R.B[ro + q] != 0.0) { const int ib = R.ib[ro + q]; lo = min(lo, ib); hi = max(hi, ib + 1); }
        gl[e][m] = lo; gh[e][m] = hi;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ne * Sact; k += blockDim.x) {
        const int e = k / Sact, m = k - e * Sact;
        int pl = -1, ph = 1 << 30;
        for (int q = 0; q < m; q++) pl = max(pl, gh[e][q]);
        for (int q = m + 1; q < Sact; q++) ph = min(ph, gl[e][q]);
        pub[((size_t)t * Sact + m) * 16 + e] = make_int2(pl, ph);
    }
}

// (G,P,N) col-major export of one pass's dpol [P][slabs][planes][G][2] into columns [n0, n0+N)
__global__ void k_wexport_dpol(const double *dpol, int G, int P, int nslab, int DW, int n0, int N, double *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)G * P * N;
    if (idx >= total) return;
    const size_t n = idx / ((size_t)G * P), rem = idx - n * (size_t)G * P, t = rem / G, pt = rem - t * G;
    const int slab = (int)n / DW, d = (int)n - slab * DW;
    const int PLn = DW >= 2 ? DW / 2 : 1, q = d / 2, h = DW >= 2 ? d & 1 : 0, w = DW >= 2 ? 2 : 1;
    out[((size_t)(n0 + n) * P + t) * G + pt] = dpol[((((size_t)t * nslab + slab) * PLn + q) * G + pt) * w + h];
}

}  // namespace hank
