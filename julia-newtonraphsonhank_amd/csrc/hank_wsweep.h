// hank_wsweep.h — the slab sweeps: the N partials of BackwardIteration / ForwardIteration as persistent launches in which
// ONE WAVE owns a slab of DW directions and a lane owns a wealth row with ALL its productivity columns.
//
// Same recurrences, record and group formation as the tangent sweeps of hank_xsweep.h (BackwardIteration.jl:90-113,
// ForwardIteration.jl:297-308 under Dual{Tag,Float64,N}; closed forms in DESIGN.md section 1). What changes is who owns what:
//   * workgroup c of an XCD's group still owns 63 wealth rows, but lane = row and the lane walks the n_e columns itself:
//     the n_e x n_e mixing (dE = dV' Pi^T backward, dD Pi forward) happens in the lane's registers — no LDS tile, no
//     workgroup barrier anywhere in the loop;
//   * wave k of the workgroup owns slab k = DW directions of the group's NW*DW; 8 groups x NW slabs x DW directions per
//     pass (256 at NW = 8, DW = 4). Slabs never exchange anything, so every wave is an independent recurrence: it
//     waits on ITS word of the source members' flag lines (one 128-B line per member, word k = wave k's episode), and the
//     CU's scheduler fills one slab's flag / L2 round trips with the other slabs' work. The record of a period is read
//     by a CU once per slab from its L1 / L2 instead of once per group of 4 directions from HBM;
//   * the state exchange (ds_t backward, dD_t forward) goes through the XCD's L2 as before: plain stores, sc1 loads,
//     per slab a ping-pong pair of [plane][row][2] buffers (16-byte lanes);
//   * forward: the sources a wave's 63 target rows draw from are one contiguous range (the policy is monotone); the wave
//     loads that range ONCE per column, 64 rows per instruction, forms each source's two lottery parts and stages them in
//     its private LDS strip; a target lane then adds its segments' parts from LDS in source order. Every source row is
//     fetched once per wave instead of once per target, with coalesced 16-byte lanes.
//   * what a period reads that is uniform over the wave (dr_t, dw_t, dtr_t of the slab, rho_t, the clamped prefix lengths,
//     the member ranges to wait for) is staged by the wave itself, WCH periods at a time, into a private LDS ring: loaded
//     a chunk ahead into registers, written half a chunk later — never a cold load on the critical path.
// Arithmetic: the expressions and summation orders of k_xtan_back / k_xtan_fwd, so dpol is bit-identical to every other
// schedule; the aggregate partial sums combine in a different order (rounding only).
#pragma once
#include "hank_xsweep.h"

namespace hank {

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
        __builtin_amdgcn_s_sleep(1);
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
    double *st;                 // [2][slabs][planes][G][2]
    double *dpol;               // [P][slabs][planes][G][2]
    const int *src, *rdr;       // [P][members] lo | hi << 8: members whose rows period t's gathers of member c read / that read member c's rows
};

// LDS need of the kernels (bytes), the same expressions the kernels carve up
static inline size_t wback_lds(int NEC, int DW, int NW) {
    return sizeof(double) * ((size_t)NEC * NEC + NEC + 4 + (size_t)NW * 2 * WCH * (6 * DW + 2));
}
static inline size_t wfwd_lds(int NEC, int DW, int NW) {
    return sizeof(double) * ((size_t)NEC * NEC + 4 + (size_t)NW * (2 * WCH * (NEC / 2 + 1) + (size_t)WSB * 64 * 2 * DW + 2 * DW));
}

template <int DW, int NEC, int MAXT>
__global__ void __launch_bounds__(MAXT) k_wtan_back(WTanBackArgs A) {
    constexpr int UE = 6 * DW + 2;                      // doubles per staged period
    constexpr int VPL = (WCH * UE + 63) / 64;
    constexpr int PL = WRows<DW>::PL;
    constexpr int EC = DW == 4 ? 2 : 4;                 // columns whose loads are in flight together
    extern __shared__ __attribute__((aligned(16))) double wl[];
    const Consts &c = A.c;
    const Record &R = A.R;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G;
    double *PT = wl;                                    // [NEC][NEC]: PT[e][k] = Pi[e, k] (the mixing of column e)
    double *zsh = PT + NEC * NEC;                       // [NEC]
    int *ctl = reinterpret_cast<int *>(zsh + NEC);      // [8]
    double *uni_all = zsh + NEC + 4;                    // per wave [2][WCH][UE]
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
    const int slab = x * A.NW + wv;
    if (slab * DW >= A.N) return;                       // (no workgroup barrier from here on: every wave runs alone)
    double *uni = uni_all + (size_t)wv * 2 * WCH * UE;
    const int a = cW * XRW + lane;
    const bool own = lane < XRW && a < na;
    const int ao = own ? a : 0;
    const double xa = c.a[ao];
    const unsigned nslab = (unsigned)(A.groups * A.NW), rows = (unsigned)G;
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
                } else {
                    int s = (Sact - 1) << 8, r = (Sact - 1) << 8;
                    const int tY = P - trip;
                    if (A.src && tY >= 0 && tY < P) s = A.src[(size_t)tY * Sact + cW];
                    if (A.rdr && tY + 1 >= 0 && tY + 1 < P) r = A.rdr[(size_t)(tY + 1) * Sact + cW];
                    val = __hiloint2double(r, s);
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
    int ibn[NEC];                                       // brackets of the next Y half (fetched a trip ahead)
#pragma unroll
    for (int e = 0; e < NEC; e++) ibn[e] = 0;
    if (own) {
#pragma unroll
        for (int e = 0; e < NEC; e++)
            if (e < ne) ibn[e] = R.ib[(size_t)(P - 1) * G + (size_t)e * na + a];
    }
    // sequence: X(P-1) | Y(P-1) X(P-2) | ... | Y(1) X(0) | Y(0); the wave publishes episode i+1 when the stores of trip i have drained
    for (int i = 0; i <= P; i++) {
        const int ci = i / WCH, ji = i - ci * WCH;
        if (ji == 0) stage_load(ci + 1);
        if (ji == WCH / 2) stage_write(ci + 1);
        const double *U = uni + ((size_t)(ci & 1) * WCH + ji) * UE;
        const double rng = U[6 * DW + 1];
        const int rs_ = __builtin_amdgcn_readfirstlane(__double2loint(rng)), rr_ = __builtin_amdgcn_readfirstlane(__double2hiint(rng));
        if (i > 0) {
            // ---- Y-tangent of period t: dg = A ds[ib] + B ds[ib+1]; dV = u dr + v ((a dr + z dw + dtr) - dg)
            const int t = P - i;
            const unsigned cur = (unsigned)((i - 1) & 1);
            if (!wpoll(A.sy, x, wv, rs_ & 255, (rs_ >> 8) & 255, (unsigned)i)) return;
            const size_t ro = (size_t)t * G + ao;
            double dr[DW], dw[DW], dt[DW];
#pragma unroll
            for (int d = 0; d < DW; d++) { dr[d] = U[d]; dw[d] = U[DW + d]; dt[d] = U[2 * DW + d]; }
#pragma unroll
            for (int e0 = 0; e0 < NEC; e0 += EC) {
                if (e0 < ne) {
                    double cA[EC], cB[EC], cu[EC], cv[EC], d0[EC][DW], d1[EC][DW];
#pragma unroll
                    for (int u = 0; u < EC; u++) {
                        const int e = e0 + u;
                        cA[u] = cB[u] = cu[u] = cv[u] = 0.0;
#pragma unroll
                        for (int d = 0; d < DW; d++) d0[u][d] = d1[u][d] = 0.0;
                        if (e < ne && own) {
                            const size_t p_ = ro + (size_t)e * na;
                            cA[u] = R.A[p_]; cB[u] = R.B[p_]; cu[u] = R.u[p_]; cv[u] = R.v[p_];
                            const unsigned el = cur * hs + sb0 + (unsigned)(e * na + ibn[e]);
                            st.load(el, rows, d0[u]);
                            st.load(el + 1, rows, d1[u]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < EC; u++) {
                        const int e = e0 + u;
                        if (e < ne) {
                            const double ze = zsh[e];
                            const bool live = cA[u] != 0.0 || cB[u] != 0.0;
                            double dg[DW];
#pragma unroll
                            for (int d = 0; d < DW; d++) {
                                dg[d] = live ? cA[u] * d0[u][d] + cB[u] * d1[u][d] : 0.0;
                                dV[e][d] = cu[u] * dr[d] + cv[u] * ((xa * dr[d] + (ze * dw[d] + dt[d])) - dg[d]);
                            }
                            if (own) wstream_store<DW>(A.dpol, (size_t)t * dts + dsl + (size_t)e * na + a, (size_t)G, dg);
                        }
                    }
                }
            }
        }
        if (i < P) {
            // ---- X-tangent of period tx: ds = kc dE - rho ((z dw + dtr) + s dr),  dE = dV' Pi^T in this lane's registers
            const int tx = P - 1 - i;
            double ck[NEC], cs[NEC];
#pragma unroll
            for (int e = 0; e < NEC; e++) {
                ck[e] = cs[e] = 0.0;
                if (e < ne && own) { ck[e] = R.kc[(size_t)tx * G + (size_t)e * na + a]; cs[e] = R.s[(size_t)tx * G + (size_t)e * na + a]; }
            }
            double dr1[DW], dw1[DW], dt1[DW];
#pragma unroll
            for (int d = 0; d < DW; d++) { dr1[d] = U[3 * DW + d]; dw1[d] = U[4 * DW + d]; dt1[d] = U[5 * DW + d]; }
            const double rho = U[6 * DW];
            // the half this trip overwrites was read by the gathers of trip i-1: every member that reads my rows has published i
            if (i >= 2 && !wpoll(A.sy, x, wv, rr_ & 255, (rr_ >> 8) & 255, (unsigned)i)) return;
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
                    const double ze = zsh[e];
#pragma unroll
                    for (int d = 0; d < DW; d++) ds[d] = ck[e] * mx[d] - rho * ((ze * dw1[d] + dt1[d]) + cs[e] * dr1[d]);
                    if (own) st.store((unsigned)(i & 1) * hs + sb0 + (unsigned)(e * na + a), rows, ds);
                }
            }
            wpublish(A.sy, x, cW, wv, (unsigned)(i + 1));
            if (own) {                  // brackets of the next trip's Y half (period tx): in flight while the sources are polled
#pragma unroll
                for (int e = 0; e < NEC; e++)
                    if (e < ne) ibn[e] = R.ib[(size_t)tx * G + (size_t)e * na + a];
            }
        }
    }
}

struct WTanFwdArgs {
    Consts c;
    Record R;                   // pol, seg, clo, lwg, Dseq of the recorded primal
    XSync *sy;
    double *st;                 // [2][slabs][planes][n_e*members*64][2]
    const double *dpol;         // [P][slabs][planes][G][2]
    int groups, NW, N;
    double *daggpart;           // [P][members][W]
    int W;                      // directions the pass's layout holds = slabs*DW
    const int *src, *rdr;       // [P][members] (forward ranges)
};

template <int DW, int NEC, int MAXT>
__global__ void __launch_bounds__(MAXT) k_wtan_fwd(WTanFwdArgs A) {
    constexpr int UE = NEC / 2 + 1;                     // doubles per staged period: the clamped prefix lengths (ints) + the ranges
    constexpr int VPL = (WCH * UE + 63) / 64;
    constexpr int PL = WRows<DW>::PL;
    extern __shared__ __attribute__((aligned(16))) double wl[];
    const Consts &c = A.c;
    const Record &R = A.R;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G;
    double *PT = wl;                                    // [NEC][NEC]: PT[e][k] = Pi[k, e]
    int *ctl = reinterpret_cast<int *>(PT + NEC * NEC); // [8]
    double *wave_all = PT + NEC * NEC + 4;
    constexpr size_t WLDS = 2 * WCH * UE + (size_t)WSB * 64 * 2 * DW + 2 * DW;
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
    const int slab = x * A.NW + wv;
    if (slab * DW >= A.N) return;
    double *uni = wave_all + (size_t)wv * WLDS;         // [2][WCH][UE]
    double *stg = uni + 2 * WCH * UE;                   // [WSB*64][2*DW]: per staged source its two lottery parts
    double *vts = stg + (size_t)WSB * 64 * 2 * DW;      // [2*DW]: the two parts of the virtual rows' sum
    const int r0 = cW * XRW, r = r0 + lane;
    const bool own = lane < XRW && r < na;
    const bool virt = lane == 63;                       // the member's virtual row (partial sums of row 0)
    const int nown = min(XRW, na - r0);
    const unsigned GM = (unsigned)(ne * Sact * 64);     // state rows per plane: [n_e][members][64]
    const unsigned nslab = (unsigned)(A.groups * A.NW);
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
        if (own || virt)
            for (int e = 0; e < ne; e++) st.store(sb0 + (unsigned)((e * Sact + cW) * 64 + lane), GM, z);
    }
    wpublish(A.sy, x, cW, wv, 1u);
    for (int t = 0; t < P; t++) {
        const int ci = t / WCH, ji = t - ci * WCH;
        if (ji == 0) stage_load(ci + 1);
        if (ji == WCH / 2) stage_write(ci + 1);
        const double *U = uni + ((size_t)(ci & 1) * WCH + ji) * UE;
        const double rng = U[NEC / 2];
        const int rs_ = __builtin_amdgcn_readfirstlane(__double2loint(rng)), rr_ = __builtin_amdgcn_readfirstlane(__double2hiint(rng));
        const bool vnz = ((rs_ >> 16) & 1) != 0;
        const size_t base = (size_t)t * G;
        const unsigned hb = (unsigned)(t & 1) * hs + sb0, hn = (unsigned)((t + 1) & 1) * hs + sb0;
        if (!wpoll(A.sy, x, wv, rs_ & 255, (rs_ >> 8) & 255, (unsigned)(t + 1))) return;   // the source members have published period t-1
        double acc[NEC][DW], polr[NEC], pd[DW];
#pragma unroll
        for (int d = 0; d < DW; d++) pd[d] = 0.0;
#pragma unroll
        for (int e = 0; e < NEC; e++) {
            polr[e] = 0.0;
#pragma unroll
            for (int d = 0; d < DW; d++) acc[e][d] = 0.0;
            if (e < ne) {
                const double cpair = U[e / 2];
                const int clo_ = __builtin_amdgcn_readfirstlane((e & 1) ? __double2hiint(cpair) : __double2loint(cpair));
                const int clo = min(max(clo_, 0), na);
                const size_t cb = base + (size_t)e * na;
                const unsigned se = (unsigned)(e * Sact) * 64u;      // this column's rows of the state
                // own-row record: the target's segments, its policy, D_t and its policy partials (the aggregate)
                int4 sg = make_int4(0, 0, 0, 0);
                double Dr = 0.0, dpr[DW];
#pragma unroll
                for (int d = 0; d < DW; d++) dpr[d] = 0.0;
                if (own) {
                    sg = R.seg[cb + r];
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
                if ((own && r < clo) || (virt && clo > 0 && vnz)) st.load(hb + se + (unsigned)(cW * 64 + lane), GM, cT);
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
                            if (j < S1 && j < na) {
                                const double2 wg = R.lwg[cb + j];
                                double dDj[DW], dpj[DW];
                                const int jm = j / XRW;
                                st.load(hb + se + (unsigned)(jm * 64 + (j - jm * XRW)), GM, dDj);
                                wstream_load<DW>(A.dpol, (size_t)t * dts + dsl + (size_t)e * na + j, (size_t)G, dpj);
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
                            double *q = stg + (size_t)(b * 64 + lane) * 2 * DW;
                            if constexpr (DW == 1) {
                                *reinterpret_cast<double2 *>(q) = make_double2(c1[0], c0[0]);
                            } else {
#pragma unroll
                                for (int d = 0; d < DW; d += 2) {
                                    *reinterpret_cast<double2 *>(q + d) = make_double2(c1[d], c1[d + 1]);
                                    *reinterpret_cast<double2 *>(q + DW + d) = make_double2(c0[d], c0[d + 1]);
                                }
                            }
                        }
                    }
                    const int jlo = max(s0, R0), jhi = min(s2, R0 + WSB * 64);
                    for (int j = jlo; j < jhi; j++) {
                        const double *q = stg + (size_t)(j - R0) * 2 * DW + (j < s1 ? 0 : DW);
                        if constexpr (DW == 1) {
                            acc[e][0] += q[0];
                        } else {
#pragma unroll
                            for (int d = 0; d < DW; d += 2) {
                                const double2 v = *reinterpret_cast<const double2 *>(q + d);
                                acc[e][d] += v.x; acc[e][d + 1] += v.y;
                            }
                        }
                    }
                }
                if (need_vT && own && s0 == 0 && s2 > 0) {
                    const double *q = vts + (0 < s1 ? 0 : DW);
#pragma unroll
                    for (int d = 0; d < DW; d++) acc[e][d] += q[d];
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
            }
        }
        // every member that read my rows in period t-1 has published it: the other half may be overwritten
        if (t >= 1 && !wpoll(A.sy, x, wv, rr_ & 255, (rr_ >> 8) & 255, (unsigned)(t + 1))) return;
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
                if (live) {
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
        wpublish(A.sy, x, cW, wv, (unsigned)(t + 2));
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
