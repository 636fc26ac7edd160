// hank_wide.h — the state-on-chip tangent sweeps for WIDE batches: ONE workgroup per tangent direction.
//
// The same two linear recurrences as k_tan_back / k_tan_fwd (hank_kernels.h) and k_xtan_back / k_xfwd (hank_xsweep.h) — the N
// partials of BackwardIteration.jl:90-113 and ForwardIteration.jl:297-308 at a recorded primal — for batches wide enough to give
// every CU a direction of its own (N >= ~100). What the other two families pay per period is the round trip of the loop-carried
// state: a launch per period reads and writes it through the fabric (132 / 192 MB per launch for 45 MB algorithmic at N = 256,
// profiles/r04n256_*), the XCD-local sweeps exchange it through an L2 behind flags (a latency chain of ~6 us per period). Here
// the state of a direction never leaves its CU:
//   * a workgroup owns ONE direction and the WHOLE n_a x n_e grid. Thread i owns the wealth rows i, i + NT, i + 2 NT, ... (R of
//     them) and ALL n_e columns of each: R * n_e doubles of loop-carried state — the first 7 columns in VGPRs, the rest (n_e = 11:
//     4 of them) in thread-private LDS slots (the registers are needed for the record that is on its way; a private slot costs
//     two LDS accesses per period).
//   * the n_e x n_e mixing (V' Pi' backward, D Pi forward) is lane-local: a row's n_e numbers are one lane's, the matrix
//     streams through scalar registers from the kernel arguments. No LDS tile, no barrier.
//   * what does cross lanes — the bracket gather ds[ib], ds[ib+1] backward, the lottery's two-target push forward — goes through a
//     double-buffered LDS COLUMN (n_a entries: 16 KB / 32 KB at n_a = 2000), one workgroup barrier per column. The forward push
//     is a gather over the recorded source segments (seg: the policy is monotone), so the sums have a fixed order.
//   * no flags, no cross-workgroup wait, no state bytes on the fabric: workgroups are independent, any grid size is legal.
//   * the record of a period is read once per CU from L2 (all CUs of an XCD read the same lines at about the same time),
//     software-pipelined one column ahead (buffer loads: scalar base + a 32-bit lane offset shared by every array); the policy
//     partials stream out / in with nontemporal accesses, one contiguous n_e * n_a * 8-byte block per direction and period:
//     dpol is laid out [P][N][n_e][n_a] here.
// Per period and CU the record costs ~44 (backward) / ~56 (forward) bytes per grid point from L2 for 8 bytes of algorithmic
// traffic: the family wins where N fills the chip, and loses below (the XCD-local sweeps share one record read between D = 4
// directions and 32 CUs).
// Arithmetic: the expressions of k_xtan_back / tan_fwd_body, with fused multiply-adds allowed (this is the issue-bound family;
// results agree with the other two to rounding, not bit for bit).
#pragma once
#include "hank_xsweep.h"

namespace hank {

constexpr int WIDE_R = 4;          // wealth rows per thread
constexpr int WIDE_MAXT = 512;     // threads per workgroup: 8 waves, 2 per SIMD -> 256 VGPRs per lane
constexpr int WIDE_KREG = 16;      // columns of the state kept in registers; any beyond live in thread-private LDS slots (none at n_e <= 16:
                                   // the single-buffered record stages below left room for the whole state)
constexpr int wide_kl(int ne) { return ne > WIDE_KREG ? ne - WIDE_KREG : 0; }
// the mixing matrix as the kernel walks it — m[k*NE + e] = the coefficient of input column k in output column e (backward: Pi[e, k]
// = P(e -> k), i.e. column-major Pi itself; forward: Pi[k, e] = P(k -> e), its transpose) — and the productivity grid. Passed BY
// VALUE: the kernel-argument segment is read with scalar loads.
template <int NE> struct WMat { double m[NE * NE]; double z[NE]; };

__device__ __forceinline__ double wide_uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// A value is computed HERE: without the pin the compiler sinks a computation with a single later use (a predicated store, the next
// period's mixing) down to that use — past the loads issued in between, whose registers then overlap the operands' — and spills.
__device__ __forceinline__ void wide_pin(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ int wide_opaque_zero(int t, int q) { return __builtin_amdgcn_readfirstlane((t >> (27 + (q & 3))) & 1); }

// y = M x for one grid row, in place: x's n_e numbers are this lane's registers, the n_e^2 coefficients stream through SCALAR
// registers, a row of coefficients per input column, the next one on its way while this one is used (all 121 at once would need
// 242 SGPRs — and the compiler, seeing the same loop-invariant loads in every row and period, hoists them all and spills: `zero`
// is a zero it cannot see through, so each row's loads stay where they are used).
// Order of the sums: k ascending, first term unrounded-added (as xtile_mix_reg).
template <int NE>
__device__ __forceinline__ void wide_mix(double (&x)[NE], const double *m, int zero) {
#pragma clang fp contract(fast)
    double y[NE], cf[2][NE];
    int off = zero;
#pragma unroll
    for (int e = 0; e < NE; e++) cf[0][e] = m[off + e];
#pragma unroll
    for (int k = 0; k < NE; k++) {
        if (k + 1 < NE) {
            // (the empty asm ties the loads to this point of the row's arithmetic: as free-floating scalar loads the instruction
            // selector lines all of a period's up at the top of the block, and they spill)
            if (k == 0) asm volatile("" : "+s"(off) : "v"(x[0]));
            else asm volatile("" : "+s"(off) : "v"(y[0]));
#pragma unroll
            for (int e = 0; e < NE; e++) cf[(k + 1) & 1][e] = m[off + (k + 1) * NE + e];
        }
        const double xk = x[k];
#pragma unroll
        for (int e = 0; e < NE; e++) y[e] = k == 0 ? cf[0][e] * xk : y[e] + cf[k & 1][e] * xk;
    }
    // (the results are pinned HERE: each has one use, inside a column's code, and the compiler would sink the whole sum — and
    // every coefficient, as a spilled scalar — down to it)
#pragma unroll
    for (int e = 0; e < NE; e++) { x[e] = y[e]; asm volatile("" : "+v"(x[e])); }
    __builtin_amdgcn_sched_barrier(0);
}

// buffer accesses: a scalar descriptor per array, a 32-bit lane offset (the row: shared by every array), a scalar offset (period and column)
typedef unsigned int wv2u __attribute__((ext_vector_type(2)));
typedef unsigned int wv4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ double wide_ld64(__amdgpu_buffer_rsrc_t rs, int vo, int so) {
    const wv2u q = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
    return __hiloint2double((int)q.y, (int)q.x);
}
__device__ __forceinline__ double wide_ld64_nt(__amdgpu_buffer_rsrc_t rs, int vo, int so) {
    const wv2u q = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 2);
    return __hiloint2double((int)q.y, (int)q.x);
}
__device__ __forceinline__ int wide_ld32(__amdgpu_buffer_rsrc_t rs, int vo, int so) { return (int)__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0); }
__device__ __forceinline__ wv4u wide_ld128(__amdgpu_buffer_rsrc_t rs, int vo, int so) { return __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0); }
__device__ __forceinline__ void wide_st64_nt(double v, __amdgpu_buffer_rsrc_t rs, int vo, int so) {
    wv2u q;
    q.x = (unsigned)__double2loint(v); q.y = (unsigned)__double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(q, rs, vo, so, 2);
}

struct WideArgs {
    Consts c;
    Record R;
    const double *xhh;      // [n_hh*P] household inputs of the recorded primal
    const double *dxhh;     // (n_hh, P, Ntot) column-major: the input tangents as the caller hands them in
    int Ntot, n0;           // workgroup b carries direction n0 + b
    double *dpol;           // [P][Ntot][G] policy partials (written backward, read forward)
    double *dagg;           // (P, Ntot) column-major (forward sweep)
    // the record is ONE allocation (hank_create): every array is reached through ONE buffer descriptor plus its byte offset from
    // `rec` (a descriptor is four scalar registers; eleven of them, next to the mixing's coefficient stream, spilled)
    const void *rec;
    unsigned o_s, o_kc, o_A, o_B, o_u, o_v, o_ib, o_lwg, o_seg, o_D, o_pol;
};

// dynamic LDS of the two kernels
static inline size_t wide_lds_back(const Consts &c) {
    return sizeof(double) * (2 * (size_t)((c.n_a + 2) & ~1) + 8 * (size_t)c.P + (size_t)wide_kl(c.n_e) * WIDE_R * WIDE_MAXT);
}
static inline size_t wide_lds_fwd(const Consts &c) {
    return sizeof(double) * (4 * ((size_t)c.n_a + 1) + 64 + (size_t)wide_kl(c.n_e) * WIDE_R * WIDE_MAXT) + sizeof(int) * ((size_t)c.P * c.n_e + 4);
}

// ---- backward: dV_t from dV_{t+1}; sequence X(P-1) Y(P-1) | X(P-2) Y(P-2) | ... (k_xtan_back's expressions) ------------------
//   X: dE = Pi dV_{t+1} (lane-local); ds = kc dE - rho ((z_e dw + dtr) + s dr)           KrusellSmith.jl:59-62 under Dual
//   Y: dg = A ds[ib] + B ds[ib+1] (through the LDS column); dV = u dr + v ((a dr + z_e dw + dtr) - dg)   :66-80
template <int NE, int R, int MAXT, bool DIET>
__global__ void __launch_bounds__(MAXT) k_wide_back(WideArgs A, WMat<NE> M) {
#pragma clang fp contract(fast)
    constexpr int KL = wide_kl(NE), KR = NE - KL;
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const int na = c.n_a, P = c.P, NT = blockDim.x, tid = threadIdx.x, G = c.G;
    const int n = A.n0 + blockIdx.x;
    const int nap = (na + 2) & ~1;          // (slot n_a takes the writes of the lanes beyond the grid)
    double *buf = xl;                       // [2][nap] the column being exchanged
    double *uni = xl + 2 * (size_t)nap;     // [P][8]: rho_t, 1 + r_t, w_t, tr_t, dr_t, dw_t, dtr_t of this direction
    double *lst = uni + 8 * (size_t)P + tid;        // [KL][R][MAXT]: this thread's slots of the LDS-resident columns of the state (compile-time strides: immediate offsets)
    for (int k = tid; k < P; k += NT) {
        const double r = A.xhh[c.n_hh * k];
        const double *dx = A.dxhh + (size_t)c.n_hh * ((size_t)k + (size_t)P * n);
        uni[8 * k] = 1.0 / (1.0 + r); uni[8 * k + 1] = 1.0 + r; uni[8 * k + 2] = A.xhh[c.n_hh * k + 1]; uni[8 * k + 3] = hh_tr(c, A.xhh, k);
        uni[8 * k + 4] = dx[0]; uni[8 * k + 5] = dx[1]; uni[8 * k + 6] = c.n_hh > 2 ? dx[2] : 0.0; uni[8 * k + 7] = 0.0;
    }
    int a[R], o8[R], o4[R];
    bool ok[R];
    double xa[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int row = q * NT + tid;
        ok[q] = row < na;
        a[q] = ok[q] ? row : na;
        o8[q] = (ok[q] ? row : 0) * 8; o4[q] = (ok[q] ? row : 0) * 4;     // (a lane beyond the grid reads row 0's record and stores nothing)
        xa[q] = c.a[ok[q] ? row : 0];
    }
    double dV[R][KR];
#pragma unroll
    for (int q = 0; q < R; q++) {
#pragma unroll
        for (int e = 0; e < KR; e++) dV[q][e] = 0.0;            // dV_T = 0 (BackwardIteration.jl:85)
#pragma unroll
        for (int k = 0; k < KL; k++) lst[(k * R + q) * MAXT] = 0.0;
    }
    const __amdgpu_buffer_rsrc_t rs = wide_rsrc(A.rec);
    // The record of a column, SINGLE-buffered: every group of loads is issued the moment the registers it lands in have been
    // used for the last time (two full stages in flight next to the state do not fit 256 VGPRs: the allocator spilled the
    // freshly loaded values, waiting for each of them). What the loads then have to hide behind: the knots `s` of the next column
    // the whole Y half and the barrier; ib, A, B the rest of the Y half, the next X half and the barrier; u, v a little more.
    struct Stage { double s[R], kc[DIET ? 1 : R], cA[R], cB[R], cu[R], cv[R]; int ib[R]; } S;
    auto col_off = [&](int t, int e, int zt) { return t * G + e * na + zt; };   // (zt: an opaque zero of the period — the column offsets are not hoisted out of the period loop)
    auto load_X = [&](int so) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            S.s[q] = wide_ld64(rs, o8[q], A.o_s + so * 8);
            if constexpr (!DIET) S.kc[q] = wide_ld64(rs, o8[q], A.o_kc + so * 8);
        }
    };
    auto load_Y1 = [&](int so) {
#pragma unroll
        for (int q = 0; q < R; q++) { S.ib[q] = wide_ld32(rs, o4[q], A.o_ib + so * 4); S.cA[q] = wide_ld64(rs, o8[q], A.o_A + so * 8); S.cB[q] = wide_ld64(rs, o8[q], A.o_B + so * 8); }
    };
    auto load_Y2 = [&](int so) {
#pragma unroll
        for (int q = 0; q < R; q++) { S.cu[q] = wide_ld64(rs, o8[q], A.o_u + so * 8); S.cv[q] = wide_ld64(rs, o8[q], A.o_v + so * 8); }
    };
    __syncthreads();
    { const int so = col_off(P - 1, 0, 0); load_X(so); load_Y1(so); load_Y2(so); }
    int pb = 0;
    for (int t = P - 1; t >= 0; t--) {
        // (uniform over the workgroup: scalar registers)
        const double rho = wide_uniform(uni[8 * t]), opr = wide_uniform(uni[8 * t + 1]), w = wide_uniform(uni[8 * t + 2]), tr = wide_uniform(uni[8 * t + 3]);
        const double dr = wide_uniform(uni[8 * t + 4]), dw = wide_uniform(uni[8 * t + 5]), dtr = wide_uniform(uni[8 * t + 6]);
        // ---- the mixing, row by row: dE[e] = sum_k Pi[e, k] dV[k]
#pragma unroll
        for (int q = 0; q < R; q++) {
            double x[NE];
#pragma unroll
            for (int e = 0; e < KR; e++) x[e] = dV[q][e];
#pragma unroll
            for (int k = 0; k < KL; k++) x[KR + k] = lst[(k * R + q) * MAXT];
            wide_mix<NE>(x, M.m, wide_opaque_zero(t, q));
#pragma unroll
            for (int e = 0; e < KR; e++) dV[q][e] = x[e];
#pragma unroll
            for (int k = 0; k < KL; k++) lst[(k * R + q) * MAXT] = x[KR + k];
        }
        const int zt = wide_opaque_zero(t, 0);
        const __amdgpu_buffer_rsrc_t rs_dp = wide_rsrc(A.dpol + ((size_t)t * A.Ntot + n) * (size_t)G);
        // ---- column by column: knot partials -> LDS column -> bracket gather -> policy partials, dV_t
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int last = e + 1 == NE;
            const int son = col_off(last ? (t > 0 ? t - 1 : 0) : t, last ? 0 : e + 1, zt);     // the next column (of the next period after the last)
            const double ze = M.z[e], wz = w * ze + tr, zd = ze * dw + dtr;
            double *const col = buf + pb * nap;
#pragma unroll
            for (int q = 0; q < R; q++) {
                const double dE = e < KR ? dV[q][e < KR ? e : 0] : lst[((e - KR) * R + q) * MAXT];
                double kc;
                if constexpr (DIET) kc = diet_kc(c, S.s[q], rho, opr, wz, xa[q]); else kc = S.kc[q];
                col[a[q]] = kc * dE - rho * (zd + S.s[q] * dr);
            }
            __builtin_amdgcn_sched_barrier(0);      // (the loads must not be scheduled above the last use of the registers they refill)
            load_X(son);
            xlds_barrier();
            double dg[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const double d0 = col[S.ib[q]], d1 = col[S.ib[q] + 1];
                dg[q] = S.cA[q] * d0 + S.cB[q] * d1;
                wide_pin(dg[q]);
            }
            __builtin_amdgcn_sched_barrier(0);      // (the loads must not be scheduled above the last use of the registers they refill)
            load_Y1(son);
#pragma unroll
            for (int q = 0; q < R; q++) {
                double dVn = S.cu[q] * dr + S.cv[q] * ((xa[q] * dr + zd) - dg[q]);
                wide_pin(dVn);
                if (e < KR) dV[q][e < KR ? e : 0] = dVn; else lst[((e - KR) * R + q) * MAXT] = dVn;
            }
            __builtin_amdgcn_sched_barrier(0);      // (the loads must not be scheduled above the last use of the registers they refill)
            load_Y2(son);
            const int sd = (e * na + zt) * 8;
#pragma unroll
            for (int q = 0; q < R; q++)
                if (ok[q]) wide_st64_nt(dg[q], rs_dp, o8[q], sd);
            pb ^= 1;
        }
    }
}

// ---- forward: dD_t from dD_{t-1} and dpol_t (tan_fwd_body's expressions, gather form) ----------------------------------------
//   per source j of column e:  cL = (1 - w) dD[j] - (ig D_{t-1}) dpol[j]  -> target lo_j,   cH = w dD[j] + (ig D_{t-1}) dpol[j] -> target lo_j + 1
//   per target r: dD_mid[r] = sum_{j in [s0, s1)} cH[j] + sum_{j in [s1, s2)} cL[j]  (+ the clamped prefix, weight one, into row 0)
//   dD_t = dD_mid Pi (lane-local);  dagg_t = sum dpol_t D_t + sum pol_t dD_t  (post-transition D_t, ForwardIteration.jl:301-307)
// The second sum of period t is taken one period later, when dD_t is walked as the source of period t + 1 (its rows are in
// registers then, and pol_t is one more coalesced load of that walk); the last period's in an epilogue.
template <int NE, int R, int MAXT>
__global__ void __launch_bounds__(MAXT) k_wide_fwd(WideArgs A, WMat<NE> M) {
#pragma clang fp contract(fast)
    constexpr int KL = wide_kl(NE), KR = NE - KL;
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const Record &Rc = A.R;
    const int na = c.n_a, P = c.P, NT = blockDim.x, tid = threadIdx.x, G = c.G;
    const int n = A.n0 + blockIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NWM = MAXT / 64;                          // wave slots of the per-wave sums (the slots of waves that do not exist stay zero)
    double2 *cb = reinterpret_cast<double2 *>(xl);          // [2][na + 1] {cL, cH} of the column being exchanged (slot n_a: the lanes beyond the grid)
    double *red = xl + 4 * ((size_t)na + 1);                // [2][16] per-wave sums of the clamped prefix
    double *aggred = red + 32;                              // [2][16] per-wave parts of a period's aggregate
    double *lst = aggred + 32 + tid;                        // [KL][R][MAXT] this thread's slots of the LDS-resident columns
    int *closh = reinterpret_cast<int *>(aggred + 32 + (size_t)KL * R * MAXT);      // [P][NE]
    for (int k = tid; k < P * NE; k += NT) closh[k] = min(max(Rc.clo[k], 0), na);
    if (tid < 64) red[tid] = 0.0;                           // red and aggred
    int a[R], o8[R], o16[R];
    bool ok[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int row = q * NT + tid;
        ok[q] = row < na;
        a[q] = ok[q] ? row : na;
        o8[q] = (ok[q] ? row : 0) * 8; o16[q] = (ok[q] ? row : 0) * 16;
    }
    double dD[R][KR];
#pragma unroll
    for (int q = 0; q < R; q++) {
#pragma unroll
        for (int e = 0; e < KR; e++) dD[q][e] = 0.0;            // D_0 carries no partials (ForwardIteration.jl:293)
#pragma unroll
        for (int k = 0; k < KL; k++) lst[(k * R + q) * MAXT] = 0.0;
    }
    const __amdgpu_buffer_rsrc_t rs = wide_rsrc(A.rec);
    // the record of a column: the SOURCE half (lottery weights, policy partials, D_t, pol_{t-1}) single-buffered — reloaded for the
    // next column the moment the push has been written to LDS; the TARGET half (the three segment bounds) double-buffered, it is
    // small and needed right behind the barrier (see k_wide_back)
    struct Src { double w[R], g[R], dp[R], Dn[R], pp[R]; } S;
    struct Seg { int s0[R], s1[R], s2[R]; } sg[2];
    auto load_src = [&](int t, int e, int zt) {
        const int pt = e * na + zt, so = t * G + pt;
        const __amdgpu_buffer_rsrc_t rs_dp = wide_rsrc(A.dpol + ((size_t)t * A.Ntot + n) * (size_t)G);
#pragma unroll
        for (int q = 0; q < R; q++) {
            const wv4u wg = wide_ld128(rs, o16[q], A.o_lwg + so * 16);
            S.w[q] = __hiloint2double((int)wg.y, (int)wg.x); S.g[q] = __hiloint2double((int)wg.w, (int)wg.z);
            S.dp[q] = wide_ld64_nt(rs_dp, o8[q], pt * 8);
            S.Dn[q] = wide_ld64(rs, o8[q], A.o_D + (so + G) * 8);                       // D_t (post-transition)
            S.pp[q] = wide_ld64(rs, o8[q], A.o_pol + (t > 0 ? so - G : so) * 8);        // pol_{t-1}: its product with dD_{t-1} belongs to dagg_{t-1}
        }
    };
    auto load_seg = [&](Seg &T, int t, int e, int zt) {
        const int so = t * G + e * na + zt;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const wv4u v = wide_ld128(rs, o16[q], A.o_seg + so * 16);
            T.s0[q] = (int)v.x; T.s1[q] = (int)v.y; T.s2[q] = (int)v.z;
        }
    };
    __syncthreads();
    load_src(0, 0, 0);
    load_seg(sg[0], 0, 0, 0);
    int pb = 0;
    double aggBp = 0.0;                     // sum dpol_{t-1} D_{t-1} over this thread's points (waiting for its other half)
    double *const outn = A.dagg + (size_t)n * P;
    for (int t = 0; t < P; t++) {
        double aggA = 0.0, aggB = 0.0;
        const int zt = wide_opaque_zero(t, 0);
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int b = e & 1, last = e + 1 == NE, nb = last ? 0 : ((e + 1) & 1);
            const int tn = last ? (t + 1 < P ? t + 1 : t) : t, en = last ? 0 : e + 1;
            __builtin_amdgcn_sched_barrier(0);      // (the loads must not be scheduled above the last use of the registers they refill)
            if (nb != b) load_seg(sg[nb], tn, en, zt);
            const Seg &T = sg[b];
            const int clo = closh[t * NE + e];
            double2 *const col = cb + pb * (na + 1);
            double cs = 0.0;
#pragma unroll
            for (int q = 0; q < R; q++) {           // (branch-free: a lane beyond the grid carries zeros, reads row 0's record and writes slot n_a)
                const double x = e < KR ? dD[q][e < KR ? e : 0] : lst[((e - KR) * R + q) * MAXT];
                const double g = S.g[q] * S.dp[q];
                col[a[q]] = make_double2((1.0 - S.w[q]) * x - g, S.w[q] * x + g);
                aggA += S.pp[q] * x;                // (t = 0: x = 0)
                aggB += ok[q] ? S.dp[q] * S.Dn[q] : 0.0;
                cs += a[q] < clo ? x : 0.0;
            }
            wide_pin(aggA); wide_pin(aggB); wide_pin(cs);
            __builtin_amdgcn_sched_barrier(0);      // (the loads must not be scheduled above the last use of the registers they refill)
            load_src(tn, en, zt);
            if (clo > 0) {                                      // (uniform) the mass point: sources clamped at the first grid point (:54-58)
                cs = xwave_reduce63(cs);
                if (lane == 63) red[pb * 16 + wv] = cs;
            }
            xlds_barrier();
            if (e == 0 && t >= 2 && tid == 0) {                 // the aggregate of period t-2, whose parts were written at the end of period t-1
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < NWM; k++) s += aggred[((t - 1) & 1) * 16 + k];
                outn[t - 2] = s;
            }
            double acc[R];
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < R; q++) { acc[q] = 0.0; cnt = max(cnt, ok[q] ? T.s2[q] - T.s0[q] : 0); }
            for (int k = 0; __any(k < cnt); k += 2) {           // (branch-free inside: a source beyond the segments reads the spare slot and adds zero)
#pragma unroll
                for (int q = 0; q < R; q++) {
                    const int j = T.s0[q] + k;
                    const bool p0 = ok[q] && j < T.s2[q], p1 = ok[q] && j + 1 < T.s2[q];
                    const double2 c0 = col[p0 ? j : na], c1 = col[p1 ? j + 1 : na];
                    acc[q] += p0 ? (j < T.s1[q] ? c0.y : c0.x) : 0.0;
                    acc[q] += p1 ? (j + 1 < T.s1[q] ? c1.y : c1.x) : 0.0;
                }
            }
            if (clo > 0 && tid == 0) {                          // thread 0 owns row 0 (q = 0)
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < NWM; k++) s += red[pb * 16 + k];
                acc[0] += s;
            }
#pragma unroll
            for (int q = 0; q < R; q++) {                       // dD_mid takes the place of the column it was pushed from
                wide_pin(acc[q]);
                if (e < KR) dD[q][e < KR ? e : 0] = acc[q]; else lst[((e - KR) * R + q) * MAXT] = acc[q];
            }
            __builtin_amdgcn_sched_barrier(0);      // (the loads must not be scheduled above the last use of the registers they refill)
            if (nb == b) load_seg(sg[nb], tn, en, zt);          // (odd n_e, last column: its own buffer is free only now)
            pb ^= 1;
        }
        // ---- exogenous transition, row by row: dD_t[e2] = sum_k dD_mid[k] Pi[k, e2] (ForwardIteration.jl:95-99)
#pragma unroll
        for (int q = 0; q < R; q++) {
            double x[NE];
#pragma unroll
            for (int e = 0; e < KR; e++) x[e] = dD[q][e];
#pragma unroll
            for (int k = 0; k < KL; k++) x[KR + k] = lst[(k * R + q) * MAXT];
            wide_mix<NE>(x, M.m, wide_opaque_zero(t, q));
#pragma unroll
            for (int e = 0; e < KR; e++) dD[q][e] = x[e];
#pragma unroll
            for (int k = 0; k < KL; k++) lst[(k * R + q) * MAXT] = x[KR + k];
        }
        if (t > 0) {                                            // dagg_{t-1} = sum pol_{t-1} dD_{t-1} + sum dpol_{t-1} D_{t-1}
            const double s = xwave_reduce63(aggA + aggBp);
            if (lane == 63) aggred[(t & 1) * 16 + wv] = s;
        }
        aggBp = aggB;
    }
    // ---- epilogue: the last period's aggregate (its first sum needs pol_{P-1} against the final dD_{P-1})
    double aggA = 0.0;
#pragma unroll
    for (int e = 0; e < NE; e++)
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double x = e < KR ? dD[q][e < KR ? e : 0] : lst[((e - KR) * R + q) * MAXT];
            aggA += wide_ld64(rs, o8[q], A.o_pol + ((P - 1) * G + e * na) * 8) * x;      // (zeros beyond the grid)
        }
    const double s = xwave_reduce63(aggA + aggBp);
    if (lane == 63) aggred[(P & 1) * 16 + wv] = s;
    __syncthreads();
    if (tid == 0) {
        if (P >= 2) { double v = 0.0; for (int k = 0; k < NWM; k++) v += aggred[((P - 1) & 1) * 16 + k]; outn[P - 2] = v; }
        double v = 0.0;
        for (int k = 0; k < NWM; k++) v += aggred[(P & 1) * 16 + k];
        outn[P - 1] = v;
    }
}

// (G, P, N) col-major export of the wide layout dpol[P][N][G]
__global__ void k_wide_export_dpol(const double *dpol, int G, int P, int N, double *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)G * P * N;
    if (idx >= total) return;
    const size_t n = idx / ((size_t)G * P), rem = idx - n * (size_t)G * P, t = rem / G, pt = rem - t * G;
    out[idx] = dpol[(t * N + n) * G + pt];
}

}  // namespace hank
