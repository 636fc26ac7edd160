// hank_wide.h — the state-on-chip tangent sweeps for WIDE batches: ONE workgroup per tangent direction.
//
// The same two linear recurrences as k_tan_back / k_tan_fwd (hank_kernels.h) and k_xtan_back / k_xfwd (hank_xsweep.h) — the N
// partials of BackwardIteration.jl:90-113 and ForwardIteration.jl:297-308 at a recorded primal — for batches wide enough to give
// every CU a direction of its own (N >= ~100). What the other two families pay per period is the round trip of the loop-carried
// state: a launch per period reads and writes it through the fabric (132 / 192 MB per launch for 45 MB algorithmic at N = 256,
// profiles/r04n256_*), the XCD-local sweeps exchange it through an L2 behind flags (a latency chain of ~6 us per period). Here
// the state of a direction never leaves its CU:
//   * a workgroup owns ONE direction and the WHOLE n_a x n_e grid. Thread i owns two PAIRS of adjacent wealth rows — 2i, 2i+1 and
//     2(i + NT), 2(i + NT) + 1 — and ALL n_e columns of each: R * n_e = 44 doubles of loop-carried state at n_e = 11, the first 7
//     columns in VGPRs, the rest in thread-private LDS slots (the registers are needed for the record that is on its way; a
//     private slot costs two LDS accesses per period). Adjacent rows make every record, policy-partial and LDS-column access a
//     16-byte one: the per-CU vector-memory path is what bounds this family (one CU streams the whole record of a period for ONE
//     direction), and 8-byte accesses cost it twice the instructions per byte.
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

// Two geometries of the same kernels: R = 4 rows per thread in workgroups of up to 512 threads (8 waves, 2 per SIMD, 256 VGPRs per
// lane) or R = 2 rows per thread in workgroups of up to 1024 (16 waves, 4 per SIMD, 128 VGPRs: more waves to put behind each other's
// memory stalls, half the registers each). Either way a workgroup covers WIDE_CS = R * MAXT = 2048 rows.
constexpr int WIDE_CS = 2048;      // row slots of a workgroup = slots of an exchanged LDS column
// KREG = columns of the state kept in registers; any beyond live in thread-private LDS slots
constexpr int wide_kl(int ne, int kreg) { return ne > kreg ? ne - kreg : 0; }
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

// y = M x for R grid rows at once, in place: a row's n_e numbers are this lane's registers (columns >= KR: its private LDS slots),
// the n_e^2 coefficients stream through SCALAR registers, one row of n_e coefficients per input column k, the next one on its
// way while this one is used by all R rows (R * n_e fused multiply-adds per scalar round trip: with one grid row per trip the
// scalar-cache latency, ~200 clocks, was exposed 44 times per period — 4.3 us of a 30 us period in the stamps). All n_e^2 at
// once would need 242 SGPRs — and the compiler, seeing the same loop-invariant loads in every period, hoists them all and
// spills: `zero` is a zero it cannot see through, the empty asm ties each row of loads to its place in the arithmetic.
// Order of the sums: k ascending, first term unrounded-added (as xtile_mix_reg).
template <int NE, int R, int KR, int LSTRIDE>
__device__ __forceinline__ void wide_mix(double (&x)[R][KR], double *lst, const double *m, int zero) {
#pragma clang fp contract(fast)
    constexpr int KL = NE - KR;
    double y[R][NE], cf[2][NE], xl[2][R];
    int off = zero;
#pragma unroll
    for (int e = 0; e < NE; e++) cf[0][e] = m[off + e];
#pragma unroll
    for (int k = 0; k < NE; k++) {
        if (k + 1 < NE) {
            // (the empty asm ties the loads of the next coefficient row to the LAST sum of the previous step, the scheduling barrier
            // at the end of a step keeps the steps apart: loose, the scalar loads of a whole period line up at the top and spill)
            if (k == 0) asm volatile("" : "+s"(off) : "v"(x[0][0]));
            else asm volatile("" : "+s"(off) : "v"(y[R - 1][NE - 1]));
#pragma unroll
            for (int e = 0; e < NE; e++) cf[(k + 1) & 1][e] = m[off + (k + 1) * NE + e];
            if (k + 1 >= KR) {
#pragma unroll
                for (int q = 0; q < R; q++) xl[(k + 1) & 1][q] = lst[((k + 1 - KR) * R + q) * LSTRIDE];
            }
        }
        if (k == 0 && KR == 0) {
#pragma unroll
            for (int q = 0; q < R; q++) xl[0][q] = lst[q * LSTRIDE];
        }
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double xk = k < KR ? x[q][k < KR ? k : 0] : xl[k & 1][q];
#pragma unroll
            for (int e = 0; e < NE; e++) { y[q][e] = k == 0 ? cf[0][e] * xk : y[q][e] + cf[k & 1][e] * xk; asm volatile("" : "+v"(y[q][e])); }
        }       // (every partial sum is pinned to its step: unpinned, all of a period's multiply-adds sink below the last step's loads)
        __builtin_amdgcn_sched_barrier(0);
    }
    // (the results are pinned HERE: each has one use, inside a column's code, and the compiler would sink the whole sum — and
    // every coefficient, as a spilled scalar — down to it)
#pragma unroll
    for (int q = 0; q < R; q++) {
#pragma unroll
        for (int e = 0; e < KR; e++) { x[q][e] = y[q][e]; asm volatile("" : "+v"(x[q][e])); }
#pragma unroll
        for (int k = 0; k < KL; k++) lst[(k * R + q) * LSTRIDE] = y[q][KR + k];
    }
    __builtin_amdgcn_sched_barrier(0);
}

// buffer accesses: ONE scalar descriptor for the whole record, a 32-bit lane offset (the row pair: shared by every array), a scalar
// offset (array, period and column)
typedef unsigned int wv2u __attribute__((ext_vector_type(2)));
typedef unsigned int wv4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ double wide_ld64(__amdgpu_buffer_rsrc_t rs, int vo, int so) {
    const wv2u q = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
    return __hiloint2double((int)q.y, (int)q.x);
}
__device__ __forceinline__ wv2u wide_ld2i(__amdgpu_buffer_rsrc_t rs, int vo, int so) { return __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0); }
template <int AUX>
__device__ __forceinline__ void wide_ld2d(__amdgpu_buffer_rsrc_t rs, int vo, int so, double &a, double &b) {    // two adjacent doubles: one 16-byte load
    const wv4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, AUX);
    a = __hiloint2double((int)q.y, (int)q.x); b = __hiloint2double((int)q.w, (int)q.z);
}
__device__ __forceinline__ wv4u wide_ld128(__amdgpu_buffer_rsrc_t rs, int vo, int so) { return __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0); }
__device__ __forceinline__ void wide_st2d_nt(double a, double b, __amdgpu_buffer_rsrc_t rs, int vo, int so) {
    wv4u q;
    q.x = (unsigned)__double2loint(a); q.y = (unsigned)__double2hiint(a); q.z = (unsigned)__double2loint(b); q.w = (unsigned)__double2hiint(b);
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, vo, so, 2);
}
__device__ __forceinline__ void wide_st64_nt(double v, __amdgpu_buffer_rsrc_t rs, int vo, int so) {
    wv2u q;
    q.x = (unsigned)__double2loint(v); q.y = (unsigned)__double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(q, rs, vo, so, 2);
}

// dev build only (make stamp): s_memtime of wave 0 / lane 0 of workgroup 0 at fixed points of periods [100, 104)
// dev timing builds (wrong numbers): HANK_WIDE_TIMING_L2 confines the record to two periods (every line an L2 hit: what the HBM
// latency of the record stream costs)
#ifdef HANK_WIDE_TIMING_L2
#define WIDE_TPER(t) ((t) & 1)
#else
#define WIDE_TPER(t) (t)
#endif
#ifdef HANK_XSTAMP
__device__ unsigned long long g_wstamps[2][4][64];
#define WSTAMP(sw, per, i)                                                                                          \
    do {                                                                                                            \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (per) >= 100 && (per) < 104) g_wstamps[sw][(per) - 100][i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define WSTAMP(sw, per, i) do {} while (0)
#endif

struct WideArgs {
    Consts c;
    Record R;
    const double *xhh;      // [n_hh*P] household inputs of the recorded primal
    const double *dxhh;     // (n_hh, P, Ntot) column-major: the input tangents as the caller hands them in
    int Ntot, n0;           // workgroup b carries direction n0 + b
    double *dpol;           // [P][Ntot][G] (+ 2 spare) policy partials (written backward, read forward)
    double *dagg;           // (P, 2 Ntot) column-major (forward sweep): the policy-weighted aggregate's columns, then the grid-weighted one's
    // the record is ONE allocation (hank_create): every array is reached through ONE buffer descriptor plus its byte offset from
    // `rec` (a descriptor is four scalar registers; eleven of them, next to the mixing's coefficient stream, spilled)
    const void *rec;
    unsigned o_s, o_kc, o_u, o_v, o_lwg, o_start, o_D, o_pol;
    const int *ibw;         // [P][G] bracket | (A == B == 0) << 31 (k_wide_prep)
    int nrank;              // backward: the workgroups that share an XCD's L2 split the next period's record lines `nrank` ways (L2 warming)
};

// dynamic LDS of the two kernels: the exchanged column holds a slot for every row a thread may own (R * WIDE_MAXT >= n_a)
static inline size_t wide_lds_back(const Consts &c, int kreg) {      // two columns of {ds, knot} pairs, the grid's spacings, the per-period inputs, the LDS-resident state columns
    return sizeof(double) * (4 * (size_t)WIDE_CS + (size_t)WIDE_CS + 8 * (size_t)c.P + (size_t)wide_kl(c.n_e, kreg) * WIDE_CS);
}
static inline size_t wide_lds_fwd(const Consts &c, int kreg) {
    return sizeof(double) * (4 * (size_t)WIDE_CS + 96 + (size_t)WIDE_CS + (size_t)wide_kl(c.n_e, kreg) * WIDE_CS) + sizeof(int) * ((size_t)c.P * c.n_e + 4);
}

// this thread's rows: two pairs of adjacent rows, pair j = rows 2 (j NT + tid), + 1. A lane whose pair lies beyond the grid reads
// row 0's record (finite numbers), keeps zeros in the forward sweep, and stores nothing.
template <int R>
struct WideRows {
    int row[R], o8[R / 2], o4[R / 2], o16[R / 2];
    bool ok[R];
    __device__ __forceinline__ void init(int NT, int tid, int na) {
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            const int r0 = 2 * (j * NT + tid);
            row[2 * j] = r0; row[2 * j + 1] = r0 + 1;
            ok[2 * j] = r0 < na; ok[2 * j + 1] = r0 + 1 < na;
            const int rc = ok[2 * j] ? r0 : 0;
            o8[j] = rc * 8; o4[j] = rc * 4; o16[j] = rc * 16;
        }
    }
};

// ---- backward: dV_t from dV_{t+1}; sequence X(P-1) Y(P-1) | X(P-2) Y(P-2) | ... (k_xtan_back's expressions) ------------------
//   X: dE = Pi dV_{t+1} (lane-local); ds = kc dE - rho ((z_e dw + dtr) + s dr)           KrusellSmith.jl:59-62 under Dual
//   Y: dg = A ds[ib] + B ds[ib+1] (through the LDS column); dV = u dr + v ((a dr + z_e dw + dtr) - dg)   :66-80
// RECORD DIET of this family: A and B are not read. The exchanged column carries the knot next to its partial ({ds, s} pairs), the
// gathering lane has the two knots of its bracket with the two partials and rebuilds the weights egm_Y recorded — h = s_{i+1} - s_i,
// f = (a - s_i) / h, sl = (a_{i+1} - a_i) / h, A = -sl (1 - f), B = -sl f (the division as a reciprocal with two Newton steps: a
// rounding away from the recorded values) — or takes zeros where the record says both were zero (bit 31 of the bracket, k_wide_prep).
// 16 of the 52 bytes a grid point and period costs this kernel's vector-memory path, for ~20 arithmetic instructions it has room for.
template <int NE, int R, int MAXT, bool DIET, int KREG>
__global__ void __launch_bounds__(MAXT) k_wide_back(WideArgs A, WMat<NE> M) {
#pragma clang fp contract(fast)
    static_assert(R % 2 == 0 && R * MAXT == WIDE_CS, "rows come in adjacent pairs; a workgroup covers WIDE_CS rows");
    constexpr int KL = wide_kl(NE, KREG), KR = NE - KL, CS = R * MAXT;      // CS: slots of an exchanged column
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const int na = c.n_a, P = c.P, NT = blockDim.x, tid = threadIdx.x, G = c.G;
    const int n = A.n0 + blockIdx.x;
    double2 *buf = reinterpret_cast<double2 *>(xl);         // [2][CS] the column being exchanged: {ds, knot}
    double *dash = xl + 4 * (size_t)CS;     // [CS] a[i+1] - a[i]
    double *uni = dash + CS;                // [P][8]: rho_t, 1 + r_t, w_t, tr_t, dr_t, dw_t, dtr_t of this direction
    double *lst = uni + 8 * (size_t)P + tid;        // [KL][R][MAXT]: this thread's slots of the LDS-resident columns of the state (compile-time strides: immediate offsets)
    for (int k = tid; k < P; k += NT) {
        const double r = A.xhh[c.n_hh * k];
        const double *dx = A.dxhh + (size_t)c.n_hh * ((size_t)k + (size_t)P * n);
        uni[8 * k] = 1.0 / (1.0 + r); uni[8 * k + 1] = 1.0 + r; uni[8 * k + 2] = A.xhh[c.n_hh * k + 1]; uni[8 * k + 3] = hh_tr(c, A.xhh, k);
        uni[8 * k + 4] = dx[0]; uni[8 * k + 5] = dx[1]; uni[8 * k + 6] = c.n_hh > 2 ? dx[2] : 0.0; uni[8 * k + 7] = 0.0;
    }
    for (int k = tid; k < CS; k += NT) dash[k] = k + 1 < na ? c.a[k + 1] - c.a[k] : 1.0;
    WideRows<R> rw;
    rw.init(NT, tid, na);
    double xa[R];
#pragma unroll
    for (int q = 0; q < R; q++) xa[q] = c.a[rw.ok[q] ? rw.row[q] : 0];
    double dV[R][KR];
#pragma unroll
    for (int q = 0; q < R; q++) {
#pragma unroll
        for (int e = 0; e < KR; e++) dV[q][e] = 0.0;            // dV_T = 0 (BackwardIteration.jl:85)
#pragma unroll
        for (int k = 0; k < KL; k++) lst[(k * R + q) * MAXT] = 0.0;
    }
    const __amdgpu_buffer_rsrc_t rs = wide_rsrc(A.rec), rs_w = wide_rsrc(A.ibw);
    // The record of a column lives in ONE register stage; each group of its arrays is re-loaded for the NEXT column the moment the
    // current column has used it for the last time (the knots after the X half, ib / A / B after the gather, u / v after dV): every
    // load has a whole column's time to land, and the column's loads are issued at three places instead of one burst — the CU's
    // vector-memory path (what bounds this family: ~30 B/clk per CU from L2) keeps working while the waves do arithmetic.
    struct Stage { double s[R], kc[DIET ? 1 : R], cu[R], cv[R]; int ib[R]; } S;
    auto load_X = [&](int so) {
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            wide_ld2d<0>(rs, rw.o8[j], A.o_s + so * 8, S.s[2 * j], S.s[2 * j + 1]);
            if constexpr (!DIET) wide_ld2d<0>(rs, rw.o8[j], A.o_kc + so * 8, S.kc[2 * j], S.kc[2 * j + 1]);
        }
    };
    auto load_Y1 = [&](int so) {
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            const wv2u iq = wide_ld2i(rs_w, rw.o4[j], so * 4);
            S.ib[2 * j] = (int)iq.x; S.ib[2 * j + 1] = (int)iq.y;
        }
    };
    auto load_Y2 = [&](int so) {
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            wide_ld2d<0>(rs, rw.o8[j], A.o_u + so * 8, S.cu[2 * j], S.cu[2 * j + 1]);
            wide_ld2d<0>(rs, rw.o8[j], A.o_v + so * 8, S.cv[2 * j], S.cv[2 * j + 1]);
        }
    };
    // L2 WARMING. Every workgroup streams the same record, a period behind the same clock: whichever workgroup of an XCD asks for a
    // line first waits for HBM (2-3 us), the others find it in that XCD's L2 — and with the loads of a column issued ONE column
    // (~1 us) ahead the front runner stalls at every column and everybody runs at its pace (stamps, profiles/r05b_wide_stamps256.log:
    // ~2 000 of a column's ~4 500 clocks; with the record confined to two periods, every load an L2 hit, the sweep takes 5.2 ms
    // instead of 7.1). Vector-memory loads return in order, so a wave cannot keep a miss in flight behind the data it needs next —
    // except where the period has slack of its own: the mixing at its top. There every workgroup asks for ONE dword of every
    // nrank-th 128-byte line of the PREVIOUS period's arrays (the next one in sweep order: its share of the XCD's 32 workgroups,
    // ~150 of 4 800 lines, one instruction per descriptor), a whole period before anybody reads them: HBM -> L2 happens once per
    // XCD, off everybody's critical path. The share is blockIdx-derived (workgroups are dealt to the XCDs round-robin): an
    // assumption about SPEED only — a line nobody warmed is an ordinary miss.
    const int tch_line = ((int)(blockIdx.x >> 3) % A.nrank + A.nrank * tid) * 128;
    const int tch_n8 = G * 8, tch_n4 = G * 4;
    constexpr int TCH_ARR = DIET ? 3 : 4;
    int tch_arr = 0, tch_off = tch_line;
#pragma unroll
    for (int k = 1; k < TCH_ARR; k++)
        if (tch_off >= tch_n8) { tch_off -= tch_n8; tch_arr = k; }
    // (branch-free: a lane without a line asks for an offset beyond the descriptor's range — no memory access, the answer is zero;
    // a divergent branch or a per-lane scalar offset around the load costs the compiler its count of the loads in flight, and
    // every wait behind it becomes vmcnt(0): the miss this is about would be waited for)
    const int tch_v8 = tch_off < tch_n8 ? (int)(tch_arr == 0 ? A.o_s : tch_arr == 1 ? A.o_u : tch_arr == 2 ? A.o_v : A.o_kc) + tch_off : (int)0x80000000;
    const int tch_v4 = tch_line < tch_n4 ? tch_line : (int)0x80000000;
    unsigned tch_j8 = 0, tch_j4 = 0;
    __syncthreads();
    { const int so = WIDE_TPER(P - 1) * G; load_X(so); load_Y1(so); load_Y2(so); }
    int pb = 0;
    for (int t = P - 1; t >= 0; t--) {
        WSTAMP(0, t, 0);
#ifndef HANK_WIDE_NO_TOUCH
        asm volatile("" :: "v"(tch_j8), "v"(tch_j4));       // (last period's two: long landed)
        {
            const int tb = WIDE_TPER(t > 0 ? t - 1 : 0) * G;
            tch_j8 = __builtin_amdgcn_raw_buffer_load_b32(rs, tch_v8, tb * 8, 0);
            tch_j4 = __builtin_amdgcn_raw_buffer_load_b32(rs_w, tch_v4, tb * 4, 0);
        }
#endif
        // (uniform over the workgroup: scalar registers)
        const double rho = wide_uniform(uni[8 * t]), opr = wide_uniform(uni[8 * t + 1]), w = wide_uniform(uni[8 * t + 2]), tr = wide_uniform(uni[8 * t + 3]);
        const double dr = wide_uniform(uni[8 * t + 4]), dw = wide_uniform(uni[8 * t + 5]), dtr = wide_uniform(uni[8 * t + 6]);
        // ---- the mixing: dE[e] = sum_k Pi[e, k] dV[k], all of this thread's rows per coefficient row
        wide_mix<NE, R, KR, MAXT>(dV, lst, M.m, wide_opaque_zero(t, 1));
        const int zt = wide_opaque_zero(t, 0);      // (an opaque zero of the period: the column offsets are not hoisted out of the period loop)
        const __amdgpu_buffer_rsrc_t rs_dp = wide_rsrc(A.dpol + ((size_t)t * A.Ntot + n) * (size_t)G);
        WSTAMP(0, t, 1);
        // ---- column by column: knot partials -> LDS column -> bracket gather -> policy partials, dV_t
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int last = e + 1 == NE;
            const int son = WIDE_TPER(last ? (t > 0 ? t - 1 : 0) : t) * G + (last ? 0 : e + 1) * na + zt;      // the next column (of the next period behind the last)
            const double ze = M.z[e], wz = w * ze + tr, zd = ze * dw + dtr;
            double2 *const col = buf + pb * CS;
            double ds[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const double dE = e < KR ? dV[q][e < KR ? e : 0] : lst[((e - KR) * R + q) * MAXT];
                double kc;
                if constexpr (DIET) kc = diet_kc(c, S.s[q], rho, opr, wz, xa[q]); else kc = S.kc[q];
                ds[q] = kc * dE - rho * (zd + S.s[q] * dr);
                wide_pin(ds[q]);
            }
#pragma unroll
            for (int q = 0; q < R; q++) col[rw.row[q]] = make_double2(ds[q], S.s[q]);
            __builtin_amdgcn_sched_barrier(0);      // (the loads must not be scheduled above the last use of the registers they refill)
            load_X(son);
            WSTAMP(0, t, 2 + 4 * e);
            xlds_barrier();
            WSTAMP(0, t, 3 + 4 * e);
            double dg[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = S.ib[q] & 0x7fffffff;
                const double2 c0 = col[i], c1 = col[i + 1];     // {ds, knot} at the bracket's two ends
                const double h = c1.y - c0.y;
                double rh = __builtin_amdgcn_rcp(h);            // 1 / h: hardware estimate + two Newton steps
                rh = rh + rh * (1.0 - h * rh);
                rh = rh + rh * (1.0 - h * rh);
                const double f = (xa[q] - c0.y) * rh, sl = dash[i] * rh;
                const double wA = S.ib[q] < 0 ? 0.0 : -sl * (1.0 - f), wB = S.ib[q] < 0 ? 0.0 : -sl * f;
                dg[q] = wA * c0.x + wB * c1.x;
                wide_pin(dg[q]);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_Y1(son);
            WSTAMP(0, t, 5 + 4 * e);
#pragma unroll
            for (int q = 0; q < R; q++) {
                double dVn = S.cu[q] * dr + S.cv[q] * ((xa[q] * dr + zd) - dg[q]);
                wide_pin(dVn);
                if (e < KR) dV[q][e < KR ? e : 0] = dVn; else lst[((e - KR) * R + q) * MAXT] = dVn;
            }
            __builtin_amdgcn_sched_barrier(0);
            load_Y2(son);
            const int sd = (e * na + zt) * 8;
#pragma unroll
            for (int j = 0; j < R / 2; j++) {
                if (rw.ok[2 * j + 1]) wide_st2d_nt(dg[2 * j], dg[2 * j + 1], rs_dp, rw.o8[j], sd);
                else if (rw.ok[2 * j]) wide_st64_nt(dg[2 * j], rs_dp, rw.o8[j], sd);
            }
            WSTAMP(0, t, 4 + 4 * e);
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
template <int NE, int R, int MAXT, int KREG>
__global__ void __launch_bounds__(MAXT) k_wide_fwd(WideArgs A, WMat<NE> M) {
#pragma clang fp contract(fast)
    static_assert(R % 2 == 0 && R * MAXT == WIDE_CS, "rows come in adjacent pairs; a workgroup covers WIDE_CS rows");
    constexpr int KL = wide_kl(NE, KREG), KR = NE - KL, CS = R * MAXT;
    constexpr int NWM = MAXT / 64;                          // wave slots of the per-wave sums (the slots of waves that do not exist stay zero)
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const Record &Rc = A.R;
    const int na = c.n_a, P = c.P, NT = blockDim.x, tid = threadIdx.x, G = c.G;
    const int n = A.n0 + blockIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    double *cLa = xl;                                       // [2][CS] a source's push to its lower target ...
    double *cHa = xl + 2 * (size_t)CS;                      // [2][CS] ... and to the upper one (two arrays: a gathered term is ONE 8-byte read)
    double *red = xl + 4 * (size_t)CS;                      // [2][16] per-wave sums of the clamped prefix
    double *aggred = red + 32;                              // [2][16] per-wave parts of a period's aggregate
    double *aggred2 = aggred + 32;                          // [2][16] ... and of the grid-weighted one (sum a dD_t: see dist_step_body)
    double *ash = aggred2 + 32;                             // [CS] the wealth grid (zeros beyond it)
    double *lst = ash + CS + tid;                           // [KL][R][MAXT] this thread's slots of the LDS-resident columns
    int *closh = reinterpret_cast<int *>(ash + CS + (size_t)KL * R * MAXT);      // [P][NE]
    for (int k = tid; k < P * NE; k += NT) closh[k] = min(max(Rc.clo[k], 0), na);
    for (int k = tid; k < CS; k += NT) ash[k] = k < na ? c.a[k] : 0.0;
    for (int k = tid; k < 96; k += NT) red[k] = 0.0;        // red, aggred and aggred2 (a small grid runs one wave)
    WideRows<R> rw;
    rw.init(NT, tid, na);
    double dD[R][KR];
#pragma unroll
    for (int q = 0; q < R; q++) {
#pragma unroll
        for (int e = 0; e < KR; e++) dD[q][e] = 0.0;            // D_0 carries no partials (ForwardIteration.jl:293)
#pragma unroll
        for (int k = 0; k < KL; k++) lst[(k * R + q) * MAXT] = 0.0;
    }
    const __amdgpu_buffer_rsrc_t rs = wide_rsrc(A.rec);
    // the record of a column in ONE register stage (see k_wide_back): the source half (lottery weights, policy partials, D_t,
    // pol_{t-1}) is re-loaded for the next column once the push is written, the target half (the segment bounds: four consecutive
    // entries of `start` serve a pair of rows) once the gather is done
    struct Src { double w[R], g[R], dp[R], Dn[R], pp[R]; int s0[R], s1[R], s2[R]; } S;
    int so4[R / 2];                                         // byte offset of start[r0 - 1] (start[0] for the first pair) within a column of `start`
#pragma unroll
    for (int j = 0; j < R / 2; j++) so4[j] = (rw.ok[2 * j] ? max(rw.row[2 * j] - 1, 0) : 0) * 4;
    auto load_src = [&](int t, int e, int zt) {
        const int pt = e * na + zt, so = WIDE_TPER(t) * G + pt;
        const __amdgpu_buffer_rsrc_t rs_dp = wide_rsrc(A.dpol + ((size_t)t * A.Ntot + n) * (size_t)G);
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            wide_ld2d<0>(rs, rw.o16[j], A.o_lwg + so * 16, S.w[2 * j], S.g[2 * j]);
            wide_ld2d<0>(rs, rw.o16[j], A.o_lwg + so * 16 + 16, S.w[2 * j + 1], S.g[2 * j + 1]);
            wide_ld2d<2>(rs_dp, rw.o8[j], pt * 8, S.dp[2 * j], S.dp[2 * j + 1]);
            wide_ld2d<0>(rs, rw.o8[j], A.o_D + (so + G) * 8, S.Dn[2 * j], S.Dn[2 * j + 1]);                        // D_t (post-transition)
            wide_ld2d<0>(rs, rw.o8[j], A.o_pol + (t > 0 ? so - G : so) * 8, S.pp[2 * j], S.pp[2 * j + 1]);         // pol_{t-1}: its product with dD_{t-1} belongs to dagg_{t-1}
        }
    };
    auto load_seg = [&](int t, int e, int zt) {
        const int so = ((WIDE_TPER(t) * NE + e) * (na + 1) + zt) * 4;
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            const wv4u v = wide_ld128(rs, so4[j], A.o_start + so);
            const bool first = rw.row[2 * j] == 0;          // the pair (0, 1): start[-1] does not exist, row 0's upper segment is empty (k_lottery's seg)
            S.s0[2 * j] = (int)v.x; S.s1[2 * j] = first ? (int)v.x : (int)v.y; S.s2[2 * j] = first ? (int)v.y : (int)v.z;
            S.s0[2 * j + 1] = first ? (int)v.x : (int)v.y; S.s1[2 * j + 1] = first ? (int)v.y : (int)v.z; S.s2[2 * j + 1] = first ? (int)v.z : (int)v.w;
        }
    };
    __syncthreads();
    load_src(0, 0, 0);
    load_seg(0, 0, 0);
    int pb = 0;
    double aggBp = 0.0;                     // sum dpol_{t-1} D_{t-1} over this thread's points (waiting for its other half)
    double *const outn = A.dagg + (size_t)n * P, *const outn2 = A.dagg + ((size_t)A.Ntot + n) * P;      // dagg: (P, 2 Ntot): both aggregates
    for (int t = 0; t < P; t++) {
        WSTAMP(1, t, 0);
        double aggA = 0.0, aggB = 0.0, aggA2 = 0.0;
        const int zt = wide_opaque_zero(t, 0);
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int last = e + 1 == NE;
            const int tn = last ? (t + 1 < P ? t + 1 : t) : t, en = last ? 0 : e + 1;
            const int clo = closh[t * NE + e];
            double *const cL = cLa + pb * CS, *const cH = cHa + pb * CS;
            double cs = 0.0, pl[R], ph[R];
#pragma unroll
            for (int q = 0; q < R; q++) {           // (branch-free: a lane beyond the grid carries zeros and reads row 0's record)
                const double x = e < KR ? dD[q][e < KR ? e : 0] : lst[((e - KR) * R + q) * MAXT];
                const double g = S.g[q] * S.dp[q];
                pl[q] = (1.0 - S.w[q]) * x - g; ph[q] = S.w[q] * x + g;
                aggA += S.pp[q] * x;                // (t = 0: x = 0)
                aggA2 += ash[rw.row[q]] * x;
                aggB += rw.ok[q] ? S.dp[q] * S.Dn[q] : 0.0;
                cs += (rw.ok[q] && rw.row[q] < clo) ? x : 0.0;
                wide_pin(pl[q]); wide_pin(ph[q]);
            }
            wide_pin(aggA); wide_pin(aggB); wide_pin(aggA2); wide_pin(cs);
            __builtin_amdgcn_sched_barrier(0);      // (the loads must not be scheduled above the last use of the registers they refill)
            load_src(tn, en, zt);
#pragma unroll
            for (int j = 0; j < R / 2; j++) {
                *reinterpret_cast<double2 *>(cL + rw.row[2 * j]) = make_double2(pl[2 * j], pl[2 * j + 1]);
                *reinterpret_cast<double2 *>(cH + rw.row[2 * j]) = make_double2(ph[2 * j], ph[2 * j + 1]);
            }
            if (clo > 0) {                                      // (uniform) the mass point: sources clamped at the first grid point (:54-58)
                cs = xwave_reduce63(cs);
                if (lane == 63) red[pb * 16 + wv] = cs;
            }
            WSTAMP(1, t, 2 + 4 * e);
            xlds_barrier();
            WSTAMP(1, t, 3 + 4 * e);
            if (e == 0 && t >= 2 && tid == 0) {                 // the aggregate of period t-2, whose parts were written at the end of period t-1
                double s = 0.0, s2 = 0.0;
#pragma unroll
                for (int k = 0; k < NWM; k++) { s += aggred[((t - 1) & 1) * 16 + k]; s2 += aggred2[((t - 1) & 1) * 16 + k]; }
                outn[t - 2] = s; outn2[t - 2] = s2;
            }
            // gather: target r sums its sources j in [s0, s2) in order — the upper parts (cH) of [s0, s1), then the lower parts (cL) of [s1, s2)
            double acc[R];
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < R; q++) { acc[q] = 0.0; cnt = max(cnt, rw.ok[q] ? S.s2[q] - S.s0[q] : 0); }
            for (int k = 0; __any(k < cnt); k += 2) {           // (branch-free inside: a term beyond the segments reads slot 0 and adds zero)
#pragma unroll
                for (int q = 0; q < R; q++) {
                    const int j = S.s0[q] + k;
                    const bool p0 = rw.ok[q] && j < S.s2[q], p1 = rw.ok[q] && j + 1 < S.s2[q];
                    const double v0 = (j < S.s1[q] ? cH : cL)[p0 ? j : 0], v1 = (j + 1 < S.s1[q] ? cH : cL)[p1 ? j + 1 : 0];
                    acc[q] += p0 ? v0 : 0.0;
                    acc[q] += p1 ? v1 : 0.0;
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
            __builtin_amdgcn_sched_barrier(0);
            load_seg(tn, en, zt);
            WSTAMP(1, t, 4 + 4 * e);
            pb ^= 1;
        }
        WSTAMP(1, t, 1);
        // ---- exogenous transition: dD_t[e2] = sum_k dD_mid[k] Pi[k, e2] (ForwardIteration.jl:95-99)
        wide_mix<NE, R, KR, MAXT>(dD, lst, M.m, wide_opaque_zero(t, 1));
        if (t > 0) {                                            // dagg_{t-1} = sum pol_{t-1} dD_{t-1} + sum dpol_{t-1} D_{t-1}; the second aggregate: sum a dD_{t-1}
            const double s = xwave_reduce63(aggA + aggBp), s2 = xwave_reduce63(aggA2);
            if (lane == 63) { aggred[(t & 1) * 16 + wv] = s; aggred2[(t & 1) * 16 + wv] = s2; }
        }
        aggBp = aggB;
    }
    // ---- epilogue: the last period's aggregate (its first sum needs pol_{P-1} against the final dD_{P-1})
    double aggA = 0.0, aggA2 = 0.0;
#pragma unroll
    for (int e = 0; e < NE; e++)
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            double p0, p1;
            wide_ld2d<0>(rs, rw.o8[j], A.o_pol + ((P - 1) * G + e * na) * 8, p0, p1);
            const double x0 = e < KR ? dD[2 * j][e < KR ? e : 0] : lst[((e - KR) * R + 2 * j) * MAXT];
            const double x1 = e < KR ? dD[2 * j + 1][e < KR ? e : 0] : lst[((e - KR) * R + 2 * j + 1) * MAXT];
            aggA += p0 * x0 + p1 * x1;                          // (zeros beyond the grid)
            aggA2 += ash[rw.row[2 * j]] * x0 + ash[rw.row[2 * j + 1]] * x1;
        }
    const double s = xwave_reduce63(aggA + aggBp), s2 = xwave_reduce63(aggA2);
    if (lane == 63) { aggred[(P & 1) * 16 + wv] = s; aggred2[(P & 1) * 16 + wv] = s2; }
    __syncthreads();
    if (tid == 0) {
        if (P >= 2) {
            double v = 0.0, v2 = 0.0;
            for (int k = 0; k < NWM; k++) { v += aggred[((P - 1) & 1) * 16 + k]; v2 += aggred2[((P - 1) & 1) * 16 + k]; }
            outn[P - 2] = v; outn2[P - 2] = v2;
        }
        double v = 0.0, v2 = 0.0;
        for (int k = 0; k < NWM; k++) { v += aggred[(P & 1) * 16 + k]; v2 += aggred2[(P & 1) * 16 + k]; }
        outn[P - 1] = v; outn2[P - 1] = v2;
    }
}

// once per recorded primal: the bracket with ONE bit of what the interpolation weights held — "both zero" (flat extrapolation, a
// blocked max rule: KrusellSmith.jl:71-76). k_wide_back rebuilds A and B from the exchanged knots and reads this instead of them.
__global__ void k_wide_prep(const int *__restrict__ ib, const double *__restrict__ A, const double *__restrict__ B, size_t n, int *__restrict__ ibw) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ibw[i] = ib[i] | ((A[i] == 0.0 && B[i] == 0.0) ? (int)0x80000000 : 0);
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
