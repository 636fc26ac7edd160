// hank_xsweep.h — the XCD-local persistent sweeps: ONE launch per sweep instead of one per period.
//
// The same recurrences as hank_kernels.h (BackwardIteration.jl:90-113, ForwardIteration.jl:297-308 and their partials
// under Dual{Tag,Float64,N}), behind the same entry points and on the same record, as four persistent kernels:
//   k_xprimal_back / k_xprimal_fwd : the Float64 recurrences (they record the linearisation),
//   k_xtan_back<D> / k_xtan_fwd<D> : the N partials as linear recurrences at that record (the whole y-iteration keeps x
//                                    fixed, NewtonRaphson.jl:91-111; so does every pass of a wide batch).
//
// Mapping (MI355X: 8 XCDs x 32 CUs, one 4 MiB L2 per XCD, L2s not coherent with each other):
//   * a launch has one workgroup per CU; each workgroup reads the XCD it actually runs on (HW_REG_XCC_ID) and takes a
//     ticket there: the workgroups of one XCD form a GROUP. Nothing assumes a placement — a group that does not have
//     the members it needs reports XERR_PLACEMENT and the host falls back to the per-period launches.
//   * inside a group, workgroup c owns 63 wealth rows (all n_e productivity columns: wave = column, lane = row).
//   * tangent sweeps: group x owns the directions [x*D, (x+1)*D) of the pass (D = 1, 2 or 4; 8*D directions per pass);
//     Float64 sweeps: the group of XCD 0 runs them, the other workgroups leave at once.
//   * the loop-carried state (EGM knots s_t / their partials ds_t backward; D_t / dD_t forward) lives in a small
//     ping-pong buffer in the XCD's L2: written with plain stores (the line stays in that L2), read back — after ONE
//     group barrier per period — with sc1 loads, which bypass the reading CU's L1 and are served by that same L2. No
//     kernel boundary, no fence, no atomics in the loop.
//   * group barrier: every storing wave drains (vmcnt(0)), raw s_barrier, the member publishes the episode number in ITS
//     flag line with a plain store, one wave polls all members' lines with one sc1 load per trip, s_barrier. It comes in
//     two halves (arrive / wait) so that loads which do not depend on the other members fly while the group meets.
//     Every spin is bounded; on timeout a status word is set and every later wait falls through: the grid always drains.
//   * what a period reads that is uniform over the workgroup (r_t, w_t, rho_t, dr/dw/dtr of the group, the clamped prefix
//     lengths) is LDS-resident for the whole sweep: as a per-period global load each is a cold round trip on the
//     critical path.
//
// Layouts:  Float64 state   st_s [2][XG][G] (backward), st_D [2][XG][G + 64*n_e] (forward; the tail holds the virtual rows)
//           tangent state   D/2 PLANES of 16-byte pairs, [plane][2][XG][rows][2] (D = 1: [2][XG][rows]): a store instruction
//                           writes 16 contiguous bytes per lane, whole lines per wave (XRows). rows = G backward; forward
//                           [n_e][members][64]: a member's 63 rows + its virtual row (slot 63) are one line-aligned block
//           dpol            [P][groups][n_e][n_a][D] — a group's stream is contiguous: whole lines, one XCD each.
// What was measured on the way (a dual kernel carrying value and partials together, a run-ahead wave, atomic-counter
// barriers, ...) is in DESIGN.md section 4.
#pragma once
#ifndef HANK_XFWD_PRECOMBINE
#define HANK_XFWD_PRECOMBINE 1     // k_xfwd: neighbouring lanes' parts for one tile row in one LDS add (dev knob: 0 = one add per part)
#endif
#include "hank_kernels.h"
#include <type_traits>

namespace hank {

constexpr int XG = 8;             // groups = XCDs
constexpr int XRW = 63;           // wealth rows per workgroup; lane 63 of every wave is the forward sweep's virtual row
enum { XERR_TIMEOUT = 1, XERR_PLACEMENT = 2 };
// Every wait is bounded in TIME: s_memrealtime counts at 100 MHz for the whole chip, a wait that has lasted longer than
// g_xwait_ticks (20 ms unless HANK_XWAIT_MS says otherwise at hank_create: four thousand periods' worth) marks the launch failed
// and every later wait falls through, so the grid always drains and the host learns how long the sweep had waited.
__device__ unsigned long long g_xwait_ticks = 2000000ull;
struct XDeadline {
    unsigned long long t0;
    __device__ __forceinline__ XDeadline() : t0(__builtin_amdgcn_s_memrealtime()) {}
    __device__ __forceinline__ bool expired(unsigned spins) const { return (spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > g_xwait_ticks; }
    __device__ __forceinline__ unsigned waited_us() const { return (unsigned)((__builtin_amdgcn_s_memrealtime() - t0) / 100ull); }
};

struct XSync {                    // zeroed by a memset node before EVERY launch (66.9 KB, a multiple of 16)
    unsigned ticket[XG][32];      // [x][0]: workgroups that arrived on XCD x (one 128-B line each)
    unsigned total[32];           // [0]: workgroups that hold a ticket
    unsigned flag[XG][64][32];    // [x][c][0]: the last barrier episode member c of group x has reached — one 128-B line per
                                  // member: 32 writers and 32 pollers on ONE line queue at one L2 channel (1.0-1.15 us per
                                  // episode against 0.71, scripts/ubench/xbar_bench.hip)
    unsigned status[32];          // [0]: XERR_* (sticky), [1]: the XCD that raised it, [2]: microseconds the wait that timed out had lasted
    unsigned pad[32];
};

typedef unsigned int xv4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double xld(const double *p) {      // 8-byte sc1 load: bypasses L1, served by the XCD's L2
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned xldu(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one row of D partials (8*D bytes, 8*D-aligned) with sc1 loads: 16-byte buffer loads where the row allows
// The state of D partials per row is kept as D/2 PLANES of 16-byte pairs ([plane][row][2]; D = 1: [row]): one store
// instruction then writes 16 contiguous bytes per lane, whole lines per wave. With [row][D] a D = 4 row took two store
// instructions that each wrote half of every 32-byte sector.
template <int D>
struct XRows {
    __amdgpu_buffer_rsrc_t rs;
    double *base;
    size_t plane;               // rows per plane = rows of the whole buffer
    __device__ __forceinline__ void init(double *p, size_t rows_total) {
        base = p; plane = rows_total;
        if (D >= 2) rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(rows_total * 8 * D), 0x00020000);
    }
    // row index counts rows from `base` (ping-pong half and group are part of the row index)
    __device__ __forceinline__ void load(size_t row, double *v) const {
        if (D == 1) {
            v[0] = xld(base + row);
        } else {
#pragma unroll
            for (int k = 0; k < D / 2; k++) {
                const xv4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((k * plane + row) * 16), 0, 16);   // aux 16 = sc1
                v[2 * k] = __hiloint2double((int)q.y, (int)q.x);
                v[2 * k + 1] = __hiloint2double((int)q.w, (int)q.z);
            }
        }
    }
    __device__ __forceinline__ double load_one(size_t row, int k) const {            // slot k of a row (sc1, 8 bytes)
        return D == 1 ? xld(base + row) : xld(base + ((size_t)(k / 2) * plane + row) * 2 + (k & 1));
    }
    __device__ __forceinline__ void store(size_t row, const double *v) const {      // plain stores: the line stays in the XCD's L2
        if (D == 1) {
            base[row] = v[0];
        } else {
#pragma unroll
            for (int k = 0; k < D / 2; k++) reinterpret_cast<double2 *>(base)[k * plane + row] = make_double2(v[2 * k], v[2 * k + 1]);
        }
    }
};
// the policy partials are a pure stream: written once by the backward sweep, read once by the forward sweep. Stores are
// nontemporal (the lines are not kept for a reader that comes a whole sweep later: backward sweep 1.36 -> 1.28 ms at
// N = 32); loads stay plain — a row is one member's own row AND a source of its neighbours in the same period, and with
// nontemporal loads the forward sweep took 3.46 ms instead of 2.25 (profiles/r03_col_nt_stream.log). L2 is write-through
// either way: every stored byte reaches the fabric once (WRITE_SIZE = stream + state with both flavours).
typedef double xv2d __attribute__((ext_vector_type(2)));
template <int D>
__device__ __forceinline__ void xstore_row(double *p, const double *v) {
    if (D == 1) {
        __builtin_nontemporal_store(v[0], p);
    } else {
#pragma unroll
        for (int k = 0; k < D / 2; k++) { xv2d q; q.x = v[2 * k]; q.y = v[2 * k + 1]; __builtin_nontemporal_store(q, reinterpret_cast<xv2d *>(p) + k); }
    }
}
template <int D>
__device__ __forceinline__ void xload_row_plain(const double *p, double *v) {  // read-only inputs (written by an earlier launch)
    if (D == 1) {
        v[0] = *p;
    } else {
#pragma unroll
        for (int k = 0; k < D / 2; k++) { const double2 q = reinterpret_cast<const double2 *>(p)[k]; v[2 * k] = q.x; v[2 * k + 1] = q.y; }
    }
}

struct XGroup { int x, c, S, ok; };

// dev build only (make stamp): s_memtime stamps of lane 0 / wave 0 of the first and last member of group 0, periods
// [XSTAMP_T0, XSTAMP_T0+8): where a period's time goes. Never compiled into the product library.
#ifdef HANK_XSTAMP
#ifndef HANK_XSTAMP_T0
#define HANK_XSTAMP_T0 100
#endif
#ifndef HANK_XSTAMP_STRIDE
#define HANK_XSTAMP_STRIDE 1       // > 1: every STRIDE-th period from T0 on is stamped (where in the sweep the time goes)
#endif
constexpr int XSTAMP_T0 = HANK_XSTAMP_T0, XSTAMP_NP = 8, XSTAMP_NS = 12, XSTAMP_ST = HANK_XSTAMP_STRIDE;
#define XSTAMP_ON(per) ((per) >= XSTAMP_T0 && (per) < XSTAMP_T0 + XSTAMP_NP * XSTAMP_ST && ((per) - XSTAMP_T0) % XSTAMP_ST == 0)
#define XSTAMP_SLOT(per) (((per) - XSTAMP_T0) / XSTAMP_ST)
__device__ unsigned long long g_xstamps[2][32][XSTAMP_NP][XSTAMP_NS];     // [sweep][member of group 0][period][stamp]
#define XSTAMP(sw, on, per, i)                                                                                   \
    do {                                                                                                         \
        if ((on) >= 0 && XSTAMP_ON(per) && threadIdx.x == 0)                 \
            g_xstamps[sw][on][XSTAMP_SLOT(per)][i] = __builtin_amdgcn_s_memrealtime();                              \
    } while (0)
// stamp taken by lane 0 of wave `wave` (arrival of the other waves at a workgroup barrier)
#define XSTAMPW(sw, on, per, i, wave)                                                                            \
    do {                                                                                                         \
        if ((on) >= 0 && XSTAMP_ON(per) && (int)threadIdx.x == 64 * (wave))  \
            g_xstamps[sw][on][XSTAMP_SLOT(per)][i] = __builtin_amdgcn_s_memrealtime();                              \
    } while (0)
// every wave's arrival at one chosen point of the period (lane 0 of each wave)
__device__ unsigned long long g_xwaves[2][32][XSTAMP_NP][16];
#define XSTAMPV(sw, on, per)                                                                                     \
    do {                                                                                                         \
        if ((on) >= 0 && XSTAMP_ON(per) && (threadIdx.x & 63) == 0)          \
            g_xwaves[sw][on][XSTAMP_SLOT(per)][threadIdx.x >> 6] = __builtin_amdgcn_s_memrealtime();             \
    } while (0)
// once per kernel (entry, prologue done, loop done): slot 0, stamps 9..11
#define XSTAMP1(sw, on, i) do { if ((on) >= 0 && threadIdx.x == 0) g_xstamps[sw][on][0][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XSTAMP1(sw, on, i) do {} while (0)
#define XSTAMPV(sw, on, per) do {} while (0)
#define XSTAMP(sw, on, per, i) do {} while (0)
#define XSTAMPW(sw, on, per, i, wave) do {} while (0)
#endif

__device__ inline void xfail(XSync *sy, unsigned code, int x, unsigned waited_us = 0u) {
    if (__hip_atomic_exchange(&sy->status[0], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __hip_atomic_store(&sy->status[1], (unsigned)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sy->status[2], waited_us, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// dev knob HANK_XFAULT: the status word of a launch's sync block pre-set by the host
__global__ void k_xpoison(XSync *base, int count, unsigned code) {
    if ((int)threadIdx.x < count) { base[threadIdx.x].status[0] = code; base[threadIdx.x].status[1] = 0u; }
}

// which XCD am I on, which member of its group am I, how many members does it have (known once EVERY workgroup of the
// launch holds a ticket: one launch-wide wait at the start of the sweep, none afterwards)
__device__ inline XGroup xgroup_join(XSync *sy, int *ctl) {
    if (threadIdx.x == 0) {
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        if (xldu(&sy->status[0]) != 0u) {       // the launch is already marked failed (a sibling gave up; the dev knob): leave at once
            ctl[0] = xcc; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0;
        } else {
            const unsigned c = __hip_atomic_fetch_add(&sy->ticket[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the ticket is taken before it is counted
            __hip_atomic_fetch_add(&sy->total[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int ok = 1;
            const XDeadline dl;
            for (unsigned spins = 0;; spins++) {
                if (xldu(&sy->total[0]) >= gridDim.x) break;
                if (dl.expired(spins) || ((spins & 255u) == 255u && xldu(&sy->status[0]) != 0u)) { xfail(sy, XERR_TIMEOUT, xcc, dl.waited_us()); ok = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            ctl[0] = xcc; ctl[1] = (int)c; ctl[2] = (int)xldu(&sy->ticket[xcc][0]); ctl[3] = ok;
        }
    }
    __syncthreads();
    XGroup g;
    g.x = __builtin_amdgcn_readfirstlane(ctl[0]); g.c = __builtin_amdgcn_readfirstlane(ctl[1]);
    g.S = __builtin_amdgcn_readfirstlane(ctl[2]); g.ok = __builtin_amdgcn_readfirstlane(ctl[3]);
    return g;
}

// workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, NOT for its global loads and stores
// (__syncthreads() carries a fence that drains vmcnt: the dpol stores and the run-ahead touches would stall here)
__device__ __forceinline__ void xlds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The group barrier. No atomics (they execute at the memory side, a fabric round trip each): member c publishes the episode
// number in ITS flag line with a plain store — the line lives in the XCD's L2 like the state itself — and one wave of
// every member polls all members' lines with ONE sc1 load per trip.
// It comes in two halves, so that loads which do not depend on the other members (next period's record) can be
// issued BETWEEN them and fly while the group meets: arrive = every storing wave drains, the workgroup meets; wait = the
// SYNC wave (an extra wave with no memory traffic of its own where the block has room for one, else wave 0) publishes
// this member's episode and polls the group's flags, then the workgroup meets again.
__device__ __forceinline__ void xbar_arrive(bool drain) {
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xlds_barrier();
}
__device__ __forceinline__ void xbar_wait(XSync *sy, int x, int c, int members, unsigned episode, bool sync_wave) {
    if (sync_wave) {
        const int lane = threadIdx.x & 63;
        if (lane == 0) *reinterpret_cast<volatile unsigned *>(&sy->flag[x][c][0]) = episode;
        const XDeadline dl;
        for (unsigned spins = 0;; spins++) {
            const unsigned f = lane < members ? xldu(&sy->flag[x][lane][0]) : episode;
            if (__all((int)(f - episode) >= 0)) break;
            if (dl.expired(spins) || ((spins & 255u) == 255u && xldu(&sy->status[0]) != 0u)) {
                if (lane == 0) xfail(sy, XERR_TIMEOUT, x, dl.waited_us());
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    xlds_barrier();
}

// The tangent sweeps split the barrier further. A member's period t needs the state rows of FEW other members (the
// brackets / lottery segments of its 63 rows: recorded by the primal, so the range of members is known in advance), and
// it may overwrite a state half only when EVERY member is done reading it. So: (1) before a period's gathers wait only
// for the source members' episode; (2) before the period's state stores — a whole gather-and-mix phase later — check that
// all members have published the previous episode; (3) publish after the stores have drained, and wait for nobody.
// Arrival skew up to the length of the first phase disappears from the critical path. xpoll: one wave, bounded.
__device__ __forceinline__ void xpoll(XSync *sy, int x, int lo, int hi, unsigned need) {
    const int lane = threadIdx.x & 63;
    const XDeadline dl;
    for (unsigned spins = 0;; spins++) {
        const unsigned f = (lane >= lo && lane <= hi) ? xldu(&sy->flag[x][lane][0]) : need;
        if (__all((int)(f - need) >= 0)) break;
        if (dl.expired(spins) || ((spins & 255u) == 255u && xldu(&sy->status[0]) != 0u)) {
            if (lane == 0) xfail(sy, XERR_TIMEOUT, x, dl.waited_us());
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void xpublish(XSync *sy, int x, int c, unsigned episode) {
    if ((threadIdx.x & 63) == 0) *reinterpret_cast<volatile unsigned *>(&sy->flag[x][c][0]) = episode;
}

// knots of one column, read from the L2-resident state. Everything egm_Y is going to look at in the usual case — the
// row's own knot and its lower neighbour (sortedness check), the column's two end knots (flat extrapolation) and the
// four knots around the guessed bracket — is fetched in ONE batch of independent loads up front: behind the branches
// of the bracket search each of them would be a round trip of its own. Anything else (a gallop after a far move) is
// loaded on demand.
struct XKnots {
    const double *p;
    int a, n, i0, i1, i2, i3;
    double sa, sam1, s0, sN, w0, w1, w2, w3;
    __device__ __forceinline__ void preload(const double *col, int a_, int n_, int guess) {
        p = col; a = a_; n = n_;
        sa = xld(col + a);
        sam1 = xld(col + (a > 0 ? a - 1 : 0));
        s0 = xld(col);
        sN = xld(col + n - 1);
        i0 = i1 = i2 = i3 = -1;
        w0 = w1 = w2 = w3 = 0.0;
        if (guess >= 0) {
            const int q = guess < n - 1 ? guess : n - 2;
            i0 = q > 0 ? q - 1 : 0; i1 = q; i2 = q + 1; i3 = q + 2 < n ? q + 2 : n - 1;
            w0 = xld(col + i0); w1 = xld(col + i1); w2 = xld(col + i2); w3 = xld(col + i3);
        }
    }
    __device__ __forceinline__ double operator[](int i) const {
        if (i == a) return sa;
        if (i == a - 1) return sam1;
        if (i == i1) return w1;
        if (i == i2) return w2;
        if (i == i0) return w0;
        if (i == i3) return w3;
        if (i == 0) return s0;
        if (i == n - 1) return sN;
        return xld(p + i);
    }
};

__device__ __forceinline__ double xwave_sum(double v) {            // butterfly: every lane ends with the same sum, fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the 64 lanes with DPP moves (no LDS traffic), fixed order; the result is valid in lane 63 ONLY
template <int CTRL, int RM, int BM>
__device__ __forceinline__ double xdpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, RM, BM, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, RM, BM, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xwave_reduce63(double v) {
    double s = v + xdpp<0x111, 0xf, 0xf>(v);     // row_shr:1
    s = s + xdpp<0x112, 0xf, 0xf>(v);            // row_shr:2
    s = s + xdpp<0x113, 0xf, 0xf>(v);            // row_shr:3
    s = s + xdpp<0x114, 0xf, 0xe>(s);            // row_shr:4, banks 1..3
    s = s + xdpp<0x118, 0xf, 0xc>(s);            // row_shr:8, banks 2..3
    s = s + xdpp<0x142, 0xa, 0xf>(s);            // row_bcast:15 into rows 1 and 3
    s = s + xdpp<0x143, 0xc, 0xf>(s);            // row_bcast:31 into rows 2 and 3
    return s;
}

// The LDS tile of the n_e-wide mixing: [column k][lane][slot], slot 0 = the value, 1..D = the partials, padded to an
// even count SL so that a lane's slots are one run of 16-byte pieces (ds_read/write_b128; lane stride 8*SL bytes is
// conflict-free for SL = 2, 4, 6). One pass over k serves the value and every partial.
template <int SL>
__device__ __forceinline__ void xtile_store(double *t, const double *v) {      // v holds SL values
    if (SL == 1) { t[0] = v[0]; return; }
#pragma unroll
    for (int q = 0; q < SL / 2; q++) reinterpret_cast<double2 *>(t)[q] = make_double2(v[2 * q], v[2 * q + 1]);
}
template <int SL, int NS>
__device__ __forceinline__ void xtile_store_n(double *t, const double *v) {    // v holds NS <= SL values; the padding slots stay unwritten
    if (SL == 1) { t[0] = v[0]; return; }
#pragma unroll
    for (int q = 0; q < NS / 2; q++) reinterpret_cast<double2 *>(t)[q] = make_double2(v[2 * q], v[2 * q + 1]);
    if (NS & 1) t[NS - 1] = v[NS - 1];
}
// out[s] = sum_k P[k*ps] * tile[k][lane][s], k ascending, first term unrounded-added (same order as mix_sum)
// FMA: the forward sweeps (round 5) fuse the multiply-adds — a rounding away from the launches' sums, half the instructions of the
// mixing (the library is built with -ffp-contract=off: a sum written p * v + acc is two instructions)
template <int SL, int NS, bool FMA = false>
__device__ __forceinline__ void xtile_mix(const double *tl, const double *P, int ps, int ne, double *out) {
    const int ks = 64 * SL;
#pragma unroll 4
    for (int k = 0; k < ne; k++) {
        double v[SL];
        if (SL == 1) v[0] = tl[(size_t)k * ks];
        else {
#pragma unroll
            for (int q = 0; q < (NS + 1) / 2; q++) { const double2 d = reinterpret_cast<const double2 *>(tl + (size_t)k * ks)[q]; v[2 * q] = d.x; v[2 * q + 1] = d.y; }
        }
        const double p = P[k * ps];
#pragma unroll
        for (int q = 0; q < NS; q++) out[q] = k == 0 ? p * v[q] : (FMA ? __builtin_fma(p, v[q], out[q]) : out[q] + p * v[q]);
    }
}

// the same with the column of Pi held in registers (16 unrolled steps, the ones beyond n_e predicated off): no LDS read for
// the coefficient, and every tile read of the pass can be in flight at once
template <int SL, int NS, bool FMA = false>
__device__ __forceinline__ void xtile_mix_reg(const double *tl, const double (&pr)[16], int ne, double *out) {
    const int ks = 64 * SL;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < ne) {
            double v[SL];
            if (SL == 1) v[0] = tl[(size_t)k * ks];
            else {
#pragma unroll
                for (int q = 0; q < (NS + 1) / 2; q++) { const double2 d = reinterpret_cast<const double2 *>(tl + (size_t)k * ks)[q]; v[2 * q] = d.x; v[2 * q + 1] = d.y; }
            }
#pragma unroll
            for (int q = 0; q < NS; q++) out[q] = k == 0 ? pr[k] * v[q] : (FMA ? __builtin_fma(pr[k], v[q], out[q]) : out[q] + pr[k] * v[q]);
        }
    }
}

// ================================ the Float64 sweeps ==========================================
// One group (the XCD with id 0) runs them; the workgroups that landed elsewhere leave at once. They write the policy
// sequence, the distribution path and the linearisation record the tangent sweeps — of either implementation — read.
struct XBackArgs {
    Consts c;
    const double *ss_value;     // [G] terminal marginal value (BackwardIteration.jl:85)
    const double *xhh;          // [n_hh*P]
    const double *rho;          // [P] 1/(1+r_t)
    XSync *sy;
    double *st_s;               // [2][XG][G] knots (see top)
    int *err;                   // device error word of the context (knots / domain)
    Record R;                   // pol; s, kc, ib, A, B, u, v: the linearisation
};

template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_xprimal_back(XBackArgs A) {
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G;
    double *Vsh = xl;                                   // [ne][64]
    double *Pish = Vsh + (size_t)ne * 64;               // [ne*ne] (registers for it would spill the 1024-thread variant)
    double *ash = Pish + ne * ne;                      // [na]: the wealth grid (the bracket's grid values are a dependent load)
    double *xsh = ash + na;                             // [P][4]: r_t, w_t, tr_t, rho_t — a cold uniform load per period otherwise
    int *ctl = reinterpret_cast<int *>(xsh + 4 * (size_t)P);
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x != 0) return;
    const int Sact = (na + XRW - 1) / XRW;              // members that own rows
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }   // the whole group agrees on S
    if (cW >= Sact) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the wave index in a scalar register: column bases become scalar)
    const bool syncw = wv >= ne;                        // see k_xtan_back
    const bool sync_duty = blockDim.x > 64 * ne ? syncw : wv == 0;
    const int e = syncw ? 0 : wv;
    const int a = cW * XRW + lane;
    const bool own = !syncw && lane < XRW && a < na;
    const size_t pt = (size_t)e * na + (own ? a : 0);
    for (int k = threadIdx.x; k < ne * ne; k += blockDim.x) Pish[k] = c.Pi[k];
    for (int k = threadIdx.x; k < na; k += blockDim.x) ash[k] = c.a[k];
    for (int k = threadIdx.x; k < P; k += blockDim.x) {
        xsh[4 * k] = A.xhh[c.n_hh * k]; xsh[4 * k + 1] = A.xhh[c.n_hh * k + 1]; xsh[4 * k + 2] = hh_tr(c, A.xhh, k); xsh[4 * k + 3] = A.rho[k];
    }
    Consts cl = c;                                      // what egm_Y sees: the same model, grid served from LDS
    cl.a = ash;
    const double ze = c.z[e], xa = c.a[own ? a : 0];
    const size_t hs = (size_t)XG * G;                   // the other half of the ping-pong state
    double *const sS = A.st_s;
    if (!syncw) Vsh[e * 64 + lane] = own ? A.ss_value[pt] : 0.0;      // terminal value (BackwardIteration.jl:85)
    __syncthreads();
    int guess = -1;
    unsigned episode = 0;
    // sequence: X(P-1) | Y(P-1) X(P-2) | ... | Y(1) X(0) | Y(0), one group barrier after every X
    for (int i = 0; i <= P; i++) {
        if (i > 0) {
            // ---- Y half of period t (KrusellSmith.jl:66-80): bracket search in the L2-resident knots -> policy, marginal value -> LDS
            const int t = P - i, cur = (i - 1) & 1;
            double V = 0.0;
            if (own) {
                XKnots kn;
                kn.preload(sS + (size_t)cur * hs + (size_t)e * na, a, na, guess);
                const YOut o = egm_Y(cl, kn, a, e, xsh[4 * t], xsh[4 * t + 1], xsh[4 * t + 2], A.err, t, guess);
                guess = o.ib;
                V = o.V;
                const size_t ro = (size_t)t * G + pt;
                A.R.pol[ro] = o.g; A.R.ib[ro] = o.ib; A.R.A[ro] = o.A; A.R.B[ro] = o.B; A.R.u[ro] = o.u; A.R.v[ro] = o.v;
            }
            if (!syncw) Vsh[e * 64 + lane] = V;
            xlds_barrier();
        }
        if (i < P) {
            // ---- X half of period tx from V_{tx+1} in LDS (KrusellSmith.jl:59-62; same expressions as egm_X): the knots s_tx -> state[i & 1]
            const int tx = P - 1 - i;
            if (own) {
                double E;
                xtile_mix<1, 1>(Vsh + lane, Pish + e, ne, ne, &E);
                const double bE = E * c.beta;
                const double ex = -1.0 / c.gamma;
                if (pow_domain_error(bE, ex)) set_err(A.err, ERR_DOMAIN, tx, e, a);
                const double cm = pow_crra(bE, ex);
                const double rho = xsh[4 * tx + 3];             // 1/(1+r_tx), once per period (k_xrho), not once per thread
                const double s1 = rho * ((cm - (xsh[4 * tx + 1] * ze + xsh[4 * tx + 2])) + xa);
                const double kc = c.diet ? diet_kc(c, s1, rho, 1.0 + xsh[4 * tx], xsh[4 * tx + 1] * ze + xsh[4 * tx + 2], xa) : rho * (c.beta * ex * (cm / bE));
                sS[(size_t)(i & 1) * hs + pt] = s1;
                A.R.s[(size_t)tx * G + pt] = s1; A.R.kc[(size_t)tx * G + pt] = kc;
            }
            episode++;
            xbar_arrive(!syncw);
            xbar_wait(A.sy, x, cW, Sact, episode, sync_duty);
        }
    }
}

// ---- the steady state's inner fixed point as ONE launch (hank_vfi; SteadyState.jl:132-141) ----------------------------
// value <- value_fn(value, xVals).Value until max|value' - value| < tol, on the group of XCD 0 exactly like k_xprimal_back
// (constant prices, nothing recorded). The group barrier carries the vote: every member publishes, in the same 16-byte
// store as its episode number, whether ITS rows have converged (compared in Float64 against tol, like k_vfi_check:
// max < tol over all rows <=> every member's max < tol; a NaN never converges), whether it has seen the device error word
// set, and its max; every member reads every line, so all of them leave the loop in the same trip.
struct XVfiArgs {
    Consts c;
    const double *V0;           // [G] start value
    double r, w, tr;
    double tol;
    int max_iter;
    XSync *sy;
    double *st_s;               // [2][XG][G] knots
    int *err;
    double *Vout, *pol;         // [G] the converged value and its policy
    int *iters;                 // [0] steps taken, [1] 1 = converged
    double *supnorm;            // max|value' - value| of the last step
};

template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_xvfi(XVfiArgs A) {
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const int ne = c.n_e, na = c.n_a, G = c.G;
    double *Vsh = xl;                                   // [ne][64]
    double *Pish = Vsh + (size_t)ne * 64;               // [ne*ne]
    double *ash = Pish + ne * ne;                       // [na]
    double *redsh = ash + na;                           // [16] per-wave max
    int *ctl = reinterpret_cast<int *>(redsh + 16);     // [4] group placement, [4..7] the vote's outcome
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x != 0) return;
    const int Sact = (na + XRW - 1) / XRW;
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }
    if (cW >= Sact) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the wave index in a scalar register: column bases become scalar)
    const bool syncw = wv >= ne;
    const bool sync_duty = blockDim.x > 64 * ne ? syncw : wv == 0;
    const int e = syncw ? 0 : wv;
    const int a = cW * XRW + lane;
    const bool own = !syncw && lane < XRW && a < na;
    const size_t pt = (size_t)e * na + (own ? a : 0);
    for (int k = threadIdx.x; k < ne * ne; k += blockDim.x) Pish[k] = c.Pi[k];
    for (int k = threadIdx.x; k < na; k += blockDim.x) ash[k] = c.a[k];
    Consts cl = c;
    cl.a = ash;
    const double ze = c.z[e], xa = c.a[own ? a : 0];
    const double rho = 1.0 / (1.0 + A.r);               // same expression as egm_X / k_xrho
    const size_t hs = (size_t)XG * G;
    double *const sS = A.st_s;
    double Vprev = own ? A.V0[pt] : 0.0, V = Vprev, pol = 0.0;
    if (!syncw) Vsh[e * 64 + lane] = Vprev;
    __syncthreads();
    int guess = -1, steps = 0, conv = 0;
    double gmax = 0.0;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(&A.sy->flag[x][0][0], 0, 64 * 32 * 4, 0x00020000);     // the group's flag lines
    unsigned episode = 0;
    // sequence: X(V_0) | Y -> V_1, vote, X(V_1) | Y -> V_2, vote, X(V_2) | ...: step k is complete after the Y half of trip k
    for (int i = 0; i <= A.max_iter; i++) {
        double dmax = 0.0;
        if (i > 0) {
            const int cur = (i - 1) & 1;
            if (own) {
                XKnots kn;
                kn.preload(sS + (size_t)cur * hs + (size_t)e * na, a, na, guess);
                const YOut o = egm_Y(cl, kn, a, e, A.r, A.w, A.tr, A.err, 0, guess);
                guess = o.ib;
                V = o.V; pol = o.g;
                dmax = fabs(V - Vprev);
                Vprev = V;
            }
            if (!syncw) Vsh[e * 64 + lane] = V;
            // this wave's max (a NaN wins and stays)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const double o2 = __shfl_xor(dmax, off, 64); dmax = (o2 > dmax || !(o2 == o2)) ? o2 : dmax; }
            if (lane == 0) redsh[wv] = syncw ? 0.0 : dmax;
            xlds_barrier();
        }
        // ---- X half from the value in LDS (it is discarded if the vote below ends the iteration)
        if (own) {
            double E;
            xtile_mix<1, 1>(Vsh + lane, Pish + e, ne, ne, &E);
            const double bE = E * c.beta;
            const double ex = -1.0 / c.gamma;
            if (pow_domain_error(bE, ex)) set_err(A.err, ERR_DOMAIN, 0, e, a);
            const double cm = pow_crra(bE, ex);
            sS[(size_t)(i & 1) * hs + pt] = rho * ((cm - (A.w * ze + A.tr)) + xa);
        }
        episode++;
        xbar_arrive(!syncw);
        if (sync_duty) {
            // publish {episode, my rows converged, error word seen, -} + my max in ONE 16-byte store, then read everybody's
            double m = 0.0;
            for (int k = 0; k < ne; k++) m = (redsh[k] > m || !(redsh[k] == redsh[k])) ? redsh[k] : m;
            const unsigned mine = (i > 0 && m < A.tol) ? 1u : 0u;
            const unsigned bad = xldu(reinterpret_cast<const unsigned *>(A.err)) != 0u ? 1u : 0u;
            if (lane == 0) {
                xv4u q;
                q.x = episode; q.y = mine | (bad << 1); q.z = (unsigned)__double2loint(m); q.w = (unsigned)__double2hiint(m);
                *reinterpret_cast<volatile xv4u *>(&A.sy->flag[x][cW][0]) = q;
            }
            xv4u f;
            f.x = episode; f.y = 1u; f.z = 0u; f.w = 0u;
            const XDeadline dl;
            for (unsigned spins = 0;; spins++) {
                if (lane < Sact) f = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 128, 0, 16);     // sc1: served by the XCD's L2
                if (__all((int)(f.x - episode) >= 0)) break;
                if (dl.expired(spins) || ((spins & 255u) == 255u && xldu(&A.sy->status[0]) != 0u)) {
                    if (lane == 0) xfail(A.sy, XERR_TIMEOUT, x, dl.waited_us());
                    f.y = 2u;                               // leave the loop: the host reads the status word
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const bool allc = __all((f.y & 1u) != 0u) != 0, anyb = __any((f.y & 2u) != 0u) != 0;
            double fm = __hiloint2double((int)f.w, (int)f.z);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const double o2 = __shfl_xor(fm, off, 64); fm = (o2 > fm || !(o2 == o2)) ? o2 : fm; }
            if (lane == 0) { ctl[4] = allc ? 1 : 0; ctl[5] = anyb ? 1 : 0; redsh[15] = fm; }
        }
        xlds_barrier();
        if (i > 0) { steps = i; gmax = redsh[15]; }
        conv = ctl[4];
        const int stop = conv | ctl[5];
        xlds_barrier();                                     // (ctl / redsh are rewritten in the next trip)
        if (i > 0 && stop) break;
    }
    if (own) { A.Vout[pt] = V; A.pol[pt] = pol; }
    if (cW == 0 && threadIdx.x == 0) { A.iters[0] = steps; A.iters[1] = conv; *A.supnorm = gmax; }
}

// ================================ tangent-only sweeps at a recorded primal ===================================
// The dual sweeps above cost what the Float64 recurrence costs — and every group repeats it (bracket search, two
// roots and four divisions per point and period: ~10x the work of one partial). At a FIXED x (the whole y-iteration,
// NewtonRaphson.jl:91-111; every pass of a wide batch) the primal sweep runs ONCE (the D = 0 instances above, which
// also record the linearisation) and the partials run as pure linear recurrences: these two kernels. Same groups, same
// state exchange through the XCD's L2, same barrier; per point and period a few FMAs per direction.
// slots of the tangent tile: D, padded where a lane stride of 8*D bytes would bank-conflict the 16-byte reads (D = 4: 32 B
// -> lanes i and i+8 collide, 31 % of the LDS cycles in profiles/r02a; 48 B is conflict-free)
template <int D> struct XTileT { static constexpr int SL = D == 4 ? 6 : (D == 8 ? 10 : D); };

// ---- the stationary distribution as ONE launch (hank_stationary_dist) ------------------------------------------------
// D <- Lambda(policy) D on the group of XCD 0 exactly like k_xprimal_fwd with ONE period's lottery record (k_lottery on the
// steady-state policy: seg, lw, clo), nothing recorded. Every `check_every` iterations each member compares its rows with
// its copy of the iterate `check_every` steps earlier (registers); the verdict rides on the group barrier as in k_xvfi.
// A member's virtual row is one of up to `members` parts of row 0: its part must move by less than tol / members, so that
// the rule is at least as strict as comparing the folded row (|sum| <= sum of |parts|).
struct XStatArgs {
    Consts c;
    Record R;                   // seg, lw, clo of the ONE lottery (period 0 of a private record)
    const double *D0;           // [G] start
    double tol;
    int max_iter, check_every;
    XSync *sy;
    double *st_D;               // [2][XG][G + 64*n_e]
    double *Dout;               // [G] the last iterate, the virtual mass folded into row 0 of its column
    int *iters;                 // [0] iterations run, [1] 1 = converged
};

template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_xstat(XStatArgs A) {
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const Record &R = A.R;
    const int ne = c.n_e, na = c.n_a, G = c.G;
    const int GV = G + 64 * ne;
    double *tile = xl;                                  // [ne][64]
    double *redsh = tile + (size_t)ne * 64;             // [16] per-wave max
    int *ctl = reinterpret_cast<int *>(redsh + 16);     // [4] placement, [4..5] the vote's outcome
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x != 0) return;
    const int Sact = (na + XRW - 1) / XRW;
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }
    if (cW >= Sact) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the wave index in a scalar register: column bases become scalar)
    const bool syncw = wv >= ne;
    const bool sync_duty = blockDim.x > 64 * ne ? syncw : wv == 0;
    const int e = syncw ? 0 : wv;
    const int r0 = cW * XRW, r = r0 + lane;
    const bool own = !syncw && lane < XRW && r < na;
    const bool virt = !syncw && lane == 63;
    const size_t pt = (size_t)e * na + (own ? r : 0);
    const size_t slot = own ? pt : (size_t)G + (size_t)e * 64 + cW;
    double pr[16];
#pragma unroll
    for (int k = 0; k < 16; k++) pr[k] = k < ne ? c.Pi[ne * e + k] : 0.0;
    const size_t hs = (size_t)XG * GV;
    double *const sP = A.st_D;
    double Dcur = own ? A.D0[pt] : 0.0;
    if (own || virt) sP[slot] = Dcur;
    double Dchk = Dcur;                                 // this lane's entry of the iterate check_every steps back
    // the lottery does not change: this lane's segments (as a target) and the clamped prefix of its column
    int sg0 = 0, sg1 = 0, sg2 = 0;
    if (own) { const int4 q = R.seg[pt]; sg0 = q.x; sg1 = q.y; sg2 = q.z; }
    const int clo = syncw ? 0 : min(max(R.clo[e], 0), na);
    bool anyclo = false;
    for (int k = 0; k < ne; k++) anyclo = anyclo || R.clo[k] > 0;
    const int s0 = max(sg0, 0), s2 = min(sg2, na);
    const bool need_vD = anyclo && clo == 0 && __any(own && (sg0 <= 0 && s2 > 0));
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(&A.sy->flag[x][0][0], 0, 64 * 32 * 4, 0x00020000);
    unsigned episode = 1;
    xbar_arrive(!syncw);
    xbar_wait(A.sy, x, cW, Sact, episode, sync_duty);
    int cur = 0, it = 0, conv = 0;
    bool vnz = false;                                   // the virtual rows hold mass from the first iteration on (if any column is clamped)
    for (it = 1; it <= A.max_iter; it++) {
        const size_t hb = (size_t)cur * hs;
        double accD = 0.0;
        if (!syncw) {
            const double *Dp = sP + hb + (size_t)e * na;
            double cD = 0.0;
            if (own && r < clo) cD = xld(Dp + r);
            if (virt && clo > 0 && vnz) cD = xld(sP + hb + slot);
            double vD = 0.0;
            if (vnz && need_vD) {
                if (lane < Sact) vD = xld(sP + hb + (size_t)G + (size_t)e * 64 + lane);
                vD = xwave_sum(vD);
            }
            if (own) {
                for (int j0 = s0; j0 < s2; j0 += 4) {
                    double wj[4], Dj[4];
                    bool on[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = j0 + u;
                        on[u] = j < s2;
                        wj[u] = 0.0; Dj[u] = 0.0;
                        if (on[u]) { wj[u] = R.lw[(size_t)e * na + j]; Dj[u] = xld(Dp + j); }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = j0 + u;
                        if (!on[u]) continue;
                        if (j == 0) Dj[u] += vD;
                        accD += (j < sg1 ? wj[u] : 1.0 - wj[u]) * Dj[u];
                    }
                }
            }
            if (clo > r0) cD = xwave_reduce63(cD);
            if (virt) accD = cD;
            tile[e * 64 + lane] = accD;
        }
        xlds_barrier();
        vnz = anyclo;
        const int nxt = cur ^ 1;
        const bool chk = (it % A.check_every) == 0 || it == A.max_iter;
        double dmax = 0.0;
        if (!syncw) {
            double Dn;
            xtile_mix_reg<1, 1>(tile + lane, pr, ne, &Dn);
            if (own || virt) sP[(size_t)nxt * hs + slot] = Dn;
            Dcur = Dn;
            if (chk) {
                if (own) dmax = fabs(Dn - Dchk);
                else if (virt) dmax = fabs(Dn - Dchk) * (double)Sact;      // one of up to Sact parts of row 0 (see top)
                Dchk = Dn;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const double o2 = __shfl_xor(dmax, off, 64); dmax = (o2 > dmax || !(o2 == o2)) ? o2 : dmax; }
            }
        }
        if (chk && lane == 0) redsh[wv] = syncw ? 0.0 : dmax;
        cur = nxt;
        episode++;
        xbar_arrive(!syncw);
        if (sync_duty) {
            double m = 0.0;
            if (chk) for (int k = 0; k < ne; k++) m = (redsh[k] > m || !(redsh[k] == redsh[k])) ? redsh[k] : m;
            if (lane == 0) {
                xv4u q;
                q.x = episode; q.y = (chk && m < A.tol) ? 1u : 0u; q.z = 0u; q.w = 0u;
                *reinterpret_cast<volatile xv4u *>(&A.sy->flag[x][cW][0]) = q;
            }
            xv4u f;
            f.x = episode; f.y = 1u; f.z = 0u; f.w = 0u;
            const XDeadline dl;
            for (unsigned spins = 0;; spins++) {
                if (lane < Sact) f = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 128, 0, 16);
                if (__all((int)(f.x - episode) >= 0)) break;
                if (dl.expired(spins) || ((spins & 255u) == 255u && xldu(&A.sy->status[0]) != 0u)) {
                    if (lane == 0) xfail(A.sy, XERR_TIMEOUT, x, dl.waited_us());
                    f.y = 2u;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const bool allc = __all((f.y & 1u) != 0u) != 0, anyb = __any((f.y & 2u) != 0u) != 0;      // (every lane votes: not inside the lane-0 branch)
            if (lane == 0) { ctl[4] = (chk && allc) ? 1 : 0; ctl[5] = anyb ? 1 : 0; }
        }
        xlds_barrier();
        conv = ctl[4];
        const int stop = conv | ctl[5];
        xlds_barrier();
        if (stop) break;
    }
    if (it > A.max_iter) it = A.max_iter;
    // the last iterate; row 0 of a column takes the mass kept on the members' virtual rows (member order fixed, like k_xfix_D)
    if (own) {
        double v = Dcur;
        if (r == 0) for (int m = 0; m < Sact; m++) v += xld(sP + (size_t)cur * hs + (size_t)G + (size_t)e * 64 + m);
        A.Dout[pt] = v;
    }
    if (cW == 0 && threadIdx.x == 0) { A.iters[0] = it; A.iters[1] = conv; }
}

struct XTanBackArgs {
    Consts c;
    Record R;                   // s, kc, ib, A, B, u, v of the recorded primal
    const double *rho;          // [P] 1/(1+r_t)
    const double *xhh;          // [n_hh*P] the household inputs of the recorded primal (record diet: see diet_kc)
    const double *dxr, *dxw, *dxt;
    int Ntot, n0, N;
    XSync *sy;
    double *st_ds;              // [2][XG][G][D]
    double *dpol;               // [P][groups][G][D]
    int groups;
    const int *src;             // [P][members] lo | hi << 8: the members whose rows period t's gathers of member c read (k_xsrc_back); null = all
    int stall;                  // dev knob HANK_XFAULT=stall: member 0 of group 0 stops publishing after three periods — a wait that really times out
};

template <int D, int MAXT>
__global__ void __launch_bounds__(MAXT) k_xtan_back(XTanBackArgs A) {
    constexpr int SL = XTileT<D>::SL;
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const Record &R = A.R;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G;
    double *tile = xl;                                  // [ne][64][SL]
    double *rhosh = tile + (size_t)SL * ne * 64;                    // [P]
    double *pxsh = rhosh + ((P + 1) & ~1);                       // [P][3]: 1 + r_t, w_t, tr_t (record diet)
    double *dxsh = pxsh + 3 * (size_t)P + (P & 1);               // [P][3][D]: this group's dr, dw, dtr (a cold uniform load per period otherwise)
    int *srcsh = reinterpret_cast<int *>(dxsh + (size_t)P * 3 * D);      // [P]: this member's source ranges
    int *ctl = srcsh + P;
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x >= A.groups) return;
    const int Sact = (na + XRW - 1) / XRW;
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }
    if (cW >= Sact) return;
    for (int k = threadIdx.x; k < P; k += blockDim.x) rhosh[k] = A.rho[k];
    for (int k = threadIdx.x; k < P; k += blockDim.x) { pxsh[3 * k] = 1.0 + A.xhh[c.n_hh * k]; pxsh[3 * k + 1] = A.xhh[c.n_hh * k + 1]; pxsh[3 * k + 2] = hh_tr(c, A.xhh, k); }
    for (int k = threadIdx.x; k < P; k += blockDim.x) srcsh[k] = A.src ? A.src[(size_t)k * Sact + cW] : ((Sact - 1) << 8);
    for (int k = threadIdx.x; k < P * D; k += blockDim.x) {
        const int t_ = k / D, d_ = k - t_ * D;
        const bool on = x * D + d_ < A.N;
        const size_t ix = (size_t)t_ * A.Ntot + A.n0 + x * D + d_;
        dxsh[(t_ * 3 + 0) * D + d_] = on ? A.dxr[ix] : 0.0;
        dxsh[(t_ * 3 + 1) * D + d_] = on ? A.dxw[ix] : 0.0;
        dxsh[(t_ * 3 + 2) * D + d_] = (on && c.n_hh > 2) ? A.dxt[ix] : 0.0;
    }
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the wave index in a scalar register: column bases become scalar)
    const bool syncw = wv >= ne;                        // the extra wave (when the block has one) only runs the group barrier's poll
    const bool sync_duty = blockDim.x > 64 * ne ? syncw : wv == 0;
    const int e = syncw ? 0 : wv;
    const int r0 = cW * XRW, a = r0 + lane;
    const bool own = !syncw && lane < XRW && a < na;
    const size_t pt = (size_t)e * na + (own ? a : 0);
    const double ze = c.z[e], xa = c.a[own ? a : 0];
    const size_t hs = (size_t)XG * G;
    XRows<D> rows;
    rows.init(A.st_ds, 2 * hs);
    const size_t gx = (size_t)x * G;                    // this group's rows within a half
    double *const myt = tile + ((size_t)e * 64 + lane) * SL;
    {
        double z[D];
#pragma unroll
        for (int k = 0; k < D; k++) z[k] = 0.0;
        if (!syncw) xtile_store_n<SL, D>(myt, z);       // dV_T = 0 (BackwardIteration.jl:85)
    }
    __syncthreads();
    double pr[16];                                      // Pi[e, k]: this wave's coefficients of the mixing
#pragma unroll
    for (int k = 0; k < 16; k++) pr[k] = k < ne ? c.Pi[e + ne * k] : 0.0;
    // the record of the period each half is about to use, fetched one trip ahead: the loads are issued between the two
    // halves of the group barrier and land while the group meets
    int ibY = 0;
    double cA = 0.0, cB = 0.0, cu = 0.0, cv = 0.0, ck = 0.0, cs = 0.0;
    if (own) { cs = R.s[(size_t)(P - 1) * G + pt]; if (!c.diet) ck = R.kc[(size_t)(P - 1) * G + pt]; }
    // ... and, since round 5, TWO trips ahead: these are cold lines of the record (2-3 us from HBM under this load) and a trip is
    // ~4 us, but the loads used to be issued behind the publish and needed right behind the next poll — a flag round (~1 us) later.
    // A second register set (n_*) is loaded a whole trip before it is needed and handed over at the end of the trip.
    int n_ib = 0;
    double n_A = 0.0, n_B = 0.0, n_u = 0.0, n_v = 0.0, n_k = 0.0, n_s = 0.0;
    auto load_next = [&](int ty) {                      // the record of Y(ty) and X(ty - 1); ty < 0: nothing left to load
        if (own && ty >= 0) {
            const size_t ro = (size_t)ty * G + pt;
            n_ib = R.ib[ro]; n_A = R.A[ro]; n_B = R.B[ro]; n_u = R.u[ro];
            if (!c.diet) n_v = R.v[ro];
            if (ty > 0) { n_s = R.s[ro - G]; if (!c.diet) n_k = R.kc[ro - G]; }
        }
    };
    // (narrow kernels only: single tangent 1.96 -> 1.90 ms, N = 16 2.40 -> 2.35; at D = 4 the second set costs 26 registers and the
    // trip — longer there — already covers most of the latency: 3.24 -> 3.28 ms, so D = 4 keeps the one-trip form)
    constexpr bool DEEP = D <= 2;
    if constexpr (DEEP) load_next(P - 1);               // what trip 1 uses
    // sequence: X(P-1) | Y(P-1) X(P-2) | ... | Y(1) X(0) | Y(0); member c publishes episode i+1 when the stores of trip i have drained
    const int son = (x == 0 && cW < 32) ? cW : -1;     // dev stamps (make stamp)
    (void)son;
    for (int i = 0; i <= P; i++) {
        XSTAMP(0, son, i, 0);
        if (i > 0) {
            // ---- Y-tangent of period t: dg = A ds[ib] + B ds[ib+1]; dV = u dr + v ((a dr + z dw + dtr) - dg)
            const int t = P - i, cur = (i - 1) & 1;
            if (sync_duty) xpoll(A.sy, x, srcsh[t] & 255, srcsh[t] >> 8, (unsigned)i);    // the members this period gathers from have published trip i-1
            xlds_barrier();
            XSTAMP(0, son, i, 1);
            double dV[D];
#pragma unroll
            for (int k = 0; k < D; k++) dV[k] = 0.0;
            if (own) {
                double d0[D], d1[D], dg[D];
#pragma unroll
                for (int k = 0; k < D; k++) d0[k] = d1[k] = 0.0;
                if (cA != 0.0 || cB != 0.0) {
                    const size_t rb = (size_t)cur * hs + gx + (size_t)e * na;
                    rows.load(rb + ibY, d0);
                    rows.load(rb + ibY + 1, d1);
                }
                const double cvt = c.diet ? diet_v(c, cu, pxsh[3 * t]) : cv;
#pragma unroll
                for (int k = 0; k < D; k++) {
                    const double dr = dxsh[(t * 3 + 0) * D + k], dw = dxsh[(t * 3 + 1) * D + k], dtr = dxsh[(t * 3 + 2) * D + k];
                    dg[k] = cA * d0[k] + cB * d1[k];
                    dV[k] = cu * dr + cvt * ((xa * dr + (ze * dw + dtr)) - dg[k]);
                }
                xstore_row<D>(A.dpol + (((size_t)t * A.groups + x) * G + pt) * D, dg);
            }
            if (!syncw) xtile_store_n<SL, D>(myt, dV);
            XSTAMP(0, son, i, 2);
            XSTAMPV(0, son, i);
            if (sync_duty) xpoll(A.sy, x, 0, Sact - 1, (unsigned)i);      // EVERY member is done reading the half the X half overwrites
            xlds_barrier();
            XSTAMP(0, son, i, 3);
        }
        if (i < P) {
            // ---- X-tangent of period tx: ds = kc dE - rho ((z dw + dtr) + s dr)
            const int tx = P - 1 - i;
            if (own) {
                const double rho = rhosh[tx];
                const double ckt = c.diet ? diet_kc(c, cs, rho, pxsh[3 * tx], pxsh[3 * tx + 1] * ze + pxsh[3 * tx + 2], xa) : ck;
                double mx[D], ds[D];
                xtile_mix_reg<SL, D>(tile + (size_t)lane * SL, pr, ne, mx);
#pragma unroll
                for (int k = 0; k < D; k++) {
                    const double dr1 = dxsh[(tx * 3 + 0) * D + k], dw1 = dxsh[(tx * 3 + 1) * D + k], dt1 = dxsh[(tx * 3 + 2) * D + k];
                    ds[k] = ckt * mx[k] - rho * ((ze * dw1 + dt1) + cs * dr1);
                }
                rows.store((size_t)(i & 1) * hs + gx + pt, ds);
            }
            XSTAMP(0, son, i, 4);
            xbar_arrive(!syncw);                                         // this member's stores have reached L2
            XSTAMP(0, son, i, 5);
            if (sync_duty && !(A.stall && x == 0 && cW == 0 && i > 2)) xpublish(A.sy, x, cW, (unsigned)(i + 1));
            // the record the next trip needs (Y of period tx, X of period tx - 1) was requested a trip ago: hand it over, and request
            // the one after (in flight through the whole next trip)
            if constexpr (DEEP) {
                ibY = n_ib; cA = n_A; cB = n_B; cu = n_u; cv = n_v;
                if (tx > 0) { cs = n_s; ck = n_k; }
                load_next(tx - 1);
            } else if (own) {      // in flight while the others arrive
                const size_t ro = (size_t)tx * G + pt;
                ibY = R.ib[ro]; cA = R.A[ro]; cB = R.B[ro]; cu = R.u[ro];
                if (!c.diet) cv = R.v[ro];
                if (tx > 0) { cs = R.s[ro - G]; if (!c.diet) ck = R.kc[ro - G]; }
            }
        }
    }
}

// slots of a sweep that carries NSL numbers per grid point (the D partials and, in a Dual pass, the value)
template <int NSL> struct XSlots {
    static constexpr int SP = NSL == 1 ? 1 : ((NSL + 1) / 2) * 2;        // slots of a state row (planes of 16-byte pairs)
    static constexpr int SL = NSL <= 2 ? NSL : (NSL <= 6 ? 6 : 10);      // slots of a tile entry (see XTileT)
};

// ---- the backward half of a Dual pass (hank_primal_jvp) as ONE launch: value AND D partials in every group -----------
// Every group repeats the Float64 EGM step (bit for bit the same in all of them: same expressions, same order) and carries
// its own D partials through it. What that buys over k_xprimal_back followed by k_xtan_back: one chain of P periods instead
// of two, and the tangent half never reads the record — the bracket and the coefficients of a row are the registers of the
// lane that has just computed them. What it costs: the EGM step's arithmetic in all eight groups at once (it is parallel,
// not serial: every XCD would otherwise idle behind XCD 0) and the partials' gather behind the bracket search instead of
// behind a prefetched bracket. The record and the policy for every later hank_jvp at this primal are written once, the arrays shared out over the groups (all hold the same numbers). The value
// rides as one more slot of the partials' LDS tile (slot D), so one pass over the columns mixes all of them.
struct XDualBackArgs {
    XBackArgs p;                // the Float64 sweep's arguments (st_s: [2][XG][G], a slice per group here)
    const double *dxr, *dxw, *dxt;
    int Ntot, n0, N;
    double *st_ds;              // [2][XG][G][D]
    double *dpol;               // [P][groups][G][D]
    int groups;
};

template <int D, int MAXT>
__global__ void __launch_bounds__(MAXT) k_xdual_back(XDualBackArgs B) {
    constexpr int NSL = D + 1, SL = XSlots<NSL>::SL, IV = D;
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const XBackArgs &A = B.p;
    const Consts &c = A.c;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G;
    double *tile = xl;                                  // [ne][64][SL]: slots 0..D-1 the partials of V, slot D the value
    double *ash = tile + (size_t)SL * ne * 64;          // [na]
    double *xsh = ash + ((na + 1) & ~1);                // [P][4]: r_t, w_t, tr_t, rho_t
    double *dxsh = xsh + 4 * (size_t)P;                 // [P][3][D]: this group's dr, dw, dtr
    int *ctl = reinterpret_cast<int *>(dxsh + (size_t)P * 3 * D);
#ifdef HANK_XSTAMP
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x >= B.groups) return;
    const int Sact = (na + XRW - 1) / XRW;
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }
    if (cW >= Sact) return;
    const int son = (x == 0 && cW < 32) ? cW : -1;     // dev stamps (make stamp)
    (void)son;
#ifdef HANK_XSTAMP
    if (son >= 0 && threadIdx.x == 0) g_xstamps[0][son][0][8] = t_entry;
#endif
    XSTAMP1(0, son, 9);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool syncw = wv >= ne;
    const bool sync_duty = blockDim.x > 64 * ne ? syncw : wv == 0;
    const int e = syncw ? 0 : wv;
    const int a = cW * XRW + lane;
    const bool own = !syncw && lane < XRW && a < na;
    // every group holds the same Float64 numbers: group x writes the record arrays j with j % groups == x (pol, ib, A, B, u, v, s, kc)
    const int ng = B.groups;
    const bool rc0 = 0 % ng == x, rc1 = 1 % ng == x, rc2 = 2 % ng == x, rc3 = 3 % ng == x, rc4 = 4 % ng == x, rc5 = 5 % ng == x, rc6 = 6 % ng == x, rc7 = 7 % ng == x;
    const size_t pt = (size_t)e * na + (own ? a : 0);
    for (int k = threadIdx.x; k < na; k += blockDim.x) ash[k] = c.a[k];
    for (int k = threadIdx.x; k < P; k += blockDim.x) {
        xsh[4 * k] = A.xhh[c.n_hh * k]; xsh[4 * k + 1] = A.xhh[c.n_hh * k + 1]; xsh[4 * k + 2] = hh_tr(c, A.xhh, k); xsh[4 * k + 3] = A.rho[k];
    }
    for (int k = threadIdx.x; k < P * D; k += blockDim.x) {
        const int t_ = k / D, d_ = k - t_ * D;
        const bool on = x * D + d_ < B.N;
        const size_t ix = (size_t)t_ * B.Ntot + B.n0 + x * D + d_;
        dxsh[(t_ * 3 + 0) * D + d_] = on ? B.dxr[ix] : 0.0;
        dxsh[(t_ * 3 + 1) * D + d_] = on ? B.dxw[ix] : 0.0;
        dxsh[(t_ * 3 + 2) * D + d_] = (on && c.n_hh > 2) ? B.dxt[ix] : 0.0;
    }
    Consts cl = c;
    cl.a = ash;
    const double ze = c.z[e], xa = c.a[own ? a : 0];
    const size_t hs = (size_t)XG * G, gx = (size_t)x * G;
    double *const sS = A.st_s;
    XRows<D> rows;
    rows.init(B.st_ds, 2 * hs);
    double *const myt = tile + ((size_t)e * 64 + lane) * SL;
    if (!syncw) {
        double z[NSL];
#pragma unroll
        for (int k = 0; k < D; k++) z[k] = 0.0;         // dV_T = 0, V_T = the terminal marginal value (BackwardIteration.jl:85)
        z[IV] = own ? A.ss_value[pt] : 0.0;
        xtile_store_n<SL, NSL>(myt, z);
    }
    __syncthreads();
    double pr[16];                                      // Pi[e, k]: this wave's coefficients of the mixing, in registers (1.94 -> 1.87 ms at
#pragma unroll                                          // N=32 against LDS reads; the kernel sits at 168 VGPRs = 3 waves per SIMD)
    for (int k = 0; k < 16; k++) pr[k] = k < ne ? c.Pi[e + ne * k] : 0.0;
    int guess = -1;
    unsigned episode = 0;
    XSTAMP1(0, son, 10);
    // sequence: X(P-1) | Y(P-1) X(P-2) | ... | Y(1) X(0) | Y(0), one group barrier after every X (as k_xprimal_back)
    for (int i = 0; i <= P; i++) {
        XSTAMP(0, son, i, 0);
        if (i > 0) {
            const int t = P - i, cur = (i - 1) & 1;
            double tv[NSL];
#pragma unroll
            for (int k = 0; k < NSL; k++) tv[k] = 0.0;
            if (own) {
                XKnots kn;
                kn.preload(sS + (size_t)cur * hs + gx + (size_t)e * na, a, na, guess);
                // the partials' rows at the bracket of the period before, in the same batch of loads as the knots: the bracket
                // rarely moves by more than a knot per period, and behind the search every gather is a round trip of its own
                const size_t rb = (size_t)cur * hs + gx + (size_t)e * na;
                const int q = guess < 0 ? 0 : (guess < na - 1 ? guess : na - 2);
                double w0[D], w1[D];
                rows.load(rb + q, w0);
                rows.load(rb + q + 1, w1);
                const YOut o = egm_Y(cl, kn, a, e, xsh[4 * t], xsh[4 * t + 1], xsh[4 * t + 2], A.err, t, guess);
                guess = o.ib;
                tv[IV] = o.V;
                XSTAMP(0, son, i, 1);
                // Y-tangent (k_xtan_back's expressions; the coefficients are this lane's own)
                double d0[D], d1[D], dg[D];
#pragma unroll
                for (int k = 0; k < D; k++) d0[k] = d1[k] = 0.0;
                if (o.A != 0.0 || o.B != 0.0) {
                    if (o.ib == q) {
#pragma unroll
                        for (int k = 0; k < D; k++) { d0[k] = w0[k]; d1[k] = w1[k]; }
                    } else if (o.ib == q + 1) {
#pragma unroll
                        for (int k = 0; k < D; k++) d0[k] = w1[k];
                        rows.load(rb + o.ib + 1, d1);
                    } else if (o.ib == q - 1) {
#pragma unroll
                        for (int k = 0; k < D; k++) d1[k] = w0[k];
                        rows.load(rb + o.ib, d0);
                    } else {
                        rows.load(rb + o.ib, d0);
                        rows.load(rb + o.ib + 1, d1);
                    }
                }
#pragma unroll
                for (int k = 0; k < D; k++) {
                    const double dr = dxsh[(t * 3 + 0) * D + k], dw = dxsh[(t * 3 + 1) * D + k], dtr = dxsh[(t * 3 + 2) * D + k];
                    dg[k] = o.A * d0[k] + o.B * d1[k];
                    tv[k] = o.u * dr + o.v * ((xa * dr + (ze * dw + dtr)) - dg[k]);
                }
                xstore_row<D>(B.dpol + (((size_t)t * B.groups + x) * G + pt) * D, dg);
                const size_t ro = (size_t)t * G + pt;
                if (rc0) A.R.pol[ro] = o.g;
                if (rc1) A.R.ib[ro] = o.ib;
                if (rc2) A.R.A[ro] = o.A;
                if (rc3) A.R.B[ro] = o.B;
                if (rc4) A.R.u[ro] = o.u;
                if (rc5) A.R.v[ro] = o.v;
            }
            if (!syncw) xtile_store_n<SL, NSL>(myt, tv);
            XSTAMP(0, son, i, 2);
            XSTAMPV(0, son, i);
            xlds_barrier();
            XSTAMP(0, son, i, 3);
        }
        if (i < P) {
            const int tx = P - 1 - i;
            if (own) {
                double mx[NSL];
                xtile_mix_reg<SL, NSL>(tile + (size_t)lane * SL, pr, ne, mx);
                const double bE = mx[IV] * c.beta;
                const double ex = -1.0 / c.gamma;
                if (pow_domain_error(bE, ex)) set_err(A.err, ERR_DOMAIN, tx, e, a);
                const double cm = pow_crra(bE, ex);
                const double rho = xsh[4 * tx + 3];
                const double s1 = rho * ((cm - (xsh[4 * tx + 1] * ze + xsh[4 * tx + 2])) + xa);
                const double kc = c.diet ? diet_kc(c, s1, rho, 1.0 + xsh[4 * tx], xsh[4 * tx + 1] * ze + xsh[4 * tx + 2], xa) : rho * (c.beta * ex * (cm / bE));
                double ds[D];
#pragma unroll
                for (int k = 0; k < D; k++) {
                    const double dr1 = dxsh[(tx * 3 + 0) * D + k], dw1 = dxsh[(tx * 3 + 1) * D + k], dt1 = dxsh[(tx * 3 + 2) * D + k];
                    ds[k] = kc * mx[k] - rho * ((ze * dw1 + dt1) + s1 * dr1);
                }
                sS[(size_t)(i & 1) * hs + gx + pt] = s1;
                rows.store((size_t)(i & 1) * hs + gx + pt, ds);
                if (rc6) A.R.s[(size_t)tx * G + pt] = s1;
                if (rc7) A.R.kc[(size_t)tx * G + pt] = kc;
            }
            episode++;
            XSTAMP(0, son, i, 4);
            xbar_arrive(!syncw);
            XSTAMP(0, son, i, 5);
            xbar_wait(A.sy, x, cW, Sact, episode, sync_duty);
        }
    }
    XSTAMP1(0, son, 11);
}

// ================================ forward sweeps, source-stationary (round 4) ========================================
// ONE kernel for the three forward recurrences (ForwardIteration.jl:297-308 and its partials):
//   k_xfwd<0, true>  the Float64 distribution sweep: group 0 only, writes D_1..D_P, {w, ig D} and the aggregate
//   k_xfwd<D, false> D partials per group at a recorded primal
//   k_xfwd<D, true>  value AND D partials per group (the forward half of the Dual pass, hank_primal_jvp): every group carries
//                    D_t itself — one more slot of the same linear step, no pow, no search — and group 0 writes the record
// A member's period in the target-stationary form of rounds 2-3 was a gather loop per target row: 2-8 sources on one row,
// three at a time, each trip a dependent L2 round trip, every source row fetched by both of its targets, and the workgroup
// waited for its slowest wave. Here the SOURCES are walked: 64 source rows per instruction, each source's two lottery parts
// added into the LDS tile of its target rows (ds_add_f64). The walk is cut into WORK UNITS in advance (k_xunits_fwd, once per
// recorded lottery): a unit is a run of <= 64 consecutive sources of one column that feeds a run of consecutive target rows
// of the member; a source on the seam between two units is walked by both, each adding only the part that lands in its own
// target run — so every tile entry is written by exactly one unit, by one wave, in program order: the sum is reproducible
// bit for bit whichever wave a unit is dealt to. Units are dealt round-robin to the member's waves, two per wave in straight-
// line code (both in flight together; the rare member with more units loops): the wave of a column whose policy is flat — up
// to 216 sources on 63 targets at 2000x11 — no longer holds the whole group back (stamps, DESIGN.md section 4: 3.2 us behind).
// The mass kept on the members' virtual rows travels as extra lanes of the unit that walks source row 0 of an open column
// (same lottery record, the members' virtual rows as state rows): no special sums. The mass point itself needs no load at
// all: a member's clamped rows and its virtual row are its own rows of the previous period — its own registers.
constexpr int XUCAP = 64;       // work units per member and period (k_xunits_fwd reports an overflow; the host then uses the launches)
struct XSweepFwdArgs {
    Consts c;
    Record R;                   // pol, lo, lw, ig (k_lottery); !VAL: lwg, Dseq of the recorded primal; VAL: group 0 writes lwg, Dseq
    XSync *sy;
    double *st;                 // [2][XG][n_e*members*64][SP] the ping-pong state
    const double *D0;           // VAL: [G] initial distribution (ForwardIteration.jl:293)
    const double *dpol;         // D > 0: [P][groups][G][D]
    int groups;
    double *Dvirt;              // VAL: [P][n_e][64] the mass kept on the virtual rows (k_xfix_D)
    double *aggpart;            // VAL: [P][members][2]
    double *daggpart;           // D > 0: [P][members][2*XG*D]
    const int *src;             // [P][members] lo | hi << 8 | (some column clamped) << 16 | (member 0 records row 0's full mass) << 17 | units << 18
    const int *overflow;        // k_xunits_fwd's flag: a member's walk did not fit XUCAP units — nothing is computed from a truncated list
    const int2 *units;          // [P][members][XUCAP] {e | ja << 4 | cnt << 16, ta | tb << 8 | nv << 16}
    int all_members;            // dev knob: every period waits for every member
};

template <int D, bool VAL, int MAXT>
__global__ void __launch_bounds__(MAXT) k_xfwd(XSweepFwdArgs A) {
#pragma clang fp contract(fast)                         // (round 5) fused multiply-adds in this kernel (the mixing: xtile_mix<.., true>)
    constexpr int NSL = D + (VAL ? 1 : 0);              // live slots: the D partials, then the value
    constexpr int SP = XSlots<NSL>::SP, SL = XSlots<NSL>::SL;
    constexpr int IV = D;                               // the value's slot
    constexpr int DD = D > 0 ? D : 1;
    constexpr bool PIREG = NSL < 4;                     // the mixing's coefficients in registers
    extern __shared__ __attribute__((aligned(16))) double xl[];
    const Consts &c = A.c;
    const Record &R = A.R;
    const int ne = c.n_e, na = c.n_a, P = c.P, G = c.G;
    double *tile = xl;                                  // [ne][64][SL]
    double *Pish = tile + (size_t)SL * ne * 64;         // [ne*ne] (read when !PIREG)
    // (round 5) the aggregate's terms of a period, per lane: the column waves leave them here and the SYNC wave — idle between its
    // polls — sums them over the columns and the lanes one period later. As eleven waves' own reductions they were 105 of the 562
    // VALU instructions a wave issues per period, three waves deep on a SIMD.
    // Two aggregates per slot (see dist_step_body): the policy-weighted one and the wealth-grid-weighted one. ONE buffer is enough:
    // the sync wave reads it between the period's first barrier and its own all-member poll, which stands in front of the barrier
    // the column waves must pass before they write the next period's terms.
    constexpr int NAP = 2 * NSL;                        // terms per lane: NSL of the first aggregate, NSL of the second
    double *aggsh = Pish + ((ne * ne + 1) & ~1);        // [ne][64][NAP]
    int *closh = reinterpret_cast<int *>(aggsh + (size_t)ne * 64 * NAP);           // [P][ne]: the clamped-prefix lengths (a cold uniform load per period otherwise)
    int *srcsh = closh + (size_t)P * ne;                // [P]
    int *ctl = srcsh + P;
    if (*A.overflow) return;                            // (every workgroup of the launch reads the same word: written by a kernel that ran before it)
    const XGroup g = xgroup_join(A.sy, ctl);
    if (!g.ok) return;
    const int x = g.x, cW = g.c;
    if (x >= A.groups) return;
    const int Sact = (na + XRW - 1) / XRW;
    if (g.S < Sact) { if (threadIdx.x == 0) xfail(A.sy, XERR_PLACEMENT, x); return; }
    if (cW >= Sact) return;
    // every group carries the same D_t: group 0 writes D_t, the virtual rows' mass and the aggregate, group 1 (where there is one)
    // the per-source record {w, ig D_{t-1}} — whose row 0 needs every member's virtual row, i.e. member 0 waits for everybody
    // (spreading the three outputs of group 0 further — one each to groups 0, 2, 3 — was slower: 2.73 against 2.63 ms at N = 32)
    const bool recD = VAL && x == 0, recL = VAL && x == 1 % A.groups;
    for (int k = threadIdx.x; k < ne * ne; k += blockDim.x) Pish[k] = c.Pi[k];
    for (int k = threadIdx.x; k < P * ne; k += blockDim.x) closh[k] = R.clo[k];
    for (int k = threadIdx.x; k < P; k += blockDim.x) {
        int w = A.src[(size_t)k * Sact + cW];
        if (A.all_members || (recL && ((w >> 17) & 1))) w = (w & ~0xffff) | ((Sact - 1) << 8);
        srcsh[k] = w;
    }
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the wave index in a scalar register: column bases become scalar)
    const bool syncw = wv >= ne;                        // see k_xtan_back
    const bool sync_duty = blockDim.x > 64 * ne ? syncw : wv == 0;
    const int e = syncw ? 0 : wv;
    const int r0 = cW * XRW, r = r0 + lane;
    const bool own = !syncw && lane < XRW && r < na;
    const bool virt = !syncw && lane == 63;             // this wave's virtual row: part of the mass point (row 0) kept by this member
    const bool live = own || virt;
    const size_t pt = (size_t)e * na + (own ? r : 0);
    const double xar = c.a[own ? r : 0];                // this row's grid point (a virtual row is a part of row 0)
    // state layout [e][member][64][SP]: a member's 63 rows and its virtual row (slot 63) are ONE line-aligned block
    const size_t GM = (size_t)ne * Sact * 64;
    const size_t gx = (size_t)x * GM;
    const size_t slot = gx + ((size_t)e * Sact + cW) * 64 + lane;
    const size_t hs = (size_t)XG * GM;
    double pr[16];                                      // Pi[k, e] as this wave's mixing uses it
#pragma unroll
    for (int k = 0; k < 16; k++) pr[k] = (PIREG && k < ne) ? c.Pi[ne * e + k] : 0.0;
    XRows<SP> rows;
    rows.init(A.st, 2 * hs);
    double *const myt = tile + ((size_t)e * 64 + lane) * SL;
    double mxp[NSL];                                    // this lane's row of the previous period (the state it stored)
#pragma unroll
    for (int k = 0; k < NSL; k++) mxp[k] = 0.0;
    if constexpr (VAL) { if (own) mxp[IV] = A.D0[pt]; }
    {
        double sv[SP];
#pragma unroll
        for (int k = 0; k < SP; k++) sv[k] = k < NSL ? mxp[k] : 0.0;
        if (live) rows.store(slot, sv);                 // D_0; it carries no partials
    }
    // this lane's own-row record of the period about to be processed (what the aggregate and the record need). At a recorded
    // primal the term dpol_t D_t of the aggregate is taken where the policy partials are loaded anyway — at the SOURCE rows —
    // and a row's own partials only count where the row is clamped (not walked as a source: its partial is zero except on a
    // knot tie). Branch-free on purpose: these loads are part of the counted batch behind the period's stores (see XFWD_NLD).
    double polr = 0.0, Dr = 0.0, lwr = 0.0, igr = 0.0, dpr[DD];
#pragma unroll
    for (int k = 0; k < DD; k++) dpr[k] = 0.0;
    // dev timing build (wrong numbers): HANK_X_TIMING_L2 confines what a period READS of the record and of the policy partials to
    // two periods — every such load an L2 / Infinity-Cache hit: what the HBM latency of the sweep's input streams costs
#ifdef HANK_X_TIMING_L2
#define XTPER(t) ((t) & 1)
#else
#define XTPER(t) (t)
#endif
    auto prefetch = [&](int t_) {
        const int t = XTPER(t_);
        const size_t ro = (size_t)t * G + (size_t)e * na + (own ? r : 0);       // (a virtual row carries row 0's policy and partials)
        polr = R.pol[ro];
        if constexpr (D > 0) xload_row_plain<DD>(A.dpol + ((size_t)t * A.groups + x) * (size_t)G * D + ((size_t)e * na + (own ? r : 0)) * D, dpr);
        if constexpr (VAL) { lwr = R.lw[ro]; igr = R.ig[ro]; }
        else { if constexpr (D > 0) Dr = R.Dseq[ro + G]; }      // D_t[r] (row 0 includes what the primal kept on its virtual rows)
    };
    // ---- the two work units of this wave, register sets 0 and 1: per lane a source's lottery record {lo, w, ig D_{t-1} | ig},
    // its policy partials and (at a recorded primal) D_t of its row; then its state row
    int2 ud[2];                                         // the units' descriptors (wave-uniform)
    int qlo[2];
    double qw[2], qg[2], qdn[2], qdp[2][DD], dd[2][SP];
    bool qon[2], qvl[2];
    const int2 *const ubase = A.units + (size_t)cW * XUCAP;
    auto unit_desc = [&](int t, int u) -> int2 {        // unit u of this member in period t (u wave-uniform): broadcast load -> scalar registers
        const int2 q = ubase[(size_t)t * Sact * XUCAP + u];
        return make_int2(__builtin_amdgcn_readfirstlane(q.x), __builtin_amdgcn_readfirstlane(q.y));
    };
    // (branch-free: every lane loads — a lane beyond the unit's sources the unit's last row, an empty unit row 0 of column 0 — and
    // qon / qvl say what counts)
    auto load_rec = [&](auto U, int t_, int2 d, int i0) {
        constexpr int u = decltype(U)::value;
        const int t = XTPER(t_);
        const int ue = d.x & 15, ja = (d.x >> 4) & 0xfff, cnt = (d.x >> 16) & 0xfff, nv = (d.y >> 16) & 0xff;
        const int i = i0 + lane;
        const bool real = i < cnt;
        qvl[u] = !real && i < cnt + nv;                 // a member's virtual row riding on source row 0's record
        qon[u] = real || qvl[u];
        const size_t cb = (size_t)t * G + (size_t)ue * na;
        const int j = real ? ja + i : (qvl[u] ? 0 : min(ja + max(cnt - 1, 0), na - 1));
        qlo[u] = R.lo[cb + j];
        if constexpr (VAL) { qw[u] = R.lw[cb + j]; if constexpr (D > 0) qg[u] = R.ig[cb + j]; }       // (ig only weights the policy partials)
        else { const double2 wg = R.lwg[cb + j]; qw[u] = wg.x; qg[u] = wg.y; if constexpr (D > 0) qdn[u] = R.Dseq[cb + G + j]; }
        if constexpr (D > 0) xload_row_plain<DD>(A.dpol + (((size_t)t * A.groups + x) * G + (size_t)ue * na + j) * D, qdp[u]);
    };
    auto load_state = [&](auto U, size_t hb, int2 d, int i0) {
        constexpr int u = decltype(U)::value;
        const int ue = d.x & 15, ja = (d.x >> 4) & 0xfff, cnt = (d.x >> 16) & 0xfff;
#pragma unroll
        for (int k = 0; k < SP; k++) dd[u][k] = 0.0;
        const int i = i0 + lane, j = ja + i;
        const size_t row = qvl[u] ? ((size_t)ue * Sact + (i - cnt)) * 64 + 63 : ((size_t)ue * Sact + j / XRW) * 64 + j % XRW;
        if (qon[u]) rows.load(hb + gx + row, dd[u]);
    };
    const std::integral_constant<int, 0> I0;
    const std::integral_constant<int, 1> I1;
    int2 udn[2];                                        // the descriptors of the period after next, on their way (loaded a period ahead: cold lines)
    bool udv[2] = {false, false};                       // ... and whether this wave has such a unit at all
    udn[0] = udn[1] = make_int2(0, 0);
    auto request_units = [&](int t) {                   // this wave's two units of period t (branch-free: an index that exists is loaded anyway)
        const int tc = min(t, P - 1);
        int nu = (srcsh[tc] >> 18) & 0xff;
        nu = t < P ? nu : 0;
        udv[0] = wv < nu; udv[1] = wv + ne < nu;
        udn[0] = ubase[(size_t)tc * Sact * XUCAP + (udv[0] ? wv : 0)];
        udn[1] = ubase[(size_t)tc * Sact * XUCAP + (udv[1] ? wv + ne : 0)];
    };
    auto take_units = [&]() {
        ud[0] = udv[0] ? make_int2(__builtin_amdgcn_readfirstlane(udn[0].x), __builtin_amdgcn_readfirstlane(udn[0].y)) : make_int2(0, 0);
        ud[1] = udv[1] ? make_int2(__builtin_amdgcn_readfirstlane(udn[1].x), __builtin_amdgcn_readfirstlane(udn[1].y)) : make_int2(0, 0);
    };
    // The loads of the NEXT period — its two units' records, the own-row record, the descriptors of the period after — are cold
    // lines of the record (2.4 us from HBM under this load) and vector-memory waits are in order. Round 5 splits them in two:
    //  * the UNIT batch (the two units' records, the descriptors of the period after) is issued BEFORE the mixing, as soon as the
    //    tile is complete: the stamps of round 4's single batch showed where its time went — not in latency but in ISSUE: twelve
    //    waves x 17 loads through one CU's vector-memory path took 2.4 us, every member's flag waited for the slowest wave's last
    //    load to be accepted, and the mixing (2 us of LDS and arithmetic that needs no memory path) sat in front of it, idle on
    //    that path. Issued here they are accepted while the waves mix, and have landed when the period's stores are drained;
    //  * the OWN-ROW batch (policy, policy partials, lottery weights of this lane's row: their registers are in use until the
    //    aggregate) stays behind the period's last store as a batch of exactly XFWD_NLD instructions, every one unconditional, and
    //    the drain waits with vmcnt(XFWD_NLD): everything older — the stores, and the unit batch — has completed, the own-row batch
    //    stays in flight through the publish, the next poll and the next state loads (tests/test_isa_hazards.py counts the
    //    instructions between the two markers in the ISA against the immediate).
    constexpr int DPI = D == 0 ? 0 : (D <= 2 ? 1 : D / 2);                 // load instructions per row of policy partials
    constexpr int XFWD_NLD = 1 + DPI + (VAL ? 2 : (D > 0 ? 1 : 0));
    auto next_units_loads = [&](int t) {                // t: the period being processed (ud holds the NEXT period's units by now)
        const int t1 = min(t + 1, P - 1);
        load_rec(I0, t1, ud[0], 0);
        load_rec(I1, t1, ud[1], 0);
        request_units(t + 2);
    };
    auto next_period_loads = [&](int t) {               // t: the period that has just been stored
        prefetch(min(t + 1, P - 1));
    };
    xbar_arrive(!syncw);                                // the initial state has reached L2: episode 1
    if (sync_duty) xpublish(A.sy, x, cW, 1u);
    ud[0] = ud[1] = make_int2(0, 0);
    if (!syncw) { request_units(0); take_units(); load_rec(I0, 0, ud[0], 0); load_rec(I1, 0, ud[1], 0); prefetch(0); request_units(1); }
    int cur = 0;
    bool vnz = false;                                   // the virtual rows may hold mass: some column was clamped last period
    const int son = (x == 0 && cW < 32) ? cW : -1;     // dev stamps (make stamp)
    (void)son;
    // the aggregate of period tp from the lanes' terms (sync wave; fixed order: columns ascending, then the lanes' DPP tree)
    auto reduce_agg = [&](int tp) {
        const double *ap = aggsh + (size_t)lane * NAP;
        double sk[NAP];
#pragma unroll
        for (int k = 0; k < NAP; k++) sk[k] = 0.0;
        for (int k2 = 0; k2 < ne; k2++) {
            double v[NAP];
            if (NAP == 1) v[0] = ap[(size_t)k2 * 64 * NAP];
            else {
#pragma unroll
                for (int q = 0; q < NAP / 2; q++) { const double2 d = reinterpret_cast<const double2 *>(ap + (size_t)k2 * 64 * NAP)[q]; v[2 * q] = d.x; v[2 * q + 1] = d.y; }
            }
#pragma unroll
            for (int k = 0; k < NAP; k++) sk[k] += v[k];
        }
        const size_t pb = (size_t)tp * Sact + cW;
#pragma unroll
        for (int k = 0; k < D; k++) {       // daggpart: [P][members][2 W], W = XG * D: the first aggregate's partials, then the second's
            const double pd = xwave_reduce63(sk[k]), pd2 = xwave_reduce63(sk[NSL + k]);
            if (lane == 63) { A.daggpart[pb * (size_t)(2 * XG * D) + x * D + k] = pd; A.daggpart[pb * (size_t)(2 * XG * D) + XG * D + x * D + k] = pd2; }
        }
        if constexpr (VAL) {
            if (recD) {
                const double pD = xwave_reduce63(sk[IV]), pD2 = xwave_reduce63(sk[NSL + IV]);
                if (lane == 63) { A.aggpart[2 * pb] = pD; A.aggpart[2 * pb + 1] = pD2; }
            }
        }
    };
    for (int t = 0; t < P; t++) {
        XSTAMP(1, son, t, 0);
        const size_t base = (size_t)t * G + (size_t)e * na;
        const size_t hb = (size_t)cur * hs;
        const int sw = srcsh[t];
        int clo = 0;
        if (!syncw) {
            clo = min(max(closh[t * ne + e], 0), na);
            double z[SL];
#pragma unroll
            for (int k = 0; k < SL; k++) z[k] = 0.0;
            xtile_store<SL>(myt, z);                    // (every wave is past the previous period's mixing: xbar_arrive)
        }
        if (sync_duty) xpoll(A.sy, x, sw & 255, (sw >> 8) & 255, (unsigned)(t + 1));   // this period's source members have published period t-1
        xlds_barrier();
        XSTAMP(1, son, t, 1);
        if (sync_duty && t > 0) reduce_agg(t - 1);      // (the sync wave has nothing else to do until the gathers are done)
        double pagg[DD];
#pragma unroll
        for (int k = 0; k < DD; k++) pagg[k] = 0.0;
        double v0 = 0.0;                                // the record's row 0: the mass on every member's virtual row of this column
        if (!syncw) {
            const bool need0 = recL && r0 == 0 && vnz && clo == 0;      // (wave-uniform)
            if constexpr (VAL) {
                if (need0 && lane < Sact) v0 = rows.load_one(hb + gx + ((size_t)e * Sact + lane) * 64 + 63, IV);
            }
            load_state(I0, hb, ud[0], 0);
            load_state(I1, hb, ud[1], 0);
            // Young lottery of a source (ForwardIteration.jl:59-73): (1-w) to row lo, w to row lo+1; the weight's partial is
            // dpol / gap (zero where clamped), times D_{t-1} of the row. Only the parts that land in the unit's own target run
            // [ta, tb) are added (the other part of a seam source belongs to the neighbouring unit).
#ifdef HANK_DEV_NOATOMIC      // timing experiment only (wrong numbers): what the same-address serialisation of the LDS adds costs
#define lds_add(p, v) (*(p) = (v))
#endif
            // PRE-COMBINE (round 5). The sources of a unit are consecutive lanes and their brackets rise with them: where the policy moves
            // one grid row per source row — most of the grid — the upper part of lane i-1 and the lower part of lane i land on the SAME
            // tile row. Lane i then adds both in one ds_add_f64 (the neighbour's part arrives by a wave_shr:1 DPP move) and lane i-1
            // skips its own: half the LDS atomics of the walk (20 per wave and period at D = 4 with the value; a timing build without
            // them put their cost at 0.33 ms of the sweep). Still one unit = one wave in program order: reproducible bit for bit; where a
            // row receives exactly these two terms the sum is the same (0 + L) + H = 0 + (L + H).
            auto process_pc = [&](auto U, int2 d) {
                constexpr int u = decltype(U)::value;
                const int ta = d.y & 0xff, tb = (d.y >> 8) & 0xff;
                double *const colt = tile + (size_t)(d.x & 15) * 64 * SL;
                const double w = qw[u], w1 = 1.0 - w;
                double gD = qg[u], Dv = 0.0;
                if constexpr (VAL) { Dv = dd[u][IV]; gD = qg[u] * Dv; }
                else { if (qvl[u]) gD = 0.0; }          // (a virtual row's mass is part of the recorded ig * D of row 0)
                const int tl = qlo[u] - r0, th = tl + 1;
                const bool doL = qon[u] && tl >= ta && tl < tb, doH = qon[u] && th >= ta && th < tb;
                const int thp = __builtin_amdgcn_update_dpp(-2, doH ? th : -2, 0x138, 0xf, 0xf, false);        // lane i-1's upper target (wave_shr:1; lane 0: none)
                const bool take = doL && thp == tl;
                const bool given = __builtin_amdgcn_update_dpp(0, take ? 1 : 0, 0x130, 0xf, 0xf, false) != 0;   // lane i+1 adds my upper part with its lower one (wave_shl:1)
                double *const tpl = colt + (size_t)(doL ? tl : 0) * SL, *const tph = colt + (size_t)(doH ? th : 0) * SL;
#pragma unroll
                for (int k = 0; k < D; k++) {
                    const double hk = w * dd[u][k] + gD * qdp[u][k];
                    const double hp = xdpp<0x138, 0xf, 0xf>(hk);
                    const double lk = w1 * dd[u][k] - gD * qdp[u][k];
                    if (doL) lds_add(tpl + k, take ? lk + hp : lk);
                    if (doH && !given) lds_add(tph + k, hk);
                }
                if constexpr (VAL) {                    // (the VALUE keeps one add per part: D_t — the record — is the same bits whichever kernel carries it)
                    if (doL) lds_add(tpl + IV, w1 * Dv);
                    if (doH) lds_add(tph + IV, w * Dv);
                }
                if constexpr (!VAL) {
                    if (doL && !qvl[u]) {               // (every unclamped source has exactly one lower target: its aggregate term is taken here, once)
#pragma unroll
                        for (int k = 0; k < D; k++) pagg[k] += qdp[u][k] * qdn[u];
                    }
                }
            };
            auto process_1 = [&](auto U, int2 d) {
                constexpr int u = decltype(U)::value;
                if (!qon[u]) return;
                const int ta = d.y & 0xff, tb = (d.y >> 8) & 0xff;
                double *const colt = tile + (size_t)(d.x & 15) * 64 * SL;
                const double w = qw[u], w1 = 1.0 - w;
                double gD = qg[u], Dv = 0.0;
                if constexpr (VAL) { Dv = dd[u][IV]; gD = qg[u] * Dv; }
                else { if (qvl[u]) gD = 0.0; }          // (a virtual row's mass is part of the recorded ig * D of row 0)
                const int tl = qlo[u] - r0, th = tl + 1;
                if (tl >= ta && tl < tb) {              // (every unclamped source has exactly one lower target: its aggregate term is taken here, once)
                    double *tp = colt + (size_t)tl * SL;
#pragma unroll
                    for (int k = 0; k < D; k++) lds_add(tp + k, w1 * dd[u][k] - gD * qdp[u][k]);
                    if constexpr (VAL) lds_add(tp + IV, w1 * Dv);
                    if constexpr (!VAL) {
                        if (!qvl[u]) {
#pragma unroll
                            for (int k = 0; k < D; k++) pagg[k] += qdp[u][k] * qdn[u];
                        }
                    }
                }
                if (th >= ta && th < tb) {
                    double *tp = colt + (size_t)th * SL;
#pragma unroll
                    for (int k = 0; k < D; k++) lds_add(tp + k, w * dd[u][k] + gD * qdp[u][k]);
                    if constexpr (VAL) lds_add(tp + IV, w * Dv);
                }
            };
            // (narrow kernels — one or two slots per tile row — keep one add per part: at D = 1 the exchange costs more than the
            // one atomic it saves, single-tangent JVP 1.98 -> 2.06 ms)
            constexpr bool PRECOMB = HANK_XFWD_PRECOMBINE && (D + (VAL ? 1 : 0)) >= 3;
            auto process = [&](auto U, int2 d) { if constexpr (PRECOMB) process_pc(U, d); else process_1(U, d); };
            process(I0, ud[0]);
            process(I1, ud[1]);
#ifdef HANK_DEV_NOATOMIC
#undef lds_add
#endif
            // (rare) a unit wider than a wave — more than 64 sources on ONE target row — and units beyond the member's 2 n_e slots
            const int nu = (sw >> 18) & 0xff;
            for (int u2 = wv; u2 < nu; u2 += ne) {
                const int2 d = u2 < 2 * ne ? ud[u2 >= ne ? 1 : 0] : unit_desc(t, u2);
                const int tot = ((d.x >> 16) & 0xfff) + ((d.y >> 16) & 0xff);
                for (int i0 = u2 < 2 * ne ? 64 : 0; i0 < tot; i0 += 64) {
                    load_rec(I0, t, d, i0);
                    load_state(I0, hb, d, i0);
                    process(I0, d);
                }
            }
            if constexpr (VAL) { if (need0) v0 = xwave_sum(v0); }
            take_units();                               // the next period's descriptors (requested a period ago; this wave's loads have all landed here)
            XSTAMP(1, son, t, 2);
            // the mass point: sources clamped at the first grid point (:54-58) go to row 0 with weight one and no weight
            // partial. Each member sums ITS clamped rows into its virtual row (never combined: everything downstream is
            // linear); while row 0 itself is clamped the old virtual row is carried along.
            double cT[NSL];
#pragma unroll
            for (int k = 0; k < NSL; k++) cT[k] = ((own && r < clo) || (virt && clo > 0 && vnz)) ? mxp[k] : 0.0;
            if (clo > r0) {
#pragma unroll
                for (int k = 0; k < NSL; k++) cT[k] = xwave_reduce63(cT[k]);
            }
            if (virt) xtile_store_n<SL, NSL>(myt, cT);
        }
        XSTAMP(1, son, t, 3);
        XSTAMPV(1, son, t);
        if (sync_duty) xpoll(A.sy, x, 0, Sact - 1, (unsigned)(t + 1));      // EVERY member is done reading the half about to be overwritten
        xlds_barrier();
        XSTAMP(1, son, t, 4);
        vnz = ((sw >> 16) & 1) != 0;
        const int nxt = cur ^ 1;
        if (!syncw) {
            next_units_loads(t);                        // accepted by the memory path while the waves mix (see above)
            // exogenous transition: D_t[., e] = sum_k D_mid[., k] Pi[k, e] (ForwardIteration.jl:95-99), value and partials in one pass
            double mx[NSL];
            if (PIREG) xtile_mix_reg<SL, NSL, true>(tile + (size_t)lane * SL, pr, ne, mx);
            else xtile_mix<SL, NSL, true>(tile + (size_t)lane * SL, Pish + ne * e, 1, ne, mx);
            if (live) {
                double sv[SP];
#pragma unroll
                for (int k = 0; k < SP; k++) sv[k] = k < NSL ? mx[k] : 0.0;
                rows.store((size_t)nxt * hs + slot, sv);
            }
            double at[NAP];
#pragma unroll
            for (int k = 0; k < NAP; k++) at[k] = 0.0;
            if constexpr (VAL) {
                if (recL && own) {
                    // what the tangent sweeps read per SOURCE: {w, ig * D_{t-1}} (k_lottery's w and ig)
                    double Dfull = mxp[IV];
                    if (r == 0 && clo == 0) Dfull += v0;
                    R.lwg[base + r] = make_double2(lwr, igr * Dfull);
                }
                if (recD) {
                    if (own) R.Dseq[(size_t)(t + 1) * G + pt] = mx[IV];
                    else if (virt) A.Dvirt[((size_t)t * ne + e) * 64 + cW] = mx[IV];
                    // aggregate on the POST-transition distribution (ForwardIteration.jl:301-307); a virtual row carries row 0's policy
                    at[IV] = live ? polr * mx[IV] : 0.0;
                    at[NSL + IV] = live ? xar * mx[IV] : 0.0;
                }
            }
            // aggregate partials: pol_t dD_t + dpol_t D_t. Value carried: a row's own policy partials meet its new D_t here (a
            // virtual row carries row 0's policy and row 0's partials). At a recorded primal: the second term was taken at the
            // source rows (pagg), clamped rows add theirs here, and a virtual row's share of D_t[0] is in the recorded D_t[0].
#pragma unroll
            for (int k = 0; k < D; k++) {
                double term = live ? polr * mx[k] : 0.0;
                if constexpr (VAL) term = live ? (polr * mx[k] + dpr[k] * mx[IV]) : 0.0;
                else term = (term + pagg[k]) + ((own && r < clo) ? dpr[k] * Dr : 0.0);
                at[k] = term;
                at[NSL + k] = live ? xar * mx[k] : 0.0;      // the grid-weighted aggregate's partial: sum a dD_t
            }
            xtile_store<NAP>(aggsh + ((size_t)e * 64 + lane) * NAP, at);      // summed by the sync wave in the next period (reduce_agg)
#pragma unroll
            for (int k = 0; k < NSL; k++) mxp[k] = mx[k];
        }
        cur = nxt;
        XSTAMP(1, son, t, 5);
        if (!syncw) {
            asm volatile("; XFWD_BATCH_BEGIN" ::: "memory");
            next_period_loads(t);
            XSTAMP(1, son, t, 7);
            asm volatile("; XFWD_BATCH_END\n\ts_waitcnt vmcnt(%0)" ::"n"(XFWD_NLD) : "memory");       // this member's stores have reached L2 ...
            XSTAMP(1, son, t, 8);
        }
        xlds_barrier();
        if (sync_duty) xpublish(A.sy, x, cW, (unsigned)(t + 2));                                       // ... episode t+2
        XSTAMP(1, son, t, 6);
    }
    if (sync_duty) reduce_agg(P - 1);                   // (every wave wrote its terms before the last barrier)
}

// the forward sweeps' work units from the lottery record (see k_xfwd), one block per (period, member): thread e cuts column
// e's walk over the sources of the member's target rows into units — a run [ta, tb) of target rows (relative to the member's
// first row) and the <= 64 consecutive sources [ja, ja + cnt) whose brackets touch it; the unit that walks source row 0 of an
// open column while the virtual rows hold mass gets `nv` = members extra lanes (see k_xfwd). Units are numbered column by
// column; unit u goes to wave u mod n_e. Per (period, member) also: the members its sources belong to (lo | hi << 8), whether
// some column is clamped in the period (<< 16: the virtual rows hold mass in the next one), whether member 0 needs every
// member's virtual row for the record of row 0 (<< 17), and the unit count (<< 18).
__global__ void k_xunits_fwd(Consts c, Record R, int Sact, int *src, int2 *units, int *overflow, int ucap) {
    __shared__ int2 su[16][XUCAP];
    __shared__ int scnt[16], slo, shi, sany0;
    __shared__ int sst[16][XRW + 3];                    // start[r0-1 .. r0+nrows] of every column: the walk below is serial per column,
    const int t = blockIdx.x, m = blockIdx.y, ne = c.n_e, na = c.n_a;      // and as dependent global loads it was 86 us per record
    const int r0 = m * XRW, nrows = min(XRW, na - r0);
    __shared__ int sclo[16], sclop[16], soff[17];       // clamped prefixes of this period and of the one before, unit offsets per column
    if (threadIdx.x == 0) { slo = m; shi = m; sany0 = 0; }
    if ((int)threadIdx.x < ne) {
        sclo[threadIdx.x] = R.clo[(size_t)t * ne + threadIdx.x];
        sclop[threadIdx.x] = t > 0 ? R.clo[(size_t)(t - 1) * ne + threadIdx.x] : 0;
    }
    for (int k = threadIdx.x; k < ne * (XRW + 3); k += blockDim.x) {
        const int e = k / (XRW + 3), q = k - e * (XRW + 3);
        sst[e][q] = min(max(R.start[((size_t)t * ne + e) * (na + 1) + min(max(r0 - 1 + q, 0), na)], 0), na);
    }
    __syncthreads();
    bool vnz = false;
    for (int e = 0; e < ne; e++) vnz = vnz || sclop[e] > 0;
    // One wave per column (the block's waves take the columns in turn). The greedy walk "extend the unit while it has <= 64 lanes"
    // is 63 dependent steps as a loop; the end of the unit that STARTS at target row ta is a monotone search, so lane ta finds it
    // by bisection (nxt), all 63 at once, and lane 0 then follows the chain nxt[0], nxt[nxt[0]], ... — a handful of hops.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int e = wave; e < ne; e += nwaves) {
        const int clo = sclo[e];
        auto S = [&](int rr) { return sst[e][min(max(rr - (r0 - 1), 0), XRW + 2)]; };     // (clamped at the fill: a record that is not a lottery must not turn into a long walk or an out-of-range row)
        // lanes of a unit that starts at `a` and ends before target row `b`: its sources, and every member's virtual row when it walks source 0
        auto lanes_of = [&](int a, int b, int *ja_out, bool *has0_out) {
            const int ja = S(r0 + a - 1 < 0 ? 0 : r0 + a - 1);
            const bool has0 = ja == 0 && clo <= 0 && vnz;        // (then source row 0 is the unit's first lane, if it has any)
            const int cnt = max(S(r0 + b) - ja, 0);
            *ja_out = ja; *has0_out = has0;
            return cnt;
        };
        int nxt = nrows;
        if (lane < nrows) {
            int ja; bool has0;
            (void)lanes_of(lane, lane + 1, &ja, &has0);
            auto fits = [&](int b) { const int c2 = max(S(r0 + b) - ja, 0); return c2 + ((has0 && c2 > 0) ? Sact : 0) <= 64; };
            int lo = lane + 1, hi = nrows;               // the unit ends at the largest b in [lane + 1, nrows] with every step up to it fitting
            if (lo < hi && fits(lo + 1)) {               // (the count is monotone in b: a bisection)
                if (fits(hi)) lo = hi;
                else {
                    lo = lo + 1;
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (fits(mid)) lo = mid; else hi = mid; }
                }
            }
            nxt = lo;
        }
        int n = 0, ta = 0;
        while (ta < nrows && n < XUCAP) {               // (wave-uniform: every lane follows the same chain)
            const int tb = __shfl(nxt, ta, 64);
            int ja; bool has0;
            const int cnt = lanes_of(ta, tb, &ja, &has0);
            const int nv = (has0 && cnt > 0) ? Sact : 0;
            if (cnt > 0) {
                if (lane == 0) {
                    su[e][n] = make_int2(e | (ja << 4) | (cnt << 16), ta | (tb << 8) | (nv << 16));
                    atomicMin(&slo, ja / XRW); atomicMax(&shi, (ja + cnt - 1) / XRW);
                    if (nv) sany0 = 1;
                }
                n++;
            }
            ta = tb;
        }
        if (lane == 0) {
            if (ta < nrows) atomicExch(overflow, 1);
            scnt[e] = n;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0;
        for (int e = 0; e < ne; e++) { soff[e] = off; off += scnt[e]; }
        soff[ne] = off;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < min(soff[ne], XUCAP); k += blockDim.x) {      // units numbered column by column
        int e = 0;
        while (e + 1 < ne && soff[e + 1] <= k) e++;
        units[((size_t)t * Sact + m) * XUCAP + k] = su[e][k - soff[e]];
    }
    if (threadIdx.x == 0) {
        int tot = soff[ne], anyclo = 0, anyopen = 0;
        for (int e = 0; e < ne; e++) {
            if (sclo[e] > 0) anyclo = 1; else anyopen = 1;
        }
        if (tot > ucap) { atomicExch(overflow, 1); tot = min(tot, XUCAP); }      // (ucap = XUCAP but for the dev knob HANK_XUCAP)
        int ml = min(slo, m), mh = max(shi, m);
        if (sany0) { ml = 0; mh = Sact - 1; }           // the units with virtual lanes read every member's virtual row
        const int allv = (m == 0 && vnz && anyopen) ? 1 : 0;
        src[(size_t)t * Sact + m] = ml | (mh << 8) | (anyclo << 16) | (allv << 17) | (tot << 18);
    }
}

// which members does member c gather from in period t? (lo | hi << 8), from the recorded primal; one block per (t, c)
__global__ void k_xsrc_back(Consts c, Record R, int Sact, int *src) {
    __shared__ int smin[256], smax[256];
    const int t = blockIdx.x, m = blockIdx.y;
    const int r0 = m * XRW, rows = min(XRW, c.n_a - r0);
    int lo = 1 << 30, hi = -1;
    for (int k = threadIdx.x; k < rows * c.n_e; k += blockDim.x) {
        const int e = k / rows, a = r0 + (k - e * rows);
        const size_t ro = (size_t)t * c.G + (size_t)e * c.n_a + a;
        if (R.A[ro] != 0.0 || R.B[ro] != 0.0) { const int ib = R.ib[ro]; lo = min(lo, ib); hi = max(hi, ib + 1); }
    }
    smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)blockDim.x; k++) { lo = min(lo, smin[k]); hi = max(hi, smax[k]); }
        int ml = m, mh = m;
        if (hi >= 0) { ml = min(m, max(lo, 0) / XRW); mh = max(m, min(hi, c.n_a - 1) / XRW); }
        src[(size_t)t * Sact + m] = ml | (mh << 8);
    }
}
// rho_t = 1/(1+r_t) for every period (the X half's discounting; same expression as egm_X)
__global__ void k_xrho(const double *xhh, int n_hh, int P, double *rho) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < P) rho[t] = 1.0 / (1.0 + xhh[n_hh * t]);
}

// row 0 of every column of D_1..D_P: add the virtual mass the forward sweep kept apart (member order fixed)
__global__ void k_xfix_D(Consts c, double *Dseq, const double *Dvirt, int Sact) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= c.P * c.n_e) return;
    const int t = idx / c.n_e, e = idx - t * c.n_e;
    double s = Dseq[(size_t)(t + 1) * c.G + (size_t)e * c.n_a];
    for (int m = 0; m < Sact; m++) s += Dvirt[(size_t)idx * 64 + m];
    Dseq[(size_t)(t + 1) * c.G + (size_t)e * c.n_a] = s;
}

// dagg of one pass [P][XG*D] -> columns [n0, n0+N) of the caller's (P, Ntot) column-major block
__global__ void k_xout(const double *__restrict__ dagg, int P, int W, int src0, int n0, int N, double *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * N) return;
    const int n = idx / P, t = idx - n * P;
    out[(size_t)(n0 + n) * P + t] = dagg[(size_t)t * W + src0 + n];
}

// ---- the persistent Dual pass's small work in TWO launches (round 5) ---------------------------------------------------------
// A one-pass Dual pass used to be wrapped in fifteen launches and copies of a few microseconds each (two input copies, two
// memsets of the sync blocks, k_zero_i32, k_xrho, k_tan_in in front; k_reduce_parts, k_tan_out, k_xfix_D, k_reduce_parts, two
// k_xout and two output copies behind): dependent launches on one stream cost ~3-5 us apiece, 0.07 ms of a 4.35 ms step.
// k_xdual_prologue: inputs in (the caller's device buffers are read where they lie; the context keeps its copies), rho_t, the
// per-direction input partials [P][N], the four sync blocks and the error word zeroed.
__global__ void k_xdual_prologue(const double *__restrict__ xhh_src, double *__restrict__ xhh_dst, const double *__restrict__ dx_src, double *__restrict__ dx_dst,
                                 int n_hh, int P, int N, double *__restrict__ rho, double *__restrict__ dxr, double *__restrict__ dxw, double *__restrict__ dxt,
                                 xv4u *__restrict__ sync, size_t sync_q, int *__restrict__ err) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < sync_q; i += step) { xv4u z; z.x = z.y = z.z = z.w = 0u; sync[i] = z; }
    if (i0 < 4) err[i0] = 0;
    for (size_t t = i0; t < (size_t)P; t += step) {
        rho[t] = 1.0 / (1.0 + xhh_src[(size_t)n_hh * t]);                   // (k_xrho's expression)
        if (xhh_dst != xhh_src)
            for (int k = 0; k < n_hh; k++) xhh_dst[(size_t)n_hh * t + k] = xhh_src[(size_t)n_hh * t + k];
    }
    for (size_t idx = i0; idx < (size_t)P * N; idx += step) {              // (k_tan_in)
        const size_t t = idx / N, n = idx - t * N;
        const size_t o = (size_t)n_hh * (t + (size_t)P * n);
        const double a = dx_src[o], b = dx_src[o + 1], c2 = n_hh > 2 ? dx_src[o + 2] : 0.0;
        dxr[idx] = a; dxw[idx] = b;
        if (n_hh > 2) dxt[idx] = c2;
        if (dx_dst != dx_src) { dx_dst[o] = a; dx_dst[o + 1] = b; if (n_hh > 2) dx_dst[o + 2] = c2; }
    }
}
// the sum k_reduce_parts takes over nb row blocks, in its order (four strided groups of eight, then the groups)
__device__ __forceinline__ double xreduce_parts_at(const double *__restrict__ parts, int nb, int N, int t, int n) {
    const double *p = parts + (size_t)t * nb * N + n;
    double sg[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        double s = 0.0;
        int b = g;
        for (; b + 28 < nb; b += 32) {
            const double v0 = p[(size_t)b * N], v1 = p[(size_t)(b + 4) * N], v2 = p[(size_t)(b + 8) * N], v3 = p[(size_t)(b + 12) * N];
            const double v4 = p[(size_t)(b + 16) * N], v5 = p[(size_t)(b + 20) * N], v6 = p[(size_t)(b + 24) * N], v7 = p[(size_t)(b + 28) * N];
            s += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
        }
        for (; b < nb; b += 4) s += p[(size_t)b * N];
        sg[g] = s;
    }
    return (sg[0] + sg[1]) + (sg[2] + sg[3]);
}
// k_xdual_epilogue: both aggregates of the value and of the pass's partials summed over the members (k_reduce_parts' order), laid
// out as the entry points return them (k_tan_out, k_xout) — into the context's buffers AND the caller's —, and row 0 of every D_t
// completed with the virtual rows' mass (k_xfix_D). One thread per (period, output).
__global__ void k_xdual_epilogue(Consts c, const double *__restrict__ aggpart, const double *__restrict__ daggpart, int Sact, int W, int n0, int N, int Ntot,
                                 double *__restrict__ agg_rm, double *__restrict__ agg_cm, double *__restrict__ dagg_pass, double *__restrict__ dagg_cm,
                                 double *__restrict__ Dseq, const double *__restrict__ Dvirt, double *__restrict__ out_agg, double *__restrict__ out_dagg) {
    const int P = c.P, per = 2 * W + 2 + c.n_e;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * per) return;
    const int t = idx / per, j = idx - t * per;
    if (j < 2 * W) {
        const double v = xreduce_parts_at(daggpart, Sact, 2 * W, t, j);
        dagg_pass[(size_t)t * 2 * W + j] = v;
        const int half = j >= W, n = j - half * W;
        if (n < N) {
            dagg_cm[(size_t)half * P * Ntot + (size_t)(n0 + n) * P + t] = v;
            if (!half && out_dagg) out_dagg[(size_t)(n0 + n) * P + t] = v;
        }
    } else if (j < 2 * W + 2) {
        const int k = j - 2 * W;
        const double v = xreduce_parts_at(aggpart, Sact, 2, t, k);
        agg_rm[(size_t)t * 2 + k] = v;
        agg_cm[(size_t)k * P + t] = v;
        if (k == 0 && out_agg) out_agg[t] = v;
    } else {
        const int e = j - 2 * W - 2;
        const size_t q = (size_t)t * c.n_e + e, o = (size_t)(t + 1) * c.G + (size_t)e * c.n_a;
        double sD = Dseq[o];
        for (int m = 0; m < Sact; m++) sD += Dvirt[q * 64 + m];
        Dseq[o] = sD;
    }
}

// (G,P,N) col-major export of one pass's dpol [P][groups][G][D] into columns [n0, n0+N)
__global__ void k_xexport_dpol(const double *dpol, int G, int P, int groups, int D, int n0, int N, double *out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)G * P * N;
    if (idx >= total) return;
    const size_t n = idx / ((size_t)G * P), rem = idx - n * (size_t)G * P, t = rem / G, pt = rem - t * G;
    const int x = (int)n / D, k = (int)n - x * D;
    out[((size_t)(n0 + n) * P + t) * G + pt] = dpol[(((size_t)t * groups + x) * G + pt) * D + k];
}

}  // namespace hank
