// hank_hip.hip — context, hipGraph-captured sweeps and the C ABI (include/hank_hip.h) of the
// MI355X household block. Both sweeps are strict recurrences in t (value_t needs value_{t+1},
// D_t needs D_{t-1}); every period is one kernel whose X half of period t-1 is fused behind the
// Y half of period t, so the loop-carried state only crosses a launch boundary once per period,
// and the 2(T-1) dependent launches are replayed from hipGraphs (no host launch cost).
#include "hank_kernels.h"
#include "hank_xsweep.h"
#include "hank_jacobian.h"
#include "hank_wide.h"
#include "../../include/hank_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <mutex>
#include <new>
#include <vector>

using namespace hank;

// Lane geometry of the tangent kernels, chosen per batch width N (measured, MI355X, 2000x11, T=300):
//  - lane width: an even batch runs TWO adjacent directions per lane (double2): every state / dpol access is
//    16 bytes, half the vector-memory instructions per byte (N=32: 4650 -> 5380 JVPs/s; N=256: 8130 -> 10940);
//  - wealth-row groups per wave: backward 2 (4 from 256 directions on) — its gathers are independent, more
//    groups = more bytes in flight per wave; forward 2 for 16-lane groups (N = 18..32, which run the
//    source-stationary form of the forward kernel, see ensure_tanwork) and from N = 64 on, 1 otherwise.
static inline int tan_rg(int NV, int forward, int ss = 0) {   // NV = lanes' worth of directions
    const char *e = getenv(forward ? "HANK_RG_F" : "HANK_RG_B");   // dev knobs
    if (e) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) return v; }
    return forward ? ((NV >= 32 || ss) ? 2 : 1) : (NV >= 128 ? 4 : 2);
}
static inline int tan_lane_width(int N, int forward) {
    const char *e = getenv(forward ? "HANK_LANE_WIDTH_F" : "HANK_LANE_WIDTH_B");   // dev knobs
    const int want = e ? atoi(e) : 2;
    return (want == 2 && N % 2 == 0) ? 2 : 1;
}

struct TanWork {
    int N = 0;
    TanGeom g{}, gf{};   // lane geometry of the backward / forward tangent kernels
    double *dxhh = nullptr;   // (2,P,N) staging for the host-pointer entry
    double *dxr = nullptr, *dxw = nullptr, *dxt = nullptr;   // [P][N] tangents of r, w (, lump-sum transfer)
    double *ds[2] = {nullptr, nullptr};
    double *dD[2] = {nullptr, nullptr};
    double *dpol = nullptr;
    double *aggpart = nullptr;
    double *dagg = nullptr;     // [P][N]
    double *dagg_cm = nullptr;  // (P,N) column-major
    int nbx = 0, nbxf = 0;
    hipGraphExec_t g_back = nullptr, g_fwd = nullptr;
    hipGraphExec_t g_fback = nullptr, g_ffwd = nullptr;   // dual-sweep graphs (primal + tangents in one chain)
    bool valid = false;  // dpol holds the partials of the current primal
    int VB = 1, VF = 1, RGB = 1, RGF = 1;   // lane widths and row groups the graphs are captured with
    unsigned nbf = 0;
};


// ---- XCD-local persistent sweeps (hank_xsweep.h): per-context workspace and per-batch-width tangent buffers ----
constexpr int XD_MAX = 4;           // directions per group and pass (D = 8 spills registers: wider batches run as passes)
constexpr int XPASS_MAX = 64;       // passes per call: N <= 8 * XD_MAX * XPASS_MAX = 2048 directions
struct XPass { int n0, N, D, groups; size_t dpol_off; };
struct XTan {                       // one per batch width N (kept in a small LRU: Jacobian assembly and Newton alternate widths)
    int N = 0;
    std::vector<XPass> passes;
    double *dxhh = nullptr, *dxr = nullptr, *dxw = nullptr, *dxt = nullptr;   // staging + [P][N] input tangents
    double *dpol = nullptr;         // per pass [P][groups][G][D]
    double *daggpart = nullptr;     // [P][Sact*n_e][XG*XD_MAX] (reused by every pass)
    double *dagg_pass = nullptr;    // [P][XG*XD_MAX]
    double *dagg_cm = nullptr;      // (P, N) column-major
    bool valid = false;             // dpol holds the partials of the current primal
};
struct XWork {
    bool ready = false;
    int grid = 0, Sact = 0, maxt = 768, dmax = XD_MAX;
    XSync *sync = nullptr;          // [2 * XPASS_MAX]: backward and forward sweep of every pass
    double *st_s = nullptr, *st_ds = nullptr, *st_D = nullptr, *st_dD = nullptr;
    double *Dvirt = nullptr, *aggpart = nullptr, *rho = nullptr;
    int *srcB = nullptr, *srcF = nullptr;     // [P][Sact] source-member ranges of the tangent sweeps at the recorded primal
    int2 *unitsF = nullptr;                   // [P][Sact][XUCAP] the forward sweeps' work units (k_xunits_fwd)
    int *unit_overflow = nullptr;             // set by k_xunits_fwd when a member has more units than XUCAP
    int ucap = 64;                            // dev knob HANK_XUCAP (read once, at hank_create): a smaller budget, to exercise the overflow path
    bool rng_valid = false;                   // srcF / unitsF belong to the recorded lottery
    bool neigh = true;                        // dev knob HANK_XNEIGH=0 (read once, at hank_create): every period waits for every member
    bool syncwave = true;                     // dev knob HANK_XSYNCWAVE=0 (read once, at hank_create): wave 0 polls instead of an extra wave
    int lds_max = 65536;
    int fault_where = 7;
    int fault = 0;                            // dev knob HANK_XFAULT=placement: every persistent launch finds its status word set ("a
                                              // group is short of members") and leaves at once — exercises the fallback paths
    bool src_valid = false;
    std::list<XTan> tans;           // most recently used first
    int last_passes = 0;            // sync blocks the last call used (their status words are checked)
};

// ---- on-chip wide sweeps (hank_wide.h): tangent buffers per batch width ----
struct WTan {
    int N = 0;
    double *dxhh = nullptr;         // (n_hh, P, N) the caller's input tangents (staging for the host-pointer entries)
    double *dpol = nullptr;         // [P][N][G]
    double *dagg_cm = nullptr;      // (P, N) column-major
    bool valid = false;             // dpol holds the partials of the current primal
};

struct hank_ctx {
    int device = 0;
    Consts c{};
    Record R{};
    int T = 0;
    double *d_a = nullptr, *d_z = nullptr, *d_Pi = nullptr;
    double *d_ss_value = nullptr, *d_ss_D = nullptr;  // d_ss_D aliases Dseq[0]
    double *d_xhh = nullptr, *d_agg = nullptr, *d_aggpart = nullptr;     // d_agg: (P, 2) column-major — the policy-weighted aggregate, then the grid-weighted one
    double *d_agg_rm = nullptr;     // [P][2] as the reduction leaves it
    double *d_zd = nullptr;         // [2][P]: sum_e z_e m_t(e) and sum_e m_t(e), m_t = Pi' m_{t-1} the productivity marginal of D_t (hank_get_het_outputs)
    int *d_err = nullptr;
    int nbp = 0;  // row blocks of the primal kernels
    bool boundary_set = false, primal_done = false;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // the primal forward sweep runs on a side stream, concurrently with the tangent backward sweep (both only
    // need the primal backward record); ev_side marks its completion, side_pending = the main stream has not
    // been made to wait for it yet
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_side = nullptr;
    bool side_pending = false;
    hipGraphExec_t g_pback = nullptr, g_pfwd = nullptr;
    hipEvent_t ev[16] = {};
    bool ev_valid[6] = {false, false, false, false, false, false};
    int launches[6] = {0, 0, 0, 0, 0, 0};
    std::list<TanWork> tws;        // per batch width, most recently used first (a small cache: Jacobian assembly and Newton alternate widths)
    TanWork *tw = nullptr;         // the current one
    // 0 = one launch per period for everything; 1 = XCD-local persistent sweeps for everything; 2 = auto (default where
    // the persistent sweeps are supported): each entry point takes the faster of the two for its shape — see sched_*
    int schedule = 2;
    bool forced_xcd = false;       // HANK_SCHEDULE=xcd at hank_create: no silent fallback to the launches
    int last_tan = 0;              // which implementation ran the last tangent sweep (0 launches, 1 persistent): hank_get_dpolicy_seq
    int xjvp_max = 64;             // auto: batches up to this width take the persistent tangent sweeps (measured crossover)
    XWork xw;
    struct { double *dpT = nullptr, *iota = nullptr, *E = nullptr, *Cp = nullptr, *F = nullptr, *Dv = nullptr; int N = 0; } fn;   // hank_fake_news workspace
    XTan *xcur = nullptr;          // tangent buffers of the last xcd-schedule JVP
    // on-chip wide sweeps: 0 = never, 1 = auto (batches of at least wide_min directions), 2 = every batch (HANK_SCHEDULE=wide: tests)
    int wide_mode = 0, wide_min = 80, num_cus = 256, wide_r = 2;     // wide_r: rows per thread of the wide kernels (2: 1024-thread workgroups, 4 waves per SIMD — since the L2 warming of round 5 the faster geometry for both sweeps, 5.16 / 6.80 ms against 5.30 / 7.10 at N=256; dev knob HANK_WIDE_R=2|4 at hank_create)
    size_t lds_max = 65536;
    char *rec_slab = nullptr;      // the record's ONE allocation
    size_t rec_bytes = 0;
    int *d_ibw = nullptr;          // the wide backward sweep's bracket record (k_wide_prep), valid for the recorded primal or not
    bool seg_valid = true;          // the record's per-target segment records match its lottery (k_lottery writes them except in the persistent Dual pass)
    bool wprep_valid = false;
    std::list<WTan> wtans;         // most recently used first
    WTan *wcur = nullptr;
    std::vector<double> h_Pi, h_z;  // host copies (the wide sweeps take the mixing matrix as a kernel argument)
    long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // see hank_stats
    // primal memo of the host-pointer hank_primal_jvp (NewtonRaphson.jl:91-95 calls JVP(fullFunction, x, y) ~21 times at one x):
    // the x whose linearisation is on record, as the host handed it in
    bool memo_on = true;                              // HANK_PRIMAL_MEMO=0 (read at hank_create) switches it off
    bool xdual_back = true;                           // dev knob HANK_XDUAL_BACK=0 (read at hank_create): a persistent Dual pass runs k_xprimal_back + k_xtan_back instead of k_xdual_back
    bool memo_valid = false;
    std::vector<double> memo_xhh;
    bool stationary = false;                          // the recorded primal is the constant steady-state path with the steady state as both boundaries (hank_fake_news)
    std::vector<double> h_ss_value, h_ss_D;           // the boundary as the host handed it in (stationarity check)
    hipEvent_t ev_stream = nullptr;
    char errmsg[512] = {0};
};

static int fail(hank_ctx *ctx, int code, const char *fmt, ...);
static void w_invalidate(hank_ctx *ctx) { for (WTan &t : ctx->wtans) t.valid = false; }
static void w_new_primal(hank_ctx *ctx) { ctx->wprep_valid = false; }      // (the record is about to be rewritten)
static void w_free_tan(WTan &w) {
    (void)hipFree(w.dxhh); (void)hipFree(w.dpol); (void)hipFree(w.dagg_cm);
    w = WTan();
}
static hipError_t join_side(hank_ctx *ctx) {
    if (!ctx->side_pending) return hipSuccess;
    ctx->side_pending = false;
    return hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0);
}

static int fail(hank_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->errmsg, sizeof(ctx->errmsg), fmt, ap);
        va_end(ap);
    }
    return code;
}

static int hip_status(hipError_t e) {
    if (e == hipErrorOutOfMemory) return HANK_ERR_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNotInitialized || e == hipErrorInsufficientDriver) return HANK_ERR_NO_DEVICE;
    return HANK_ERR_LAUNCH;
}
#define HIPC(ctx, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, hip_status(e_),                                                   \
                        "HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_),       \
                        __FILE__, __LINE__, #call);                                             \
    } while (0)

// a context belongs to ONE HIP device (hank_create: the current one; hank_create_on: the one named). Every entry point
// makes that device current for the duration of the call and restores the caller's, so one host thread can drive one
// context per GPU of a node (GeneralStructures.jl:542-550 has no notion of a device: the shim owns the placement).
static void x_section_release(int dev);
struct DeviceGuard {
    int prev = -1, dev = -1;
    bool switched = false, ok = true;
    explicit DeviceGuard(const hank_ctx *ctx);
    ~DeviceGuard() { if (dev >= 0) x_section_release(dev); if (switched) (void)hipSetDevice(prev); }
};
DeviceGuard::DeviceGuard(const hank_ctx *ctx) {
    if (!ctx) return;
    dev = ctx->device & 63;
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != ctx->device) { switched = hipSetDevice(ctx->device) == hipSuccess; ok = switched; }
}
// (a call must never run on the caller's device with another device's pointers: a failed switch fails the call)
#define ENTER(ctx)                                                                                                       \
    DeviceGuard dev_guard_(ctx);                                                                                         \
    if (!dev_guard_.ok) return fail(ctx, HANK_ERR_NO_DEVICE, "HIP device %d of this context could not be made current", (ctx) ? (ctx)->device : -1)

template <typename T>
static hipError_t dmalloc(T **p, size_t count) {
    return hipMalloc((void **)p, count * sizeof(T) > 0 ? count * sizeof(T) : 8);
}

static size_t primal_lds(const Consts &c) { return sizeof(double) * ((size_t)c.n_e * RBP + (size_t)c.n_e * c.n_e + 16); }

static void free_tanwork(TanWork &w) {
    if (w.g_back) (void)hipGraphExecDestroy(w.g_back);
    if (w.g_fwd) (void)hipGraphExecDestroy(w.g_fwd);
    if (w.g_fback) (void)hipGraphExecDestroy(w.g_fback);
    if (w.g_ffwd) (void)hipGraphExecDestroy(w.g_ffwd);
    (void)hipFree(w.dxhh); (void)hipFree(w.dxr); (void)hipFree(w.dxw); (void)hipFree(w.dxt);
    (void)hipFree(w.ds[0]); (void)hipFree(w.ds[1]); (void)hipFree(w.dD[0]); (void)hipFree(w.dD[1]);
    (void)hipFree(w.dpol); (void)hipFree(w.aggpart); (void)hipFree(w.dagg); (void)hipFree(w.dagg_cm);
    w = TanWork();
}

// ---- graph construction -----------------------------------------------------------------------
static int end_capture(hank_ctx *ctx, hipGraphExec_t *out) {
    hipGraph_t graph = nullptr;
    HIPC(ctx, hipStreamEndCapture(ctx->own_stream, &graph));
    HIPC(ctx, hipGraphInstantiate(out, graph, nullptr, nullptr, 0));
    HIPC(ctx, hipGraphDestroy(graph));
    HIPC(ctx, hipGetLastError());
    ctx->stats[2]++;
    return HANK_OK;
}

static int build_primal_graphs(hank_ctx *ctx) {
    const Consts &c = ctx->c;
    const int P = c.P;
    hipStream_t s = ctx->own_stream;
    const dim3 blk(RBP * c.n_e), grd(ctx->nbp);
    const size_t lds = primal_lds(c);
    // backward: X of the last period from the terminal value, then P fused Y;X steps, then lottery
    HIPC(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, ctx->d_err, 4);
    hipLaunchKernelGGL(k_egm_X, grd, blk, lds, s, c, ctx->d_ss_value, ctx->d_xhh + c.n_hh * (P - 1),
                       ctx->R.s + (size_t)(P - 1) * c.G, ctx->R.kc + (size_t)(P - 1) * c.G, ctx->d_err, P - 1, (const int *)nullptr);
    for (int t = P - 1; t >= 0; t--)
        hipLaunchKernelGGL(k_egm_step, grd, blk, lds, s, c, ctx->R, ctx->d_xhh, t, ctx->d_err);
    hipLaunchKernelGGL(k_lottery, dim3(P * c.n_e), dim3(256), sizeof(int) * (2 * (size_t)c.n_a + 2), s, c, ctx->R, P * c.n_e, ctx->d_err, 1, 1);
    int rc = end_capture(ctx, &ctx->g_pback);
    if (rc) return rc;
    // forward
    HIPC(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < P; t++)
        hipLaunchKernelGGL(k_dist_step, grd, blk, lds, s, c, ctx->R, t, ctx->d_aggpart);
    hipLaunchKernelGGL(k_reduce_parts, dim3(P, 1), dim3(256), 0, s, ctx->d_aggpart, ctx->nbp, 2, ctx->d_agg_rm);
    hipLaunchKernelGGL(k_tan_out, dim3((2 * P + 255) / 256), dim3(256), 0, s, ctx->d_agg_rm, P, 2, ctx->d_agg);
    rc = end_capture(ctx, &ctx->g_pfwd);
    ctx->launches[0] = P + 2;
    ctx->launches[1] = P + 1;
    return rc;
}

// Captures the four tangent graphs for lane type VT (double: one direction per lane; double2: two).
// the row-group count is a template parameter of the kernels and a run-time choice here
#define LAUNCH_RG(RGV, KERNEL, VTYPE, ...)                                                      \
    do {                                                                                        \
        if ((RGV) == 4) hipLaunchKernelGGL((KERNEL<4, VTYPE>), __VA_ARGS__);                    \
        else if ((RGV) == 2) hipLaunchKernelGGL((KERNEL<2, VTYPE>), __VA_ARGS__);               \
        else hipLaunchKernelGGL((KERNEL<1, VTYPE>), __VA_ARGS__);                               \
    } while (0)

#define LAUNCH_RG_SS(RGV, SSV, KERNEL, VTYPE, ...)                                              \
    do {                                                                                        \
        if (SSV) {                                                                              \
            if ((RGV) == 4) hipLaunchKernelGGL((KERNEL<4, VTYPE, true>), __VA_ARGS__);          \
            else if ((RGV) == 2) hipLaunchKernelGGL((KERNEL<2, VTYPE, true>), __VA_ARGS__);     \
            else hipLaunchKernelGGL((KERNEL<1, VTYPE, true>), __VA_ARGS__);                     \
        } else {                                                                                \
            if ((RGV) == 4) hipLaunchKernelGGL((KERNEL<4, VTYPE, false>), __VA_ARGS__);         \
            else if ((RGV) == 2) hipLaunchKernelGGL((KERNEL<2, VTYPE, false>), __VA_ARGS__);    \
            else hipLaunchKernelGGL((KERNEL<1, VTYPE, false>), __VA_ARGS__);                    \
        }                                                                                       \
    } while (0)

// captures ONE pair of tangent graphs, on first use: which = 0 the tangent-only sweeps (hank_jvp), 1 the dual-sweep
// launches (hank_primal_jvp)
template <typename VT, typename VF>
static int capture_tangent_graphs(hank_ctx *ctx, TanWork &w, int which) {
    const int RGB = w.RGB, RGF = w.RGF;
    const unsigned nbf = w.nbf;
    const Consts &c = ctx->c;
    const size_t P = c.P;
    const int N = w.N;
    const size_t GV = (size_t)(c.n_a + KV) * c.n_e;
    hipStream_t s = ctx->own_stream;
    const dim3 blk(64 * c.n_e);
    const unsigned ny = (w.g.N + w.g.NC - 1) / w.g.NC, nyf = (w.gf.N + w.gf.NC - 1) / w.gf.NC;
    const int PN = (int)(P * N);
    const VT *dxr = reinterpret_cast<const VT *>(w.dxr), *dxw = reinterpret_cast<const VT *>(w.dxw), *dxt = reinterpret_cast<const VT *>(w.dxt);
    VT *ds[2] = {reinterpret_cast<VT *>(w.ds[0]), reinterpret_cast<VT *>(w.ds[1])};
    VF *dD[2] = {reinterpret_cast<VF *>(w.dD[0]), reinterpret_cast<VF *>(w.dD[1])};
    VT *dpol = reinterpret_cast<VT *>(w.dpol);
    VF *dpolf = reinterpret_cast<VF *>(w.dpol), *aggpart = reinterpret_cast<VF *>(w.aggpart);
    const unsigned nbt = (w.nbx + RGB - 1) / RGB;
    int rc = HANK_OK, cur = 0;
    if (which == 0) {
    // backward tangent sweep
    HIPC(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_tan_in, dim3((PN + 255) / 256), dim3(256), 0, s, w.dxhh, c.n_hh, (int)P, N, w.dxr, w.dxw, w.dxt);
    LAUNCH_RG(RGB, k_tan_back, VT, dim3(nbt, ny), blk, 0, s, c, ctx->R, ctx->d_xhh, dxr, dxw, dxt, w.g, (int)P - 1, 1,
                       ds[1], ds[0], dpol);
    cur = 0;
    for (int t = (int)P - 1; t >= 0; t--) {
        LAUNCH_RG(RGB, k_tan_back, VT, dim3(nbt, ny), blk, 0, s, c, ctx->R, ctx->d_xhh, dxr, dxw, dxt, w.g, t, 0,
                           ds[cur], ds[cur ^ 1], dpol);
        cur ^= 1;
    }
    rc = end_capture(ctx, &w.g_back);
    if (rc) return rc;
    // forward tangent sweep
    HIPC(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_zero_f64, dim3(512), dim3(256), 0, s, w.dD[0], GV * N);  // dD_0 = 0 (ForwardIteration.jl:293)
    cur = 0;
    for (int t = 0; t < (int)P; t++) {
        LAUNCH_RG_SS(RGF, w.gf.ss, k_tan_fwd, VF, dim3(nbf, nyf), blk, 0, s, c, ctx->R, w.gf, t, dD[cur], dD[cur ^ 1], dpolf, aggpart);
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)P, (2 * N + 63) / 64), dim3(256), 0, s, w.aggpart, (int)nbf, 2 * N, w.dagg);
    hipLaunchKernelGGL(k_tan_out, dim3((2 * PN + 255) / 256), dim3(256), 0, s, w.dagg, (int)P, 2 * N, w.dagg_cm);
    rc = end_capture(ctx, &w.g_fwd);
    if (rc) return rc;
    ctx->launches[2] = (int)P + 2;
    ctx->launches[3] = (int)P + 3;
    return HANK_OK;
    }

    // ---- dual-sweep graphs: the primal recurrence and the tangent recurrence advance in the SAME
    // chain of launches, the tangent one period behind (it reads the record the previous launch wrote):
    // T launches per direction instead of 2(T-1).
    const dim3 pblk(RBP * c.n_e), pgrd(ctx->nbp);
    const size_t lds = primal_lds(c);
    HIPC(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, ctx->d_err, 4);
    hipLaunchKernelGGL(k_tan_in, dim3((PN + 255) / 256), dim3(256), 0, s, w.dxhh, c.n_hh, (int)P, N, w.dxr, w.dxw, w.dxt);
    hipLaunchKernelGGL(k_egm_X, pgrd, pblk, lds, s, c, ctx->d_ss_value, ctx->d_xhh + c.n_hh * (P - 1),
                       ctx->R.s + (size_t)(P - 1) * c.G, ctx->R.kc + (size_t)(P - 1) * c.G, ctx->d_err, (int)P - 1, (const int *)nullptr);
    cur = 0;
    for (int k = 0; k <= (int)P; k++) {
        const int tp = k < (int)P ? (int)P - 1 - k : -1;
        if (k == 0) {
            LAUNCH_RG(RGB, k_fused_back, VT, dim3(ctx->nbp + nbt, ny), blk, 0, s, c, ctx->R, ctx->d_xhh, ctx->d_err, tp, ctx->nbp,
                               dxr, dxw, dxt, w.g, (int)P - 1, 1, ds[1], ds[0], dpol);
        } else {
            LAUNCH_RG(RGB, k_fused_back, VT, dim3(ctx->nbp + nbt, ny), blk, 0, s, c, ctx->R, ctx->d_xhh, ctx->d_err, tp, ctx->nbp,
                               dxr, dxw, dxt, w.g, (int)P - k, 0, ds[cur], ds[cur ^ 1], dpol);
            cur ^= 1;
        }
    }
    hipLaunchKernelGGL(k_lottery, dim3(P * c.n_e), dim3(256), sizeof(int) * (2 * (size_t)c.n_a + 2), s, c, ctx->R, (int)P * c.n_e, ctx->d_err, 1, 1);
    rc = end_capture(ctx, &w.g_fback);
    if (rc) return rc;
    HIPC(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_zero_f64, dim3(512), dim3(256), 0, s, w.dD[0], GV * N);
    cur = 0;
    for (int k = 0; k <= (int)P; k++) {
        const int tp = k < (int)P ? k : -1, tt = k - 1;
        LAUNCH_RG_SS(RGF, w.gf.ss, k_fused_fwd, VF, dim3(ctx->nbp + nbf, nyf), blk, 0, s, c, ctx->R, tp, ctx->nbp, ctx->d_aggpart, w.gf, tt,
                  dD[cur], dD[cur ^ 1], dpolf, aggpart);
        if (tt >= 0) cur ^= 1;
    }
    hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)P, 1), dim3(256), 0, s, ctx->d_aggpart, ctx->nbp, 2, ctx->d_agg_rm);
    hipLaunchKernelGGL(k_tan_out, dim3((unsigned)((2 * P + 255) / 256)), dim3(256), 0, s, ctx->d_agg_rm, (int)P, 2, ctx->d_agg);
    hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)P, (2 * N + 63) / 64), dim3(256), 0, s, w.aggpart, (int)nbf, 2 * N, w.dagg);
    hipLaunchKernelGGL(k_tan_out, dim3((2 * PN + 255) / 256), dim3(256), 0, s, w.dagg, (int)P, 2 * N, w.dagg_cm);
    rc = end_capture(ctx, &w.g_ffwd);
    ctx->launches[4] = (int)P + 5;
    ctx->launches[5] = (int)P + 5;
    return rc;
}

static int ensure_tanwork(hank_ctx *ctx, int N) {
    for (auto it = ctx->tws.begin(); it != ctx->tws.end(); ++it)
        if (it->N == N) { ctx->tws.splice(ctx->tws.begin(), ctx->tws, it); ctx->tw = &ctx->tws.front(); return HANK_OK; }
    const char *ce = getenv("HANK_TAN_CACHE");
    const size_t keep = ce ? (size_t)atoi(ce) : 3;
    while (ctx->tws.size() >= (keep ? keep : 1)) {     // evict the least recently used — the async entries may still have its graphs in flight
        HIPC(ctx, join_side(ctx));
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
        free_tanwork(ctx->tws.back());
        ctx->tws.pop_back();
    }
    ctx->tws.emplace_front();
    ctx->tw = &ctx->tws.front();
    TanWork &w = *ctx->tw;
    ctx->stats[1]++;
    const Consts &c = ctx->c;
    const size_t P = c.P, G = c.G;
    w.N = N;
    // an even batch can run two directions per lane (16-byte accesses): the [..][N] layout is the same, so
    // each sweep picks its own lane width
    const int VB = tan_lane_width(N, 0), VF = tan_lane_width(N, 1);
    auto geom = [&](int V, TanGeom &g) {
        const int NV = N / V;
        int NC = 1, lg = 0;
        while (NC < NV && NC < 64) { NC <<= 1; lg++; }
        const int RB = 64 / NC;
        g.N = NV; g.NC = NC; g.lgNC = lg; g.nbx = (c.n_a + RB - 1) / RB;
        g.ss = 0;
    };
    geom(VB, w.g); geom(VF, w.gf);
    w.nbx = w.g.nbx; w.nbxf = w.gf.nbx;
    const size_t GV = (size_t)(c.n_a + KV) * c.n_e;   // dD state carries KV virtual rows per column
    // forward kernel form: source-stationary from 16-lane groups on (N >= 18: forward sweep 3.42 -> 3.05 ms at N=32 with 2 row
    // groups; round 4, with the column index in a scalar register: 4.58 -> 4.28 ms at N=64, 12.5 -> 11.6 ms at N=256,
    // profiles/r04_wide_knobs.log), target-stationary gather below
    const char *se = getenv("HANK_FWD_SS");   // dev knob
    w.gf.ss = se ? atoi(se) : (w.gf.NC >= 16 ? 1 : 0);
    const int RGB = tan_rg(w.g.N, 0), RGF = tan_rg(w.gf.N, 1, w.gf.ss);
    const unsigned nbf = (w.nbxf + RGF - 1) / RGF + KV;   // forward blocks: regular + mass-point
    w.VB = VB; w.VF = VF; w.RGB = RGB; w.RGF = RGF; w.nbf = nbf;
    // a failed allocation must not leave a half-built entry in the cache: a retry with this N would find it, return
    // HANK_OK and launch on null pointers
    auto alloc = [&]() -> int {
        HIPC(ctx, dmalloc(&w.dxhh, (size_t)c.n_hh * P * N));
        HIPC(ctx, dmalloc(&w.dxr, P * N));
        HIPC(ctx, dmalloc(&w.dxw, P * N));
        HIPC(ctx, dmalloc(&w.dxt, P * N));
        for (int k = 0; k < 2; k++) {
            HIPC(ctx, dmalloc(&w.ds[k], G * N));
            HIPC(ctx, dmalloc(&w.dD[k], GV * N));
        }
        HIPC(ctx, dmalloc(&w.dpol, P * G * N));
        HIPC(ctx, dmalloc(&w.aggpart, 2 * P * (size_t)nbf * N));      // both aggregates: [P][blocks][2 N]
        HIPC(ctx, dmalloc(&w.dagg, 2 * P * N));
        HIPC(ctx, dmalloc(&w.dagg_cm, 2 * P * N));                    // (P, 2 N) column-major: the policy-weighted aggregate's N columns, then the grid-weighted one's
        return HANK_OK;
    };
    const int rc = alloc();
    if (rc) { free_tanwork(w); ctx->tws.pop_front(); ctx->tw = ctx->tws.empty() ? nullptr : &ctx->tws.front(); (void)hipGetLastError(); return rc; }
    return HANK_OK;
}

// the graph pair of one schedule, captured the first time that schedule runs at this batch width
static int ensure_graphs(hank_ctx *ctx, TanWork &w, int which) {
    if (which == 0 ? w.g_back != nullptr : w.g_fback != nullptr) return HANK_OK;
    if (w.VB == 2) return w.VF == 2 ? capture_tangent_graphs<double2, double2>(ctx, w, which) : capture_tangent_graphs<double2, double>(ctx, w, which);
    return w.VF == 2 ? capture_tangent_graphs<double, double2>(ctx, w, which) : capture_tangent_graphs<double, double>(ctx, w, which);
}

static int x_status(hank_ctx *ctx);
static int fetch_device_error(hank_ctx *ctx) {
    int e[4] = {0, 0, 0, 0};
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    HIPC(ctx, hipMemcpy(e, ctx->d_err, sizeof(e), hipMemcpyDeviceToHost));
    if (ctx->schedule >= 1) {
        // a persistent sweep that did not run (its groups did not form, a wait timed out) leaves the record unwritten: what
        // the kernels behind it then found in it (a "non-monotone policy", say) is not an error of the model — report the sweep
        const int xs = x_status(ctx);
        if (xs) {
            if (e[0] != 0) HIPC(ctx, hipMemsetAsync(ctx->d_err, 0, sizeof(e), ctx->stream));
            return xs;
        }
    }
    if (e[0] == 0) return HANK_OK;
    ctx->primal_done = false;
    HIPC(ctx, hipMemsetAsync(ctx->d_err, 0, sizeof(e), ctx->stream));      // reported once: the next call starts clean
    switch (e[0]) {
    case ERR_KNOTS:
        return fail(ctx, HANK_ERR_KNOTS,
                    "knot-vectors must be unique and sorted in increasing order (EGM implied state, "
                    "period %d, productivity state %d, wealth index %d)", e[1] + 1, e[2] + 1, e[3] + 1);
    case ERR_DOMAIN:
        return fail(ctx, HANK_ERR_DOMAIN,
                    "DomainError: negative base under a non-integer power (period %d, productivity "
                    "state %d, wealth index %d)", e[1] + 1, e[2] + 1, e[3] + 1);
    case ERR_NONMONO:
        return fail(ctx, HANK_ERR_NONMONOTONE,
                    "savings policy is not monotone in wealth (period %d, productivity state %d, "
                    "wealth index %d)", e[1] + 1, e[2] + 1, e[3] + 1);
    default:
        return fail(ctx, HANK_ERR_BAD_ARG, "unknown device error %d (%d,%d,%d) d_err=%p", e[0], e[1], e[2], e[3], (void*)ctx->d_err);
    }
}

// ================================ XCD-local persistent sweeps: host side ==========================
// Two persistent launches must never share the chip half-resident (each would wait for workgroups the other's
// spinning workgroups keep out): every sweep launch of this process, from any context or stream, is ordered behind
// the previous one through one event per device.
static std::mutex g_xmutex;
static hipEvent_t g_xlast[64] = {};
// ... and the launches of ONE call are enqueued as one block: between x_serialize_begin and x_serialize_end this thread holds the
// device's section lock, so a second host thread (another context on the same GPU: parallel.DeviceGroup) cannot wait for the event
// of the call BEFORE and then enqueue its sweeps beside this call's. Nested sections of one thread count (the one-pass Dual pass
// opens one around its prologue, its two halves open their own inside); a section left open by an early return is closed when the
// entry point returns (DeviceGuard).
static std::mutex g_xsection[64];
static thread_local int t_xdepth[64];
static void x_section_release(int dev) {
    if (t_xdepth[dev] > 0) { t_xdepth[dev] = 0; g_xsection[dev].unlock(); }
}

// dynamic LDS of the persistent kernels (the expressions the kernels carve up): it grows with the horizon P
static size_t x_lds_primal_back(const Consts &c) { return sizeof(double) * ((size_t)c.n_e * 64 + (size_t)c.n_e * c.n_e + c.n_a + 4 * (size_t)c.P) + 64; }
static size_t x_lds_tan_back(const Consts &c, int D) {
    const int SLt = D == 4 ? 6 : D;      // XTileT<D>::SL
    return sizeof(double) * ((size_t)SLt * c.n_e * 64 + c.P + 1 + 3 * (size_t)c.P + 1 + 3 * (size_t)c.P * D) + sizeof(int) * (size_t)c.P + 64;
}
// k_xdual_back<D>: tile of D + 1 slots, the grid, the household inputs and this group's input tangents of every period
static size_t x_lds_dual_back(const Consts &c, int D) {
    const int NSL = D + 1, SLt = NSL <= 2 ? NSL : (NSL <= 6 ? 6 : 10);
    return sizeof(double) * ((size_t)SLt * c.n_e * 64 + c.n_a + 1 + 4 * (size_t)c.P + 3 * (size_t)c.P * D) + 64;
}
// k_xfwd with NSL live slots (the D partials + the value): tile, Pi, {source range, clamped prefix} and source members of every period
static size_t x_lds_fwd(const Consts &c, int NSL) {
    const int SLt = NSL <= 2 ? NSL : (NSL <= 6 ? 6 : 10);      // XSlots<NSL>::SL
    const int NAP = 2 * NSL;                                   // the two aggregates' terms per lane (k_xfwd: aggsh)
    return sizeof(double) * ((size_t)SLt * c.n_e * 64 + (size_t)c.n_e * c.n_e + 1 + (size_t)c.n_e * 64 * NAP) + sizeof(int) * ((size_t)c.P * c.n_e + c.P) + 64;
}
// the grid fits the XCD-local schedule: a 63-row slab per CU of an XCD, and the Float64 sweeps' LDS (which holds the
// per-period inputs of the WHOLE horizon) fits a workgroup
static bool x_supported(const hank_ctx *ctx, int cus, size_t lds_max) {
    const Consts &c = ctx->c;
    const int Sact = (c.n_a + XRW - 1) / XRW;
    return cus >= XG && Sact <= cus / XG && c.n_e <= 16 && std::max(x_lds_primal_back(c), x_lds_fwd(c, 1)) <= lds_max;
}

static void x_free_tan(XTan &w) {
    (void)hipFree(w.dxhh); (void)hipFree(w.dxr); (void)hipFree(w.dxw); (void)hipFree(w.dxt);
    (void)hipFree(w.dpol); (void)hipFree(w.daggpart); (void)hipFree(w.dagg_pass); (void)hipFree(w.dagg_cm);
    w = XTan();
}

static void x_free(hank_ctx *ctx) {
    XWork &X = ctx->xw;
    for (XTan &w : X.tans) x_free_tan(w);
    X.tans.clear();
    ctx->xcur = nullptr;
    (void)hipFree(X.sync); (void)hipFree(X.st_s); (void)hipFree(X.st_ds); (void)hipFree(X.st_D); (void)hipFree(X.st_dD);
    (void)hipFree(X.Dvirt); (void)hipFree(X.aggpart); (void)hipFree(X.rho); (void)hipFree(X.srcB); (void)hipFree(X.srcF); (void)hipFree(X.unitsF); (void)hipFree(X.unit_overflow);
    X = XWork();
}

static int x_setup(hank_ctx *ctx) {
    XWork &X = ctx->xw;
    if (X.ready) return HANK_OK;
    const Consts &c = ctx->c;
    hipDeviceProp_t prop;
    HIPC(ctx, hipGetDeviceProperties(&prop, ctx->device));
    X.grid = (prop.multiProcessorCount / XG) * XG;
    X.Sact = (c.n_a + XRW - 1) / XRW;
    // 64*n_e threads per workgroup (+ one run-ahead wave in the forward sweep); the register budget follows the bound
    X.maxt = 64 * (c.n_e + 1) <= 768 ? 768 : 1024;
    X.dmax = X.maxt == 768 ? XD_MAX : 2;
    const size_t G = c.G, GV = G + 64 * (size_t)c.n_e, P = c.P;
    HIPC(ctx, dmalloc(&X.sync, (size_t)2 + 2 * XPASS_MAX));
    HIPC(ctx, hipMemsetAsync(X.sync, 0, sizeof(XSync) * ((size_t)2 + 2 * XPASS_MAX), ctx->stream));      // x_status reads blocks 0, 1 also when the launches recorded the primal
    HIPC(ctx, dmalloc(&X.st_s, 2 * XG * G));
    HIPC(ctx, dmalloc(&X.st_ds, 2 * XG * G * X.dmax));
    const size_t GM = (size_t)c.n_e * X.Sact * 64;       // member-major state of the forward sweeps: [n_e][members][64]
    HIPC(ctx, dmalloc(&X.st_D, 2 * XG * std::max(GV, GM)));
    HIPC(ctx, dmalloc(&X.st_dD, 2 * XG * GM * (X.dmax + 2)));      // D partials + the value, padded to pairs
    HIPC(ctx, dmalloc(&X.Dvirt, P * c.n_e * 64));
    HIPC(ctx, dmalloc(&X.aggpart, 2 * P * (size_t)X.Sact));       // [P][members][2]
    HIPC(ctx, dmalloc(&X.rho, P));
    HIPC(ctx, dmalloc(&X.srcB, P * X.Sact));
    HIPC(ctx, dmalloc(&X.srcF, P * X.Sact));
    HIPC(ctx, dmalloc(&X.unitsF, P * X.Sact * XUCAP));
    HIPC(ctx, dmalloc(&X.unit_overflow, 1));
    HIPC(ctx, hipMemsetAsync(X.unit_overflow, 0, sizeof(int), ctx->stream));
    if (const char *ng = getenv("HANK_XNEIGH")) X.neigh = atoi(ng) != 0;
    if (const char *uc = getenv("HANK_XUCAP")) X.ucap = std::min(XUCAP, std::max(1, atoi(uc)));
    {   // the bound on every wait inside a persistent sweep, in 100 MHz ticks (read once, here)
        double ms = 20.0;
        if (const char *wm = getenv("HANK_XWAIT_MS")) ms = atof(wm);
        const unsigned long long ticks = (unsigned long long)(std::max(ms, 0.01) * 1e5);
        HIPC(ctx, hipMemcpyToSymbol(HIP_SYMBOL(hank::g_xwait_ticks), &ticks, sizeof(ticks)));
    }
    if (const char *sv = getenv("HANK_XSYNCWAVE")) X.syncwave = atoi(sv) != 0;
    X.lds_max = (int)prop.sharedMemPerBlock;
    if (const char *xf = getenv("HANK_XFAULT")) {      // "placement" | "timeout", optionally ":primal" | ":tangent" | ":fixedpoint" (default: every persistent launch)
        X.fault = strncmp(xf, "placement", 9) == 0 ? XERR_PLACEMENT : (strncmp(xf, "timeout", 7) == 0 ? XERR_TIMEOUT : (strncmp(xf, "stall", 5) == 0 ? 3 : 0));
        const char *w = strchr(xf, ':');
        X.fault_where = !w ? 7 : (strcmp(w, ":primal") == 0 ? 1 : (strcmp(w, ":tangent") == 0 ? 2 : (strcmp(w, ":fixedpoint") == 0 ? 4 : 7)));
    }
    HIPC(ctx, hipMemsetAsync(X.Dvirt, 0, sizeof(double) * P * c.n_e * 64, ctx->stream));
    X.ready = true;
    return HANK_OK;
}

// tangent buffers for a batch of N directions, from a small most-recently-used cache (Jacobian assembly at N = 256
// and the Newton inner loop at N = 1 alternate: neither re-allocates)
static int x_ensure_tan(hank_ctx *ctx, int N, XTan **out) {
    XWork &X = ctx->xw;
    for (auto it = X.tans.begin(); it != X.tans.end(); ++it)
        if (it->N == N) { X.tans.splice(X.tans.begin(), X.tans, it); *out = &X.tans.front(); return HANK_OK; }
    if (N > 8 * X.dmax * XPASS_MAX) return fail(ctx, HANK_ERR_BAD_ARG, "N=%d exceeds %d directions per call", N, 8 * X.dmax * XPASS_MAX);
    const char *ce = getenv("HANK_TAN_CACHE");
    const size_t keep = ce ? (size_t)atoi(ce) : 3;
    while (X.tans.size() >= (keep ? keep : 1)) {       // evict the least recently used — after the stream has drained
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->xcur == &X.tans.back()) ctx->xcur = nullptr;
        x_free_tan(X.tans.back());
        X.tans.pop_back();
    }
    X.tans.emplace_front();
    XTan &w = X.tans.front();
    const Consts &c = ctx->c;
    const size_t P = c.P, G = c.G;
    w.N = N;
    size_t off = 0;
    for (int n0 = 0; n0 < N; n0 += XG * X.dmax) {
        XPass ps;
        ps.n0 = n0; ps.N = std::min(N - n0, XG * X.dmax);
        ps.D = 1; while (XG * ps.D < ps.N) ps.D *= 2;
        ps.groups = (ps.N + ps.D - 1) / ps.D;
        ps.dpol_off = off;
        off += P * ps.groups * G * ps.D;
        w.passes.push_back(ps);
    }
    if ((int)w.passes.size() > XPASS_MAX) {
        const int np_ = (int)w.passes.size();
        X.tans.pop_front();
        return fail(ctx, HANK_ERR_BAD_ARG, "N=%d needs %d passes, at most %d per call", N, np_, XPASS_MAX);
    }
    ctx->stats[1]++;
    // (a failed allocation leaves no half-built entry behind: the next call with this N must not find it in the cache)
    auto alloc = [&]() -> int {
        HIPC(ctx, dmalloc(&w.dxhh, (size_t)c.n_hh * P * N));
        HIPC(ctx, dmalloc(&w.dxr, P * N)); HIPC(ctx, dmalloc(&w.dxw, P * N)); HIPC(ctx, dmalloc(&w.dxt, P * N));
        HIPC(ctx, dmalloc(&w.dpol, off));
        const size_t W = (size_t)XG * X.dmax, nb = (size_t)X.Sact;
        HIPC(ctx, dmalloc(&w.daggpart, 2 * P * nb * W));              // both aggregates: [P][members][2 W]
        HIPC(ctx, hipMemsetAsync(w.daggpart, 0, sizeof(double) * 2 * P * nb * W, ctx->stream));
        HIPC(ctx, dmalloc(&w.dagg_pass, 2 * P * W));
        HIPC(ctx, dmalloc(&w.dagg_cm, 2 * P * N));                    // (P, 2 N) column-major
        return HANK_OK;
    };
    const int rc = alloc();
    if (rc) { x_free_tan(w); X.tans.pop_front(); (void)hipGetLastError(); return rc; }
    *out = &w;
    return HANK_OK;
}


template <int MAXT>
static void x_launch_tan_back(int D, dim3 grd, dim3 blk, size_t lds, hipStream_t s, const XTanBackArgs &ab) {
    if (D == 1) hipLaunchKernelGGL((k_xtan_back<1, MAXT>), grd, blk, lds, s, ab);
    else if (D == 2) hipLaunchKernelGGL((k_xtan_back<2, MAXT>), grd, blk, lds, s, ab);
    else if (D == 4) { if constexpr (MAXT == 768) hipLaunchKernelGGL((k_xtan_back<4, MAXT>), grd, blk, lds, s, ab); }
}

static void x_launch_dual_back(int D, dim3 grd, dim3 blk, size_t lds, hipStream_t s, const XDualBackArgs &ab) {
    if (D == 1) hipLaunchKernelGGL((k_xdual_back<1, 768>), grd, blk, lds, s, ab);
    else if (D == 2) hipLaunchKernelGGL((k_xdual_back<2, 768>), grd, blk, lds, s, ab);
    else hipLaunchKernelGGL((k_xdual_back<4, 768>), grd, blk, lds, s, ab);
}

// k_xfwd<D, VAL>: D = 0 (the Float64 sweep alone, VAL), 1, 2, 4
template <int MAXT>
static void x_launch_fwd(int D, bool val, dim3 grd, dim3 blk, size_t lds, hipStream_t s, const XSweepFwdArgs &a) {
#define XF(DV, VV) hipLaunchKernelGGL((k_xfwd<DV, VV, MAXT>), grd, blk, lds, s, a)
    if (D == 0) XF(0, true);
    else if (D == 1) { if (val) XF(1, true); else XF(1, false); }
    else if (D == 2) { if (val) XF(2, true); else XF(2, false); }
    else if (D == 4) { if constexpr (MAXT == 768) { if (val) XF(4, true); else XF(4, false); } }
#undef XF
}
static void x_launch_fwd(const XWork &X, int D, bool val, dim3 grd, dim3 blk, size_t lds, hipStream_t s, const XSweepFwdArgs &a) {
    if (X.maxt == 768) x_launch_fwd<768>(D, val, grd, blk, lds, s, a);
    else x_launch_fwd<1024>(D, val, grd, blk, lds, s, a);
}
// the forward sweeps' geometry at the recorded lottery (once per primal)
static void x_ensure_rng(hank_ctx *ctx) {
    XWork &X = ctx->xw;
    if (X.rng_valid) return;
    hipLaunchKernelGGL(k_xunits_fwd, dim3((unsigned)ctx->c.P, (unsigned)X.Sact), dim3(256), 0, ctx->stream, ctx->c, ctx->R, X.Sact, X.srcF, X.unitsF, X.unit_overflow, X.ucap);
    X.rng_valid = true;
}

static int ensure_seg(hank_ctx *ctx) {       // before a reader of R.seg (launch-family forward tangent sweeps, hank_fake_news)
    if (ctx->seg_valid) return HANK_OK;
    hipLaunchKernelGGL(k_seg_build, dim3((unsigned)(ctx->c.P * ctx->c.n_e)), dim3(256), 0, ctx->stream, ctx->c, ctx->R, ctx->c.P * ctx->c.n_e);
    HIPC(ctx, hipGetLastError());
    ctx->seg_valid = true;
    return HANK_OK;
}
static int x_serialize_begin(hank_ctx *ctx) {
    const int d = ctx->device & 63;
    if (t_xdepth[d]++ == 0) g_xsection[d].lock();
    std::lock_guard<std::mutex> lk(g_xmutex);
    if (g_xlast[d]) HIPC(ctx, hipStreamWaitEvent(ctx->stream, g_xlast[d], 0));
    return HANK_OK;
}
static int x_serialize_end(hank_ctx *ctx) {
    const int d = ctx->device & 63;
    {
        std::lock_guard<std::mutex> lk(g_xmutex);
        hipEvent_t &e = g_xlast[d];
        if (!e) HIPC(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPC(ctx, hipEventRecord(e, ctx->stream));
    }
    if (t_xdepth[d] > 0 && --t_xdepth[d] == 0) g_xsection[d].unlock();
    return HANK_OK;
}

// zero the sync blocks of the launches about to be enqueued (and, under the dev knob, pre-set their status words)
static int x_sync_reset(hank_ctx *ctx, XSync *base, int count, int where) {     // where: 1 primal sweeps, 2 tangent sweeps, 4 the steady state's fixed points
    HIPC(ctx, hipMemsetAsync(base, 0, sizeof(XSync) * (size_t)count, ctx->stream));
    if (ctx->xw.fault && ctx->xw.fault != 3 && (ctx->xw.fault_where & where)) hipLaunchKernelGGL(k_xpoison, dim3(1), dim3(64), 0, ctx->stream, base, count, (unsigned)ctx->xw.fault);
    return HANK_OK;
}

// the Float64 recurrences at the context's current x (d_xhh) and boundary: two persistent launches on ONE XCD's
// workgroups (the policy sequence, the distribution path and the linearisation record the tangent sweeps read)
// skip_fwd: the distribution sweep travels with the tangents' forward sweep instead (k_xfwd<D, true>, x_run_tangent(.., val))
// dual: the backward sweep carries the partials of this ONE-pass batch too (k_xdual_back; implies skip_fwd — the caller follows
// with x_run_tangent(.., val, skip_back))
static int x_run_primal(hank_ctx *ctx, bool skip_fwd = false, XTan *dual = nullptr, bool pro_done = false) {      // pro_done: k_xdual_prologue has run (x_dual)
    XWork &X = ctx->xw;
    const Consts &c = ctx->c;
    const size_t P = c.P;
    hipStream_t s = ctx->stream;
    int rc = x_serialize_begin(ctx);
    if (rc) return rc;
    if (!pro_done) {
        rc = x_sync_reset(ctx, X.sync, 2, 1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, ctx->d_err, 4);
        hipLaunchKernelGGL(k_xrho, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, ctx->d_xhh, c.n_hh, (int)P, X.rho);
        if (dual) hipLaunchKernelGGL(k_tan_in, dim3((unsigned)((P * dual->N + 255) / 256)), dim3(256), 0, s, dual->dxhh, c.n_hh, (int)P, dual->N, dual->dxr, dual->dxw, dual->dxt);
    }
    // + one wave that only runs the group barrier's poll, where the block has room (dev knob HANK_XSYNCWAVE=0: wave 0 polls)
    const bool fits = 64 * (c.n_e + 1) <= X.maxt && X.syncwave;
    const dim3 grd(X.grid), blk(fits ? 64 * (c.n_e + 1) : 64 * c.n_e), blkf = blk;
    XBackArgs ab{};
    ab.c = c; ab.ss_value = ctx->d_ss_value; ab.xhh = ctx->d_xhh; ab.rho = X.rho; ab.sy = X.sync; ab.st_s = X.st_s;
    ab.err = ctx->d_err; ab.R = ctx->R;
    const size_t ldsb = x_lds_primal_back(c);
    HIPC(ctx, hipEventRecord(ctx->ev[0], s));
    if (dual) {
        const XPass &ps = dual->passes[0];
        XDualBackArgs db{};
        db.p = ab; db.dxr = dual->dxr; db.dxw = dual->dxw; db.dxt = dual->dxt; db.Ntot = dual->N; db.n0 = ps.n0; db.N = ps.N;
        db.st_ds = X.st_ds; db.dpol = dual->dpol + ps.dpol_off; db.groups = ps.groups;
        x_launch_dual_back(ps.D, grd, blk, x_lds_dual_back(c, ps.D), s, db);
    } else if (X.maxt == 768) hipLaunchKernelGGL((k_xprimal_back<768>), grd, blk, ldsb, s, ab);
    else hipLaunchKernelGGL((k_xprimal_back<1024>), grd, blk, ldsb, s, ab);
    HIPC(ctx, hipEventRecord(ctx->ev[1], s));
    // (the Dual pass's forward half reads the lottery through its work units: the per-target segment records are built when somebody asks)
    hipLaunchKernelGGL(k_lottery, dim3((unsigned)(P * c.n_e)), dim3(256), sizeof(int) * (2 * (size_t)c.n_a + 2), s, c, ctx->R, (int)P * c.n_e, ctx->d_err, dual ? 0 : 1, 1);
    ctx->seg_valid = !dual;
    X.rng_valid = false;
    x_ensure_rng(ctx);
    HIPC(ctx, hipEventRecord(ctx->ev[6], s));
    if (!skip_fwd) {
        XSweepFwdArgs fa{};
        fa.c = c; fa.R = ctx->R; fa.sy = X.sync + 1; fa.st = X.st_D; fa.D0 = ctx->d_ss_D; fa.groups = 1; fa.Dvirt = X.Dvirt; fa.aggpart = X.aggpart;
        fa.src = X.srcF; fa.units = X.unitsF; fa.overflow = X.unit_overflow; fa.all_members = X.neigh ? 0 : 1;
        x_launch_fwd(X, 0, true, grd, blkf, x_lds_fwd(c, 1), s, fa);
    }
    HIPC(ctx, hipEventRecord(ctx->ev[2], s));
    if (!skip_fwd) {
        hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)P, 1), dim3(256), 0, s, X.aggpart, X.Sact, 2, ctx->d_agg_rm);
            hipLaunchKernelGGL(k_tan_out, dim3((unsigned)((2 * P + 255) / 256)), dim3(256), 0, s, ctx->d_agg_rm, (int)P, 2, ctx->d_agg);
        hipLaunchKernelGGL(k_xfix_D, dim3((unsigned)((P * c.n_e + 255) / 256)), dim3(256), 0, s, c, ctx->R.Dseq, X.Dvirt, X.Sact);
    }
    HIPC(ctx, hipGetLastError());
    rc = x_serialize_end(ctx);
    if (rc) return rc;
    ctx->stats[0] += skip_fwd ? 1 : 2;
    X.last_passes = 1;
    ctx->launches[0] = ctx->launches[1] = 1;
    ctx->ev_valid[0] = true; ctx->ev_valid[1] = !skip_fwd;
    ctx->ev_valid[2] = ctx->ev_valid[3] = ctx->ev_valid[4] = ctx->ev_valid[5] = false;
    ctx->primal_done = true; w_new_primal(ctx);
    X.src_valid = false;
    for (XTan &t : X.tans) t.valid = false;
    w_invalidate(ctx);
    ctx->xcur = nullptr;
    return HANK_OK;
}

// the N partials of `w` at the recorded primal: two persistent launches per pass of up to 8*dmax directions, every XCD a group
// val: the first pass's forward sweep carries the value too (the Float64 distribution sweep of a Dual pass) and writes the record
// skip_back: the backward sweep of this (one-pass) batch has run with the Float64 sweep (k_xdual_back, x_run_primal(.., dual))
struct XOut { double *agg = nullptr, *dagg = nullptr; bool pro_done = false, done = false; };      // x_dual's merged launches: the caller's output buffers, and what has been done
static int x_run_tangent(hank_ctx *ctx, XTan *w, bool val = false, bool skip_back = false, XOut *xo = nullptr) {
    XWork &X = ctx->xw;
    const Consts &c = ctx->c;
    const size_t P = c.P;
    hipStream_t s = ctx->stream;
    int rc = x_serialize_begin(ctx);
    if (rc) return rc;
    const int np = (int)w->passes.size(), N = w->N;
    // sync blocks 0, 1 belong to the primal sweeps (their status is checked with this call's when both ran unchecked)
    if (!(xo && xo->pro_done)) {
        rc = x_sync_reset(ctx, X.sync + 2, 2 * np, 2);
        if (rc) return rc;
    }
    if (!skip_back) hipLaunchKernelGGL(k_tan_in, dim3((unsigned)((P * N + 255) / 256)), dim3(256), 0, s, w->dxhh, c.n_hh, (int)P, N, w->dxr, w->dxw, w->dxt);
    if (!skip_back) hipLaunchKernelGGL(k_xrho, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, ctx->d_xhh, c.n_hh, (int)P, X.rho);     // (the primal may have been recorded by the launches)
    if (!X.src_valid && !skip_back) {      // once per recorded primal: which members each member's gathers read, period by period
        hipLaunchKernelGGL(k_xsrc_back, dim3((unsigned)P, X.Sact), dim3(256), 0, s, c, ctx->R, X.Sact, X.srcB);
        X.src_valid = true;
    }
    x_ensure_rng(ctx);
    const bool neigh = X.neigh;
    const dim3 grd(X.grid);
    // + one wave that only runs the group barrier's poll, where the block has room (dev knob HANK_XSYNCWAVE=0: wave 0 polls)
    const bool fits = 64 * (c.n_e + 1) <= X.maxt && X.syncwave;
    const dim3 blk(fits ? 64 * (c.n_e + 1) : 64 * c.n_e), blkF = blk;
    XTanBackArgs ab{};
    ab.c = c; ab.R = ctx->R; ab.rho = X.rho; ab.xhh = ctx->d_xhh; ab.dxr = w->dxr; ab.dxw = w->dxw; ab.dxt = w->dxt; ab.Ntot = N; ab.st_ds = X.st_ds;
    ab.src = neigh ? X.srcB : nullptr;
    ab.stall = X.fault == 3 ? 1 : 0;
    XSweepFwdArgs fa{};
    fa.c = c; fa.R = ctx->R; fa.st = X.st_dD; fa.daggpart = w->daggpart; fa.src = X.srcF; fa.units = X.unitsF; fa.overflow = X.unit_overflow; fa.all_members = neigh ? 0 : 1;
    HIPC(ctx, hipEventRecord(ctx->ev[3], s));
    for (int p = 0; p < np && !skip_back; p++) {
        const XPass &ps = w->passes[p];
        ab.n0 = ps.n0; ab.N = ps.N; ab.groups = ps.groups; ab.sy = X.sync + 2 + 2 * p; ab.dpol = w->dpol + ps.dpol_off;
        const size_t lds = x_lds_tan_back(c, ps.D);
        if (X.maxt == 768) x_launch_tan_back<768>(ps.D, grd, blk, lds, s, ab);
        else x_launch_tan_back<1024>(ps.D, grd, blk, lds, s, ab);
    }
    HIPC(ctx, hipEventRecord(ctx->ev[4], s));
    HIPC(ctx, hipEventRecord(ctx->ev[7], s));
    const int nb = X.Sact;                  // (one row of partials per member: the sync wave sums a member's columns)
    for (int p = 0; p < np; p++) {
        const XPass &ps = w->passes[p];
        fa.sy = X.sync + 2 + 2 * p + 1; fa.groups = ps.groups; fa.dpol = w->dpol + ps.dpol_off;
        const bool v = val && p == 0;
        if (v) { fa.D0 = ctx->d_ss_D; fa.Dvirt = X.Dvirt; fa.aggpart = X.aggpart; }
        x_launch_fwd(X, ps.D, v, grd, blkF, x_lds_fwd(c, ps.D + (v ? 1 : 0)), s, fa);
        const int W = XG * ps.D;
        if (v && np == 1 && xo) {               // the one-pass Dual pass: everything behind the sweep in ONE launch (k_xdual_epilogue)
            HIPC(ctx, hipEventRecord(ctx->ev[5], s));
            const int per = 2 * W + 2 + c.n_e;
            hipLaunchKernelGGL(k_xdual_epilogue, dim3((unsigned)((P * per + 255) / 256)), dim3(256), 0, s, c, X.aggpart, w->daggpart, X.Sact, W, ps.n0, ps.N, N,
                               ctx->d_agg_rm, ctx->d_agg, w->dagg_pass, w->dagg_cm, ctx->R.Dseq, X.Dvirt, xo->agg, xo->dagg);
            xo->done = true;
            break;
        }
        if (v) {
            hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)P, 1), dim3(256), 0, s, X.aggpart, X.Sact, 2, ctx->d_agg_rm);
            hipLaunchKernelGGL(k_tan_out, dim3((unsigned)((2 * P + 255) / 256)), dim3(256), 0, s, ctx->d_agg_rm, (int)P, 2, ctx->d_agg);
            hipLaunchKernelGGL(k_xfix_D, dim3((unsigned)((P * c.n_e + 255) / 256)), dim3(256), 0, s, c, ctx->R.Dseq, X.Dvirt, X.Sact);
        }
        if (p == np - 1) HIPC(ctx, hipEventRecord(ctx->ev[5], s));
        hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)P, (2 * W + 63) / 64), dim3(256), 0, s, w->daggpart, nb, 2 * W, w->dagg_pass);
        hipLaunchKernelGGL(k_xout, dim3((unsigned)((P * ps.N + 255) / 256)), dim3(256), 0, s, w->dagg_pass, (int)P, 2 * W, 0, ps.n0, ps.N, w->dagg_cm);
        hipLaunchKernelGGL(k_xout, dim3((unsigned)((P * ps.N + 255) / 256)), dim3(256), 0, s, w->dagg_pass, (int)P, 2 * W, W, ps.n0, ps.N, w->dagg_cm + P * (size_t)N);
    }
    HIPC(ctx, hipGetLastError());
    rc = x_serialize_end(ctx);
    if (rc) return rc;
    ctx->stats[0] += skip_back ? np : 2 * np;
    X.last_passes = 1 + np;
    ctx->launches[2] = ctx->launches[3] = np;
    ctx->ev_valid[2] = ctx->ev_valid[3] = true;
    ctx->ev_valid[4] = ctx->ev_valid[5] = false;
    for (XTan &t : X.tans) t.valid = false;
    w->valid = true;
    ctx->xcur = w;
    ctx->last_tan = 1;
    for (TanWork &t : ctx->tws) t.valid = false; w_invalidate(ctx);
    return HANK_OK;
}

// after a synchronisation: did every sweep of the last call form its groups and meet all its barriers?
static int x_status(hank_ctx *ctx) {
    XWork &X = ctx->xw;
    if (!X.ready || X.last_passes == 0) return HANK_OK;
    int uo = 0;
    HIPC(ctx, hipMemcpy(&uo, X.unit_overflow, sizeof(int), hipMemcpyDeviceToHost));
    if (uo) {
        HIPC(ctx, hipMemsetAsync(X.unit_overflow, 0, sizeof(int), ctx->stream));
        ctx->primal_done = false;
        X.rng_valid = false;
        for (XTan &t : X.tans) t.valid = false;
        return fail(ctx, HANK_ERR_SWEEP, "persistent forward sweep: a member's walk over its sources needs more than %d work units (a savings policy this flat is served by the per-period launches)", XUCAP);
    }
    std::vector<XSync> h(2 * (size_t)X.last_passes);
    HIPC(ctx, hipMemcpy(h.data(), X.sync, sizeof(XSync) * h.size(), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < h.size(); k++)
        if (h[k].status[0] != 0) {
            ctx->primal_done = false;
            for (XTan &t : X.tans) t.valid = false;
            char waited[64] = "";
            if (h[k].status[0] == XERR_TIMEOUT) snprintf(waited, sizeof(waited), " after %.1f ms", h[k].status[2] / 1000.0);
            return fail(ctx, HANK_ERR_SWEEP, "persistent %s sweep %zu (0 = primal, then one per tangent pass): %s%s on XCD %u (workgroups per XCD: %u %u %u %u %u %u %u %u)",
                        (k & 1) ? "forward" : "backward", k / 2, h[k].status[0] == XERR_PLACEMENT ? "a group is short of members" : "a wait timed out", waited,
                        h[k].status[1], h[k].ticket[0][0], h[k].ticket[1][0], h[k].ticket[2][0], h[k].ticket[3][0], h[k].ticket[4][0],
                        h[k].ticket[5][0], h[k].ticket[6][0], h[k].ticket[7][0]);
        }
    return HANK_OK;
}

// ================================ on-chip wide sweeps (hank_wide.h): host side ====================
// value-function families share the kernels (n_hh and the record diet are run-time / template switches); the number of
// productivity states is a template parameter (the state of a grid row is a register array)
#define HANK_WIDE_NE_LIST(X) X(2) X(3) X(4) X(5) X(7) X(11)
static bool w_ne_instantiated(int ne) {
#define X(NEV) if (ne == NEV) return true;
    HANK_WIDE_NE_LIST(X)
#undef X
    return false;
}
// geometry of the wide kernels (hank_wide.h): rows per thread 4 (workgroups of <= 512 threads) or 2 (<= 1024); which columns of the
// state stay in registers follows from the register budget of each: all of them, except the forward kernel of the 2-row geometry
struct WGeom { int R, maxt, kreg_b, kreg_f; };
static WGeom w_geom(const hank_ctx *ctx) { return ctx->wide_r == 2 ? WGeom{2, 1024, 16, 7} : WGeom{4, 512, 16, 16}; }
static int w_threads(const hank_ctx *ctx) { const WGeom g = w_geom(ctx); return std::max(64, ((ctx->c.n_a + g.R - 1) / g.R + 63) / 64 * 64); }
static size_t w_lds(const hank_ctx *ctx) { const WGeom g = w_geom(ctx); return std::max(wide_lds_back(ctx->c, g.kreg_b), wide_lds_fwd(ctx->c, g.kreg_f)); }
static bool w_supported(const hank_ctx *ctx) {
    const Consts &c = ctx->c;
    return w_ne_instantiated(c.n_e) && c.n_a <= WIDE_CS && w_lds(ctx) <= ctx->lds_max &&
           ctx->rec_bytes < 0x7fffffffull && (size_t)c.G * sizeof(double) < 0x7fffffffull;
}
// auto: a workgroup (= a direction) per CU and round; a round costs what ~80 directions cost the per-period launches (measured at
// 2000x11, T=300, profiles/r05b_wide_crossover.log: 12.9 ms per round with the primal, 10.3 at a recorded primal, whatever its
// fill; the launches 13.1 ms at N = 80, 14.0 at 128 and 23.5 from 144 on), so the batch goes to the wide sweeps when its last round
// holds at least that many (before the L2 warming of k_wide_back a round cost 13.9 ms at a recorded primal and the crossover was 152).
// wide_min (HANK_WIDE_MIN) is that fill, in directions.
static bool use_wide(const hank_ctx *ctx, int N) {
    if (ctx->wide_mode == 2) return true;
    if (ctx->wide_mode != 1) return false;
    const int rounds = (N + ctx->num_cus - 1) / ctx->num_cus;
    return (long long)N * 256 >= (long long)rounds * ctx->wide_min * ctx->num_cus;      // (wide_min is quoted for a 256-CU chip)
}

template <int NE, int R, int MAXT, int KB, int KF>
static int w_launch_geom(hank_ctx *ctx, bool fwd, int N, const WideArgs &a) {
    const Consts &c = ctx->c;
    WMat<NE> M;
    for (int k = 0; k < NE; k++)
        for (int e = 0; e < NE; e++) M.m[k * NE + e] = fwd ? ctx->h_Pi[k + NE * e] : ctx->h_Pi[e + NE * k];
    for (int e = 0; e < NE; e++) M.z[e] = ctx->h_z[e];
    const dim3 grd((unsigned)N), blk((unsigned)w_threads(ctx));
    if (fwd) {
        const size_t lds = wide_lds_fwd(c, KF);
        HIPC(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wide_fwd<NE, R, MAXT, KF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_wide_fwd<NE, R, MAXT, KF>), grd, blk, lds, ctx->stream, a, M);
    } else {
        const size_t lds = wide_lds_back(c, KB);
        if (c.diet) {
            HIPC(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wide_back<NE, R, MAXT, true, KB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_wide_back<NE, R, MAXT, true, KB>), grd, blk, lds, ctx->stream, a, M);
        } else {
            HIPC(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wide_back<NE, R, MAXT, false, KB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_wide_back<NE, R, MAXT, false, KB>), grd, blk, lds, ctx->stream, a, M);
        }
    }
    return HANK_OK;
}
static int w_launch(hank_ctx *ctx, bool fwd, int N, const WideArgs &a) {
#define X(NEV) if (ctx->c.n_e == NEV) return ctx->wide_r == 2 ? w_launch_geom<NEV, 2, 1024, 16, 7>(ctx, fwd, N, a) : w_launch_geom<NEV, 4, 512, 16, 16>(ctx, fwd, N, a);
    HANK_WIDE_NE_LIST(X)
#undef X
    return fail(ctx, HANK_ERR_BAD_ARG, "on-chip wide sweeps: n_e=%d is not instantiated", ctx->c.n_e);
}

static int w_ensure_tan(hank_ctx *ctx, int N, bool staging, WTan **out) {
    for (auto it = ctx->wtans.begin(); it != ctx->wtans.end(); ++it)
        if (it->N == N) {
            ctx->wtans.splice(ctx->wtans.begin(), ctx->wtans, it);
            WTan &w = ctx->wtans.front();
            if (staging && !w.dxhh) HIPC(ctx, dmalloc(&w.dxhh, (size_t)ctx->c.n_hh * ctx->c.P * N));
            *out = &w;
            return HANK_OK;
        }
    const char *ce = getenv("HANK_TAN_CACHE");
    const size_t keep = ce ? (size_t)atoi(ce) : 3;
    while (ctx->wtans.size() >= (keep ? keep : 1)) {       // evict the least recently used — after the stream has drained
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->wcur == &ctx->wtans.back()) ctx->wcur = nullptr;
        w_free_tan(ctx->wtans.back());
        ctx->wtans.pop_back();
    }
    ctx->wtans.emplace_front();
    WTan &w = ctx->wtans.front();
    const Consts &c = ctx->c;
    const size_t P = c.P, G = c.G;
    w.N = N;
    ctx->stats[1]++;
    auto alloc = [&]() -> int {
        if (staging) HIPC(ctx, dmalloc(&w.dxhh, (size_t)c.n_hh * P * N));
        HIPC(ctx, dmalloc(&w.dpol, P * (size_t)N * G + 2));       // (+ 2: the last 16-byte load of an odd-sized grid reads 8 bytes past its row)
        HIPC(ctx, dmalloc(&w.dagg_cm, 2 * P * (size_t)N));            // (P, 2 N) column-major: both aggregates
        return HANK_OK;
    };
    const int rc = alloc();
    if (rc) { w_free_tan(w); ctx->wtans.pop_front(); (void)hipGetLastError(); return rc; }
    *out = &w;
    return HANK_OK;
}

// the N partials at the recorded primal (either schedule may have recorded it): two launches, one workgroup per direction
static int w_run_tangent(hank_ctx *ctx, WTan *w, const double *d_dxhh) {
    const Consts &c = ctx->c;
    hipStream_t s = ctx->stream;
    int rc = x_serialize_begin(ctx);        // (a persistent sweep of another context must not find the chip half full of these workgroups)
    if (rc) return rc;
    if (!ctx->d_ibw) HIPC(ctx, dmalloc(&ctx->d_ibw, (size_t)c.P * c.G + 4));      // (before the kernel arguments are filled in)
    WideArgs a{};
    a.c = c; a.R = ctx->R; a.xhh = ctx->d_xhh; a.dxhh = d_dxhh; a.Ntot = w->N; a.n0 = 0; a.dpol = w->dpol; a.dagg = w->dagg_cm;
    a.rec = ctx->rec_slab;
    auto off = [&](const void *p) { return (unsigned)((const char *)p - ctx->rec_slab); };
    const Record &R = ctx->R;
    a.o_s = off(R.s); a.o_kc = off(R.kc); a.o_u = off(R.u); a.o_v = off(R.v); a.ibw = ctx->d_ibw;
    a.nrank = std::max(1, std::min(ctx->num_cus / 8, w->N / 8));       // workgroups per XCD of the backward launch (hank_wide.h: L2 warming)
    a.o_lwg = off(R.lwg); a.o_start = off(R.start); a.o_D = off(R.Dseq); a.o_pol = off(R.pol);
    if (!ctx->wprep_valid) {                // once per recorded primal
        const size_t npt = (size_t)c.P * c.G;
        hipLaunchKernelGGL(k_wide_prep, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, s, R.ib, R.A, R.B, npt, ctx->d_ibw);
        ctx->wprep_valid = true;
    }
    HIPC(ctx, hipEventRecord(ctx->ev[3], s));
    rc = w_launch(ctx, false, w->N, a);
    if (rc) return rc;
    HIPC(ctx, hipEventRecord(ctx->ev[4], s));
    HIPC(ctx, join_side(ctx));              // the forward sweep reads D_t and the {w, ig D} record of the primal's forward sweep
    HIPC(ctx, hipEventRecord(ctx->ev[7], s));
    rc = w_launch(ctx, true, w->N, a);
    if (rc) return rc;
    HIPC(ctx, hipEventRecord(ctx->ev[5], s));
    HIPC(ctx, hipGetLastError());
    rc = x_serialize_end(ctx);
    if (rc) return rc;
    ctx->launches[2] = ctx->launches[3] = 1;
    ctx->ev_valid[2] = ctx->ev_valid[3] = true;
    ctx->ev_valid[4] = ctx->ev_valid[5] = false;
    for (TanWork &t : ctx->tws) t.valid = false;
    for (XTan &t : ctx->xw.tans) t.valid = false;
    w_invalidate(ctx);
    w->valid = true;
    ctx->wcur = w;
    ctx->last_tan = 2;
    ctx->stats[0] += 2;
    return HANK_OK;
}
static int w_jvp(hank_ctx *ctx, const double *dxhh, hipMemcpyKind kind, int N, double *d_dagg_out) {
    WTan *w = nullptr;
    const bool staging = kind != hipMemcpyDeviceToDevice;
    int rc = w_ensure_tan(ctx, N, staging, &w);
    if (rc) return rc;
    const size_t P = ctx->c.P;
    if (staging) HIPC(ctx, hipMemcpyAsync(w->dxhh, dxhh, sizeof(double) * ctx->c.n_hh * P * N, kind, ctx->stream));
    rc = w_run_tangent(ctx, w, staging ? w->dxhh : dxhh);
    if (rc) return rc;
    if (d_dagg_out) HIPC(ctx, hipMemcpyAsync(d_dagg_out, w->dagg_cm, sizeof(double) * P * N, hipMemcpyDeviceToDevice, ctx->stream));
    return HANK_OK;
}

// ================================ C ABI =========================================================
extern "C" {

const char *hank_last_error(const hank_ctx *ctx) { return ctx ? ctx->errmsg : "null context"; }
int hank_device_available(void) {
    int ndev = 0, dev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
int hank_n_hh(const hank_ctx *ctx) { return ctx ? ctx->c.n_hh : 0; }

int hank_create(const hank_model *m, hank_ctx **out) {
    int dev = 0;
    if (out) *out = nullptr;
    if (hipGetDevice(&dev) != hipSuccess) return HANK_ERR_NO_DEVICE;
    return hank_create_on(m, dev, out);
}

int hank_create_on(const hank_model *m, int32_t device, hank_ctx **out) {
    if (!m || !out) return HANK_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return HANK_ERR_NO_DEVICE;
    hank_ctx *ctx = new (std::nothrow) hank_ctx();
    if (!ctx) return HANK_ERR_NOMEM;
    *out = ctx;  // returned even on failure so the caller can read hank_last_error, then destroy
    if (device < 0 || device >= ndev) return fail(ctx, HANK_ERR_BAD_ARG, "device %d: the process sees %d HIP device(s)", device, ndev);
    ctx->device = device;
    ENTER(ctx);
    if (m->n_a < 2 || m->n_e < 1 || m->n_e > 16 || m->T < 2)
        return fail(ctx, HANK_ERR_BAD_ARG, "bad shape: n_a=%d (>=2), n_e=%d (1..16), T=%d (>=2)", m->n_a, m->n_e, m->T);
    if (m->value_fn_id != HANK_VF_KRUSELL_SMITH && m->value_fn_id != HANK_VF_ONE_ASSET_HANK)
        return fail(ctx, HANK_ERR_BAD_ARG, "unknown value function id %d", m->value_fn_id);
    if (!m->a_grid || !m->z_grid || !m->Pi) return fail(ctx, HANK_ERR_BAD_ARG, "null grid pointer");
    for (int i = 1; i < m->n_a; i++)
        if (!(m->a_grid[i] > m->a_grid[i - 1])) return fail(ctx, HANK_ERR_BAD_ARG, "wealth grid must be strictly increasing (index %d)", i + 1);
    hipDeviceProp_t prop;
    HIPC(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ctx, HANK_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", ctx->device, prop.gcnArchName);
    Consts &c = ctx->c;
    c.n_a = m->n_a; c.n_e = m->n_e; c.G = m->n_a * m->n_e; c.P = m->T - 1;
    c.beta = m->beta; c.gamma = m->gamma; c.bc = m->borrow_cons;
    c.n_hh = m->value_fn_id == HANK_VF_ONE_ASSET_HANK ? 3 : 2;
    {   // record diet (hank_kernels.h:diet_kc): the tangent sweeps rebuild kc and v where CRRA's powers are products and a root
        const char *rd = getenv("HANK_RECORD_DIET");      // dev knob (A-B): 0 = read kc and v from the record
        c.diet = ((c.gamma == 1.0 || c.gamma == 2.0) && !(rd && atoi(rd) == 0)) ? 1 : 0;
    }
    ctx->T = m->T;
    const size_t P = c.P, G = c.G;
    if ((2 * (size_t)c.n_a + 2) * sizeof(int) > 150 * 1024) return fail(ctx, HANK_ERR_BAD_ARG, "n_a=%d too large for the LDS-staged lottery", c.n_a);
    HIPC(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIPC(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    HIPC(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPC(ctx, hipEventCreate(&ctx->ev_side));
    for (int k = 0; k < 16; k++) HIPC(ctx, hipEventCreate(&ctx->ev[k]));
    HIPC(ctx, dmalloc(&ctx->d_a, c.n_a));
    HIPC(ctx, dmalloc(&ctx->d_z, c.n_e));
    HIPC(ctx, dmalloc(&ctx->d_Pi, (size_t)c.n_e * c.n_e));
    HIPC(ctx, hipMemcpy(ctx->d_a, m->a_grid, sizeof(double) * c.n_a, hipMemcpyHostToDevice));
    HIPC(ctx, hipMemcpy(ctx->d_z, m->z_grid, sizeof(double) * c.n_e, hipMemcpyHostToDevice));
    HIPC(ctx, hipMemcpy(ctx->d_Pi, m->Pi, sizeof(double) * c.n_e * c.n_e, hipMemcpyHostToDevice));
    c.a = ctx->d_a; c.z = ctx->d_z; c.Pi = ctx->d_Pi;
    ctx->h_Pi.assign(m->Pi, m->Pi + (size_t)c.n_e * c.n_e);
    ctx->h_z.assign(m->z_grid, m->z_grid + c.n_e);
    ctx->lds_max = prop.sharedMemPerBlock;
    ctx->num_cus = std::max(1, prop.multiProcessorCount);
    Record &R = ctx->R;
    {   // the record is ONE allocation: the on-chip wide sweeps reach every array through one buffer descriptor (hank_wide.h)
        size_t off = 0;
        auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t d8 = P * G * sizeof(double);
        const size_t o_s = carve(d8), o_kc = carve(d8), o_A = carve(d8), o_B = carve(d8), o_u = carve(d8), o_v = carve(d8), o_pol = carve(d8),
                     o_lw = carve(d8), o_ig = carve(d8), o_D = carve((P + 1) * G * sizeof(double)), o_ib = carve(P * G * sizeof(int)),
                     o_lo = carve(P * G * sizeof(int)), o_st = carve(P * (size_t)c.n_e * (c.n_a + 1) * sizeof(int)),
                     o_clo = carve(P * (size_t)c.n_e * sizeof(int)), o_lwg = carve(P * G * sizeof(double2)), o_seg = carve(P * G * sizeof(int4));
        HIPC(ctx, hipMalloc((void **)&ctx->rec_slab, off));
        ctx->rec_bytes = off;
        char *b = ctx->rec_slab;
        R.s = (double *)(b + o_s); R.kc = (double *)(b + o_kc); R.A = (double *)(b + o_A); R.B = (double *)(b + o_B);
        R.u = (double *)(b + o_u); R.v = (double *)(b + o_v); R.pol = (double *)(b + o_pol); R.lw = (double *)(b + o_lw);
        R.ig = (double *)(b + o_ig); R.Dseq = (double *)(b + o_D); R.ib = (int *)(b + o_ib); R.lo = (int *)(b + o_lo);
        R.start = (int *)(b + o_st); R.clo = (int *)(b + o_clo); R.lwg = (double2 *)(b + o_lwg); R.seg = (int4 *)(b + o_seg);
    }
    HIPC(ctx, dmalloc(&ctx->d_ss_value, G));
    ctx->d_ss_D = R.Dseq;
    ctx->nbp = (c.n_a + RBP - 1) / RBP;
    HIPC(ctx, dmalloc(&ctx->d_xhh, (size_t)c.n_hh * P));
    HIPC(ctx, dmalloc(&ctx->d_agg, 2 * P));
    HIPC(ctx, dmalloc(&ctx->d_agg_rm, 2 * P));
    HIPC(ctx, dmalloc(&ctx->d_zd, 2 * P));
    HIPC(ctx, dmalloc(&ctx->d_aggpart, 2 * P * (size_t)ctx->nbp));
    HIPC(ctx, dmalloc(&ctx->d_err, 4));
    HIPC(ctx, hipMemsetAsync(ctx->d_err, 0, 4 * sizeof(int), ctx->stream));
    HIPC(ctx, hipEventCreateWithFlags(&ctx->ev_stream, hipEventDisableTiming));
    // schedule (measured on MI355X): "auto" wherever the grid fits one 63-row slab per CU of an XCD —
    // the Float64 sweeps alone (hank_primal) and narrow tangent batches at a recorded primal (hank_jvp, N <= 64) run as
    // XCD-local persistent sweeps, the dual pass (hank_primal_jvp) and wide batches as per-period launches; both
    // read and write the same record. HANK_SCHEDULE=launch|xcd forces one implementation for everything (A-B, tests).
    const char *se = getenv("HANK_SCHEDULE");
    ctx->schedule = x_supported(ctx, prop.multiProcessorCount, prop.sharedMemPerBlock) ? 2 : 0;
    if (se && strcmp(se, "launch") == 0) ctx->schedule = 0;
    if (se && strcmp(se, "xcd") == 0) {
        if (ctx->schedule == 0)
            return fail(ctx, HANK_ERR_BAD_ARG, "HANK_SCHEDULE=xcd: n_a=%d needs %d workgroups per XCD (the device has %d) and %zu bytes of LDS per workgroup (it has %zu)", c.n_a,
                        (c.n_a + XRW - 1) / XRW, prop.multiProcessorCount / XG, std::max(x_lds_primal_back(c), x_lds_fwd(c, 1)), (size_t)prop.sharedMemPerBlock);
        ctx->schedule = 1;
        ctx->forced_xcd = true;
    }
    // on-chip wide sweeps (hank_wide.h): "auto" sends tangent batches of at least wide_min directions to them (measured crossover,
    // DESIGN.md section 4); a forced schedule (launch | xcd) keeps its one implementation; HANK_SCHEDULE=wide sends every batch (tests)
    if (const char *wr = getenv("HANK_WIDE_R")) ctx->wide_r = atoi(wr) == 4 ? 4 : 2;
    ctx->wide_mode = (w_supported(ctx) && !(se && (strcmp(se, "launch") == 0 || strcmp(se, "xcd") == 0))) ? 1 : 0;
    if (se && strcmp(se, "wide") == 0) {
        if (!w_supported(ctx))
            return fail(ctx, HANK_ERR_BAD_ARG, "HANK_SCHEDULE=wide: %dx%d, T=%d does not fit the on-chip wide sweeps (n_e instantiated: 2,3,4,5,7,11; n_a <= %d; %zu bytes of LDS per workgroup, the device has %zu)",
                        c.n_a, c.n_e, ctx->T, WIDE_CS, w_lds(ctx), ctx->lds_max);
        ctx->wide_mode = 2;
    }
    if (const char *wm = getenv("HANK_WIDE_MIN")) ctx->wide_min = std::max(1, atoi(wm));
    if (const char *xm = getenv("HANK_XJVP_MAX")) ctx->xjvp_max = atoi(xm);
    if (const char *pm = getenv("HANK_PRIMAL_MEMO")) ctx->memo_on = atoi(pm) != 0;
    if (const char *xd = getenv("HANK_XDUAL_BACK")) ctx->xdual_back = atoi(xd) != 0;
    int rc = HANK_OK;
    if (ctx->schedule == 0) rc = build_primal_graphs(ctx);
    else rc = x_setup(ctx);
    if (rc) return rc;
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

int hank_destroy(hank_ctx *ctx) {
    if (!ctx) return HANK_OK;
    ENTER(ctx);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    for (TanWork &t : ctx->tws) free_tanwork(t);
    ctx->tws.clear();
    ctx->tw = nullptr;
    x_free(ctx);
    (void)hipFree(ctx->fn.dpT); (void)hipFree(ctx->fn.iota); (void)hipFree(ctx->fn.E); (void)hipFree(ctx->fn.Cp); (void)hipFree(ctx->fn.F); (void)hipFree(ctx->fn.Dv);
    if (ctx->ev_stream) (void)hipEventDestroy(ctx->ev_stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_side) (void)hipEventDestroy(ctx->ev_side);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->g_pback) (void)hipGraphExecDestroy(ctx->g_pback);
    if (ctx->g_pfwd) (void)hipGraphExecDestroy(ctx->g_pfwd);
    for (WTan &t : ctx->wtans) w_free_tan(t);
    ctx->wtans.clear();
    ctx->wcur = nullptr;
    (void)hipFree(ctx->rec_slab); (void)hipFree(ctx->d_ibw);
    (void)hipFree(ctx->d_a); (void)hipFree(ctx->d_z); (void)hipFree(ctx->d_Pi); (void)hipFree(ctx->d_ss_value);
    (void)hipFree(ctx->d_xhh); (void)hipFree(ctx->d_agg); (void)hipFree(ctx->d_agg_rm); (void)hipFree(ctx->d_zd); (void)hipFree(ctx->d_aggpart); (void)hipFree(ctx->d_err);
    for (int k = 0; k < 16; k++)
        if (ctx->ev[k]) (void)hipEventDestroy(ctx->ev[k]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return HANK_OK;
}

int hank_set_stream(hank_ctx *ctx, void *hip_stream) {
    if (!ctx) return HANK_ERR_BAD_ARG;
    ENTER(ctx);
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next != ctx->stream) {      // what is still queued on the old stream (and on the side stream) happens before the new one's work
        HIPC(ctx, join_side(ctx));
        HIPC(ctx, hipEventRecord(ctx->ev_stream, ctx->stream));
        HIPC(ctx, hipStreamWaitEvent(next, ctx->ev_stream, 0));
    }
    ctx->stream = next;
    return HANK_OK;
}

int hank_sync(hank_ctx *ctx) {
    if (!ctx) return HANK_ERR_BAD_ARG;
    ENTER(ctx);
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return HANK_OK;
}

int hank_set_boundary(hank_ctx *ctx, const double *ss_end_value, const double *ss_init_D) {
    ENTER(ctx);
    if (!ctx || !ss_end_value || !ss_init_D) return fail(ctx, HANK_ERR_BAD_ARG, "null boundary pointer");
    const size_t G = ctx->c.G;
    // the same boundary again (the reference's closures pass ss_end / ss_initial on every call, NewtonRaphson.jl:78-79): nothing
    // to do, and the recorded primal stays valid
    if (ctx->boundary_set && ctx->h_ss_value.size() == G && ctx->h_ss_D.size() == G &&
        memcmp(ctx->h_ss_value.data(), ss_end_value, sizeof(double) * G) == 0 && memcmp(ctx->h_ss_D.data(), ss_init_D, sizeof(double) * G) == 0) {
        ctx->errmsg[0] = 0;
        return HANK_OK;
    }
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipMemcpyAsync(ctx->d_ss_value, ss_end_value, sizeof(double) * G, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipMemcpyAsync(ctx->d_ss_D, ss_init_D, sizeof(double) * G, hipMemcpyHostToDevice, ctx->stream));
    {   // the productivity marginal of D_t does not depend on the policies (the lottery moves mass within a column, the exogenous
        // step mixes the columns: ForwardIteration.jl:95-99): m_t = Pi' m_{t-1} from D_0's
        const int ne = ctx->c.n_e, na = ctx->c.n_a, P = ctx->c.P;
        std::vector<double> m(ne, 0.0), m2(ne), zd(2 * (size_t)P);
        for (int e = 0; e < ne; e++) for (int i = 0; i < na; i++) m[e] += ss_init_D[(size_t)e * na + i];
        for (int t = 0; t < P; t++) {
            for (int e2 = 0; e2 < ne; e2++) { double v = 0.0; for (int e = 0; e < ne; e++) v += ctx->h_Pi[e + (size_t)ne * e2] * m[e]; m2[e2] = v; }
            m.swap(m2);
            double z = 0.0, one = 0.0;
            for (int e = 0; e < ne; e++) { z += ctx->h_z[e] * m[e]; one += m[e]; }
            zd[t] = z; zd[(size_t)P + t] = one;
        }
        HIPC(ctx, hipMemcpyAsync(ctx->d_zd, zd.data(), sizeof(double) * 2 * P, hipMemcpyHostToDevice, ctx->stream));   // (synchronised below: zd may go)
    }
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    ctx->boundary_set = true;
    ctx->primal_done = false;
    ctx->memo_valid = false;
    ctx->stationary = false;
    ctx->h_ss_value.assign(ss_end_value, ss_end_value + G);
    ctx->h_ss_D.assign(ss_init_D, ss_init_D + G);
    for (TanWork &t : ctx->tws) t.valid = false; w_invalidate(ctx);
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

// which x the record belongs to: the host-pointer entries know it (and whether it is a constant path), the device-pointer
// entries do not
static void note_primal_x(hank_ctx *ctx, const double *xhh) {
    const size_t n = (size_t)ctx->c.n_hh * ctx->c.P, nh = ctx->c.n_hh;
    if (!xhh) { ctx->memo_valid = false; ctx->stationary = false; return; }
    ctx->memo_xhh.assign(xhh, xhh + n);
    ctx->memo_valid = true;
    bool constant = true;
    for (size_t k = nh; k < n && constant; k++) constant = xhh[k] == xhh[k - nh];
    ctx->stationary = constant;      // (hank_fake_news also compares the first and the last recorded policy on the device)
}

static int run_primal(hank_ctx *ctx, double *d_agg_out) {
    HIPC(ctx, join_side(ctx));     // a previous forward sweep still reads the record this one overwrites
    HIPC(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    HIPC(ctx, hipGraphLaunch(ctx->g_pback, ctx->stream));
    HIPC(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    // fork: the distribution sweep goes to the side stream; the main stream is free for the tangent
    // backward sweep and joins (join_side) before anything that needs D_t or the aggregates
    HIPC(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPC(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    HIPC(ctx, hipEventRecord(ctx->ev[6], ctx->side_stream));
    HIPC(ctx, hipGraphLaunch(ctx->g_pfwd, ctx->side_stream));
    HIPC(ctx, hipEventRecord(ctx->ev[2], ctx->side_stream));
    if (d_agg_out)
        HIPC(ctx, hipMemcpyAsync(d_agg_out, ctx->d_agg, sizeof(double) * ctx->c.P, hipMemcpyDeviceToDevice, ctx->side_stream));
    HIPC(ctx, hipEventRecord(ctx->ev_side, ctx->side_stream));
    ctx->side_pending = true;
    ctx->ev_valid[0] = ctx->ev_valid[1] = true;
    ctx->ev_valid[4] = ctx->ev_valid[5] = false;
    ctx->primal_done = true; ctx->seg_valid = true; w_new_primal(ctx);
    ctx->xw.src_valid = false; ctx->xw.rng_valid = false;
    for (TanWork &t : ctx->tws) t.valid = false; w_invalidate(ctx);
    return HANK_OK;
}

// ---- xcd schedule: entry-point bodies --------------------------------------------------------------
static int x_primal(hank_ctx *ctx, const double *xhh, hipMemcpyKind kind, double *d_agg_out) {
    int rc = x_setup(ctx);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(ctx->d_xhh, xhh, sizeof(double) * ctx->c.n_hh * ctx->c.P, kind, ctx->stream));
    rc = x_run_primal(ctx);
    if (rc) return rc;
    if (d_agg_out) HIPC(ctx, hipMemcpyAsync(d_agg_out, ctx->d_agg, sizeof(double) * ctx->c.P, hipMemcpyDeviceToDevice, ctx->stream));
    return HANK_OK;
}
// value and N partials; xhh == nullptr keeps the recorded primal (hank_jvp): the partials are linear recurrences at
// that record, so a y-iteration pays the Float64 sweeps once (NewtonRaphson.jl:91) and each JVP (:95) only the tangent sweeps
static bool x_dual_back_fits(const hank_ctx *ctx, int D);
static int x_dual(hank_ctx *ctx, const double *xhh, const double *dxhh, hipMemcpyKind kind, int N, double *d_agg_out, double *d_dagg_out) {
    int rc = x_setup(ctx);
    if (rc) return rc;
    XTan *w = nullptr;
    rc = x_ensure_tan(ctx, N, &w);
    if (rc) return rc;
    const size_t P = ctx->c.P;
    // a Dual pass of one pass (N <= 8 groups x 4): value and partials together in BOTH sweeps (k_xdual_back, k_xfwd<D, true>)
    const bool fused_back = xhh && w->passes.size() == 1 && x_dual_back_fits(ctx, w->passes[0].D);
    XOut xo;
    xo.agg = d_agg_out; xo.dagg = d_dagg_out;
    XWork &X = ctx->xw;
    if (fused_back && kind == hipMemcpyDeviceToDevice && !X.fault) {
        // the one-pass Dual pass on device-resident inputs (what bench.py times): ONE launch in front of the sweeps instead of seven
        // launches and copies (k_xdual_prologue); a fault-injection run (HANK_XFAULT) keeps the separate launches
        rc = x_serialize_begin(ctx);        // (the sync blocks about to be zeroed may belong to a sweep still in flight on another stream)
        if (rc) return rc;
        hipLaunchKernelGGL(k_xdual_prologue, dim3(64), dim3(256), 0, ctx->stream, xhh, ctx->d_xhh, dxhh, w->dxhh, ctx->c.n_hh, (int)P, N, X.rho, w->dxr, w->dxw, w->dxt,
                           reinterpret_cast<xv4u *>(X.sync), sizeof(XSync) * 4 / sizeof(xv4u), ctx->d_err);
        xo.pro_done = true;
    } else {
        HIPC(ctx, hipMemcpyAsync(w->dxhh, dxhh, sizeof(double) * ctx->c.n_hh * P * N, kind, ctx->stream));
        if (xhh) HIPC(ctx, hipMemcpyAsync(ctx->d_xhh, xhh, sizeof(double) * ctx->c.n_hh * P, kind, ctx->stream));
    }
    if (xhh) {
        rc = x_run_primal(ctx, true, fused_back ? w : nullptr, xo.pro_done);        // the distribution sweep rides on the tangents' forward sweep (value + partials)
        if (rc) return rc;
    }
    rc = x_run_tangent(ctx, w, xhh != nullptr, fused_back, &xo);
    if (rc) return rc;
    if (!xo.done) {
        if (d_agg_out) HIPC(ctx, hipMemcpyAsync(d_agg_out, ctx->d_agg, sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_dagg_out) HIPC(ctx, hipMemcpyAsync(d_dagg_out, w->dagg_cm, sizeof(double) * P * N, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return HANK_OK;
}
static bool use_x_primal(const hank_ctx *ctx) { return ctx->schedule >= 1; }
// the persistent tangent sweeps' LDS grows with the horizon too (the group's dr/dw/dtr of every period): a long horizon goes
// to the per-period launches, which take any T
static bool x_tan_fits(const hank_ctx *ctx, int N) {
    const XWork &X = ctx->xw;
    int D = 1;
    while (XG * D < N && D < X.dmax) D *= 2;
    return std::max(x_lds_tan_back(ctx->c, D), x_lds_fwd(ctx->c, D + 1)) <= (size_t)X.lds_max;
}
static bool use_x_jvp(const hank_ctx *ctx, int N) { return (ctx->schedule == 1 || (ctx->schedule == 2 && N <= ctx->xjvp_max)) && x_tan_fits(ctx, N); }
// hank_primal_jvp on the persistent sweeps. Forced schedule: always. auto: a batch of ONE pass (N <= 8 groups x 4) runs value and
// partials together in both sweeps (k_xdual_back, k_xfwd<D, true>: 4.8 ms against 5.13 for the dual-sweep launches at 2000x11,
// T=300, N=32; 3.3 against 4.2 at N=1); wider batches would be two backward sweeps at the same record: the launches keep them
static bool x_dual_back_fits(const hank_ctx *ctx, int D) {
    const XWork &X = ctx->xw;
    return ctx->xdual_back && X.maxt == 768 && 64 * (ctx->c.n_e + 1) <= X.maxt && x_lds_dual_back(ctx->c, D) <= (size_t)X.lds_max;
}
static bool use_x_fused(const hank_ctx *ctx, int N) {
    if (ctx->schedule < 1 || !x_tan_fits(ctx, N)) return false;
    if (ctx->schedule == 1) return true;
    return N <= XG * ctx->xw.dmax && x_dual_back_fits(ctx, ctx->xw.dmax);
}

// a sweep could not form its groups (or timed out): this context continues on the per-period launches
static int to_launch_schedule(hank_ctx *ctx) {
    ctx->primal_done = false;
    if (!ctx->g_pback) {                    // (a context that has only run persistent sweeps has never captured them)
        const int rc = build_primal_graphs(ctx);
        if (rc != HANK_OK) return rc;       // the schedule is left as it was: the next call reports the sweep's failure again, not a null graph
    }
    ctx->schedule = 0;
    ctx->stats[4]++;
    return HANK_OK;
}
static bool x_fallback_allowed(const hank_ctx *ctx) { return !ctx->forced_xcd; }      // a schedule forced at hank_create fails loudly instead

int hank_primal_dev(hank_ctx *ctx, const double *d_xhh, double *d_agg_out) {
    ENTER(ctx);
    if (!ctx || !d_xhh) return fail(ctx, HANK_ERR_BAD_ARG, "null pointer");
    if (!ctx->boundary_set) return fail(ctx, HANK_ERR_NOT_READY, "hank_set_boundary must be called first");
    note_primal_x(ctx, nullptr);
    ctx->stats[7]++;
    if (use_x_primal(ctx)) return x_primal(ctx, d_xhh, hipMemcpyDeviceToDevice, d_agg_out);
    const size_t P = ctx->c.P;
    HIPC(ctx, hipMemcpyAsync(ctx->d_xhh, d_xhh, sizeof(double) * ctx->c.n_hh * P, hipMemcpyDeviceToDevice, ctx->stream));
    return run_primal(ctx, d_agg_out);
}

int hank_check(hank_ctx *ctx) {
    ENTER(ctx);
    if (!ctx) return HANK_ERR_BAD_ARG;
    const int rc = fetch_device_error(ctx);
    // the asynchronous entries cannot re-run a call: the error is reported (once), and a context whose schedule was not forced
    // continues on the per-period launches, so the caller's next call succeeds
    if (rc == HANK_ERR_SWEEP && ctx->schedule >= 1 && x_fallback_allowed(ctx)) {
        char keep[sizeof(ctx->errmsg)];
        memcpy(keep, ctx->errmsg, sizeof(keep));
        const int rc2 = to_launch_schedule(ctx);
        if (rc2 != HANK_OK) return rc2;     // the launches' graphs could not be built either: THAT is what the caller has to see (the
                                            // message names it); the context stays unusable until a later call succeeds in building them
        memcpy(ctx->errmsg, keep, sizeof(keep));
    }
    return rc;
}

int hank_primal(hank_ctx *ctx, const double *xhh, double *agg_out) {
    ENTER(ctx);
    if (!ctx || !xhh) return fail(ctx, HANK_ERR_BAD_ARG, "null pointer");
    if (!ctx->boundary_set) return fail(ctx, HANK_ERR_NOT_READY, "hank_set_boundary must be called first");
    const size_t P = ctx->c.P;
    for (size_t t = 0; t < P; t++)
        if (!(1.0 + xhh[ctx->c.n_hh * t] > 0.0)) return fail(ctx, HANK_ERR_DOMAIN, "1 + r must be positive (period %zu)", t + 1);
    note_primal_x(ctx, nullptr);
    int rc = HANK_OK;
    bool done = false;
    if (use_x_primal(ctx)) {
        rc = x_primal(ctx, xhh, hipMemcpyHostToDevice, nullptr);
        if (rc) return rc;
        rc = fetch_device_error(ctx);
        if (rc == HANK_ERR_SWEEP && x_fallback_allowed(ctx)) rc = to_launch_schedule(ctx);
        else if (rc) return rc;
        else done = true;
        if (rc) return rc;
    }
    if (!done) {
        HIPC(ctx, hipMemcpyAsync(ctx->d_xhh, xhh, sizeof(double) * ctx->c.n_hh * P, hipMemcpyHostToDevice, ctx->stream));
        rc = run_primal(ctx, nullptr);
        if (rc) return rc;
        rc = fetch_device_error(ctx);
        if (rc) return rc;
    }
    if (agg_out) {
        HIPC(ctx, hipMemcpyAsync(agg_out, ctx->d_agg, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream));
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
    }
    note_primal_x(ctx, xhh);
    ctx->stats[7]++;
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

static int run_jvp(hank_ctx *ctx) {
    TanWork &w = *ctx->tw;
    int grc = ensure_graphs(ctx, w, 0);
    if (grc) return grc;
    ctx->launches[2] = ctx->c.P + 2; ctx->launches[3] = ctx->c.P + 3;
    HIPC(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    HIPC(ctx, hipGraphLaunch(w.g_back, ctx->stream));
    HIPC(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    HIPC(ctx, join_side(ctx));      // the tangent forward sweep needs D_t
    { const int src = ensure_seg(ctx); if (src) return src; }
    HIPC(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    HIPC(ctx, hipGraphLaunch(w.g_fwd, ctx->stream));
    HIPC(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->ev_valid[2] = ctx->ev_valid[3] = true;
    for (TanWork &t : ctx->tws) t.valid = false; w_invalidate(ctx);
    w.valid = true;
    ctx->last_tan = 0;
    for (XTan &t : ctx->xw.tans) t.valid = false;
    return HANK_OK;
}

int hank_jvp_dev(hank_ctx *ctx, const double *d_dxhh, int32_t N, double *d_dagg_out) {
    ENTER(ctx);
    if (!ctx || !d_dxhh || N < 1) return fail(ctx, HANK_ERR_BAD_ARG, "bad argument (N=%d)", N);
    if (!ctx->primal_done) return fail(ctx, HANK_ERR_NOT_READY, "hank_primal must be called before hank_jvp");
    if (use_wide(ctx, N)) return w_jvp(ctx, d_dxhh, hipMemcpyDeviceToDevice, N, d_dagg_out);
    if (use_x_jvp(ctx, N)) return x_dual(ctx, nullptr, d_dxhh, hipMemcpyDeviceToDevice, N, nullptr, d_dagg_out);
    int rc = ensure_tanwork(ctx, N);
    if (rc) return rc;
    const size_t P = ctx->c.P;
    HIPC(ctx, hipMemcpyAsync(ctx->tw->dxhh, d_dxhh, sizeof(double) * ctx->c.n_hh * P * N, hipMemcpyDeviceToDevice, ctx->stream));
    rc = run_jvp(ctx);
    if (rc) return rc;
    if (d_dagg_out) HIPC(ctx, hipMemcpyAsync(d_dagg_out, ctx->tw->dagg_cm, sizeof(double) * P * N, hipMemcpyDeviceToDevice, ctx->stream));
    return HANK_OK;
}

int hank_jvp(hank_ctx *ctx, const double *dxhh, int32_t N, double *dagg_out) {
    ENTER(ctx);
    if (!ctx || !dxhh || !dagg_out || N < 1) return fail(ctx, HANK_ERR_BAD_ARG, "bad argument (N=%d)", N);
    if (!ctx->primal_done) return fail(ctx, HANK_ERR_NOT_READY, "hank_primal must be called before hank_jvp");
    const size_t P = ctx->c.P;
    int rc = HANK_OK;
    if (use_wide(ctx, N)) {      // a wide batch at the recorded primal: the on-chip sweeps (no cross-workgroup waits: nothing to fall back from)
        rc = w_jvp(ctx, dxhh, hipMemcpyHostToDevice, N, nullptr);
        if (rc) return rc;
        rc = fetch_device_error(ctx);
        if (rc) return rc;
        HIPC(ctx, hipMemcpyAsync(dagg_out, ctx->wcur->dagg_cm, sizeof(double) * P * N, hipMemcpyDeviceToHost, ctx->stream));
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
        ctx->errmsg[0] = 0;
        return HANK_OK;
    }
    if (use_x_jvp(ctx, N)) {
        rc = x_dual(ctx, nullptr, dxhh, hipMemcpyHostToDevice, N, nullptr, nullptr);
        if (rc) return rc;
        rc = fetch_device_error(ctx);
        const bool fell_back = rc == HANK_ERR_SWEEP && x_fallback_allowed(ctx);
        if (fell_back) {
            rc = to_launch_schedule(ctx);
            if (rc) return rc;
            rc = run_primal(ctx, nullptr);       // re-record the primal at the current x with the launches
            if (rc) return rc;
            rc = fetch_device_error(ctx);
        }
        if (rc) return rc;
        if (!fell_back) {
            HIPC(ctx, hipMemcpyAsync(dagg_out, ctx->xcur->dagg_cm, sizeof(double) * P * N, hipMemcpyDeviceToHost, ctx->stream));
            HIPC(ctx, hipStreamSynchronize(ctx->stream));
            ctx->errmsg[0] = 0;
            return HANK_OK;
        }
    }
    rc = ensure_tanwork(ctx, N);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(ctx->tw->dxhh, dxhh, sizeof(double) * ctx->c.n_hh * P * N, hipMemcpyHostToDevice, ctx->stream));
    rc = run_jvp(ctx);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(dagg_out, ctx->tw->dagg_cm, sizeof(double) * P * N, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

static int run_fused(hank_ctx *ctx) {
    TanWork &w = *ctx->tw;
    int grc = ensure_graphs(ctx, w, 1);
    if (grc) return grc;
    ctx->launches[4] = ctx->launches[5] = ctx->c.P + 5;
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipEventRecord(ctx->ev[8], ctx->stream));
    HIPC(ctx, hipGraphLaunch(w.g_fback, ctx->stream));
    HIPC(ctx, hipEventRecord(ctx->ev[9], ctx->stream));
    HIPC(ctx, hipGraphLaunch(w.g_ffwd, ctx->stream));
    HIPC(ctx, hipEventRecord(ctx->ev[10], ctx->stream));
    ctx->ev_valid[4] = ctx->ev_valid[5] = true;
    ctx->ev_valid[0] = ctx->ev_valid[1] = ctx->ev_valid[2] = ctx->ev_valid[3] = false;
    ctx->primal_done = true; ctx->seg_valid = true; w_new_primal(ctx);
    ctx->xw.src_valid = false; ctx->xw.rng_valid = false;
    for (TanWork &t : ctx->tws) t.valid = false; w_invalidate(ctx);
    w.valid = true;
    ctx->last_tan = 0;
    for (XTan &t : ctx->xw.tans) t.valid = false;
    return HANK_OK;
}

int hank_primal_jvp_dev(hank_ctx *ctx, const double *d_xhh, const double *d_dxhh, int32_t N, double *d_agg_out,
                        double *d_dagg_out) {
    ENTER(ctx);
    if (!ctx || !d_xhh || !d_dxhh || N < 1) return fail(ctx, HANK_ERR_BAD_ARG, "bad argument (N=%d)", N);
    if (!ctx->boundary_set) return fail(ctx, HANK_ERR_NOT_READY, "hank_set_boundary must be called first");
    if (use_wide(ctx, N)) {           // a wide batch: the Float64 sweeps of the context's schedule, then the on-chip tangent sweeps
        const int rc = hank_primal_dev(ctx, d_xhh, d_agg_out);
        return rc ? rc : hank_jvp_dev(ctx, d_dxhh, N, d_dagg_out);
    }
    note_primal_x(ctx, nullptr);      // (this entry never skips work: bench.py times it)
    ctx->stats[7]++;
    if (use_x_fused(ctx, N)) return x_dual(ctx, d_xhh, d_dxhh, hipMemcpyDeviceToDevice, N, d_agg_out, d_dagg_out);
    int rc = ensure_tanwork(ctx, N);
    if (rc) return rc;
    const size_t P = ctx->c.P;
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipMemcpyAsync(ctx->d_xhh, d_xhh, sizeof(double) * ctx->c.n_hh * P, hipMemcpyDeviceToDevice, ctx->stream));
    HIPC(ctx, hipMemcpyAsync(ctx->tw->dxhh, d_dxhh, sizeof(double) * ctx->c.n_hh * P * N, hipMemcpyDeviceToDevice, ctx->stream));
    rc = run_fused(ctx);
    if (rc) return rc;
    if (d_agg_out) HIPC(ctx, hipMemcpyAsync(d_agg_out, ctx->d_agg, sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_dagg_out) HIPC(ctx, hipMemcpyAsync(d_dagg_out, ctx->tw->dagg_cm, sizeof(double) * P * N, hipMemcpyDeviceToDevice, ctx->stream));
    return HANK_OK;
}

int hank_primal_jvp(hank_ctx *ctx, const double *xhh, const double *dxhh, int32_t N, double *agg_out, double *dagg_out) {
    ENTER(ctx);
    if (!ctx || !xhh || !dxhh || !dagg_out || N < 1) return fail(ctx, HANK_ERR_BAD_ARG, "bad argument (N=%d)", N);
    if (!ctx->boundary_set) return fail(ctx, HANK_ERR_NOT_READY, "hank_set_boundary must be called first");
    const size_t P = ctx->c.P;
    for (size_t t = 0; t < P; t++)
        if (!(1.0 + xhh[ctx->c.n_hh * t] > 0.0)) return fail(ctx, HANK_ERR_DOMAIN, "1 + r must be positive (period %zu)", t + 1);
    int rc = HANK_OK;
    // The reference calls JVP(fullFunction, x, y) about 21 times per Newton step at ONE x (NewtonRaphson.jl:91-95) and its Dual
    // pass recomputes the primal every time (GeneralStructures.jl:546-547). The linearisation of the x on record is still
    // valid when the same x comes in again: only the tangent sweeps run (what hank_jvp does), the value is the recorded one.
    if (ctx->memo_on && ctx->primal_done && ctx->memo_valid && ctx->memo_xhh.size() == (size_t)ctx->c.n_hh * P &&
        memcmp(ctx->memo_xhh.data(), xhh, sizeof(double) * ctx->c.n_hh * P) == 0) {
        ctx->stats[6]++;
        rc = hank_jvp(ctx, dxhh, N, dagg_out);
        if (rc) return rc;
        if (agg_out) {
            HIPC(ctx, join_side(ctx));
            HIPC(ctx, hipMemcpyAsync(agg_out, ctx->d_agg, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream));
            HIPC(ctx, hipStreamSynchronize(ctx->stream));
        }
        return HANK_OK;
    }
    if (use_wide(ctx, N)) {
        rc = hank_primal(ctx, xhh, agg_out);
        return rc ? rc : hank_jvp(ctx, dxhh, N, dagg_out);
    }
    note_primal_x(ctx, nullptr);
    const double *d_dagg = nullptr;
    if (use_x_fused(ctx, N)) {
        rc = x_dual(ctx, xhh, dxhh, hipMemcpyHostToDevice, N, nullptr, nullptr);
        if (rc) return rc;
        rc = fetch_device_error(ctx);
        if (rc == HANK_ERR_SWEEP && x_fallback_allowed(ctx)) rc = to_launch_schedule(ctx);
        else if (rc) return rc;
        else d_dagg = ctx->xcur->dagg_cm;
        if (rc) return rc;
    }
    if (!d_dagg) {
        rc = ensure_tanwork(ctx, N);
        if (rc) return rc;
        HIPC(ctx, join_side(ctx));
        HIPC(ctx, hipMemcpyAsync(ctx->d_xhh, xhh, sizeof(double) * ctx->c.n_hh * P, hipMemcpyHostToDevice, ctx->stream));
        HIPC(ctx, hipMemcpyAsync(ctx->tw->dxhh, dxhh, sizeof(double) * ctx->c.n_hh * P * N, hipMemcpyHostToDevice, ctx->stream));
        rc = run_fused(ctx);
        if (rc) return rc;
        rc = fetch_device_error(ctx);
        if (rc) { ctx->tw->valid = false; return rc; }
        d_dagg = ctx->tw->dagg_cm;
    }
    if (agg_out) HIPC(ctx, hipMemcpyAsync(agg_out, ctx->d_agg, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipMemcpyAsync(dagg_out, d_dagg, sizeof(double) * P * N, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    note_primal_x(ctx, xhh);
    ctx->stats[7]++;
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

// The household block's sequence-space Jacobian at a stationary primal from its Toeplitz structure (hank_jacobian.h):
// F (P, P, n_hh) and Dv (P, n_hh), column-major. Requires hank_primal at the constant steady-state path with the steady
// state as both boundaries (what getSteadyStateJacobian builds, SteadyStateJacobian.jl:53-57).
int hank_fake_news(hank_ctx *ctx, double *F_out, double *Dv_out) {
    ENTER(ctx);
    if (!ctx || !F_out || !Dv_out) return fail(ctx, HANK_ERR_BAD_ARG, "null pointer");
    if (!ctx->primal_done) return fail(ctx, HANK_ERR_NOT_READY, "hank_primal must be called before hank_fake_news");
    if (!ctx->stationary)
        return fail(ctx, HANK_ERR_NOT_READY, "hank_fake_news needs the recorded primal to be hank_primal at a CONSTANT path (the steady state; SteadyStateJacobian.jl:53-57): "
                    "the last primal was recorded at a path that varies over time, or through a device-pointer entry");
    const Consts &c = ctx->c;
    const int P = c.P, G = c.G, N = c.n_hh, NP = P * N, S = 16;
    hipStream_t s = ctx->stream;
    {   // ... and to be stationary: the policy of the first period equals the policy of the last (a constant path that is not the
        // steady state of the boundary drifts; 1e-6 of the policy's scale is far above a converged value iteration's 1e-11)
        std::vector<double> p0((size_t)G), p1((size_t)G);
        HIPC(ctx, join_side(ctx));
        HIPC(ctx, hipMemcpyAsync(p0.data(), ctx->R.pol, sizeof(double) * G, hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipMemcpyAsync(p1.data(), ctx->R.pol + (size_t)(P - 1) * G, sizeof(double) * G, hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipStreamSynchronize(s));
        double scale = 0.0, diff = 0.0;
        for (int k = 0; k < G; k++) { scale = std::max(scale, fabs(p1[k])); diff = std::max(diff, fabs(p0[k] - p1[k])); }
        if (!(diff <= 1e-6 * std::max(scale, 1e-300)))
            return fail(ctx, HANK_ERR_NOT_READY, "hank_fake_news: the recorded primal is not stationary (policy of period 1 and of period %d differ by %.3g): "
                        "it needs hank_primal at the steady state with the steady state as both boundaries", P, diff);
    }
    { const int src = ensure_seg(ctx); if (src) return src; }
    // 1. n_hh backward tangent sweeps (one batch) seeded at the last period: every lag of the policy response
    int rc = ensure_tanwork(ctx, N);
    if (rc) return rc;
    TanWork &w = *ctx->tw;
    rc = ensure_graphs(ctx, w, 0);
    if (rc) return rc;
    if (!ctx->fn.dpT) {
        auto alloc = [&]() -> int {
            HIPC(ctx, dmalloc(&ctx->fn.dpT, (size_t)G * NP)); HIPC(ctx, dmalloc(&ctx->fn.iota, (size_t)G * NP));
            HIPC(ctx, dmalloc(&ctx->fn.E, (size_t)P * G)); HIPC(ctx, dmalloc(&ctx->fn.Cp, (size_t)S * P * NP));
            HIPC(ctx, dmalloc(&ctx->fn.F, (size_t)P * NP)); HIPC(ctx, dmalloc(&ctx->fn.Dv, (size_t)NP));
            return HANK_OK;
        };
        rc = alloc();
        if (rc) {
            (void)hipFree(ctx->fn.dpT); (void)hipFree(ctx->fn.iota); (void)hipFree(ctx->fn.E); (void)hipFree(ctx->fn.Cp); (void)hipFree(ctx->fn.F); (void)hipFree(ctx->fn.Dv);
            ctx->fn = {};
            (void)hipGetLastError();
            return rc;
        }
    }
    HIPC(ctx, join_side(ctx));      // D_1 and the {w, ig D} records come from the primal's forward sweep
    hipLaunchKernelGGL(k_fn_seed, dim3((unsigned)((N * P * N + 255) / 256)), dim3(256), 0, s, w.dxhh, N, P);
    HIPC(ctx, hipGraphLaunch(w.g_back, s));
    for (TanWork &t : ctx->tws) t.valid = false; w_invalidate(ctx);      // (w.dpol no longer belongs to a caller's batch)
    for (XTan &t : ctx->xw.tans) t.valid = false;
    // 2. the lottery impulse of every lag and input at once
    hipLaunchKernelGGL(k_fn_transpose, dim3((unsigned)((G + 31) / 32), (unsigned)((P + 31) / 32), (unsigned)N), dim3(256), 0, s, w.dpol, P, G, N, ctx->fn.dpT);
    hipLaunchKernelGGL(k_fn_impulse, dim3((unsigned)c.n_a, (unsigned)((NP + 255) / 256)), dim3(256), 0, s, c, ctx->R, ctx->fn.dpT, NP, ctx->fn.iota);
    // 3. the expectation vectors E_u = (T')^u pol_ss
    HIPC(ctx, hipMemcpyAsync(ctx->fn.E, ctx->R.pol, sizeof(double) * G, hipMemcpyDeviceToDevice, s));
    for (int u = 0; u + 1 < P; u++)
        hipLaunchKernelGGL(k_fn_expect, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, c, ctx->R, ctx->fn.E + (size_t)u * G, ctx->fn.E + (size_t)(u + 1) * G);
    // 4. F = E iota, Dv = D_ss' dpT
    const int kchunk = ((G + S - 1) / S + 15) / 16 * 16;
    hipLaunchKernelGGL(k_fn_gemm, dim3((unsigned)((NP + 63) / 64), (unsigned)((P + 63) / 64), (unsigned)S), dim3(256), 0, s, ctx->fn.E, ctx->fn.iota, ctx->fn.Cp, P, NP, G, kchunk);
    hipLaunchKernelGGL(k_fn_reduce, dim3((unsigned)((P * NP + 255) / 256)), dim3(256), 0, s, ctx->fn.Cp, P * NP, S, ctx->fn.F);
    hipLaunchKernelGGL(k_fn_gemm, dim3((unsigned)((NP + 63) / 64), 1, (unsigned)S), dim3(256), 0, s, ctx->R.Dseq + G, ctx->fn.dpT, ctx->fn.Cp, 1, NP, G, kchunk);
    hipLaunchKernelGGL(k_fn_reduce, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, s, ctx->fn.Cp, NP, S, ctx->fn.Dv);
    HIPC(ctx, hipGetLastError());
    std::vector<double> hF((size_t)P * NP), hD((size_t)NP);
    HIPC(ctx, hipMemcpyAsync(hF.data(), ctx->fn.F, sizeof(double) * hF.size(), hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipMemcpyAsync(hD.data(), ctx->fn.Dv, sizeof(double) * hD.size(), hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipStreamSynchronize(s));
    rc = fetch_device_error(ctx);
    if (rc) return rc;
    // column n' = t*N + k of the device arrays is lag j = P-1-t of input k
    for (int k = 0; k < N; k++)
        for (int t = 0; t < P; t++) {
            const int j = P - 1 - t;
            Dv_out[j + (size_t)P * k] = hD[(size_t)t * N + k];
            for (int u = 0; u < P; u++) F_out[u + (size_t)P * (j + (size_t)P * k)] = hF[(size_t)u * NP + (size_t)t * N + k];
        }
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

#ifdef HANK_XSTAMP
int hank_debug_wstamps(hank_ctx *ctx, unsigned long long *out) {
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    HIPC(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(hank::g_wstamps), sizeof(unsigned long long) * 2 * 4 * 64));
    return HANK_OK;
}
// dev build: the stamps of the last sweeps (see hank_xsweep.h)
int hank_debug_stamps(hank_ctx *ctx, unsigned long long *out) {
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    HIPC(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(hank::g_xstamps), sizeof(unsigned long long) * 2 * 32 * XSTAMP_NP * XSTAMP_NS));
    HIPC(ctx, hipMemcpyFromSymbol(out + 2 * 32 * XSTAMP_NP * XSTAMP_NS, HIP_SYMBOL(hank::g_xwaves), sizeof(unsigned long long) * 2 * 32 * XSTAMP_NP * 16));
    return HANK_OK;
}
#endif

int hank_stats(hank_ctx *ctx, int64_t out[8]) {
    if (!ctx || !out) return HANK_ERR_BAD_ARG;
    ctx->stats[3] = ctx->schedule;
    for (int k = 0; k < 8; k++) out[k] = ctx->stats[k];
    return HANK_OK;
}

int hank_gather_columns(hank_ctx *const *ctxs, int32_t n, const double *const *d_blocks, const int32_t *N_k, double *d_out) {
    if (!ctxs || n < 1 || !ctxs[0]) return HANK_ERR_BAD_ARG;
    hank_ctx *c0 = ctxs[0];
    if (!d_blocks || !N_k || !d_out) return fail(c0, HANK_ERR_BAD_ARG, "hank_gather_columns: null pointer");
    const size_t P = c0->c.P;
    size_t col = 0;
    for (int k = 0; k < n; k++) {
        hank_ctx *ck = ctxs[k];
        if (!ck || ck->c.P != c0->c.P || N_k[k] < 0 || (N_k[k] > 0 && !d_blocks[k])) return fail(c0, HANK_ERR_BAD_ARG, "hank_gather_columns: bad block %d", k);
        const size_t bytes = sizeof(double) * P * (size_t)N_k[k];
        double *dst = d_out + P * col;
        col += (size_t)N_k[k];
        if (bytes == 0) continue;
        ENTER(ck);                          // the copy is enqueued on the SOURCE context's stream, behind the sweeps that write the block
        if (ck->device == c0->device) {
            HIPC(c0, hipMemcpyAsync(dst, d_blocks[k], bytes, hipMemcpyDeviceToDevice, ck->stream));
        } else {
            const hipError_t pe = hipDeviceEnablePeerAccess(c0->device, 0);      // (from the source device: it writes into device 0's memory over xGMI)
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return fail(c0, HANK_ERR_NO_DEVICE, "no peer access from device %d to device %d (%s)", ck->device, c0->device, hipGetErrorString(pe)); }
            (void)hipGetLastError();
            HIPC(c0, hipMemcpyPeerAsync(dst, c0->device, d_blocks[k], ck->device, bytes, ck->stream));
        }
        if (ck != c0 && ck->stream != c0->stream) {
            HIPC(c0, hipEventRecord(ck->ev_stream, ck->stream));
            HIPC(c0, hipStreamWaitEvent(c0->stream, ck->ev_stream, 0));
        }
    }
    c0->errmsg[0] = 0;
    return HANK_OK;
}

int hank_info(hank_ctx *ctx, int64_t out[8]) {
    if (!ctx || !out) return HANK_ERR_BAD_ARG;
    out[0] = ctx->last_tan; out[1] = ctx->wide_mode; out[2] = ctx->wide_min; out[3] = w_supported(ctx) ? 1 : 0;
    out[4] = ctx->xjvp_max; out[5] = ctx->c.diet; out[6] = (int64_t)ctx->rec_bytes; out[7] = 0;
    return HANK_OK;
}

int hank_last_timings(hank_ctx *ctx, double out_ms[6], int32_t launches[6]) {
    if (!ctx || !out_ms) return HANK_ERR_BAD_ARG;
    ENTER(ctx);
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    int a[6] = {0, 6, 3, 7, 8, 9}, b[6] = {1, 2, 4, 5, 9, 10};
    for (int k = 0; k < 6; k++) {
        out_ms[k] = -1.0;
        if (ctx->ev_valid[k]) {
            float ms = 0.f;
            HIPC(ctx, hipEventElapsedTime(&ms, ctx->ev[a[k]], ctx->ev[b[k]]));
            out_ms[k] = ms;
        }
        if (launches) launches[k] = ctx->launches[k];
    }
    return HANK_OK;
}

int hank_get_policy_seq(hank_ctx *ctx, double *out) {
    if (!ctx || !out) return HANK_ERR_BAD_ARG;
    ENTER(ctx);
    if (!ctx->primal_done) return fail(ctx, HANK_ERR_NOT_READY, "no primal sweep has been run");
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipMemcpyAsync(out, ctx->R.pol, sizeof(double) * (size_t)ctx->c.P * ctx->c.G, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return HANK_OK;
}

int hank_get_dist_seq(hank_ctx *ctx, double *out) {
    if (!ctx || !out) return HANK_ERR_BAD_ARG;
    ENTER(ctx);
    if (!ctx->primal_done) return fail(ctx, HANK_ERR_NOT_READY, "no primal sweep has been run");
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipMemcpyAsync(out, ctx->R.Dseq + ctx->c.G, sizeof(double) * (size_t)ctx->c.P * ctx->c.G, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return HANK_OK;
}

// The grid-weighted aggregate AD_t = sum_pt a(pt) D_t(pt) of the last primal sweep and its N tangents dAD_t = sum a dD_t of the
// last tangent sweep (whichever family ran it): every forward kernel reduces it next to the policy-weighted one. dev != 0:
// the outputs are device pointers (copies ordered on the context's stream).
static int grid_aggregates(hank_ctx *ctx, double *agg2_out, int32_t N, double *dagg2_out, bool dev) {
    if (!ctx || (!agg2_out && !dagg2_out) || N < 0) return HANK_ERR_BAD_ARG;
    ENTER(ctx);
    const size_t P = ctx->c.P;
    const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (agg2_out) {
        if (!ctx->primal_done) return fail(ctx, HANK_ERR_NOT_READY, "no primal sweep has been run");
        HIPC(ctx, join_side(ctx));
        HIPC(ctx, hipMemcpyAsync(agg2_out, ctx->d_agg + P, sizeof(double) * P, kind, ctx->stream));
    }
    if (dagg2_out && N > 0) {
        const double *src = nullptr;
        if (ctx->last_tan == 2) { if (ctx->wcur && ctx->wcur->valid && ctx->wcur->N == N) src = ctx->wcur->dagg_cm; }
        else if (ctx->last_tan == 1) { if (ctx->xcur && ctx->xcur->valid && ctx->xcur->N == N) src = ctx->xcur->dagg_cm; }
        else if (ctx->tw && ctx->tw->valid && ctx->tw->N == N) src = ctx->tw->dagg_cm;
        if (!src) return fail(ctx, HANK_ERR_NOT_READY, "no tangent sweep with N=%d is current", N);
        HIPC(ctx, hipMemcpyAsync(dagg2_out, src + P * (size_t)N, sizeof(double) * P * N, kind, ctx->stream));
    }
    if (!dev) HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return HANK_OK;
}
int hank_get_grid_aggregates(hank_ctx *ctx, double *agg2_out, int32_t N, double *dagg2_out) { return grid_aggregates(ctx, agg2_out, N, dagg2_out, false); }
int hank_get_grid_aggregates_dev(hank_ctx *ctx, double *d_agg2_out, int32_t N, double *d_dagg2_out) { return grid_aggregates(ctx, d_agg2_out, N, d_dagg2_out, true); }

int hank_get_dpolicy_seq(hank_ctx *ctx, int32_t N, double *out) {
    if (!ctx || !out) return HANK_ERR_BAD_ARG;
    ENTER(ctx);
    const size_t total = (size_t)ctx->c.P * ctx->c.G * N;
    double *tmp = nullptr;
    if (ctx->last_tan == 2) {
        WTan *w = ctx->wcur;
        if (!w || !w->valid || w->N != N) return fail(ctx, HANK_ERR_NOT_READY, "no tangent sweep with N=%d is current", N);
        HIPC(ctx, dmalloc(&tmp, total));
        hipLaunchKernelGGL(k_wide_export_dpol, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, w->dpol, ctx->c.G, ctx->c.P, N, tmp);
        hipError_t e1 = hipMemcpyAsync(out, tmp, sizeof(double) * total, hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
        HIPC(ctx, e1);
        HIPC(ctx, e2);
        return HANK_OK;
    }
    if (ctx->last_tan == 1) {
        XTan *x = ctx->xcur;
        if (!x || !x->valid || x->N != N) return fail(ctx, HANK_ERR_NOT_READY, "no tangent sweep with N=%d is current", N);
        HIPC(ctx, dmalloc(&tmp, total));
        for (const XPass &ps : x->passes) {
            const size_t cnt = (size_t)ctx->c.P * ctx->c.G * ps.N;
            hipLaunchKernelGGL(k_xexport_dpol, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, x->dpol + ps.dpol_off, ctx->c.G, ctx->c.P,
                               ps.groups, ps.D, ps.n0, ps.N, tmp);
        }
        hipError_t e1 = hipMemcpyAsync(out, tmp, sizeof(double) * total, hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
        HIPC(ctx, e1);
        HIPC(ctx, e2);
        return HANK_OK;
    }
    if (!ctx->tw || !ctx->tw->valid || ctx->tw->N != N) return fail(ctx, HANK_ERR_NOT_READY, "no tangent sweep with N=%d is current", N);
    TanWork &w = *ctx->tw;
    HIPC(ctx, dmalloc(&tmp, total));
    hipLaunchKernelGGL(k_export_dpol, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, w.dpol, ctx->c.G, ctx->c.P, N, tmp);
    hipError_t e1 = hipMemcpyAsync(out, tmp, sizeof(double) * total, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    HIPC(ctx, e1);
    HIPC(ctx, e2);
    return HANK_OK;
}

}  // extern "C"

// ---- granular steps ---------------------------------------------------------------------------
struct Scratch {  // frees its device buffers on scope exit
    std::vector<void *> p;
    ~Scratch() { for (void *q : p) (void)hipFree(q); }
    template <typename T> hipError_t alloc(T **out, size_t n) {
        hipError_t e = dmalloc(out, n);
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};

static int granular_backward(hank_ctx *ctx, const double *value_next, const double *dvalue_next,
                             const double *xhh_t, const double *dxhh_t, int N, double *value_out,
                             double *dvalue_out, double *policy_out, double *dpolicy_out) {
    if (!ctx || !value_next || !xhh_t || !value_out || !policy_out) return fail(ctx, HANK_ERR_BAD_ARG, "null pointer");
    if (N > 0 && (!dvalue_next || !dxhh_t || !dvalue_out || !dpolicy_out)) return fail(ctx, HANK_ERR_BAD_ARG, "null tangent pointer");
    ENTER(ctx);
    { const bool pd = ctx->primal_done; const int pend = fetch_device_error(ctx); if (pend) return pend; ctx->primal_done = pd; }   // an error a preceding async sweep left is reported, not overwritten
    const Consts &c = ctx->c;
    const size_t G = c.G;
    hipStream_t s = ctx->stream;
    Scratch sc;
    double *Vin, *xt, *sK, *kc, *A, *B, *u, *v, *pol, *Vout;
    int *ib;
    HIPC(ctx, sc.alloc(&Vin, G)); HIPC(ctx, sc.alloc(&xt, 3)); HIPC(ctx, sc.alloc(&sK, G)); HIPC(ctx, sc.alloc(&kc, G));
    HIPC(ctx, sc.alloc(&A, G)); HIPC(ctx, sc.alloc(&B, G)); HIPC(ctx, sc.alloc(&u, G)); HIPC(ctx, sc.alloc(&v, G));
    HIPC(ctx, sc.alloc(&pol, G)); HIPC(ctx, sc.alloc(&Vout, G)); HIPC(ctx, sc.alloc(&ib, G));
    if (!(1.0 + xhh_t[0] > 0.0)) return fail(ctx, HANK_ERR_DOMAIN, "1 + r must be positive");
    HIPC(ctx, hipMemcpyAsync(Vin, value_next, sizeof(double) * G, hipMemcpyHostToDevice, s));
    HIPC(ctx, hipMemcpyAsync(xt, xhh_t, sizeof(double) * c.n_hh, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, ctx->d_err, 4);
    const dim3 blk(RBP * c.n_e), grd(ctx->nbp);
    hipLaunchKernelGGL(k_egm_X, grd, blk, primal_lds(c), s, c, Vin, xt, sK, kc, ctx->d_err, 0, (const int *)nullptr);
    hipLaunchKernelGGL(k_egm_Y, grd, blk, 0, s, c, sK, xhh_t[0], xhh_t[1], c.n_hh > 2 ? xhh_t[2] : 0.0, pol, ib, A, B, u, v, Vout, ctx->d_err, 0, (const int *)nullptr);
    HIPC(ctx, hipGetLastError());
    const bool was_done = ctx->primal_done;
    int rc = fetch_device_error(ctx);
    ctx->primal_done = was_done;
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(value_out, Vout, sizeof(double) * G, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipMemcpyAsync(policy_out, pol, sizeof(double) * G, hipMemcpyDeviceToHost, s));
    if (N > 0) {
        double *dVin, *dr, *dw, *dtr = nullptr, *ds, *dpol, *dV;
        HIPC(ctx, sc.alloc(&dVin, G * N)); HIPC(ctx, sc.alloc(&dr, N)); HIPC(ctx, sc.alloc(&dw, N));
        HIPC(ctx, sc.alloc(&ds, G * N)); HIPC(ctx, sc.alloc(&dpol, G * N)); HIPC(ctx, sc.alloc(&dV, G * N));
        std::vector<double> hr(N), hw(N), ht(N);
        for (int n = 0; n < N; n++) { hr[n] = dxhh_t[c.n_hh * n]; hw[n] = dxhh_t[c.n_hh * n + 1]; ht[n] = c.n_hh > 2 ? dxhh_t[c.n_hh * n + 2] : 0.0; }
        if (c.n_hh > 2) {
            HIPC(ctx, sc.alloc(&dtr, N));
            HIPC(ctx, hipMemcpyAsync(dtr, ht.data(), sizeof(double) * N, hipMemcpyHostToDevice, s));
        }
        HIPC(ctx, hipMemcpyAsync(dVin, dvalue_next, sizeof(double) * G * N, hipMemcpyHostToDevice, s));
        HIPC(ctx, hipMemcpyAsync(dr, hr.data(), sizeof(double) * N, hipMemcpyHostToDevice, s));
        HIPC(ctx, hipMemcpyAsync(dw, hw.data(), sizeof(double) * N, hipMemcpyHostToDevice, s));
        const unsigned nb = (unsigned)((G * N + 255) / 256);
        hipLaunchKernelGGL(k_tan_X, dim3(nb), dim3(256), 0, s, c, kc, sK, xhh_t[0], dr, dw, dtr, N, dVin, ds);
        hipLaunchKernelGGL(k_tan_Y, dim3(nb), dim3(256), 0, s, c, ib, A, B, u, v, dr, dw, dtr, N, ds, dpol, dV);
        HIPC(ctx, hipGetLastError());
        HIPC(ctx, hipMemcpyAsync(dvalue_out, dV, sizeof(double) * G * N, hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipMemcpyAsync(dpolicy_out, dpol, sizeof(double) * G * N, hipMemcpyDeviceToHost, s));
    }
    HIPC(ctx, hipStreamSynchronize(s));
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

// ---- n_het heterogeneous outputs ---------------------------------------------------------------------------------------------
// ForwardIteration aggregates EVERY heterogeneous variable with the same D_t: agg_j[t] = dot(vec(policy_j[t]), D_t)
// (ForwardIteration.jl:303-307; BackwardIteration.jl:99-112 keeps one policy sequence per variable). Output 0 is the policy variable
// of the endogenous dimension (the savings a', KD / A). Output 1 is consumption, the budget residual c = (1+r_t) a + w_t z_e + tr_t
// - a' (KrusellSmith.jl:80): its aggregate is affine in what the sweeps already reduce,
//     C_t  = (1+r_t) AD_t + w_t ZD_t + tr_t MD_t - KD_t,      AD_t = sum a D_t (the grid-weighted aggregate), ZD_t = sum z_e D_t, MD_t = sum D_t
//     dC_t = dr_t AD_t + dw_t ZD_t + dtr_t MD_t + (1+r_t) dAD_t - dKD_t      (ZD_t, MD_t carry no partials: see hank_set_boundary)
// agg (P, 2) and dagg (P, 2 N) as the sweeps leave them -> out_agg (P, n_het), out_dagg (P, n_het, N) column-major.
__global__ void k_het_outputs(int P, int n_hh, int n_het, int N, const double *__restrict__ xhh, const double *__restrict__ dxhh,
                              const double *__restrict__ agg, const double *__restrict__ dagg, const double *__restrict__ zd,
                              double *__restrict__ out_agg, double *__restrict__ out_dagg) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)P * (N + 1)) return;
    const int t = (int)(idx % P), n = (int)(idx / P) - 1;
    const double r = xhh[n_hh * t], w = xhh[n_hh * t + 1], tr = n_hh > 2 ? xhh[n_hh * t + 2] : 0.0;
    const double KD = agg[t], AD = agg[P + t], ZD = zd[t], MD = zd[P + t];
    if (n < 0) {
        if (!out_agg) return;
        out_agg[t] = KD;
        if (n_het > 1) out_agg[P + t] = ((1.0 + r) * AD + w * ZD + tr * MD) - KD;
        return;
    }
    if (!out_dagg) return;
    const double dKD = dagg[(size_t)n * P + t], dAD = dagg[((size_t)N + n) * P + t];
    const double *dx = dxhh + ((size_t)n * P + t) * n_hh;
    out_dagg[((size_t)n * n_het) * P + t] = dKD;
    if (n_het > 1) out_dagg[((size_t)n * n_het + 1) * P + t] = (dx[0] * AD + dx[1] * ZD + (n_hh > 2 ? dx[2] : 0.0) * MD + (1.0 + r) * dAD) - dKD;
}

static int het_outputs(hank_ctx *ctx, int32_t n_het, const double *dxhh, int32_t N, double *agg_out, double *dagg_out, bool dev) {
    if (!ctx) return HANK_ERR_BAD_ARG;
    ENTER(ctx);
    if (n_het < 1 || n_het > 2 || N < 0 || (!agg_out && !dagg_out)) return fail(ctx, HANK_ERR_BAD_ARG, "n_het must be 1 or 2 (the policy variable, consumption), N >= 0");
    if (!ctx->primal_done) return fail(ctx, HANK_ERR_NOT_READY, "no primal sweep has been run");
    const size_t P = ctx->c.P, nh = ctx->c.n_hh;
    const bool tan = dagg_out && N > 0;
    if (tan && !dxhh) return fail(ctx, HANK_ERR_BAD_ARG, "the tangent outputs need the dxhh of the last tangent sweep");
    const double *src = nullptr;
    if (tan) {
        if (ctx->last_tan == 2) { if (ctx->wcur && ctx->wcur->valid && ctx->wcur->N == N) src = ctx->wcur->dagg_cm; }
        else if (ctx->last_tan == 1) { if (ctx->xcur && ctx->xcur->valid && ctx->xcur->N == N) src = ctx->xcur->dagg_cm; }
        else if (ctx->tw && ctx->tw->valid && ctx->tw->N == N) src = ctx->tw->dagg_cm;
        if (!src) return fail(ctx, HANK_ERR_NOT_READY, "no tangent sweep with N=%d is current", N);
    }
    HIPC(ctx, join_side(ctx));
    Scratch sc;
    const double *d_dx = dxhh;
    double *d_a = agg_out, *d_da = dagg_out;
    if (!dev) {
        double *tmp = nullptr;
        if (tan) {
            HIPC(ctx, sc.alloc(&tmp, nh * P * N));
            HIPC(ctx, hipMemcpyAsync(tmp, dxhh, sizeof(double) * nh * P * N, hipMemcpyHostToDevice, ctx->stream));
            d_dx = tmp;
            HIPC(ctx, sc.alloc(&d_da, P * n_het * N));
        }
        if (agg_out) HIPC(ctx, sc.alloc(&d_a, P * n_het));
    }
    const int Nk = tan ? N : 0;
    hipLaunchKernelGGL(k_het_outputs, dim3((unsigned)((P * (Nk + 1) + 255) / 256)), dim3(256), 0, ctx->stream, (int)P, (int)nh, (int)n_het, Nk, ctx->d_xhh,
                       d_dx, ctx->d_agg, src, ctx->d_zd, d_a, tan ? d_da : nullptr);
    HIPC(ctx, hipGetLastError());
    if (!dev) {
        if (agg_out) HIPC(ctx, hipMemcpyAsync(agg_out, d_a, sizeof(double) * P * n_het, hipMemcpyDeviceToHost, ctx->stream));
        if (tan) HIPC(ctx, hipMemcpyAsync(dagg_out, d_da, sizeof(double) * P * n_het * N, hipMemcpyDeviceToHost, ctx->stream));
        HIPC(ctx, hipStreamSynchronize(ctx->stream));
    }
    return HANK_OK;
}
extern "C" {
int hank_get_het_outputs(hank_ctx *ctx, int32_t n_het, const double *dxhh, int32_t N, double *agg_out, double *dagg_out) {
    return het_outputs(ctx, n_het, dxhh, N, agg_out, dagg_out, false);
}
int hank_get_het_outputs_dev(hank_ctx *ctx, int32_t n_het, const double *d_dxhh, int32_t N, double *d_agg_out, double *d_dagg_out) {
    return het_outputs(ctx, n_het, d_dxhh, N, d_agg_out, d_dagg_out, true);
}
}

extern "C" {
int hank_backward_step(hank_ctx *ctx, const double *value_next, const double *xhh_t, double *value_out, double *policy_out) {
    return granular_backward(ctx, value_next, nullptr, xhh_t, nullptr, 0, value_out, nullptr, policy_out, nullptr);
}
int hank_backward_step_dual(hank_ctx *ctx, const double *value_next, const double *dvalue_next,
                            const double *xhh_t, const double *dxhh_t, int32_t N, double *value_out,
                            double *dvalue_out, double *policy_out, double *dpolicy_out) {
    if (N < 1) return fail(ctx, HANK_ERR_BAD_ARG, "N must be >= 1");
    return granular_backward(ctx, value_next, dvalue_next, xhh_t, dxhh_t, N, value_out, dvalue_out, policy_out, dpolicy_out);
}
}  // extern "C"

// ---- steady state: the inner fixed point of get_xVals on the device (SteadyState.jl:132-141) ------------------
extern "C" int hank_vfi(hank_ctx *ctx, const double *xhh_t, double tol, int32_t max_iter, double *value_io, double *policy_out,
                        int32_t *iters_out, double *supnorm_out) {
    if (!ctx || !xhh_t || !value_io || !policy_out || max_iter < 1) return fail(ctx, HANK_ERR_BAD_ARG, "bad argument");
    if (!(1.0 + xhh_t[0] > 0.0)) return fail(ctx, HANK_ERR_DOMAIN, "1 + r must be positive");
    ENTER(ctx);
    { const bool pd = ctx->primal_done; const int pend = fetch_device_error(ctx); if (pend) return pend; ctx->primal_done = pd; }
    const Consts &c = ctx->c;
    const size_t G = c.G;
    hipStream_t s = ctx->stream;
    Scratch sc;
    double *V[2], *xt, *sK, *kc, *A, *B, *u, *v, *pol, *norm;
    int *ib, *state;
    HIPC(ctx, sc.alloc(&V[0], G)); HIPC(ctx, sc.alloc(&V[1], G)); HIPC(ctx, sc.alloc(&xt, 4)); HIPC(ctx, sc.alloc(&sK, G));
    HIPC(ctx, sc.alloc(&kc, G)); HIPC(ctx, sc.alloc(&A, G)); HIPC(ctx, sc.alloc(&B, G)); HIPC(ctx, sc.alloc(&u, G));
    HIPC(ctx, sc.alloc(&v, G)); HIPC(ctx, sc.alloc(&pol, G)); HIPC(ctx, sc.alloc(&norm, 1)); HIPC(ctx, sc.alloc(&ib, G));
    HIPC(ctx, sc.alloc(&state, 2));
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipMemcpyAsync(V[0], value_io, sizeof(double) * G, hipMemcpyHostToDevice, s));
    HIPC(ctx, hipMemcpyAsync(xt, xhh_t, sizeof(double) * c.n_hh, hipMemcpyHostToDevice, s));
    HIPC(ctx, hipMemsetAsync(state, 0, 2 * sizeof(int), s));
    hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, ctx->d_err, 4);
    const dim3 blk(RBP * c.n_e), grd(ctx->nbp);
    const double r = xhh_t[0], w = xhh_t[1], tr = c.n_hh > 2 ? xhh_t[2] : 0.0;
    int hstate[2] = {0, 0};
    const char *xve = getenv("HANK_XVFI");            // dev knob: 0 = per-step launches
    if (use_x_primal(ctx) && !(xve && atoi(xve) == 0)) {
        // the whole iteration as ONE persistent launch on the group of XCD 0 (k_xvfi): the vote on convergence rides on the group barrier
        int rc = x_setup(ctx);
        if (rc) return rc;
        XWork &X = ctx->xw;
        rc = x_serialize_begin(ctx);
        if (rc) return rc;
        { const int rc_ = x_sync_reset(ctx, X.sync, 1, 4); if (rc_) return rc_; }
        XVfiArgs va{};
        va.c = c; va.V0 = V[0]; va.r = r; va.w = w; va.tr = tr; va.tol = tol; va.max_iter = max_iter; va.sy = X.sync; va.st_s = X.st_s;
        va.err = ctx->d_err; va.Vout = V[1]; va.pol = pol; va.iters = state; va.supnorm = norm;
        const bool fits = 64 * (c.n_e + 1) <= X.maxt && X.syncwave;
        const dim3 xblk(fits ? 64 * (c.n_e + 1) : 64 * c.n_e);
        const size_t lds = sizeof(double) * ((size_t)c.n_e * 64 + (size_t)c.n_e * c.n_e + c.n_a + 16) + 64;
        if (X.maxt == 768) hipLaunchKernelGGL((k_xvfi<768>), dim3(X.grid), xblk, lds, s, va);
        else hipLaunchKernelGGL((k_xvfi<1024>), dim3(X.grid), xblk, lds, s, va);
        HIPC(ctx, hipGetLastError());
        rc = x_serialize_end(ctx);
        if (rc) return rc;
        int xs[2] = {0, 0};      // {steps, converged}
        XSync hsy;
        HIPC(ctx, hipMemcpyAsync(xs, state, sizeof(xs), hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipMemcpyAsync(&hsy, X.sync, sizeof(XSync), hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipStreamSynchronize(s));
        X.last_passes = 0;       // (this sync block has been checked here)
        if (hsy.status[0] == 0) {
            int e[4];
            HIPC(ctx, hipMemcpy(e, ctx->d_err, sizeof(e), hipMemcpyDeviceToHost));
            if (e[0] != 0) HIPC(ctx, hipMemsetAsync(ctx->d_err, 0, sizeof(e), ctx->stream));
            if (e[0] == ERR_KNOTS)
                return fail(ctx, HANK_ERR_KNOTS, "knot-vectors must be unique and sorted in increasing order (steady-state value iteration, step %d, "
                            "productivity state %d, wealth index %d)", xs[0], e[2] + 1, e[3] + 1);
            if (e[0] == ERR_DOMAIN)
                return fail(ctx, HANK_ERR_DOMAIN, "DomainError: negative base under a non-integer power (steady-state value iteration, step %d)", xs[0]);
            double hn = 0.0;
            HIPC(ctx, hipMemcpyAsync(value_io, V[1], sizeof(double) * G, hipMemcpyDeviceToHost, s));
            HIPC(ctx, hipMemcpyAsync(policy_out, pol, sizeof(double) * G, hipMemcpyDeviceToHost, s));
            HIPC(ctx, hipMemcpyAsync(&hn, norm, sizeof(double), hipMemcpyDeviceToHost, s));
            HIPC(ctx, hipStreamSynchronize(s));
            if (iters_out) *iters_out = xs[0];
            if (supnorm_out) *supnorm_out = hn;
            ctx->stats[5] += xs[0];
            ctx->errmsg[0] = 0;
            return HANK_OK;
        }
        // the group did not form or a wait timed out: this context continues on the launches (a forced schedule fails loudly)
        if (!x_fallback_allowed(ctx))
            return fail(ctx, HANK_ERR_SWEEP, "persistent value iteration: %s on XCD %u", hsy.status[0] == XERR_PLACEMENT ? "the group is short of members" : "a wait timed out", hsy.status[1]);
        rc = to_launch_schedule(ctx);
        if (rc) return rc;
        HIPC(ctx, hipMemsetAsync(state, 0, 2 * sizeof(int), s));
        hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, ctx->d_err, 4);
    }
    int done = 0;                      // steps enqueued
    const int chunk = 64;              // the stop flag travels to the host once per chunk; converged steps freeze the state
    while (!hstate[0] && done < max_iter) {
        const int n = std::min(chunk, max_iter - done);
        for (int k = 0; k < n; k++) {
            const int cur = (done + k) & 1;
            hipLaunchKernelGGL(k_egm_X, grd, blk, primal_lds(c), s, c, V[cur], xt, sK, kc, ctx->d_err, 0, (const int *)state);
            hipLaunchKernelGGL(k_egm_Y, grd, blk, 0, s, c, sK, r, w, tr, pol, ib, A, B, u, v, V[cur ^ 1], ctx->d_err, 0, (const int *)state);
            hipLaunchKernelGGL(k_vfi_check, dim3(1), dim3(1024), 0, s, V[cur ^ 1], V[cur], (int)G, tol, state, norm);
        }
        done += n;
        HIPC(ctx, hipGetLastError());
        HIPC(ctx, hipMemcpyAsync(hstate, state, sizeof(hstate), hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipStreamSynchronize(s));
        int e[4];
        HIPC(ctx, hipMemcpy(e, ctx->d_err, sizeof(e), hipMemcpyDeviceToHost));
        if (e[0] != 0) HIPC(ctx, hipMemsetAsync(ctx->d_err, 0, sizeof(e), ctx->stream));
        if (e[0] == ERR_KNOTS)
            return fail(ctx, HANK_ERR_KNOTS, "knot-vectors must be unique and sorted in increasing order (steady-state value iteration, step %d, "
                        "productivity state %d, wealth index %d)", hstate[1] + 1, e[2] + 1, e[3] + 1);
        if (e[0] == ERR_DOMAIN)
            return fail(ctx, HANK_ERR_DOMAIN, "DomainError: negative base under a non-integer power (steady-state value iteration, step %d)", hstate[1] + 1);
    }
    const int fin = hstate[1] & 1;     // step k reads V[(k-1)&1] and writes V[k&1]
    double hn = 0.0;
    HIPC(ctx, hipMemcpyAsync(value_io, V[fin], sizeof(double) * G, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipMemcpyAsync(policy_out, pol, sizeof(double) * G, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipMemcpyAsync(&hn, norm, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipStreamSynchronize(s));
    if (iters_out) *iters_out = hstate[1];
    if (supnorm_out) *supnorm_out = hn;
    ctx->stats[5] += hstate[1];
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

// ---- steady state: the stationary distribution by the power method on the device --------------------------------
// D <- Lambda(policy) D (Young lottery + exogenous transition: the forward step kernel of the hot path) until two
// iterates `check_every` steps apart differ by less than tol in the max norm, the rule of the host's power method
// (GeneralStructures.py:invariant_dist, used where the reference's direct solve of ForwardIteration.jl:436-442 is too
// large). The caller normalises the result.
extern "C" int hank_stationary_dist(hank_ctx *ctx, const double *policy, double *D_io, double tol, int32_t max_iter, int32_t check_every,
                                    int32_t *iters_out) {
    if (!ctx || !policy || !D_io || max_iter < 1 || check_every < 1) return fail(ctx, HANK_ERR_BAD_ARG, "bad argument");
    ENTER(ctx);
    { const bool pd = ctx->primal_done; const int pend = fetch_device_error(ctx); if (pend) return pend; ctx->primal_done = pd; }
    const Consts &c = ctx->c;
    const size_t G = c.G;
    hipStream_t s = ctx->stream;
    Scratch sc;
    Record R{};      // ONE period of lottery record
    double *D[2], *Dchk, *aggp, *norm;
    int *state;
    HIPC(ctx, sc.alloc(&R.pol, G)); HIPC(ctx, sc.alloc(&R.lw, G)); HIPC(ctx, sc.alloc(&R.ig, G)); HIPC(ctx, sc.alloc(&R.lo, G));
    HIPC(ctx, sc.alloc(&R.start, (size_t)c.n_e * (c.n_a + 1))); HIPC(ctx, sc.alloc(&R.clo, c.n_e)); HIPC(ctx, sc.alloc(&R.seg, G));
    HIPC(ctx, sc.alloc(&R.lwg, G));
    HIPC(ctx, sc.alloc(&D[0], G)); HIPC(ctx, sc.alloc(&D[1], G)); HIPC(ctx, sc.alloc(&Dchk, G)); HIPC(ctx, sc.alloc(&aggp, 2 * (size_t)ctx->nbp));
    HIPC(ctx, sc.alloc(&norm, 1)); HIPC(ctx, sc.alloc(&state, 2));
    HIPC(ctx, join_side(ctx));
    HIPC(ctx, hipMemcpyAsync(R.pol, policy, sizeof(double) * G, hipMemcpyHostToDevice, s));
    HIPC(ctx, hipMemcpyAsync(D[0], D_io, sizeof(double) * G, hipMemcpyHostToDevice, s));
    HIPC(ctx, hipMemcpyAsync(Dchk, D_io, sizeof(double) * G, hipMemcpyHostToDevice, s));
    HIPC(ctx, hipMemsetAsync(state, 0, 2 * sizeof(int), s));
    hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, ctx->d_err, 4);
    hipLaunchKernelGGL(k_lottery, dim3((unsigned)c.n_e), dim3(256), sizeof(int) * (2 * (size_t)c.n_a + 2), s, c, R, c.n_e, ctx->d_err, 1, 0);
    HIPC(ctx, hipGetLastError());
    const dim3 blk(RBP * c.n_e), grd(ctx->nbp);
    int hstate[2] = {0, 0}, done = 0;
    const char *xse = getenv("HANK_XSTAT");           // dev knob: 0 = one launch per iteration
    if (use_x_primal(ctx) && !(xse && atoi(xse) == 0)) {
        // the whole power method as ONE persistent launch on the group of XCD 0 (k_xstat)
        int rc = x_setup(ctx);
        if (rc) return rc;
        XWork &X = ctx->xw;
        rc = x_serialize_begin(ctx);
        if (rc) return rc;
        { const int rc_ = x_sync_reset(ctx, X.sync, 1, 4); if (rc_) return rc_; }
        XStatArgs sa{};
        sa.c = c; sa.R = R; sa.D0 = D[0]; sa.tol = tol; sa.max_iter = max_iter; sa.check_every = check_every; sa.sy = X.sync;
        sa.st_D = X.st_D; sa.Dout = D[1]; sa.iters = state;
        const bool fits = 64 * (c.n_e + 1) <= X.maxt && X.syncwave;
        const dim3 xblk(fits ? 64 * (c.n_e + 1) : 64 * c.n_e);
        const size_t lds = sizeof(double) * ((size_t)c.n_e * 64 + 16) + 64;
        if (X.maxt == 768) hipLaunchKernelGGL((k_xstat<768>), dim3(X.grid), xblk, lds, s, sa);
        else hipLaunchKernelGGL((k_xstat<1024>), dim3(X.grid), xblk, lds, s, sa);
        HIPC(ctx, hipGetLastError());
        rc = x_serialize_end(ctx);
        if (rc) return rc;
        int xs[2] = {0, 0};
        XSync hsy;
        HIPC(ctx, hipMemcpyAsync(xs, state, sizeof(xs), hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipMemcpyAsync(&hsy, X.sync, sizeof(XSync), hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipStreamSynchronize(s));
        X.last_passes = 0;
        if (hsy.status[0] == 0) {
            int e[4];
            HIPC(ctx, hipMemcpy(e, ctx->d_err, sizeof(e), hipMemcpyDeviceToHost));
            if (e[0] != 0) HIPC(ctx, hipMemsetAsync(ctx->d_err, 0, sizeof(e), ctx->stream));
            if (e[0] == ERR_NONMONO) return fail(ctx, HANK_ERR_NONMONOTONE, "savings policy is not monotone in wealth (productivity state %d, wealth index %d)", e[2] + 1, e[3] + 1);
            HIPC(ctx, hipMemcpyAsync(D_io, D[1], sizeof(double) * G, hipMemcpyDeviceToHost, s));
            HIPC(ctx, hipStreamSynchronize(s));
            if (iters_out) *iters_out = xs[0];
            ctx->errmsg[0] = 0;
            return HANK_OK;
        }
        if (!x_fallback_allowed(ctx))
            return fail(ctx, HANK_ERR_SWEEP, "persistent power method: %s on XCD %u", hsy.status[0] == XERR_PLACEMENT ? "the group is short of members" : "a wait timed out", hsy.status[1]);
        rc = to_launch_schedule(ctx);
        if (rc) return rc;
        HIPC(ctx, hipMemsetAsync(state, 0, 2 * sizeof(int), s));
    }
    while (!hstate[0] && done < max_iter) {
        // a chunk = several checks; every check compares the iterate with the one check_every steps earlier
        for (int q = 0; q < 16 && done < max_iter; q++) {
            const int n = std::min((int)check_every, max_iter - done);
            for (int k = 0; k < n; k++, done++)
                hipLaunchKernelGGL(k_dist_iter, grd, blk, primal_lds(c), s, c, R, (const double *)D[done & 1], D[(done + 1) & 1], aggp, (const int *)state);
            hipLaunchKernelGGL(k_vfi_check, dim3(1), dim3(1024), 0, s, (const double *)D[done & 1], (const double *)Dchk, (int)G, tol, state, norm);
            HIPC(ctx, hipMemcpyAsync(Dchk, D[done & 1], sizeof(double) * G, hipMemcpyDeviceToDevice, s));
        }
        HIPC(ctx, hipGetLastError());
        HIPC(ctx, hipMemcpyAsync(hstate, state, sizeof(hstate), hipMemcpyDeviceToHost, s));
        HIPC(ctx, hipStreamSynchronize(s));
    }
    int e[4];
    HIPC(ctx, hipMemcpy(e, ctx->d_err, sizeof(e), hipMemcpyDeviceToHost));
    if (e[0] != 0) HIPC(ctx, hipMemsetAsync(ctx->d_err, 0, sizeof(e), ctx->stream));
    if (e[0] == ERR_NONMONO) return fail(ctx, HANK_ERR_NONMONOTONE, "savings policy is not monotone in wealth (productivity state %d, wealth index %d)", e[2] + 1, e[3] + 1);
    // once converged the iteration kernels stop touching the buffers: Dchk holds the last checked iterate
    HIPC(ctx, hipMemcpyAsync(D_io, Dchk, sizeof(double) * G, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipStreamSynchronize(s));
    if (iters_out) *iters_out = done;
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

static int granular_forward(hank_ctx *ctx, const double *policy, const double *dpolicy, const double *D_prev,
                            const double *dD_prev, int N, double *D_out, double *dD_out, double *agg_out, double *dagg_out) {
    if (!ctx || !policy || !D_prev || !D_out) return fail(ctx, HANK_ERR_BAD_ARG, "null pointer");
    if (N > 0 && (!dpolicy || !dD_prev || !dD_out)) return fail(ctx, HANK_ERR_BAD_ARG, "null tangent pointer");
    const Consts &c = ctx->c;
    const size_t G = c.G, W = 1 + (size_t)N;
    hipStream_t s = ctx->stream;
    Scratch sc;
    double *pol, *dpol = nullptr, *Dp, *dDp = nullptr, *Dmid, *Do, *dDo = nullptr, *aggterm, *aggv;
    HIPC(ctx, sc.alloc(&pol, G)); HIPC(ctx, sc.alloc(&Dp, G)); HIPC(ctx, sc.alloc(&Dmid, G * W));
    HIPC(ctx, sc.alloc(&Do, G)); HIPC(ctx, sc.alloc(&aggterm, G * W)); HIPC(ctx, sc.alloc(&aggv, W));
    HIPC(ctx, sc.alloc(&dpol, G * (N ? N : 1))); HIPC(ctx, sc.alloc(&dDp, G * (N ? N : 1))); HIPC(ctx, sc.alloc(&dDo, G * (N ? N : 1)));
    HIPC(ctx, hipMemcpyAsync(pol, policy, sizeof(double) * G, hipMemcpyHostToDevice, s));
    HIPC(ctx, hipMemcpyAsync(Dp, D_prev, sizeof(double) * G, hipMemcpyHostToDevice, s));
    if (N > 0) {
        HIPC(ctx, hipMemcpyAsync(dpol, dpolicy, sizeof(double) * G * N, hipMemcpyHostToDevice, s));
        HIPC(ctx, hipMemcpyAsync(dDp, dD_prev, sizeof(double) * G * N, hipMemcpyHostToDevice, s));
    }
    HIPC(ctx, hipMemsetAsync(Dmid, 0, sizeof(double) * G * W, s));
    const unsigned nb = (unsigned)((G * W + 255) / 256);
    hipLaunchKernelGGL(k_scatter_general, dim3(nb), dim3(256), 0, s, c, pol, dpol, Dp, dDp, N, Dmid);
    hipLaunchKernelGGL(k_mix_general, dim3(nb), dim3(256), 0, s, c, Dmid, pol, dpol, N, Do, dDo, aggterm);
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)W), dim3(256), 0, s, aggterm, (int)G, (int)W, aggv);
    HIPC(ctx, hipGetLastError());
    std::vector<double> hagg(W);
    HIPC(ctx, hipMemcpyAsync(D_out, Do, sizeof(double) * G, hipMemcpyDeviceToHost, s));
    if (N > 0) HIPC(ctx, hipMemcpyAsync(dD_out, dDo, sizeof(double) * G * N, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipMemcpyAsync(hagg.data(), aggv, sizeof(double) * W, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipStreamSynchronize(s));
    if (agg_out) *agg_out = hagg[0];
    if (dagg_out) for (int n = 0; n < N; n++) dagg_out[n] = hagg[1 + n];
    ctx->errmsg[0] = 0;
    return HANK_OK;
}

extern "C" {
int hank_forward_step(hank_ctx *ctx, const double *policy, const double *D_prev, double *D_out, double *agg_out) {
    return granular_forward(ctx, policy, nullptr, D_prev, nullptr, 0, D_out, nullptr, agg_out, nullptr);
}
int hank_forward_step_dual(hank_ctx *ctx, const double *policy, const double *dpolicy, const double *D_prev,
                           const double *dD_prev, int32_t N, double *D_out, double *dD_out, double *agg_out, double *dagg_out) {
    if (N < 1) return fail(ctx, HANK_ERR_BAD_ARG, "N must be >= 1");
    return granular_forward(ctx, policy, dpolicy, D_prev, dD_prev, N, D_out, dD_out, agg_out, dagg_out);
}

}  // extern "C"
