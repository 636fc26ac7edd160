// ubench: can the workgroups of the other XCDs TRAIL a producer group that runs on one XCD, inside ONE launch? The producer group
// (XCD 0, 32 workgroups) writes one "period" of a record per step (NARR arrays of G doubles, each member its slice), meets at its
// XCD-local barrier and publishes a progress word; the consumers (XCDs 1..7) wait for progress >= step + LAG, read the period and
// check every value. Variants: record stores plain / sc1 (write-through); progress poll by sc1 load / by an agent-scope atomic
// (executes at the memory side). Per-XCD L2s are not coherent with each other: a consumer's L2 keeps a line it has fetched —
// G not a multiple of 16 doubles makes the tail line of period i hold the head of period i+1 (LAG = 1 then reads stale heads).
// Reports: mismatching values, and how far behind the producer the consumers finish.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NARR = 6;
struct Sync { unsigned ticket[8][32]; unsigned total[32]; unsigned flagw[8][64][32]; unsigned prog[32]; unsigned long long bad[32]; };
__device__ __forceinline__ unsigned ldu(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double val(int i, int a, int k) { return (double)(i * 1000003 + a * 7919 + k) * 0.5; }
template <int ST, int POLL>
__global__ void k(Sync *sy, double *rec, int G, int E, int LAG, unsigned long long *tout) {
    __shared__ int ctl[4];
    if (threadIdx.x == 0) {
        int xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7;
        unsigned c = __hip_atomic_fetch_add(&sy->ticket[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&sy->total[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ldu(&sy->total[0]) < gridDim.x) __builtin_amdgcn_s_sleep(2);
        ctl[0] = xcc; ctl[1] = (int)c; ctl[2] = (int)ldu(&sy->ticket[xcc][0]);
    }
    __syncthreads();
    const int x = ctl[0], c = ctl[1], S = ctl[2];
    const int per = (G + S - 1) / S, k0 = c * per, k1 = min(G, k0 + per);
    unsigned long long bad = 0;
    for (int i = 0; i < E; i++) {
        if (x == 0) {
            for (int a = 0; a < NARR; a++)
                for (int kk = k0 + threadIdx.x; kk < k1; kk += blockDim.x) {
                    double *p = rec + ((size_t)i * NARR + a) * G + kk;
                    const double v = val(i, a, kk);
                    if (ST == 1) __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *p = v;
                }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                if (lane == 0) *reinterpret_cast<volatile unsigned *>(&sy->flagw[x][c][0]) = (unsigned)(i + 1);
                for (;;) { const unsigned f = lane < S ? ldu(&sy->flagw[x][lane][0]) : (unsigned)(i + 1); if (__all((int)(f - (unsigned)(i + 1)) >= 0)) break; __builtin_amdgcn_s_sleep(1); }
                if (c == 0 && lane == 0) __hip_atomic_store(&sy->prog[0], (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every member's stores have been acknowledged
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            if (threadIdx.x == 0) {
                const unsigned need = (unsigned)min(i + LAG, E);
                for (;;) {
                    const unsigned p = POLL == 1 ? __hip_atomic_fetch_add(&sy->prog[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ldu(&sy->prog[0]);
                    if ((int)(p - need) >= 0) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            __syncthreads();
            for (int a = 0; a < NARR; a++)
                for (int kk = k0 + threadIdx.x; kk < k1; kk += blockDim.x)
                    if (rec[((size_t)i * NARR + a) * G + kk] != val(i, a, kk)) bad++;
        }
    }
    if (bad) atomicAdd(&sy->bad[0], bad);
    if (threadIdx.x == 0) tout[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}
int main() {
    Sync *sy; double *rec; unsigned long long *tout; const int E = 300;
    hipMalloc(&sy, sizeof(Sync)); hipMalloc(&tout, 256 * 8);
    const size_t maxG = 22016;
    hipMalloc(&rec, sizeof(double) * E * NARR * maxG);
    std::vector<unsigned long long> h(256);
    for (int G : {22000, 22001}) for (int LAG : {1, 2, 3}) for (int st = 0; st < 2; st++) for (int poll = 0; poll < 2; poll++) {
        hipMemset(sy, 0, sizeof(Sync)); hipMemset(rec, 0xff, sizeof(double) * E * NARR * maxG);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
        if (st == 0 && poll == 0) hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(704), 0, 0, sy, rec, G, E, LAG, tout);
        if (st == 0 && poll == 1) hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(704), 0, 0, sy, rec, G, E, LAG, tout);
        if (st == 1 && poll == 0) hipLaunchKernelGGL((k<1, 0>), dim3(256), dim3(704), 0, 0, sy, rec, G, E, LAG, tout);
        if (st == 1 && poll == 1) hipLaunchKernelGGL((k<1, 1>), dim3(256), dim3(704), 0, 0, sy, rec, G, E, LAG, tout);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        Sync hs; hipMemcpy(&hs, sy, sizeof(Sync), hipMemcpyDeviceToHost);
        printf("G %5d LAG %d stores %s poll %s: %8llu stale values, %.3f ms per launch (%.2f us per period)\n", G, LAG, st ? "sc1  " : "plain", poll ? "atomic" : "sc1   ",
               hs.bad[0], ms, 1e3 * ms / E);
    }
    return 0;
}
