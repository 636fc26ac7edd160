// ubench: what does ONE episode of the XCD-local group barrier of hank_xsweep.h cost, alone? 256 workgroups (one per CU),
// grouped by the XCD they run on, E episodes, nothing between the barriers except an optional few-hundred-cycle delay.
// Variants: 0 = flags in one line per group + one polling wave (the product's xbarrier), 1 = one 128-B line per member,
// 2 = agent-scope atomic counter + sc1 poll (round 1's form), 3 = variant 0 without s_sleep in the poll loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Sync { unsigned ticket[8][32]; unsigned total[32]; unsigned flag[8][64]; unsigned flagw[8][64][32]; unsigned ctr[8][32]; };
__device__ __forceinline__ unsigned ldu(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int V>
__global__ void k(Sync *sy, int E, int delay, unsigned long long *out) {
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
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int ep = 1; ep <= E; ep++) {
        for (int d = 0; d < delay; d++) __builtin_amdgcn_s_sleep(1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            if (V == 2) {
                if (lane == 0) __hip_atomic_fetch_add(&sy->ctr[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (;;) { unsigned v = lane == 0 ? ldu(&sy->ctr[x][0]) : 0u; v = __builtin_amdgcn_readfirstlane(v); if ((int)(v - (unsigned)(ep * S)) >= 0) break; __builtin_amdgcn_s_sleep(1); }
            } else if (V == 1) {
                if (lane == 0) *reinterpret_cast<volatile unsigned *>(&sy->flagw[x][c][0]) = (unsigned)ep;
                for (;;) { const unsigned f = lane < S ? ldu(&sy->flagw[x][lane][0]) : (unsigned)ep; if (__all((int)(f - (unsigned)ep) >= 0)) break; __builtin_amdgcn_s_sleep(1); }
            } else {
                if (lane == 0) *reinterpret_cast<volatile unsigned *>(&sy->flag[x][c]) = (unsigned)ep;
                for (;;) { const unsigned f = lane < S ? ldu(&sy->flag[x][lane]) : (unsigned)ep; if (__all((int)(f - (unsigned)ep) >= 0)) break; if (V == 0) __builtin_amdgcn_s_sleep(1); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}
int main() {
    Sync *sy; unsigned long long *out; const int E = 2000;
    hipMalloc(&sy, sizeof(Sync)); hipMalloc(&out, 256 * 8);
    std::vector<unsigned long long> h(256);
    for (int threads : {64, 704}) for (int delay : {0, 20}) for (int v = 0; v < 4; v++) {
        hipMemset(sy, 0, sizeof(Sync));
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, sy, E, delay, out);
        if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, sy, E, delay, out);
        if (v == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, sy, E, delay, out);
        if (v == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, sy, E, delay, out);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (auto t : h) mx = t > mx ? t : mx;
        printf("threads %4d delay %2d variant %d: %.3f us per episode (incl. %d x s_sleep 1 of work)\n", threads, delay, v, mx * 0.01 / E, delay);   // s_memrealtime: 100 MHz
    }
    return 0;
}
