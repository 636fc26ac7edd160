// micro-benchmark: does data written by kernel A stay in the writing XCD's L2 for kernel B?
// A: block b writes chunk b (CH bytes). B: block b reads chunk (b + shift) with a dependent chain, timing itself.
// shift = 0 -> same block index (same XCD under round-robin dispatch); shift = 1 -> a neighbouring XCD's data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int CH = 2048;   // doubles per chunk (16 KB)
__global__ void kA(double *buf, double v) {
    double *p = buf + (size_t)blockIdx.x * CH;
    for (int i = threadIdx.x; i < CH; i += blockDim.x) p[i] = v + i;
}
__global__ void kB(const double *buf, int nblk, int shift, double *out, unsigned long long *cyc) {
    const double *p = buf + (size_t)((blockIdx.x + shift) % nblk) * CH;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int i = threadIdx.x; i < CH; i += blockDim.x) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = s; cyc[blockIdx.x] = t1 - t0; }
}
int main() {
    const int nblk = 1024;   // 16 MB total: fits the 8 x 4 MB L2s
    double *buf, *out; unsigned long long *cyc;
    CK(hipMalloc(&buf, sizeof(double) * CH * nblk)); CK(hipMalloc(&out, sizeof(double) * nblk)); CK(hipMalloc(&cyc, 8 * nblk));
    std::vector<unsigned long long> h(nblk);
    for (int shift : {0, 1, 8, 9}) {
        double tot = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipLaunchKernelGGL(kA, dim3(nblk), dim3(64), 0, 0, buf, (double)rep);
            hipLaunchKernelGGL(kB, dim3(nblk), dim3(64), 0, 0, buf, nblk, shift, out, cyc);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), cyc, 8 * nblk, hipMemcpyDeviceToHost));
            double m = 0; for (auto c : h) m += c; tot += m / nblk;
        }
        printf("shift %d: mean read time per block %.0f cycles (16 KB by one wave)\n", shift, tot / 20);
    }
    return 0;
}
