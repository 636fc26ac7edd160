// micro-benchmark: per-launch cost of a dependent chain of small kernels (hipGraph replay) by block size and block count,
// with and without one workgroup barrier + one dependent global round trip. What is the floor under a per-period launch?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_chain(const double *in, double *out, int n, int work) {
    __shared__ double sh[1024];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (work >= 1 && i < n) v = in[i];                  // one global round trip
    sh[threadIdx.x] = v;
    if (work >= 2) { __syncthreads(); v += sh[(threadIdx.x + 64) % blockDim.x]; }   // one barrier
    if (work >= 3 && i < n) v += in[(i + 4097) % n];    // a second, dependent-ish round trip
    if (i < n) out[i] = v + 1.0;
}
int main() {
    const int n = 1 << 20;
    double *a, *b;
    CK(hipMalloc(&a, 8 * n)); CK(hipMalloc(&b, 8 * n));
    CK(hipMemset(a, 0, 8 * n)); CK(hipMemset(b, 0, 8 * n));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int L = 300;
    for (int work = 0; work <= 3; work++)
        for (int bs : {256, 704, 1024})
            for (int nb : {64, 512, 2048}) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int l = 0; l < L; l++) hipLaunchKernelGGL(k_chain, dim3(nb), dim3(bs), 0, s, (l & 1) ? b : a, (l & 1) ? a : b, n, work);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < 5; r++) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("work %d  block %4d  blocks %4d : %.2f us per launch\n", work, bs, nb, 1e3 * ms / (5 * L));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
    return 0;
}
