// calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the tangent kernels:
// a known byte count is read once (buffer >> Infinity Cache) with 8 B/lane, 16 B/lane, and 16 B/lane in
// 256-byte row pieces (16 lanes per row, 4 rows per wave, rows visited in a shuffled order).
// Run: rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib ; expected bytes per kernel are printed.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_read8(const double *p, size_t n, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void k_read16(const double2 *p, size_t n, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
// rows of 16 double2 (256 B); wave = 4 rows x 16 lanes; row order shuffled by a multiplicative hash within the buffer
__global__ void k_rows16(const double2 *p, size_t nrows, double *out) {
    const int lane = threadIdx.x & 63, nl = lane & 15, rl = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    double s = 0.0;
    for (size_t g = wave; g * 4 < nrows; g += nw) {
        const size_t r = ((g * 4 + rl) * 2654435761ull) % nrows;   // odd multiplier, nrows a power of two: a permutation
        const double2 v = p[r * 16 + nl];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}
int main() {
    const size_t bytes = 1ull << 30;   // 1 GiB
    double *buf, *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, buf, bytes / 8, out);
        hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const double2 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_rows16, dim3(4096), dim3(256), 0, 0, (const double2 *)buf, bytes / 256, out);
    }
    CK(hipDeviceSynchronize());
    printf("each kernel reads %zu bytes = %.1f KB once\n", bytes, bytes / 1024.0);
    return 0;
}
