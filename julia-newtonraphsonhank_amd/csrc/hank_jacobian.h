// hank_jacobian.h — the household block's sequence-space Jacobian at a STATIONARY primal from its Toeplitz structure
// (SteadyStateJacobian.jl:187-256, :293-323, :358-387; Boehl 2021 / Auclert et al. 2021 "fake news").
//
// At the steady state the response of the policy at t to a shock to household input k at s depends on s - t only, and is
// zero for t > s: ONE backward tangent sweep per input, seeded at the LAST period (the reference's JBI, :240-243), gives every
// lag: Y_j = dpol_{P-1-j}. Each Y_j perturbs the lottery of its period: the impulse iota_j = (d Lambda / d pol [Y_j]) D_ss —
// one single-period forward push for all P*n_hh of them at once (k_fn_impulse). The aggregate u periods later is
// E_u . iota_j with the expectation vectors E_0 = pol_ss, E_{u+1} = T' E_u (T' = the transposed forward step: mix with
// Pi, then read at the lottery's two targets — the shape of the EGM expectation; the reference gets these from Zygote
// pullbacks through ForwardIteration, :249-253). F[u, j] = E_u . iota_j (the reference's `helper`, :300-305) is ONE
// (P x G) x (G x P n_hh) product, Dv[j] = Y_j . D_ss the direct term, and the host finishes with the recursion
// J[t, s] = J[t-1, s-1] + F[t, s] (:363-371). 598 unit tangents become 2.
#pragma once
#include "hank_kernels.h"

namespace hank {

// dpol [P][G][N] (direction fastest) -> dpT [G][P*N], column n' = t*N + k
__global__ void k_fn_transpose(const double *__restrict__ dpol, int P, int G, int N, double *__restrict__ dpT) {
    __shared__ double tile[32][33];
    const int g0 = blockIdx.x * 32, t0 = blockIdx.y * 32, k = blockIdx.z;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;      // 256 threads: 8 rows per pass
    for (int q = ly; q < 32; q += 8) {
        const int t = t0 + q, g = g0 + lx;
        tile[q][lx] = (t < P && g < G) ? dpol[((size_t)t * G + g) * N + k] : 0.0;
    }
    __syncthreads();
    for (int q = ly; q < 32; q += 8) {
        const int g = g0 + q, t = t0 + lx;
        if (t < P && g < G) dpT[(size_t)g * ((size_t)P * N) + (size_t)t * N + k] = tile[lx][q];
    }
}

// iota[(e2*na + r)][n'] = sum_e mid[r, e][n'] Pi[e, e2],  mid[r, e] = sum_{j in first segment of r} g_j dp_j - sum_{second} g_j dp_j
// (the weight tangent of source j moves g_j dp_j = (dp_j / gap) D_ss[j] from its lower to its upper target, ForwardIteration.jl:64-73;
// clamped sources carry no weight tangent). One block per target row and 256 columns.
__global__ void __launch_bounds__(256) k_fn_impulse(Consts c, Record R, const double *__restrict__ dpT, int NP, double *__restrict__ iota) {
    const int r = blockIdx.x, n = blockIdx.y * 256 + threadIdx.x;
    if (n >= NP) return;
    const int ne = c.n_e, na = c.n_a;
    double mid[16];
    for (int e = 0; e < ne; e++) {
        const int4 sg = R.seg[(size_t)e * na + r];
        const int s0 = max(sg.x, 0), s1 = sg.y, s2 = min(sg.z, na);
        double acc = 0.0;
        for (int j = s0; j < s2; j++) {
            const double g = R.lwg[(size_t)e * na + j].y, v = g * dpT[((size_t)e * na + j) * NP + n];
            acc = j < s1 ? acc + v : acc - v;
        }
        mid[e] = acc;
    }
    for (int e2 = 0; e2 < ne; e2++) {
        double s = 0.0;
        for (int e = 0; e < ne; e++) s += mid[e] * c.Pi[e + ne * e2];
        iota[((size_t)e2 * na + r) * NP + n] = s;
    }
}

// E_next[j, e] = (1 - w_j) U[lo_j, e] + w_j U[lo_j + 1, e],  U[r, e] = sum_e2 Pi[e, e2] E[r, e2]   (the transposed forward step)
__global__ void k_fn_expect(Consts c, Record R, const double *__restrict__ E, double *__restrict__ En) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= c.G) return;
    const int ne = c.n_e, na = c.n_a, e = idx / na;
    const int lo = R.lo[idx];
    const double w = R.lw[idx];
    double u0 = 0.0, u1 = 0.0;
    for (int e2 = 0; e2 < ne; e2++) {
        const double p = c.Pi[e + ne * e2];
        u0 += p * E[(size_t)e2 * na + lo];
        u1 += p * E[(size_t)e2 * na + lo + 1];
    }
    En[idx] = (1.0 - w) * u0 + w * u1;
}

// Cp[z][M][N] = A[M][K-slice z] * B[K-slice z][N]: a plain fp64 product (row-major operands), 64 x 64 tiles, 4 x 4 per thread,
// K split over blockIdx.z so that a 5 x 10-tile product still fills the chip; k_fn_reduce sums the slices in fixed order
__global__ void __launch_bounds__(256) k_fn_gemm(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ Cp, int M, int N, int K, int kchunk) {
    __shared__ double As[16][68], Bs[16][64];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64, z = blockIdx.z;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int kb = z * kchunk, ke = min(K, kb + kchunk);
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
    for (int k0 = kb; k0 < ke; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int rr = (tid >> 4) + 16 * i, kk = tid & 15, m = m0 + rr, k = k0 + kk;
            As[kk][rr] = (m < M && k < ke) ? A[(size_t)m * K + k] : 0.0;
            const int kk2 = (tid >> 6) + 4 * i, cc = tid & 63, k2 = k0 + kk2, nn = n0 + cc;
            Bs[kk2][cc] = (k2 < ke && nn < N) ? B[(size_t)k2 * N + nn] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = m0 + ty * 4 + i, nn = n0 + tx * 4 + j;
            if (m < M && nn < N) Cp[((size_t)z * M + m) * N + nn] = acc[i][j];
        }
}
__global__ void k_fn_reduce(const double *__restrict__ Cp, int MN, int S, double *__restrict__ C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= MN) return;
    double s = 0.0;
    for (int z = 0; z < S; z++) s += Cp[(size_t)z * MN + idx];
    C[idx] = s;
}
// dxhh (n_hh, P, N = n_hh) column-major: direction k = a unit shock to input k in the LAST period
__global__ void k_fn_seed(double *dxhh, int n_hh, int P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_hh * P * n_hh) return;
    const int k = idx % n_hh, t = (idx / n_hh) % P, n = idx / (n_hh * P);
    dxhh[idx] = (k == n && t == P - 1) ? 1.0 : 0.0;
}

}  // namespace hank
