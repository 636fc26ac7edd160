// ref_dual_shim.cpp — TEST INFRASTRUCTURE. A C-ABI window onto the REFERENCE's own C++
// dual-number classes (ForwardDiff.jl/benchmarks/cpp/benchmarks.h:10-118, dual1.cpp..dual5.cpp),
// which are compiled from /root/reference in place (see Makefile target `ref`).
// Used only by tests/test_oracle_dual_vs_ref.py to pin the oracle's dual arithmetic.
#include "benchmarks.h"

template <typename D> static D rosen(const std::vector<D> &x) {
    // same test function as the reference's known-answer check (benchmarks.cpp:24-34)
    double a = 100.0, b = 1.0;
    D result(0.0);
    for (size_t i = 0; i + 1 < x.size(); i++) {
        D t1 = b - x[i];
        D t2 = x[i + 1] - x[i] * x[i];
        result = result + t1 * t1 + a * t2 * t2;
    }
    return result;
}

extern "C" {
// op: 0 add, 1 sub, 2 mul, 3 real*dual (x[0]*y), 4 real-dual (x[0]-y), 5 sqrt(x), 6 exp(x)
void ref_dual1_op(int op, const double *x, const double *y, double *out) {
    Dual1 a(x[0], x[1]), b(y[0], y[1]), r;
    switch (op) {
    case 0: r = a + b; break;
    case 1: r = a - b; break;
    case 2: r = a * b; break;
    case 3: r = x[0] * b; break;
    case 4: r = x[0] - b; break;
    case 5: r = sqrt(a); break;
    case 6: r = exp(a); break;
    }
    out[0] = r.real; out[1] = r.eps1;
}
void ref_dual3_op(int op, const double *x, const double *y, double *out) {
    Dual3 a(x[0], x[1], x[2], x[3]), b(y[0], y[1], y[2], y[3]), r;
    switch (op) {
    case 0: r = a + b; break;
    case 1: r = a - b; break;
    case 2: r = a * b; break;
    case 3: r = x[0] * b; break;
    case 4: r = x[0] - b; break;
    case 5: r = sqrt(a); break;
    case 6: r = exp(a); break;
    }
    out[0] = r.real; out[1] = r.eps1; out[2] = r.eps2; out[3] = r.eps3;
}
// rosenbrock gradient through the reference's chunk-1 `gradient` template (benchmarks.h:120-131)
void ref_rosenbrock_grad1(const double *in, int n, double *out) {
    std::vector<double> input(in, in + n), result(n);
    std::vector<Dual1> dv(n);
    gradient<rosen<Dual1>>(result, dv, input);
    for (int i = 0; i < n; i++) out[i] = result[i];
}
}
