// micro-test: accuracy of v_rcp_f64 and of the one-correction quotient used by the LU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, const double *y, double *r0, double *q1, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double rc = __builtin_amdgcn_rcp(x[i]);
    r0[i] = rc;
    double q = y[i] * rc;
    double e = fma(-q, x[i], y[i]);
    q1[i] = fma(e, rc, q);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), y(n), r0(n), q1(n);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); };
    for (int i = 0; i < n; ++i) { x[i] = std::pow(10.0, -30 + 40 * rnd()) * (rnd() < 0.5 ? -1 : 1) * (1 + rnd()); y[i] = std::pow(10.0, -30 + 40 * rnd()) * (1 + rnd()); }
    double *dx, *dy, *dr, *dq;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&dq, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, dr, dq, n);
    hipMemcpy(r0.data(), dr, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q1.data(), dq, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0; long exact = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        m0 = std::fmax(m0, (double)fabsl(((long double)r0[i] - t) / t));
        long double qq = (long double)y[i] / (long double)x[i];
        m1 = std::fmax(m1, (double)fabsl(((long double)q1[i] - qq) / qq));
        if (q1[i] == y[i] / x[i]) exact++;
    }
    printf("v_rcp_f64 max rel err %.3e (2^%.1f); quotient after one correction: max rel err %.3e, bit-identical to IEEE division in %.4f%% of %d cases\n",
           m0, std::log2(m0), m1, 100.0 * exact / n, n);
    return 0;
}
