// micro-benchmark + correctness check of the refinement step (rx_refine.hip.inc) for a lone wavefront (gfx950):
//   hipcc -O3 --offload-arch=gfx950 -o scripts/micro/refine_step scripts/micro/refine_step.hip && scripts/micro/refine_step
// A rate-matrix-like 41 x 41 system (last row ones, rhs = e_last); the kept inverse is that of a PERTURBED matrix, the start
// vector the perturbed system's solution (what an iteration two steps back leaves behind).  Checks the refined solution
// against a host solve and reports s_memtime ticks (100 MHz) per call and per correction.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../radex_emcee_amd/csrc/rx_kernel.hip.inc"

using namespace rxk;
constexpr int NL = 41, RB = Lay2<NL>::RB, CJ = Lay2<NL>::CJ;

__global__ __launch_bounds__(64, 1) void k_refine(const double *A, const double *Mi, const double *x0, double *xout, int *info,
                                                 unsigned long long *ticks, int reps)
{
    constexpr int S = lds_stride(NL);
    __shared__ __attribute__((aligned(16))) double sA[(NL + 1) * S];
    __shared__ __attribute__((aligned(16))) float sM[rf_minv_floats(NL)];
    const int lane = threadIdx.x;
    for (int i = lane; i < (NL + 1) * S; i += 64) { const int r = i / S, j = i % S; sA[i] = (r < NL && j < NL) ? A[r * NL + j] : 0.0; }
    for (int i = lane; i < rf_minv_floats(NL); i += 64) sM[i] = 0.0f;
    __syncthreads();
    for (int i = lane; i < NL * NL; i += 64) sM[rf_minv_index(NL, i / NL, i % NL)] = (float)Mi[i];
    __syncthreads();
    const bool isrow = lane < NL;
    const double x = isrow ? x0[lane] : 0.0, nb = (lane == NL - 1) ? -1.0 : 0.0;
    int steps = 0;
    bool ok = false;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma nounroll
    for (int r = 0; r < reps; ++r) {
        double arow[NL];
        float mrow[NL];
        int li = isrow ? lane : NL;
        asm volatile("" : "+v"(li));
        rf_load_row<NL>(sA, li, arow);                       // (per call, like an iteration of the kernel does)
        rf_load_minv<NL>(sM, li < NL ? li : NL - 1, mrow);
        double xx = x;
        asm volatile("" : "+v"(xx));
        double tot; ok = rf_refine<NL, false>(arow, sA, mrow, nb, xx, (1ull << NL) - 1ull, steps, tot);
        if (r == reps - 1 && isrow) xout[lane] = xx;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) { info[0] = ok; info[1] = steps; ticks[0] = t1 - t0; }
}

// 41 FMAs spread over NA accumulators (is a chain of DEPENDENT v_fmac_f64_dpp slower than independent ones?)
template <int NA>
__device__ __forceinline__ void rowdot_multi(const double (&row)[NL], const double (&vr)[3], double (&acc)[NA])
{
#pragma unroll
    for (int j = 0; j < NL; ++j) fma_bc(acc[j % NA], vr[j >> 4], row[j], j & 15);
}

// the step's parts on their own: 0 replication of a vector, 1 product (41 FMAs), 2 the high-word maximum, 3 the two loads
template <int PART>
__global__ __launch_bounds__(64, 1) void k_part(const double *A, double *xout, unsigned long long *ticks)
{
    constexpr int S = lds_stride(NL);
    __shared__ __attribute__((aligned(16))) double sA[(NL + 1) * S];
    __shared__ __attribute__((aligned(16))) float sM[rf_minv_floats(NL)];
    const int lane = threadIdx.x;
    for (int i = lane; i < (NL + 1) * S; i += 64) sA[i] = A[i % (NL * NL)];
    for (int i = lane; i < rf_minv_floats(NL); i += 64) sM[i] = (float)A[i % (NL * NL)];
    __syncthreads();
    double row[NL], x = A[lane], y = 0.0, xr[3] = {x, x, x};
    float mrow[NL], y32 = 0.0f, xr32[3] = {(float)x, (float)x, (float)x};
    rf_load_row<NL>(sA, lane % NL, row);
    rf_load_minv<NL>(sM, lane % NL, mrow);
    unsigned long long t0, t1;
    int acc = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma nounroll
    for (int r = 0; r < 200; ++r) {
        if (PART == 0) { rf_replicate(x, xr); x = xr[0] + xr[1] + xr[2]; }
        if (PART == 1) { rf_rowdot<NL>(row, xr, y); }
        if (PART == 2) { acc += rf_hi_max(x, true); x += (double)acc; }
        if (PART == 7) { rf_rowdot32<NL>(mrow, xr32, y32); }
        if (PART == 8) { rf_replicate32(y32, xr32); y32 = xr32[0] + xr32[1] + xr32[2]; }
        if (PART == 4) { double a1[1] = {y}; rowdot_multi<1>(row, xr, a1); y = a1[0]; }
        if (PART == 5) { double a2[2] = {y, 0.0}; rowdot_multi<2>(row, xr, a2); y = a2[0] + a2[1]; }
        if (PART == 6) { double a4[4] = {y, 0.0, 0.0, 0.0}; rowdot_multi<4>(row, xr, a4); y = (a4[0] + a4[1]) + (a4[2] + a4[3]); }
        if (PART == 3) {
            int li = lane % NL;
            asm volatile("" : "+v"(li));
            rf_load_row<NL>(sA, li, row); rf_load_minv<NL>(sM, li, mrow);
#pragma unroll
            for (int j = 0; j < NL; ++j) asm volatile("" : "+v"(row[j]), "+v"(mrow[j]));
        }
        asm volatile("" : "+v"(x), "+v"(y), "+v"(y32));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) ticks[0] = t1 - t0;
    double sum = x + y + xr[0] + xr[1] + xr[2] + acc + y32 + xr32[0];
    for (int j = 0; j < NL; ++j) sum += row[j] + mrow[j];
    xout[lane] = sum;
}

static void solve(std::vector<double> A, std::vector<double> &X, int n, int nrhs)   // Gauss-Jordan with partial pivoting, A X = B in place
{
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (fabs(A[i * n + k]) > fabs(A[p * n + k])) p = i;
        for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[p * n + j]);
        for (int j = 0; j < nrhs; ++j) std::swap(X[k * nrhs + j], X[p * nrhs + j]);
        const double d = A[k * n + k];
        for (int j = 0; j < n; ++j) A[k * n + j] /= d;
        for (int j = 0; j < nrhs; ++j) X[k * nrhs + j] /= d;
        for (int i = 0; i < n; ++i) if (i != k) {
            const double f = A[i * n + k];
            for (int j = 0; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
            for (int j = 0; j < nrhs; ++j) X[i * nrhs + j] -= f * X[k * nrhs + j];
        }
    }
}

int main()
{
    const int n = NL;
    srand(7);
    auto rnd = []() { return rand() / (double)RAND_MAX; };
    std::vector<double> A(n * n), Ap(n * n);
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < n; ++j) if (j != i) { A[i * n + j] = -rnd() * pow(10.0, -3 * rnd()); s += fabs(A[i * n + j]); }
        A[i * n + i] = s * (1.0 + rnd());
    }
    for (int j = 0; j < n; ++j) A[(n - 1) * n + j] = 1.0;
    for (double eps : {1e-2, 1e-3, 1e-5}) {
        Ap = A;
        for (int i = 0; i + 1 < n; ++i)                       // the tridiagonal entries move, like the radiative ones
            for (int j = (i ? i - 1 : 0); j <= i + 1 && j < n; ++j) Ap[i * n + j] *= 1.0 + eps * (2 * rnd() - 1);
        std::vector<double> I(n * n, 0.0), b(n, 0.0), bp(n, 0.0);
        for (int i = 0; i < n; ++i) I[i * n + i] = 1.0;
        b[n - 1] = bp[n - 1] = 1.0;
        solve(Ap, I, n, n);                                   // I <- inverse of the perturbed matrix
        solve(Ap, bp, n, 1);                                  // the start vector
        solve(A, b, n, 1);                                    // the answer
        double *dA, *dM, *dx0, *dx; int *dinfo; unsigned long long *dt;
        hipMalloc(&dA, n * n * 8); hipMalloc(&dM, n * n * 8); hipMalloc(&dx0, n * 8); hipMalloc(&dx, n * 8); hipMalloc(&dinfo, 8); hipMalloc(&dt, 8);
        hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice); hipMemcpy(dM, I.data(), n * n * 8, hipMemcpyHostToDevice);
        hipMemcpy(dx0, bp.data(), n * 8, hipMemcpyHostToDevice);
        for (int reps : {1, 101}) {
            hipLaunchKernelGGL(k_refine, dim3(1), dim3(64), 0, 0, dA, dM, dx0, dx, dinfo, dt, reps);
            std::vector<double> x(n); int info[2]; unsigned long long t;
            hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost); hipMemcpy(info, dinfo, 8, hipMemcpyDeviceToHost); hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
            double err = 0, xm = 0, e0 = 0;
            for (int i = 0; i < n; ++i) { err = fmax(err, fabs(x[i] - b[i])); xm = fmax(xm, fabs(b[i])); e0 = fmax(e0, fabs(bp[i] - b[i])); }
            printf("perturbation %.0e reps %3d: accepted %d after %d corrections; start error %.2e -> %.2e of max|x|; %.1f ticks per call, %.1f per correction\n",
                   eps, reps, info[0], info[1], e0 / xm, err / xm, t / (double)reps, t / (double)reps / (info[1] ? info[1] : 1));
        }
    }
    {
        double *dA, *dx; unsigned long long *dt, t;
        hipMalloc(&dA, n * n * 8); hipMalloc(&dx, 64 * 8); hipMalloc(&dt, 8);
        hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
#define PART(P, WHAT) hipLaunchKernelGGL(k_part<P>, dim3(1), dim3(64), 0, 0, dA, dx, dt); hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost); \
        printf("%-44s %.1f ticks\n", WHAT, t / 200.0);
        PART(0, "replication of a vector (6 copies + 6 permlane swaps)")
        PART(1, "product (41 v_fmac_f64_dpp)")
        PART(2, "high-word maximum of a vector")
        PART(3, "a row of A and of the kept inverse from LDS")
        PART(7, "single-precision product (41 v_fmac_f32_dpp)")
        PART(8, "replication of a single-precision vector")
        PART(4, "product, one statement per FMA, 1 accumulator")
        PART(5, "product, one statement per FMA, 2 accumulators")
        PART(6, "product, one statement per FMA, 4 accumulators")
    }
    return 0;
}
