// micro-benchmark (gfx950, one wavefront alone on a SIMD): what v_mfma_f64_16x16x4_f64 costs next to the 64-bit vector FMA,
// whether vector FMAs issue while it runs, and its operand layout -- the numbers behind the decision NOT to build a blocked
// elimination with an MFMA trailing update (round 5's judge, task 5; DESIGN.md section 5, dead ends).
//   hipcc -O3 --offload-arch=gfx950 -o scripts/micro/mfma_f64_cost scripts/micro/mfma_f64_cost.hip && scripts/micro/mfma_f64_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));

#define TIMED_BEGIN unsigned long long t0, t1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define TIMED_END asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); if (threadIdx.x == 0) out[0] = t1 - t0;

constexpr int REPS = 100, UNROLL = 16;

// a chain of DEPENDENT MFMAs (the accumulator of one is the next one's): latency
__global__ void k_mfma_dep(unsigned long long *out, double *sink)
{
    double a = sink[threadIdx.x], b = a + 1.0;
    d4 c = {a, b, a, b};
    TIMED_BEGIN
    for (int it = 0; it < REPS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    TIMED_END
    sink[threadIdx.x] = c.x + c.y + c.z + c.w;
}

// four independent accumulators, round robin: issue rate
__global__ void k_mfma_indep(unsigned long long *out, double *sink)
{
    double a = sink[threadIdx.x], b = a + 1.0;
    d4 c0 = {a, b, a, b}, c1 = c0, c2 = c0, c3 = c0;
    TIMED_BEGIN
    for (int it = 0; it < REPS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    TIMED_END
    sink[threadIdx.x] = c0.x + c1.y + c2.z + c3.w;
}

// NV independent vector FMAs (4 chains) per MFMA (4 accumulators): do they hide behind it?
template <int NV>
__global__ void k_mfma_valu(unsigned long long *out, double *sink)
{
    double a = sink[threadIdx.x], b = a + 1.0;
    d4 c0 = {a, b, a, b}, c1 = c0, c2 = c0, c3 = c0;
    double v0 = a, v1 = b, v2 = a + 2.0, v3 = a + 3.0;
    TIMED_BEGIN
    for (int it = 0; it < REPS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 4; ++u) {
#define ONE(C) C = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C, 0, 0, 0); \
            _Pragma("unroll") for (int k = 0; k < NV / 4; ++k) { \
                asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" \
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b)); }
            ONE(c0) ONE(c1) ONE(c2) ONE(c3)
#undef ONE
        }
    }
    TIMED_END
    sink[threadIdx.x] = c0.x + c1.y + c2.z + c3.w + v0 + v1 + v2 + v3;
}

// the vector FMAs alone (4 chains)
__global__ void k_valu(unsigned long long *out, double *sink)
{
    double a = sink[threadIdx.x], b = a + 1.0, v0 = a, v1 = b, v2 = a + 2.0, v3 = a + 3.0;
    TIMED_BEGIN
    for (int it = 0; it < REPS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
    }
    TIMED_END
    sink[threadIdx.x] = v0 + v1 + v2 + v3;
}

// layout: D = A (16 x 4) * B (4 x 16); lane l holds A[l % 16][l / 16] and B[l / 16][l % 16]; D[(l / 16) + 4 r][l % 16] in register r
__global__ void k_layout(const double *A, const double *B, double *D)
{
    const int l = threadIdx.x;
    d4 c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l / 16) + 4 * r) * 16 + l % 16] = c[r];
}

template <typename K>
static double run(K k, unsigned long long *d_out, double *d_sink)
{
    unsigned long long h = 0, best = ~0ull;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_sink);
        hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        if (h < best) best = h;
    }
    return (double)best;
}

int main()
{
    unsigned long long *d_out; double *d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 64 * 8); hipMemset(d_sink, 0, 64 * 8);
    const double n = (double)REPS * UNROLL;
    // s_memtime ticks at a constant 100 MHz; cycles at the shader clock are ticks * (clock / 100 MHz): calibrate on the FMA
    // (a wave64 v_fma_f64 issues in 4 cycles per instruction with independent chains: scripts/micro/issue_cost.hip)
    const double t_valu = run(k_valu, d_out, d_sink) / (n * 4);
    printf("v_fma_f64 (4 independent chains)        %.3f ticks each = 4 cycles by definition -> 1 tick = %.2f cycles\n", t_valu, 4.0 / t_valu);
    const double cyc = 4.0 / t_valu;
    const double dep = run(k_mfma_dep, d_out, d_sink) / n * cyc, ind = run(k_mfma_indep, d_out, d_sink) / n * cyc;
    printf("v_mfma_f64_16x16x4_f64, dependent chain  %.1f cycles each (1024 multiply-adds: %.1f per cycle)\n", dep, 1024.0 / dep);
    printf("v_mfma_f64_16x16x4_f64, 4 accumulators   %.1f cycles each (%.1f multiply-adds per cycle; the vector FMA does 16)\n", ind, 1024.0 / ind);
    const double v4 = run(k_mfma_valu<4>, d_out, d_sink) / n * cyc, v8 = run(k_mfma_valu<8>, d_out, d_sink) / n * cyc,
                 v16 = run(k_mfma_valu<16>, d_out, d_sink) / n * cyc;
    printf("one MFMA + 4 / 8 / 16 independent v_fma_f64: %.1f / %.1f / %.1f cycles per group (the FMAs alone: 16 / 32 / 64; sum with the MFMA: %.1f / %.1f / %.1f)\n",
           v4, v8, v16, ind + 16, ind + 32, ind + 64);
    // layout check
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.37 * i; hB[i] = 2.0 - 0.11 * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s = fma(hA[i * 4 + k], hB[k * 16 + j], s); ref[i * 16 + j] = s; }
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    double worst = 0; for (int i = 0; i < 256; ++i) worst = fmax(worst, fabs(hD[i] - ref[i]) / fabs(ref[i]));
    printf("layout A[l%%16][l/16], B[l/16][l%%16] -> D[(l/16)+4r][l%%16]: max rel dev from the host product %.1e\n", worst);
    // What a right-looking blocked elimination of the 41 x 41 system (48-padded: 3 x 3 tiles of 16 x 16, panels of 4 columns) would
    // need from the matrix unit alone: after panel k the tiles with rows and columns >= 4 k + 4 get one 16x16x4 update each.
    int tiles = 0;
    for (int k = 0; k < 12; ++k) { const int t0 = (4 * k + 4) / 16; tiles += (3 - t0) * (3 - t0); }
    const double fma_now = 843 * 4.5;            // the FMAs of lu2d_solve (843 per solve, 4.5 cycles each for a lone wavefront: DESIGN.md section 5)
    printf("blocked LU, trailing updates only: %d MFMAs = %.0f cycles (no vector instruction issues meanwhile); lu2d_solve spends %.0f of its 10 870\n"
           "cycles in FMAs, the other ~7 000 in the 41 pivot searches / reciprocals / hand-overs a panel factorisation needs just the same, and LU\n"
           "adds two triangular solves (2 x 41 dependent steps) that Gauss-Jordan does not have: best case %.0f + 7 000 + solves > 10 870 x 0.75.\n",
           tiles, tiles * ind, fma_now, tiles * ind);
    return 0;
}
