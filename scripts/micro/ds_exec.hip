// micro-benchmark: what does a lone wavefront's LDS store cost when half of EXEC is off? (gfx950)
// The elimination's column hand-over stores 24 bytes per lane every other step, and only two of the four 16-lane groups' data is read.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE, int KIND>
__global__ void k(unsigned long long *out, double *sink)
{
    __shared__ double buf[64 * 4 + 64];
    double a = sink[threadIdx.x] + 1.5, b = a + 1.0, c = a + 2.0;
    const unsigned ad2 = (unsigned)(size_t)buf + threadIdx.x * 16, ad1 = (unsigned)(size_t)buf + 64 * 16 + threadIdx.x * 8;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 w = {a, b};
    unsigned long long t0, t1;
    if (MODE == 1) asm volatile("s_mov_b32 exec_hi, 0");          // lanes 0..31
    if (MODE == 2) asm volatile("s_mov_b32 exec_lo, 0");          // lanes 32..63
    if (MODE == 3) asm volatile("s_mov_b64 exec, 0xffff");        // lanes 0..15
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 100; ++it) {
        if (KIND == 0) asm volatile(REP64("ds_write_b128 %0, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %2, %2, %2, %2\n\t") : : "v"(ad2), "v"(w), "v"(c) : "memory");
        if (KIND == 1) asm volatile(REP64("ds_write_b64 %0, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %2, %2, %2, %2\n\t") : : "v"(ad1), "v"(a), "v"(c) : "memory");
        if (KIND == 2) asm volatile(REP64("ds_write_b128 %0, %1\n\tds_write_b64 %3, %4\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %2, %2, %2, %2\n\t") : : "v"(ad2), "v"(w), "v"(c), "v"(ad1), "v"(a) : "memory");
        if (KIND == 3) asm volatile(REP64("v_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %2, %2, %2, %2\n\t") : : "v"(ad2), "v"(w), "v"(c) : "memory");
        if (KIND == 4) asm volatile(REP64("s_mov_b32 exec_hi, 0\n\tds_write_b128 %0, %1\n\tds_write_b64 %3, %4\n\ts_mov_b32 exec_hi, -1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %2, %2, %2, %2\n\t") : : "v"(ad2), "v"(w), "v"(c), "v"(ad1), "v"(a) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_mov_b64 exec, -1");
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = a + b + c + buf[threadIdx.x];
}
template <int KIND> void run(const char *what, unsigned long long *d_out, double *d_sink)
{
    unsigned long long h[4];
    hipLaunchKernelGGL((k<0, KIND>), dim3(1), dim3(64), 0, 0, d_out, d_sink); hipMemcpy(&h[0], d_out, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k<1, KIND>), dim3(1), dim3(64), 0, 0, d_out, d_sink); hipMemcpy(&h[1], d_out, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k<2, KIND>), dim3(1), dim3(64), 0, 0, d_out, d_sink); hipMemcpy(&h[2], d_out, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k<3, KIND>), dim3(1), dim3(64), 0, 0, d_out, d_sink); hipMemcpy(&h[3], d_out, 8, hipMemcpyDeviceToHost);
    printf("%-46s ticks per group: all lanes %.1f | lanes 0-31 %.1f | lanes 32-63 %.1f | lanes 0-15 %.1f\n", what,
           h[0] / 6400.0, h[1] / 6400.0, h[2] / 6400.0, h[3] / 6400.0);
}
int main()
{
    unsigned long long *d_out; double *d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 64 * 8); hipMemset(d_sink, 0, 64 * 8);
    run<3>("2 x v_fma_f64 (the filler alone)", d_out, d_sink);
    run<0>("ds_write_b128 + 2 x v_fma_f64", d_out, d_sink);
    run<1>("ds_write_b64 + 2 x v_fma_f64", d_out, d_sink);
    run<2>("ds_write_b128 + ds_write_b64 + 2 x v_fma_f64", d_out, d_sink);
    run<4>("the same with exec_hi = 0 around the stores", d_out, d_sink);
    return 0;
}
