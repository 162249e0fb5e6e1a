// micro-benchmark: does a wave64 VALU instruction get cheaper when whole 16-lane quarters of EXEC are off? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE>
__global__ void k(unsigned long long *out, double *sink)
{
    double a = sink[threadIdx.x] + 1.5, b = a + 1.0, c = a + 2.0, d = a + 3.0;
    unsigned long long t0, t1;
    if (MODE == 1) asm volatile("s_mov_b64 exec, 0xffffffff");
    if (MODE == 2) asm volatile("s_mov_b64 exec, 0xffff");
    if (MODE == 3) asm volatile("s_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, 0xffff");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 100; ++it)
        asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %2, %2, %3, %0\n\tv_fma_f64 %3, %3, %0, %1\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_mov_b64 exec, -1");
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = a + b + c + d;
}
int main()
{
    unsigned long long *d_out, h; double *d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 64 * 8); hipMemset(d_sink, 0, 64 * 8);
    const char *names[] = {"exec = all 64 lanes", "exec = lanes 0-31", "exec = lanes 0-15", "exec = lanes 0-47"};
    for (int m = 0; m < 4; ++m) {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d_out, d_sink);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d_out, d_sink);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d_out, d_sink);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d_out, d_sink);
        hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        printf("%-22s %.2f ticks per v_fma_f64\n", names[m], (double)h / (100.0 * 64 * 4));
    }
    return 0;
}
