// micro-benchmark: a lone wavefront loading 33 doubles per lane from LDS -- 33 x ds_read_b64 against 16 x ds_read_b128 + 1 x ds_read_b64
// (the matrix load of Phase C), each followed by one wait and a few dependent FMAs.  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(unsigned long long *out, double *sink)
{
    __shared__ __attribute__((aligned(16))) double buf[64 * 48];
    for (int i = threadIdx.x; i < 64 * 48; i += 64) buf[i] = (double)i;
    __syncthreads();
    const unsigned ad = (unsigned)(size_t)buf + (threadIdx.x & 15) * 336 + (threadIdx.x >> 4) * 8;   // row stride 42 doubles, group offset
    const unsigned adc = (unsigned)(size_t)buf + (threadIdx.x & 15) * 400 + (threadIdx.x >> 4) * 96; // contiguous 11 doubles per group, 16-byte aligned, conflict-free lane stride
    double acc = sink[threadIdx.x];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 200; ++it) {
        double v[33]; d2 w[16];
        if (KIND == 0) {
#define R64(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"(ad), "i"(32 * (i % 11) + 4032 * (i / 11)) : "memory");
            R64(0) R64(1) R64(2) R64(3) R64(4) R64(5) R64(6) R64(7) R64(8) R64(9) R64(10) R64(11) R64(12) R64(13) R64(14) R64(15) R64(16)
            R64(17) R64(18) R64(19) R64(20) R64(21) R64(22) R64(23) R64(24) R64(25) R64(26) R64(27) R64(28) R64(29) R64(30) R64(31) R64(32)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i = 0; i < 33; ++i) asm volatile("" : "+v"(v[i]));
            for (int i = 0; i < 33; ++i) acc += v[i];
        } else {
#define R128(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[i]) : "v"(adc), "i"(16 * (i % 5) + 6400 * (i / 5)) : "memory");
            R128(0) R128(1) R128(2) R128(3) R128(4) R128(5) R128(6) R128(7) R128(8) R128(9) R128(10) R128(11) R128(12) R128(13) R128(14)
            asm volatile("ds_read_b64 %0, %1 offset:80" : "=v"(v[0]) : "v"(adc) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:6480" : "=v"(v[1]) : "v"(adc) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:12880" : "=v"(v[2]) : "v"(adc) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i = 0; i < 15; ++i) asm volatile("" : "+v"(w[i]));
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
            for (int i = 0; i < 15; ++i) acc += w[i].x + w[i].y;
            acc += v[0] + v[1] + v[2];
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main()
{
    unsigned long long *d_out, h; double *d_sink;
    (void)hipMalloc(&d_out, 8); (void)hipMalloc(&d_sink, 64 * 8); (void)hipMemset(d_sink, 0, 64 * 8);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d_out, d_sink); (void)hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    printf("33 x ds_read_b64 (strided, as Phase C) + wait + 33 adds: %.1f ticks per round\n", h / 200.0);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d_out, d_sink); (void)hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    printf("15 x ds_read_b128 + 3 x ds_read_b64 (contiguous per group) + wait + 33 adds: %.1f ticks per round\n", h / 200.0);
    return 0;
}
