// micro-benchmark: cycles per instruction for one wavefront alone on a SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define TEST(NAME, BODY, PER)                                                                    \
    __global__ void NAME(unsigned long long *out, double *sink, int p)                            \
    {                                                                                             \
        double a = sink[threadIdx.x], b = a + 1.0, c = a + 2.0, d = a + 3.0, e = a + 4.0;         \
        unsigned u = (unsigned)threadIdx.x; int sp = __builtin_amdgcn_readfirstlane(p);           \
        unsigned long long t0, t1;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");              \
        for (int it = 0; it < 100; ++it) { asm volatile(REP64(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(u) : "s"(sp) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "vcc"); } \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");              \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                   \
        sink[threadIdx.x] = a + b + c + d + e + u;                                                \
    }                                                                                             \
    static const double NAME##_per = PER;
TEST(k_fma_indep, "v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %2, %2, %3, %4\n\tv_fma_f64 %3, %3, %4, %0\n\t", 4)
TEST(k_fma_dep, "v_fma_f64 %0, %0, %1, %0\n\t", 1)
TEST(k_fma_dep2, "v_fma_f64 %0, %0, %1, %0\n\tv_fma_f64 %2, %2, %1, %2\n\t", 2)
TEST(k_fmac_sgpr, "v_fmac_f64 %0, s[40:41], %1\n\tv_fmac_f64 %2, s[42:43], %1\n\tv_fmac_f64 %3, s[44:45], %1\n\tv_fmac_f64 %4, s[46:47], %1\n\t", 4)
TEST(k_readlane, "v_readlane_b32 s40, %5, %6\n\tv_readlane_b32 s41, %5, %6\n\tv_readlane_b32 s42, %5, %6\n\tv_readlane_b32 s43, %5, %6\n\t", 4)
TEST(k_group, "v_readlane_b32 s40, %5, %6\n\tv_readlane_b32 s41, %5, %6\n\tv_readlane_b32 s42, %5, %6\n\tv_readlane_b32 s43, %5, %6\n\tv_readlane_b32 s44, %5, %6\n\tv_readlane_b32 s45, %5, %6\n\tv_readlane_b32 s46, %5, %6\n\tv_readlane_b32 s47, %5, %6\n\tv_fmac_f64 %0, s[40:41], %4\n\tv_fmac_f64 %1, s[42:43], %4\n\tv_fmac_f64 %2, s[44:45], %4\n\tv_fmac_f64 %3, s[46:47], %4\n\t", 12)
TEST(k_mov32, "v_add_u32 %5, %5, %5\n\t", 1)
TEST(k_add_indep, "v_add_f64 %0, %0, %1\n\tv_add_f64 %2, %2, %1\n\tv_add_f64 %3, %3, %1\n\tv_add_f64 %4, %4, %1\n\t", 4)
TEST(k_cndmask, "v_cndmask_b32 %5, %5, %5, vcc\n\t", 1)
TEST(k_dppmax, "s_nop 1\n\tv_max_u32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t", 1)
TEST(k_salu, "s_mov_b32 s40, s41\n\t", 1)
TEST(k_snop, "s_nop 0\n\t", 1)
TEST(k_rcp, "v_rcp_f64 %0, %0\n\t", 1)
TEST(k_rcp_indep, "v_rcp_f64 %0, %1\n\tv_rcp_f64 %2, %1\n\tv_rcp_f64 %3, %1\n\tv_rcp_f64 %4, %1\n\t", 4)
TEST(k_mul_dep, "v_mul_f64 %0, %0, %1\n\t", 1)
TEST(k_cmp_ff1, "v_cmp_eq_u32 vcc, %5, %5\n\ts_ff1_i32_b64 s40, vcc\n\tv_readlane_b32 s41, %5, s40\n\t", 3)
TEST(k_readfirst, "v_readfirstlane_b32 s40, %5\n\tv_readfirstlane_b32 s41, %5\n\tv_readfirstlane_b32 s42, %5\n\tv_readfirstlane_b32 s43, %5\n\t", 4)
TEST(k_readlane_imm, "v_readlane_b32 s40, %5, 7\n\tv_readlane_b32 s41, %5, 7\n\tv_readlane_b32 s42, %5, 7\n\tv_readlane_b32 s43, %5, 7\n\t", 4)
TEST(k_rl_fmac, "v_readlane_b32 s40, %5, %6\n\tv_readlane_b32 s41, %5, %6\n\tv_fmac_f64 %0, s[40:41], %4\n\t", 3)
TEST(k_writelane, "v_writelane_b32 %5, s40, 3\n\t", 1)
__global__ void k_lds(unsigned long long *out, double *sink, int p)
{
    __shared__ double buf[256];
    double a = sink[threadIdx.x];
    buf[threadIdx.x] = a; buf[threadIdx.x + 64] = a; __syncthreads();
    unsigned long long t0, t1, t2, t3;
    double2 acc = {0.0, 0.0};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    // 64 single-lane 8-byte stores
    if (threadIdx.x == p) {
#pragma unroll
        for (int i = 0; i < 64; ++i) { asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(0), "v"(a), "n"(8 * 64) : "memory"); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    // 64 broadcast 16-byte reads (all lanes same address), no dependence between them
    double2 v[8];
#pragma unroll
    for (int i = 0; i < 64; ++i) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i & 7]) : "v"(0), "n"(16 * 4) : "memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    // write-then-read round trip latency
    asm volatile("ds_write_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(0), "v"(a) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3)::"memory");
    for (int i = 0; i < 8; ++i) { acc.x += v[i].x; acc.y += v[i].y; }
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; }
    sink[threadIdx.x] = a + acc.x + acc.y;
}
#define RUN(NAME) do { if (which >= 0 && which != idx++) break; hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, d_out, d_sink, 3); hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost); \
    printf("%-14s %7.2f cycles/instr\n", #NAME, (double)h / (100.0 * 64.0 * NAME##_per)); } while (0)
int main(int argc, char **argv)
{
    int which = argc > 1 ? atoi(argv[1]) : -1; int idx = 0;
    unsigned long long *d_out, h; double *d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 64 * 8); hipMemset(d_sink, 0, 64 * 8);
    RUN(k_fma_indep); RUN(k_fma_indep); RUN(k_fma_dep); RUN(k_fma_dep2); RUN(k_fmac_sgpr); RUN(k_readlane); RUN(k_group); RUN(k_mov32);
    RUN(k_add_indep); RUN(k_cndmask); RUN(k_dppmax); RUN(k_salu); RUN(k_snop); RUN(k_rcp); RUN(k_rcp_indep); RUN(k_mul_dep); RUN(k_cmp_ff1); RUN(k_readfirst); RUN(k_readlane_imm); RUN(k_rl_fmac); RUN(k_writelane);
    if (which < 0 || which == 99) {
        unsigned long long hh[3]; unsigned long long *d3; hipMalloc(&d3, 24);
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d3, d_sink, 3); hipMemcpy(hh, d3, 24, hipMemcpyDeviceToHost);
        printf("LDS: single-lane ds_write_b64 %.2f cyc each; broadcast ds_read_b128 %.2f cyc each (incl. final wait); write->read round trip %llu cyc\n", hh[0] / 64.0, hh[1] / 64.0, hh[2]);
    }
    return 0;
}
