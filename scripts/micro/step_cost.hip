// micro-benchmark: cost of the per-step fixed part of the LU (pivot confirmation, multipliers,
// bookkeeping) for one wavefront alone on a SIMD (gfx950), piece by piece and as a whole.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define TEST(NAME, BODY, PER)                                                                    \
    __global__ void NAME(unsigned long long *out, double *sink, int p)                            \
    {                                                                                             \
        double a = sink[threadIdx.x] + 1.5, b = a + 1.0, c = a + 2.0, d = a + 3.0, e = a + 4.0;   \
        unsigned u = (unsigned)threadIdx.x + 0x3ff00000u; int sp = __builtin_amdgcn_readfirstlane(p); \
        unsigned long long t0, t1;                                                                \
        asm volatile("s_mov_b64 s[44:45], exec\n\ts_mov_b64 s[46:47], 0x80" ::: "s44", "s45", "s46", "s47"); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");              \
        for (int it = 0; it < 100; ++it) { asm volatile(REP64(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(u) : "s"(sp) : "s40", "s41", "s42", "s43", "s48", "s49", "vcc", "scc"); } \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");              \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                   \
        sink[threadIdx.x] = a + b + c + d + e + u;                                                \
    }                                                                                             \
    static const double NAME##_per = PER;

#define POSCHK "v_readlane_b32 s40, %5, 7\n\ts_cmp_eq_u32 s40, s40\n\ts_cbranch_scc0 1f\n1:\n\t"
#define WRLANE "v_writelane_b32 %5, s42, 3\n\tv_writelane_b32 %5, s43, 3\n\t"
#define KEYSEL "v_cndmask_b32_e64 %5, 0, |%5|, s[44:45]\n\t"
#define DPP4 "s_nop 1\n\tv_max_u32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
             "s_nop 1\n\tv_max_u32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
             "s_nop 1\n\tv_max_u32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t" \
             "s_nop 1\n\tv_max_u32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
#define ARGMAX "s_nop 0\n\tv_readlane_b32 s40, %5, 16\n\tv_readlane_b32 s41, %5, 32\n\ts_max_u32 s40, s40, s41\n\tv_cmp_eq_u32_e64 s[48:49], s40, %5\n\t"
#define CONFIRM "s_and_b32 s48, s48, 0x80\n\ts_cmp_eq_u64 s[48:49], s[46:47]\n\ts_cbranch_scc0 1f\n1:\n\t"
#define MULT "v_readlane_b32 s42, %5, 9\n\tv_readlane_b32 s43, %5, 9\n\tv_rcp_f64 %0, s[42:43]\n\t" \
             "v_fma_f64 %1, -s[42:43], %0, 1.0\n\tv_fmac_f64 %0, %1, %0\n\tv_fma_f64 %1, -s[42:43], %0, 1.0\n\tv_fmac_f64 %0, %1, %0\n\t" \
             "v_mul_f64 %2, %3, -%0\n\tv_cmp_lt_i32_e64 s[40:41], 31, %5\n\ts_nop 0\n\tv_cndmask_b32_e64 %5, 0, %5, s[40:41]\n\tv_cndmask_b32_e64 %5, 0, %5, s[40:41]\n\t"
#define COL "v_readlane_b32 s40, %5, 5\n\tv_readlane_b32 s41, %5, 5\n\tv_fmac_f64 %4, s[40:41], %3\n\t"

TEST(k_poschk, POSCHK, 3)
TEST(k_wrlane, WRLANE, 2)
TEST(k_keysel, KEYSEL, 1)
TEST(k_dpp4, DPP4, 8)
TEST(k_argmax, ARGMAX, 5)
TEST(k_confirm, CONFIRM, 3)
TEST(k_mult, MULT, 12)
TEST(k_col, COL, 3)
TEST(k_waitcnt, "s_waitcnt lgkmcnt(0)\n\t", 1)
TEST(k_branch, "s_cbranch_scc0 1f\n1:\n\t", 1)
TEST(k_scmp_branch, "s_cmp_eq_u32 s40, s40\n\ts_cbranch_scc0 1f\n1:\n\t", 2)
TEST(k_rl_scmp, "v_readlane_b32 s40, %5, 7\n\ts_cmp_eq_u32 s40, s40\n\t", 2)
TEST(k_vcmp_sand, "v_cmp_eq_u32_e64 s[48:49], s40, %5\n\ts_and_b32 s48, s48, 0x80\n\t", 2)
TEST(k_step, "s_waitcnt lgkmcnt(0)\n\t" POSCHK WRLANE COL COL KEYSEL DPP4 ARGMAX MULT CONFIRM, 1)
TEST(k_step_nobranch, "s_waitcnt lgkmcnt(0)\n\tv_readlane_b32 s40, %5, 7\n\ts_cmp_eq_u32 s40, s40\n\t" WRLANE COL COL KEYSEL DPP4 ARGMAX MULT "s_and_b32 s48, s48, 0x80\n\ts_cmp_eq_u64 s[48:49], s[46:47]\n\t", 1)

#define RUN(NAME) do { if (which >= 0 && which != idx++) break; hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, d_out, d_sink, 3); hipDeviceSynchronize(); \
    hipEventRecord(e0); hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, d_out, d_sink, 3); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost); \
    printf("%-16s %8.2f ticks per %g instr = %6.2f /instr   (kernel %.3f ms incl. launch, %.0f ticks/us)\n", #NAME, (double)h / (100.0 * 64.0), NAME##_per, (double)h / (100.0 * 64.0 * NAME##_per), ms, (double)h / (ms * 1e3)); } while (0)
int main(int argc, char **argv)
{
    int which = argc > 1 ? atoi(argv[1]) : -1; int idx = 0;
    unsigned long long *d_out, h; double *d_sink; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 64 * 8); hipMemset(d_sink, 0, 64 * 8);
    RUN(k_poschk); RUN(k_wrlane); RUN(k_keysel); RUN(k_dpp4); RUN(k_argmax); RUN(k_confirm); RUN(k_mult); RUN(k_col);
    RUN(k_waitcnt); RUN(k_branch); RUN(k_scmp_branch); RUN(k_rl_scmp); RUN(k_vcmp_sand); RUN(k_step); RUN(k_step_nobranch);
    return 0;
}
