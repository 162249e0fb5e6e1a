// micro-benchmark: 64-bit DPP (row_newbcast) on gfx950 -- semantics and issue cost for one
// wavefront alone on a SIMD; LDS column broadcast (write own 3 doubles, read another group's).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double d2v __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define TEST(NAME, BODY, PER)                                                                    \
    __global__ void NAME(unsigned long long *out, double *sink, int p)                            \
    {                                                                                             \
        __shared__ double buf[1024];                                                              \
        double a = sink[threadIdx.x] + 1.5, b = a + 1.0, c = a + 2.0, d = a + 3.0, e = a + 4.0;   \
        unsigned u = (unsigned)threadIdx.x * 16u; int sp = __builtin_amdgcn_readfirstlane(p);     \
        buf[threadIdx.x] = a; buf[threadIdx.x + 64] = b; __syncthreads();                         \
        d2v q = {a, b}; d2v q2 = {c, d};                                                  \
        unsigned long long t0, t1;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");              \
        for (int it = 0; it < 100; ++it) { asm volatile(REP64(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(u), "+v"(q), "+v"(q2) : "s"(sp) : "s40", "s41", "s42", "s43", "s48", "s49", "vcc", "scc", "memory"); } \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");              \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                   \
        sink[threadIdx.x] = a + b + c + d + e + u + q.x + q.y + q2.x + q2.y + buf[threadIdx.x];   \
    }                                                                                             \
    static const double NAME##_per = PER;

#define DPPF(D, S, M, L) "v_fmac_f64_dpp " D ", " S ", " M " row_newbcast:" L " row_mask:0xf bank_mask:0xf\n\t"
TEST(k_fmac_plain, "v_fmac_f64 %0, %4, %3\n\tv_fmac_f64 %1, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\t", 3)
TEST(k_fmac_dpp, DPPF("%0", "%4", "%3", "5") DPPF("%1", "%4", "%3", "5") DPPF("%2", "%4", "%3", "5"), 3)
TEST(k_fmac_dpp_dep, DPPF("%0", "%4", "%3", "5"), 1)
TEST(k_fmac_dpp_selfsrc, DPPF("%0", "%0", "%3", "5") DPPF("%1", "%1", "%3", "5") DPPF("%2", "%2", "%3", "5"), 3)
TEST(k_mov64_dpp, "v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t", 2)
TEST(k_mov64, "v_mov_b64 %0, %4\n\tv_mov_b64 %1, %4\n\t", 2)
TEST(k_mul64, "v_mul_f64 %0, %4, %3\n\tv_mul_f64 %1, %4, %3\n\t", 2)
TEST(k_cmp64, "v_cmp_ge_f64 s[40:41], |%4|, |%3|\n\tv_cmp_ge_f64 s[42:43], |%4|, |%3|\n\t", 2)
TEST(k_read2_b64, "ds_read2_b64 %6, %5 offset0:1 offset1:5\n\tds_read2_b64 %7, %5 offset0:9 offset1:13\n\t", 2)
TEST(k_read_b128, "ds_read_b128 %6, %5\n\tds_read_b128 %7, %5 offset:1024\n\t", 2)
TEST(k_read_b64, "ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:1024\n\t", 2)
TEST(k_write_b128, "ds_write_b128 %5, %6\n\tds_write_b128 %5, %7 offset:1024\n\t", 2)
TEST(k_write_b64_uni, "ds_write_b64 %5, %0 offset:4096\n\t", 1)
// column broadcast of approach R: own 3 doubles out, another group's 3 doubles in, then wait
TEST(k_bcast_rt, "ds_write_b128 %5, %6\n\tds_write_b64 %5, %0 offset:2048\n\tds_read_b128 %7, %5\n\tds_read_b64 %1, %5 offset:2048\n\ts_waitcnt lgkmcnt(0)\n\t", 1)
// the same with twelve independent FMAs between issue and wait (latency hidden?)
#define F12 "v_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\tv_fmac_f64 %2, %4, %3\n\t"
TEST(k_bcast_hidden, "ds_write_b128 %5, %6\n\tds_write_b64 %5, %0 offset:2048\n\tds_read_b128 %7, %5\n\tds_read_b64 %1, %5 offset:2048\n\t" F12 F12 "s_waitcnt lgkmcnt(0)\n\t", 1)
TEST(k_f24, F12 F12, 24)
// pivot chain of one step: broadcast, compare, reciprocal (2 Newton steps), 3 multipliers, zero the pivot lane
#define CHAIN "v_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
              "v_cmp_ge_f64 s[40:41], |%0|, |%1|\n\tv_cmp_ge_f64 s[42:43], |%3|, |%1|\n\ts_and_b32 s40, s40, 0xfff0fff0\n\ts_or_b32 s40, s40, s42\n\t" \
              "v_rcp_f64 %2, %1\n\tv_fma_f64 %4, -%1, %2, 1.0\n\tv_fmac_f64 %2, %4, %2\n\tv_fma_f64 %4, -%1, %2, 1.0\n\tv_fmac_f64 %2, %4, %2\n\t" \
              "v_mul_f64 %3, %0, -%2\n\tv_mul_f64 %4, %0, -%2\n\tv_mul_f64 %0, %0, -%2\n\t" \
              "v_cmp_ne_u32 vcc, 48, %5\n\tv_cndmask_b32 %5, 0, %5, vcc\n\tv_cndmask_b32 %5, 0, %5, vcc\n\ts_cmp_eq_u32 s40, 0\n\ts_cbranch_scc0 1f\n1:\n\t"
TEST(k_chain, CHAIN, 1)

TEST(k_mix_3dpp_1fmac, DPPF("%0", "%4", "%3", "5") DPPF("%1", "%4", "%3", "5") DPPF("%2", "%4", "%3", "5") "v_fmac_f64 %4, %3, %3\n\t", 4)
TEST(k_mix_3dpp_1nop, DPPF("%0", "%4", "%3", "5") DPPF("%1", "%4", "%3", "5") DPPF("%2", "%4", "%3", "5") "s_nop 0\n\t", 4)
TEST(k_mix_2dpp_1fmac, DPPF("%0", "%4", "%3", "5") DPPF("%1", "%4", "%3", "5") "v_fmac_f64 %2, %3, %3\n\t", 3)
TEST(k_mix_mul_fmac, "v_mul_f64 %0, %4, %3\n\tv_fmac_f64 %1, %3, %3\n\t", 2)
TEST(k_mix_mul_nop, "v_mul_f64 %0, %4, %3\n\ts_nop 0\n\t", 2)
TEST(k_snop, "s_nop 0\n\t", 1)
TEST(k_salu4, "s_or_b32 s40, s41, s42\n\t", 1)
TEST(k_salu8, "s_and_b32 s40, s41, 0xff80\n\t", 1)
TEST(k_mix_mul_salu, "v_mul_f64 %0, %4, %3\n\ts_or_b32 s40, s41, s42\n\t", 2)

TEST(k_rcp64, "v_rcp_f64 %0, %4\n\tv_rcp_f64 %1, %4\n\t", 2)
TEST(k_rsq64, "v_rsq_f64 %0, %4\n\tv_rsq_f64 %1, %4\n\t", 2)
TEST(k_ldexp64, "v_ldexp_f64 %0, %4, %5\n\tv_ldexp_f64 %1, %4, %5\n\t", 2)
TEST(k_frexp64, "v_frexp_mant_f64 %0, %4\n\tv_frexp_mant_f64 %1, %4\n\t", 2)
TEST(k_divfmas, "v_div_fmas_f64 %0, %4, %3, %2\n\tv_div_fmas_f64 %1, %4, %3, %2\n\t", 2)
TEST(k_divfixup, "v_div_fixup_f64 %0, %4, %3, %2\n\tv_div_fixup_f64 %1, %4, %3, %2\n\t", 2)
TEST(k_divscale, "v_div_scale_f64 %0, vcc, %4, %3, %4\n\tv_div_scale_f64 %1, vcc, %4, %3, %4\n\t", 2)
TEST(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %5\n\tv_cvt_f64_i32 %1, %5\n\t", 2)
TEST(k_fma_dep1, "v_fma_f64 %0, %0, %4, %3\n\t", 1)
TEST(k_fma_dep_mix, "v_fma_f64 %0, %0, %4, %3\n\tv_fma_f64 %1, %1, %4, %3\n\t", 2)
TEST(k_cmp_cnd, "v_cmp_gt_f64 vcc, %4, %3\n\tv_cndmask_b32 %5, %5, %5, vcc\n\t", 2)
TEST(k_rdlane_salu, "v_readlane_b32 s40, %5, 3\n\ts_add_u32 s41, s40, 1\n\t", 2)
TEST(k_accread, "v_accvgpr_read_b32 %5, a0\n\t", 1)
TEST(k_add64, "v_add_f64 %0, %4, %3\n\tv_add_f64 %1, %4, %3\n\t", 2)
TEST(k_add_dep, "v_add_f64 %0, %0, %3\n\t", 1)

TEST(k_br_scc0_nt, "s_cmp_eq_u32 s40, s40\n\ts_cbranch_scc0 1f\n\tv_fmac_f64 %0, %4, %3\n1:\n\t", 3)
TEST(k_br_scc1_nt, "s_cmp_lg_u32 s40, s40\n\ts_cbranch_scc1 1f\n\tv_fmac_f64 %0, %4, %3\n1:\n\t", 3)
TEST(k_br_vccnz_nt, "s_cbranch_vccnz 1f\n\tv_fmac_f64 %0, %4, %3\n1:\n\t", 2)
TEST(k_br_execz_nt, "s_cbranch_execz 1f\n\tv_fmac_f64 %0, %4, %3\n1:\n\t", 2)
TEST(k_br_taken, "s_cmp_eq_u32 s40, s40\n\ts_cbranch_scc1 1f\n\tv_fmac_f64 %0, %4, %3\n1:\n\t", 2)
TEST(k_nobr, "s_cmp_eq_u32 s40, s40\n\tv_fmac_f64 %0, %4, %3\n\t", 2)

__global__ void k_sem(double *out)
{
    const int lane = threadIdx.x;
    double src = 100.0 + lane, m = 2.0, acc = 1000.0 * lane;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m));
    out[lane] = acc;                       // expect 1000*lane + 2*(100 + 16*(lane/16) + 5)
    double mv = -1.0;
    asm volatile("s_nop 4\n\tv_mov_b64_dpp %0, %1 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(mv) : "v"(src));
    out[64 + lane] = mv;                   // expect 100 + 16*(lane/16) + 9
    double acc2 = 0.0;                     // row_mask 0x5: only rows 0 and 2 written
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0x5 bank_mask:0xf" : "+v"(acc2) : "v"(src), "v"(m));
    out[128 + lane] = acc2;
}

#define RUN(NAME) do { hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, d_out, d_sink, 3); hipDeviceSynchronize(); \
    hipEventRecord(e0); hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, d_out, d_sink, 3); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost); \
    printf("%-20s %8.2f ticks per %g instr = %6.2f /instr   (kernel %.3f ms incl. launch, %.0f ticks/us)\n", #NAME, (double)h / (100.0 * 64.0), NAME##_per, (double)h / (100.0 * 64.0 * NAME##_per), ms, (double)h / (ms * 1e3)); } while (0)
#define RUN8(NAME) do { hipLaunchKernelGGL(NAME, dim3(1), dim3(512), 0, 0, d_out, d_sink, 3); hipDeviceSynchronize(); \
    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost); \
    printf("%-20s 8 waves in the workgroup (2 per SIMD): %6.2f ticks/instr for wave 0\n", #NAME, (double)h / (100.0 * 64.0 * NAME##_per)); } while (0)
int main()
{
    unsigned long long *d_out, h; double *d_sink; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 512 * 8); hipMemset(d_sink, 0, 512 * 8);
    double *d_sem, hs[192]; hipMalloc(&d_sem, sizeof(hs));
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d_sem); hipMemcpy(hs, d_sem, sizeof(hs), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (hs[l] != 1000.0 * l + 2.0 * (100 + 16 * (l / 16) + 5)) bad |= 1;
        if (hs[64 + l] != 100 + 16 * (l / 16) + 9) bad |= 2;
        const double e2 = ((l / 16) % 2 == 0) ? 2.0 * (100 + 16 * (l / 16)) : 0.0;
        if (hs[128 + l] != e2) bad |= 4;
    }
    printf("semantics: fmac_dpp/mov_dpp/row_mask %s (mask %d)  sample: lane 37 -> %.1f, %.1f, %.1f\n", bad ? "UNEXPECTED" : "as expected", bad, hs[37], hs[64 + 37], hs[128 + 37]);
    RUN(k_fmac_plain); RUN(k_fmac_plain); RUN(k_fmac_dpp); RUN(k_fmac_dpp_dep); RUN(k_fmac_dpp_selfsrc); RUN(k_mov64_dpp); RUN(k_mov64); RUN(k_mul64); RUN(k_cmp64);
    RUN(k_read2_b64); RUN(k_read_b128); RUN(k_read_b64); RUN(k_write_b128); RUN(k_write_b64_uni); RUN(k_bcast_rt); RUN(k_bcast_hidden); RUN(k_f24); RUN(k_chain);
    RUN(k_mix_3dpp_1fmac); RUN(k_mix_3dpp_1nop); RUN(k_mix_2dpp_1fmac); RUN(k_mix_mul_fmac); RUN(k_mix_mul_nop); RUN(k_snop); RUN(k_salu4); RUN(k_salu8); RUN(k_mix_mul_salu);
    RUN(k_rcp64); RUN(k_rsq64); RUN(k_ldexp64); RUN(k_frexp64); RUN(k_divfmas); RUN(k_divfixup); RUN(k_divscale); RUN(k_cvt_f64_i32); RUN(k_fma_dep1); RUN(k_fma_dep_mix); RUN(k_cmp_cnd); RUN(k_rdlane_salu); RUN(k_accread); RUN(k_add64); RUN(k_add_dep);
    RUN(k_br_scc0_nt); RUN(k_br_scc1_nt); RUN(k_br_vccnz_nt); RUN(k_br_execz_nt); RUN(k_br_taken); RUN(k_nobr);
    RUN8(k_mul64); RUN8(k_fmac_dpp); RUN8(k_mix_mul_salu);
    return 0;
}
