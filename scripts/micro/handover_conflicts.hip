// micro-benchmark for the PMC counters SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: the LDS accesses of the elimination's column
// hand-over, one kind per kernel (gfx950).  Round 4's profiles showed 31 % conflict cycles where DESIGN.md said "no bank
// conflicts" for the [l][rb][g] layout: which instruction is it?
//   hipcc -O3 --offload-arch=gfx950 -o scripts/micro/handover_conflicts scripts/micro/handover_conflicts.hip
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d out -- scripts/micro/handover_conflicts
// Kernels (one wavefront each, 20000 rounds):
//   k<0>  new layout, the stores: ds_write2_b64 (offset0:0 offset1:4) + ds_write_b64 offset:64 at l * 112 + g * 8
//   k<1>  new layout, the loads: 3 x ds_read_b128 at l * 112 + 32 rb + 8 gp
//   k<2>  old layout, the stores: ds_write_b128 at lane * 16 + ds_write_b64 at 1024 + lane * 8
//   k<3>  old layout, the loads: 2 x (ds_read_b128 at l * 16 + go * 256, ds_read_b64 at 1024 + l * 8 + go * 128)
//   k<4>  Phase C: 33 x ds_read_b64 at row (l + 16 rb) * 336 + g * 8 + 32 c (the natural row order)
//   k<6..9> candidates [rb][l][g] with 48 / 32 bytes per l; k<10>, k<11>: [rb][g / 2][l][g % 2] stores / loads
//   k<5>  the stores alone, ds_write_b64 x 3 at l * 112 + g * 8 + {0, 32, 64}   (is it the write2 form?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));
template <int K>
__global__ __launch_bounds__(64) void k(double *sink)
{
    __shared__ __attribute__((aligned(16))) double buf[4096];
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
    for (int i = lane; i < 4096; i += 64) buf[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)buf;
    double a = sink[lane], b = a + 1.0, c = a + 2.0, acc = 0.0;
    d2v w = {a, b}, q0, q1, q2;
    const unsigned wad3 = base + l * 112u + g * 8u, rad3 = base + l * 112u;
    const unsigned wad2 = base + lane * 16u, wad1 = base + 1024u + lane * 8u, rad2 = base + l * 16u, rad1 = base + 1024u + l * 8u;
    for (int it = 0; it < 20000; ++it) {
        if (K == 0) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:4\n\tds_write_b64 %0, %3 offset:64" :: "v"(wad3), "v"(a), "v"(b), "v"(c) : "memory");
        if (K == 1) asm volatile("ds_read_b128 %0, %3 offset:0\n\tds_read_b128 %1, %3 offset:32\n\tds_read_b128 %2, %3 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2) : "v"(rad3) : "memory");
        if (K == 2) asm volatile("ds_write_b128 %0, %1\n\tds_write_b64 %2, %3" :: "v"(wad2), "v"(w), "v"(wad1), "v"(c) : "memory");
        if (K == 3) asm volatile("ds_read_b128 %0, %3 offset:256\n\tds_read_b64 %1, %4 offset:128\n\tds_read_b128 %2, %3 offset:512\n\ts_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(c), "=v"(q2) : "v"(rad2), "v"(rad1) : "memory");
        if (K == 4) {
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const unsigned ad = base + (unsigned)((l + 16 * rb) % 41) * 336u + g * 8u;
#pragma unroll
                for (int cc = 0; cc < 11; ++cc) { double t; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(ad), "i"(32 * cc) : "memory"); acc += t; }
            }
        }
        if (K == 5) asm volatile("ds_write_b64 %0, %1 offset:0\n\tds_write_b64 %0, %2 offset:32\n\tds_write_b64 %0, %3 offset:64" :: "v"(wad3), "v"(a), "v"(b), "v"(c) : "memory");
        // candidates: [rb][l][g] with 48 bytes per l (k<6> stores, k<7> loads), with 32 bytes per l (k<8>, k<9>)
        if (K == 6) asm volatile("ds_write_b64 %0, %1 offset:0\n\tds_write_b64 %0, %2 offset:768\n\tds_write_b64 %0, %3 offset:1536" :: "v"(base + l * 48u + g * 8u), "v"(a), "v"(b), "v"(c) : "memory");
        if (K == 7) asm volatile("ds_read_b128 %0, %3 offset:0\n\tds_read_b128 %1, %3 offset:768\n\tds_read_b128 %2, %3 offset:1536\n\ts_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2) : "v"(base + l * 48u) : "memory");
        if (K == 8) asm volatile("ds_write_b64 %0, %1 offset:0\n\tds_write_b64 %0, %2 offset:512\n\tds_write_b64 %0, %3 offset:1024" :: "v"(base + l * 32u + g * 8u), "v"(a), "v"(b), "v"(c) : "memory");
        if (K == 9) asm volatile("ds_read_b128 %0, %3 offset:0\n\tds_read_b128 %1, %3 offset:512\n\tds_read_b128 %2, %3 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2) : "v"(base + l * 32u) : "memory");
        // [rb][g / 2][l][g % 2]: the stores of two groups interleave into 256 contiguous bytes, a pair's two words are adjacent
        if (K == 10) asm volatile("ds_write_b64 %0, %1 offset:0\n\tds_write_b64 %0, %2 offset:512\n\tds_write_b64 %0, %3 offset:1024" :: "v"(base + (g >> 1) * 256u + l * 16u + (g & 1) * 8u), "v"(a), "v"(b), "v"(c) : "memory");
        if (K == 11) asm volatile("ds_read_b128 %0, %3 offset:256\n\tds_read_b128 %1, %3 offset:768\n\tds_read_b128 %2, %3 offset:1280\n\ts_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2) : "v"(base + l * 16u) : "memory");
        if (K == 1 || K == 3 || K == 7 || K == 9 || K == 11) acc += q0.x + q2.y;
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    sink[lane] = acc + buf[lane] + c;
}
int main()
{
    double *d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<9>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<10>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<11>, dim3(1), dim3(64), 0, 0, d);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
