// Micro-benchmark: issue cost of plain vs packed f32 VALU instructions on gfx950, at a given
// number of waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    f2 bb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent plain multiplies
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
        } else if (MODE == 1) {  // 4 independent packed multiplies (8 lanes-ops)
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
        } else if (MODE == 2) {  // 8 plain fma
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
        } else if (MODE == 3) {  // 8 rcp
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 4) {  // 8 div_fixup
            asm volatile("v_div_fixup_f32 %0, %0, %8, %8\n v_div_fixup_f32 %1, %1, %8, %8\n v_div_fixup_f32 %2, %2, %8, %8\n v_div_fixup_f32 %3, %3, %8, %8\n"
                         "v_div_fixup_f32 %4, %4, %8, %8\n v_div_fixup_f32 %5, %5, %8, %8\n v_div_fixup_f32 %6, %6, %8, %8\n v_div_fixup_f32 %7, %7, %8, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
        } else if (MODE == 5) {  // 8 mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                         "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
        } else if (MODE == 6) {  // 8 cndmask
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
        } else if (MODE == 7) {  // 4 pk_add
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
        } else if (MODE == 9) {  // 4 independent packed fma (8 lane-ops)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
        } else if (MODE == 10) {  // 4 full 32 x 32 -> 64 products
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %6, 0\n v_mad_u64_u32 %2, vcc, %4, %7, 0\n v_mad_u64_u32 %3, vcc, %4, %8, 0\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");
        } else if (MODE == 11) {  // 8 mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
        } else if (MODE == 12) {  // 8 sqrt
            asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 13) {  // 8 cndmask with the mask in an SGPR pair
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                         "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "s20", "s21");
        } else if (MODE == 14) {  // 8 cndmask writing other registers than they read
            asm volatile("v_cndmask_b32 %0, %1, %8, vcc\n v_cndmask_b32 %1, %2, %8, vcc\n v_cndmask_b32 %2, %3, %8, vcc\n v_cndmask_b32 %3, %4, %8, vcc\n"
                         "v_cndmask_b32 %4, %5, %8, vcc\n v_cndmask_b32 %5, %6, %8, vcc\n v_cndmask_b32 %6, %7, %8, vcc\n v_cndmask_b32 %7, %0, %8, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
        } else if (MODE == 15) {  // 8 compares into vcc
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
        } else if (MODE == 16) {  // 4 x (compare into vcc, select on vcc)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
        } else if (MODE == 17) {  // 4 x (compare into an SGPR pair, select on it)
            asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n s_nop 1\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %2, %8\n s_nop 1\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n"
                         "v_cmp_lt_f32_e64 s[20:21], %4, %8\n s_nop 1\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %6, %8\n s_nop 1\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "s20", "s21", "s22", "s23");
        } else if (MODE == 8) {  // 8 dependent plain multiplies (one chain)
            asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                         "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                         : "+v"(x0) : "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
double run(float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f, 1.0000001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1.0000001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float *d; hipMalloc(&d, 256 * 4096 * 4 * 4);
    const char *names[] = {"v_mul_f32 x8", "v_pk_mul_f32 x4", "v_fma_f32 x8", "v_rcp_f32 x8", "v_div_fixup x8", "v_mul_hi_u32 x8", "v_cndmask x8", "v_pk_add_f32 x4", "v_mul_f32 dep x8",
                           "v_pk_fma_f32 x4", "v_mad_u64_u32 x4", "v_mul_lo_u32 x8", "v_sqrt_f32 x8",
                           "v_cndmask sgpr x8", "v_cndmask rot x8", "v_cmp_lt_f32 x8",
                           "cmp+cndmask vcc x4", "cmp+cndmask sgpr x4"};
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {            // waves per SIMD
        int blocks = 256 * wps;                        // 256 CUs x wps blocks of 4 waves
        printf("waves/SIMD=%d\n", wps);
        double ms[18];
        ms[0] = run<0>(d, blocks, iters); ms[1] = run<1>(d, blocks, iters); ms[2] = run<2>(d, blocks, iters);
        ms[3] = run<3>(d, blocks, iters); ms[4] = run<4>(d, blocks, iters); ms[5] = run<5>(d, blocks, iters);
        ms[6] = run<6>(d, blocks, iters); ms[7] = run<7>(d, blocks, iters); ms[8] = run<8>(d, blocks, iters);
        ms[9] = run<9>(d, blocks, iters); ms[10] = run<10>(d, blocks, iters); ms[11] = run<11>(d, blocks, iters);
        ms[12] = run<12>(d, blocks, iters); ms[13] = run<13>(d, blocks, iters); ms[14] = run<14>(d, blocks, iters);
        ms[15] = run<15>(d, blocks, iters); ms[16] = run<16>(d, blocks, iters); ms[17] = run<17>(d, blocks, iters);
        for (int m = 0; m < 18; ++m) {
            int n_instr = (m == 1 || m == 7 || m == 9 || m == 10) ? 4 : 8;
            // cycles per instruction per SIMD at 2.4 GHz: each SIMD runs wps waves x iters x n_instr
            double cyc = ms[m] * 1e-3 * 2.4e9 / ((double)wps * iters * n_instr);
            printf("  %-18s %8.3f ms  %.2f cycles/instr/SIMD (at 2.4 GHz)\n", names[m], ms[m], cyc);
        }
    }
    return 0;
}
