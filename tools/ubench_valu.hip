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
    const char *names[] = {"v_mul_f32 x8", "v_pk_mul_f32 x4", "v_fma_f32 x8", "v_rcp_f32 x8", "v_div_fixup x8", "v_mul_hi_u32 x8", "v_cndmask x8", "v_pk_add_f32 x4", "v_mul_f32 dep x8"};
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {            // waves per SIMD
        int blocks = 256 * wps;                        // 256 CUs x wps blocks of 4 waves
        printf("waves/SIMD=%d\n", wps);
        double ms[9];
        ms[0] = run<0>(d, blocks, iters); ms[1] = run<1>(d, blocks, iters); ms[2] = run<2>(d, blocks, iters);
        ms[3] = run<3>(d, blocks, iters); ms[4] = run<4>(d, blocks, iters); ms[5] = run<5>(d, blocks, iters);
        ms[6] = run<6>(d, blocks, iters); ms[7] = run<7>(d, blocks, iters); ms[8] = run<8>(d, blocks, iters);
        for (int m = 0; m < 9; ++m) {
            int n_instr = (m == 1 || m == 7) ? 4 : 8;
            // cycles per instruction per SIMD at 2.4 GHz: each SIMD runs wps waves x iters x n_instr
            double cyc = ms[m] * 1e-3 * 2.4e9 / ((double)wps * iters * n_instr);
            printf("  %-18s %8.3f ms  %.2f cycles/instr/SIMD (at 2.4 GHz)\n", names[m], ms[m], cyc);
        }
    }
    return 0;
}
