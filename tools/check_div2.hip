// check_div2 -- the packed two-quotient division of rtx_math.hpp against the compiler's own
// binary32 division, bit for bit, over random bit patterns, random ordinary numbers and every
// pair of a table of special values.  Built by raytracing_amd/csrc/Makefile as
// raytracing_amd/check_div2; exit status 0 when every quotient has the bits of a / b.
#include "rtx_math.hpp"
#include <cstdio>
#include <vector>

__device__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ bool same(float x, float y)
{
    if (x != x && y != y) return true;   // any NaN equals any NaN
    return __float_as_uint(x) == __float_as_uint(y);
}
// mode 0: raw bit patterns; 1: exponents near 127 (ordinary numbers); 2: same denominator
__global__ void k_check(uint32_t rounds, int mode, unsigned long long *bad, float *example)
{
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t h = mix(id * 2654435761u + r * 40503u + (uint32_t)mode);
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) { h = mix(h + 0x9e3779b9u); w[k] = h; }
        if (mode >= 1)
            for (int k = 0; k < 4; ++k) w[k] = (w[k] & 0x807fffffu) | ((96u + (mix(w[k]) % 64u)) << 23);
        float a1 = __uint_as_float(w[0]), b1 = __uint_as_float(w[1]), a2 = __uint_as_float(w[2]), b2 = __uint_as_float(w[3]);
        if (mode == 2) b2 = b1;
        float q1, q2;
        rtx::div2(a1, b1, a2, b2, q1, q2);
        if (!same(q1, a1 / b1) || !same(q2, a2 / b2)) {
            if (atomicAdd(bad, 1ull) == 0ull) { example[0] = a1; example[1] = b1; example[2] = a2; example[3] = b2; }
        }
    }
}
__global__ void k_table(const float *v, int n, unsigned long long *bad, float *example)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n) return;
    float a = v[i / n], b = v[i % n];
    for (int j = 0; j < n; ++j) {
        float q1, q2;
        rtx::div2(a, b, v[j], b, q1, q2);
        if (!same(q1, a / b) || !same(q2, v[j] / b))
            if (atomicAdd(bad, 1ull) == 0ull) { example[0] = a; example[1] = b; example[2] = v[j]; example[3] = b; }
        rtx::div2(b, a, b, v[j], q1, q2);
        if (!same(q1, b / a) || !same(q2, b / v[j]))
            if (atomicAdd(bad, 1ull) == 0ull) { example[0] = b; example[1] = a; example[2] = b; example[3] = v[j]; }
    }
}

int main()
{
    unsigned long long *bad;
    float *example;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&example, 16) != hipSuccess) return 2;
    hipMemset(bad, 0, 8);
    hipMemset(example, 0, 16);
    for (int mode = 0; mode < 3; ++mode) k_check<<<4096, 256>>>(1024, mode, bad, example);
    std::vector<float> v;
    const float base[] = {0.0f, 1.0f, 2.0f, 3.0f, 0.1f, 1e-4f, 1e-6f, 16777216.0f, 1.17549435e-38f, 1e-45f, 5e-39f,
                          3.40282347e38f, 1e38f, 1e-30f, 1e30f, 1.5e-19f, 7e18f, 0.99999994f, 1.00000012f};
    for (float x : base) { v.push_back(x); v.push_back(-x); }
    v.push_back(__builtin_huge_valf()); v.push_back(-__builtin_huge_valf()); v.push_back(__builtin_nanf(""));
    float *dv;
    hipMalloc(&dv, v.size() * 4);
    hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    const int n = (int)v.size();
    k_table<<<(n * n + 255) / 256, 256>>>(dv, n, bad, example);
    unsigned long long h = 0;
    float ex[4];
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(ex, example, 16, hipMemcpyDeviceToHost);
    printf("%llu quotients differ from a / b", h);
    if (h) printf(" (first: %a / %a, %a / %a)", ex[0], ex[1], ex[2], ex[3]);
    printf("; checked %.2e random pairs and %d special-value pairs\n", 3.0 * 4096 * 256 * 1024, 2 * n * n * n);
    return h ? 1 : 0;
}
