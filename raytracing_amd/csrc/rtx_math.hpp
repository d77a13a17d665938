// rtx_math.hpp -- device-side binary32 building blocks of the trace loop.
//
// Every expression is written in the order the reference evaluates it (rustc never
// fuses a multiply-add; nalgebra's 3-vector kernels: dot = (x+y)+z, normalize = three
// divisions by the length, quaternion rotation = ((t*w) + u x t) + v with t = (u x v)*2).
// The translation unit must be compiled with -ffp-contract=off and with HIP's default
// correctly rounded f32 division and square root, so results are bit-identical to an
// IEEE host evaluation of the same expressions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtx {

#define RTX_DEV __device__ __forceinline__

constexpr float kPi = 3.14159274101257324f;   // std::f32::consts::PI
constexpr float kRayEps = 1e-4f;              // src/ray.rs:3
constexpr float kInf = __builtin_huge_valf();

typedef float F2 __attribute__((ext_vector_type(2)));

#ifndef RTX_V3_PACKED
#define RTX_V3_PACKED 0   // experiment, see below
#endif
#if RTX_V3_PACKED
// Experiment: x and y share a register pair, so that componentwise sums, differences and products take
// one packed instruction for the two of them (the same IEEE operation on each half) and one scalar one
// for z.  Static count over the reference scene's kernel: 392 scalar operations fewer, 199 packed ones and
// 92 moves more, 7 more registers and a 16-byte stack slot for the union -- about what it saves, so off.
struct V3 {
    union {
        F2 xy;
        struct {
            float x, y;
        };
    };
    float z;
};
#else
struct V3 {
    float x, y, z;
};
#endif

// q1 = a1 / b1 and q2 = a2 / b2, both correctly rounded: the instruction sequence the compiler
// emits for one binary32 division (scale both operands, reciprocal, two Newton steps on it, three
// on the quotient, v_div_fmas, v_div_fixup), written out so that the six multiply / FMA steps of
// the two quotients run as packed instructions (v_pk_fma_f32: 5.5 cycles for both against 4.1 each).
RTX_DEV void div2(float a1, float b1, float a2, float b2, float &q1, float &q2)
{
#ifdef RTX_NO_DIV2
    q1 = a1 / b1;
    q2 = a2 / b2;
    return;
#endif
    bool scaled1, scaled2, ignore;
    const F2 d = {__builtin_amdgcn_div_scalef(a1, b1, false, &ignore), __builtin_amdgcn_div_scalef(a2, b2, false, &ignore)};
    const F2 n = {__builtin_amdgcn_div_scalef(a1, b1, true, &scaled1), __builtin_amdgcn_div_scalef(a2, b2, true, &scaled2)};
    const F2 r0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const F2 one = {1.0f, 1.0f};
    const F2 e0 = __builtin_elementwise_fma(-d, r0, one);
    const F2 r = __builtin_elementwise_fma(e0, r0, r0);
    const F2 m = n * r;
    const F2 e1 = __builtin_elementwise_fma(-d, m, n);
    const F2 q = __builtin_elementwise_fma(e1, r, m);
    const F2 e2 = __builtin_elementwise_fma(-d, q, n);
    q1 = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.x, r.x, q.x, scaled1), b1, a1);
    q2 = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.y, r.y, q.y, scaled2), b2, a2);
}

#if RTX_V3_PACKED
RTX_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RTX_DEV V3 v3(F2 xy, float z) { V3 r; r.xy = xy; r.z = z; return r; }
RTX_DEV V3 operator+(V3 a, V3 b) { return v3(a.xy + b.xy, a.z + b.z); }
RTX_DEV V3 operator-(V3 a, V3 b) { return v3(a.xy - b.xy, a.z - b.z); }
RTX_DEV V3 operator-(V3 a) { return v3(-a.xy, -a.z); }
RTX_DEV V3 operator*(V3 a, float s) { return v3(a.xy * s, a.z * s); }
#else
RTX_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RTX_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RTX_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RTX_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RTX_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
#endif
RTX_DEV V3 operator/(V3 a, float s)
{
    V3 q;
    div2(a.x, s, a.y, s, q.x, q.y);
    q.z = a.z / s;
    return q;
}
#if RTX_V3_PACKED
RTX_DEV V3 cmul(V3 a, V3 b) { return v3(a.xy * b.xy, a.z * b.z); }
#else
RTX_DEV V3 cmul(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
#endif
RTX_DEV V3 cdiv(V3 a, V3 b)
{
    V3 q;
    div2(a.x, b.x, a.y, b.y, q.x, q.y);
    q.z = a.z / b.z;
    return q;
}
#if RTX_V3_PACKED
RTX_DEV float dot(V3 a, V3 b)
{
    const F2 p = a.xy * b.xy;
    return (p.x + p.y) + a.z * b.z;
}
#else
RTX_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
#endif
RTX_DEV float len2(V3 a) { return dot(a, a); }
RTX_DEV float len(V3 a) { return __builtin_sqrtf(dot(a, a)); }
RTX_DEV V3 normalize(V3 a) { return a / len(a); }
RTX_DEV V3 cross(V3 a, V3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RTX_DEV float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// f32::min / f32::max of the reference's language: a NaN operand is ignored; operands that
// compare equal (+0 / -0) may return either, and v_min_f32 / v_max_f32 order -0 below +0.
RTX_DEV float rmin(float a, float b) { return __builtin_fminf(a, b); }
RTX_DEV float rmax(float a, float b) { return __builtin_fmaxf(a, b); }
RTX_DEV bool finite_f(float x) { return (x - x) == 0.0f; }

// unit quaternion (x, y, z, w) applied to a vector; inverse = conjugate
RTX_DEV V3 rotate(float4 q, V3 v)
{
    V3 u = v3(q.x, q.y, q.z);
    V3 t = cross(u, v) * 2.0f;
    V3 c = cross(u, t);
    return (t * q.w + c) + v;
}
RTX_DEV V3 rotate_inv(float4 q, V3 v) { return rotate(make_float4(-q.x, -q.y, -q.z, q.w), v); }
// rotate_inv of two vectors by one quaternion -- a ray's origin and direction on their way into an
// object's frame -- with every operation of the formula done once for both as a packed instruction
// (v_pk_mul_f32 / v_pk_add_f32: 5.5 cycles for the pair against 4.1 each).  The same operations on the
// same values in the same order as two calls of rotate_inv: the same bits.
#ifndef RTX_PACKED_ROTATIONS
#define RTX_PACKED_ROTATIONS 1
#endif
RTX_DEV void rotate_inv2(float4 q, V3 a, V3 b, V3 &ra, V3 &rb)
{
#if RTX_PACKED_ROTATIONS
    const float ux = -q.x, uy = -q.y, uz = -q.z, w = q.w;
    const F2 vx = {a.x, b.x}, vy = {a.y, b.y}, vz = {a.z, b.z};
    const F2 tx = (uy * vz - uz * vy) * 2.0f, ty = (uz * vx - ux * vz) * 2.0f, tz = (ux * vy - uy * vx) * 2.0f;
    const F2 cx = uy * tz - uz * ty, cy = uz * tx - ux * tz, cz = ux * ty - uy * tx;
    const F2 rx = (tx * w + cx) + vx, ry = (ty * w + cy) + vy, rz = (tz * w + cz) + vz;
    ra = v3(rx.x, ry.x, rz.x);
    rb = v3(rx.y, ry.y, rz.y);
#else
    ra = rotate_inv(q, a);
    rb = rotate_inv(q, b);
#endif
}

// sin and cos of x in [0, 2*pi]: Cody-Waite reduction by pi/2 in three steps, then the
// Cephes single-precision polynomials on [-pi/4, pi/4].  The same operation sequence is
// evaluated by the host-side checker, so both agree bit for bit.
RTX_DEV void sincos_2pi(float x, float &s_out, float &c_out)
{
    int k = (int)(x * 0.636619747f + 0.5f);
    float kf = (float)k;
    float r = x - kf * 1.5703125f;
    r = r - kf * 4.83751297e-4f;
    r = r - kf * 7.54978942e-8f;
    float r2 = r * r;
    float s = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
    float c = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2
              - 0.5f * r2 + 1.0f;
    int q = k & 3;
    float ss = (q & 1) ? c : s;
    float cc = (q & 1) ? s : c;
    s_out = (q & 2) ? -ss : ss;
    c_out = (q == 1 || q == 2) ? -cc : cc;
}

#ifndef RTX_PHILOX_KEYS_PER_CALL
#define RTX_PHILOX_KEYS_PER_CALL 1
#endif
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
// KEYS_PER_CALL: the key is the same in every call of a kernel, so the compiler works the ten rounds' keys out once
// and keeps all eighteen in scalar registers for the kernel's whole life.  Behind the barrier it cannot know that:
// every call forms them again, with scalar additions that cost no vector issue slot.
template <bool KEYS_PER_CALL = false>
RTX_DEV uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    if constexpr (KEYS_PER_CALL && RTX_PHILOX_KEYS_PER_CALL) asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply (v_mad_u64_u32) per product instead of a high and a low one
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// uniform draws as the reference's RNG crate forms them
RTX_DEV float u24(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-8f; }   // gen::<f32>()
RTX_DEV float u23(uint32_t w) { return (float)(w >> 9) * 1.1920928955078125e-7f; }
RTX_DEV float gen_range(uint32_t w, float lo, float hi) { return u23(w) * (hi - lo) + lo; }  // gen_range(lo..hi)

}  // namespace rtx
