// rtx_trace.hpp -- per-lane pieces of the trace loop, shared by every kernel variant.
//
// What each piece follows in the reference:
//   figure_t / figure_normal / figure_inside   src/objects/geometry.rs:56-147
//   to_local / world_normal                     src/objects/geometry.rs:22-37
//   closest_hit_scan                            src/trace.rs:102-121
//   cosine_sample, light_surface_sample         src/random.rs:33-53, src/objects/sample.rs:13-85
//   light_pdf, mis_pdf                          src/random.rs:83-128, 143-161
//   reflect_dir, refract_dir, schlick           src/trace.rs:123-149
//   scatter (one surface interaction)           src/trace.rs:21-100
// Operation order inside every expression is the reference's; see rtx_math.hpp.
#pragma once

#include "rtx_device.h"
#include "rtx_math.hpp"

namespace rtx {

// ------------------------------------------------------------------ counters
template <bool ON>
struct Counters {
    uint32_t v[ON ? C_COUNT : 1];
    RTX_DEV void clear()
    {
        if constexpr (ON) {
#pragma unroll
            for (int i = 0; i < C_COUNT; ++i) v[i] = 0;
        }
    }
    template <int I>
    RTX_DEV void add(bool cond = true)
    {
        if constexpr (ON) v[I] += cond ? 1u : 0u;
    }
};

// ------------------------------------------------------------- object tables
// View of the object table: pointers into LDS (staged copy) or global memory.
struct Table {
    const float4 *geom;
    const float4 *mat;
    const int32_t *lights;
    const float4 *light_bounds;  // two per light: (bounding-sphere centre xyz, squared radius), (slack, -, -, -)
    const int32_t *order;        // the objects by category (RenderParams::scan_order), or nullptr
    const float *uniforms;       // LDS copy of the camera, the background colour and the constant draw words
    bool has_uniforms;           // (stage_uniforms below)
    bool keys_per_call;          // the caller is a megakernel (see draw_block)
    bool in_lds;                 // geom and order are LDS copies
    uint32_t n_prims, n_lights;
    bool ellipsoids_pow2;   // every ellipsoid's radii are powers of two (kGeomPow2): their normals need no division
};

struct Geom {
    float4 rot;
    V3 pos;
    int kind;
    int flags;   // kGeomIdentity: rotation is exactly (+0, +0, +0, 1) and the position is tame;
                 // kGeomPow2: every component of param is a positive power of two;
                 // kGeomPlainBoxLight: a box with kGeomIdentity and half-sizes in (0, 1e30)
    V3 param;
};
RTX_DEV Geom load_geom(const float4 *g, uint32_t i)
{
    float4 a = g[3 * i + 0], b = g[3 * i + 1], c = g[3 * i + 2];
    Geom r;
    r.rot = a;
    r.pos = v3(b.x, b.y, b.z);
    int bits = __float_as_int(b.w);
    r.kind = bits & 0xff;
    r.flags = bits >> 8;
    r.param = v3(c.x, c.y, c.z);
    return r;
}

// A ray is "plain" when every component of its origin and direction is finite, below 1e30 in
// magnitude and not -0.  For a plain ray and an object whose rotation is exactly the identity
// quaternion, the rotation formula returns its argument bit for bit: every product with the
// quaternion's zero vector part is a zero, a zero added to a non-zero x gives x, and a zero
// added to +0 gives +0 -- only a -0 component (or a non-finite one, 0 * inf) could come out
// different, and o - p cannot be -0 unless o is.  Such objects skip the two rotations.
RTX_DEV bool ray_is_plain(V3 o, V3 d)
{
    float m = ((fabsf(o.x) + fabsf(o.y)) + fabsf(o.z)) + ((fabsf(d.x) + fabsf(d.y)) + fabsf(d.z));
    bool ok = m < 1e30f;
    ok = ok && __float_as_uint(o.x) != 0x80000000u && __float_as_uint(o.y) != 0x80000000u &&
         __float_as_uint(o.z) != 0x80000000u;
    ok = ok && __float_as_uint(d.x) != 0x80000000u && __float_as_uint(d.y) != 0x80000000u &&
         __float_as_uint(d.z) != 0x80000000u;
    return ok;
}

struct LocalRay {
    V3 o, d;
};
// src/objects/geometry.rs:24-27: the direction is rotated only, so it keeps unit length
RTX_DEV LocalRay to_local(const Geom &g, V3 o, V3 d)
{
    LocalRay r;
    rotate_inv2(g.rot, o - g.pos, d, r.o, r.d);
    return r;
}
// the same for an identity-rotation object and a plain ray (see ray_is_plain)
RTX_DEV LocalRay to_local_identity(const Geom &g, V3 o, V3 d)
{
    LocalRay r;
    r.o = o - g.pos;
    r.d = d;
    return r;
}

// Distance along the local ray to the figure, or "no hit".
RTX_DEV bool plane_t(V3 n, const LocalRay &r, float &t)
{
    t = -dot(r.o, n) / dot(r.d, n);
    return !(t < 0.0f); // t = 0, -0 and NaN pass, as in the reference
}
// 1 / p for p = 2^e: the exponent field mirrored about the bias
RTX_DEV V3 pow2_inverse(V3 p)
{
    return v3(__int_as_float(0x7f000000 - __float_as_int(p.x)), __int_as_float(0x7f000000 - __float_as_int(p.y)),
              __int_as_float(0x7f000000 - __float_as_int(p.z)));
}
// x / p for the object's parameter vector.  When every component of p is a power of two
// (kGeomPow2, set at upload) 1 / p is exact, and x * (1 / p) and x / p are the same real number
// rounded once: identical bits for every x, subnormal results and non-finite x included.
// `pow2` must be wave-uniform; the division costs about ten instructions, the product one.
RTX_DEV V3 div_param(V3 x, V3 p, bool pow2)
{
    if (pow2) return cmul(x, pow2_inverse(p));
    return cdiv(x, p);
}
RTX_DEV bool ellipsoid_t(V3 rad, const LocalRay &r, float &t, bool pow2)
{
    V3 u = div_param(r.o, rad, pow2);
    V3 v = div_param(r.d, rad, pow2);
    float a = len2(v);
    float b = dot(u, v);
    float c = len2(u) - 1.0f;
    float det = b * b - a * c;
    if (det < 0.0f) return false;
    float sq = __builtin_sqrtf(det);
    float t1, t2;
    div2(-b + sq, a, -b - sq, a, t1, t2);
    float lo = rmin(t1, t2), hi = rmax(t1, t2);
    if (lo > 0.0f) { t = lo; return true; }
    if (hi > 0.0f) { t = hi; return true; }
    return false;
}
RTX_DEV bool box_t(V3 s, const LocalRay &r, float &t)
{
    float ax1, ax2, ay1, ay2, az1, az2;
    div2(s.x - r.o.x, r.d.x, -s.x - r.o.x, r.d.x, ax1, ax2);
    div2(s.y - r.o.y, r.d.y, -s.y - r.o.y, r.d.y, ay1, ay2);
    div2(s.z - r.o.z, r.d.z, -s.z - r.o.z, r.d.z, az1, az2);
    float t1 = rmax(rmax(rmin(ax1, ax2), rmin(ay1, ay2)), rmin(az1, az2));
    float t2 = rmin(rmin(rmax(ax1, ax2), rmax(ay1, ay2)), rmax(az1, az2));
    if (t1 > t2) return false;
    if (t1 >= 0.0f) { t = t1; return true; }
    if (t2 >= 0.0f) { t = t2; return true; }
    return false;
}
// ellipsoid_t or box_t for lanes that hold different kinds of object (BVH leaves, the lights of a
// many-lights scene: never planes).  Both tests begin with six quotients -- o / r and d / r for the
// ellipsoid, (+-s - o) / d for the box -- and a division is a dozen instructions whatever its
// operands, so every lane picks its own operands and the six divisions run once for both kinds
// instead of once per kind.  Each lane evaluates exactly the expressions of its kind's test.
RTX_DEV bool ellipsoid_or_box_t(int kind, V3 s, const LocalRay &r, float &t)
{
    const bool box = kind == kBox;
    float x1, x2, y1, y2, z1, z2;
    div2(box ? s.x - r.o.x : r.o.x, box ? r.d.x : s.x, box ? -s.x - r.o.x : r.d.x, box ? r.d.x : s.x, x1, x2);
    div2(box ? s.y - r.o.y : r.o.y, box ? r.d.y : s.y, box ? -s.y - r.o.y : r.d.y, box ? r.d.y : s.y, y1, y2);
    div2(box ? s.z - r.o.z : r.o.z, box ? r.d.z : s.z, box ? -s.z - r.o.z : r.d.z, box ? r.d.z : s.z, z1, z2);
    if (box) {
        float t1 = rmax(rmax(rmin(x1, x2), rmin(y1, y2)), rmin(z1, z2));
        float t2 = rmin(rmin(rmax(x1, x2), rmax(y1, y2)), rmax(z1, z2));
        if (t1 > t2) return false;
        if (t1 >= 0.0f) { t = t1; return true; }
        if (t2 >= 0.0f) { t = t2; return true; }
        return false;
    }
    const V3 u = v3(x1, y1, z1), v = v3(x2, y2, z2);
    float a = len2(v);
    float b = dot(u, v);
    float c = len2(u) - 1.0f;
    float det = b * b - a * c;
    if (det < 0.0f) return false;
    float sq = __builtin_sqrtf(det);
    float r1, r2;
    div2(-b + sq, a, -b - sq, a, r1, r2);
    float lo = rmin(r1, r2), hi = rmax(r1, r2);
    if (lo > 0.0f) { t = lo; return true; }
    if (hi > 0.0f) { t = hi; return true; }
    return false;
}
// A plane whose normal is +-e_k, for a ray that went through to_local_identity (finite components,
// none of them -0).  The two dot products of plane_t are then s * o_k and s * d_k with s = +-1, up
// to the sign of a zero: the other products are zeros, and a zero added to a non-zero x gives x.
// Division rounds symmetrically in both signs, so -(s o_k) / (s d_k) has the bits of (-o_k) / d_k.
// When o_k or d_k is zero the sign of that zero may differ from plane_t's:
//   d_k = 0: t is an infinity or a NaN either way, and both are rejected by the range filter;
//   o_k = 0: t is a zero of either sign, accepted either way, and t only enters comparisons and
//   the hit point o + d * t, where d * (+-0) is a zero added to components that are not -0.
RTX_DEV bool plane_axis_t(float o_k, float d_k, float &t)
{
    t = -o_k / d_k;
    return !(t < 0.0f);
}
// `kind` may carry kGeomPow2 << 8 (wave-uniform callers only): an ellipsoid whose radii are
// powers of two is one more case of the dispatch, so other objects pay nothing for it.
constexpr int kEllipsoidPow2 = kEllipsoid | (kGeomPow2 << 8);
RTX_DEV bool figure_t(int kind, V3 param, const LocalRay &r, float &t)
{
    if (kind == kPlane) return plane_t(param, r, t);
    if (kind == kEllipsoid) return ellipsoid_t(param, r, t, false);
    if (kind == kEllipsoidPow2) return ellipsoid_t(param, r, t, true);
    return box_t(param, r, t);
}
// Local (unnormalised) normal at distance t.  `pow2` (the same in every lane): every component of
// an ellipsoid's param is a power of two (kGeomPow2), so the nine divisions are exact products (see
// div_param).
RTX_DEV V3 figure_normal(int kind, V3 param, const LocalRay &r, float t, bool pow2 = false)
{
    if (kind == kPlane) return param;
    if (kind == kEllipsoid) {
        if (pow2) {
            const V3 inv = pow2_inverse(param);
            return cmul(cmul(r.o, inv) + cmul(r.d, inv) * t, inv);
        }
        V3 u = cdiv(r.o, param);
        V3 v = cdiv(r.d, param);
        return cdiv(u + v * t, param);
    }
    V3 n = cdiv(r.o + r.d * t, param);
    // keep the first component of largest magnitude, zero the other two
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    int im = 0;
    float vm = ax;
    if (ay > vm) { vm = ay; im = 1; }
    if (az > vm) { vm = az; im = 2; }
    return v3(im == 0 ? n.x : 0.0f, im == 1 ? n.y : 0.0f, im == 2 ? n.z : 0.0f);
}
RTX_DEV bool figure_inside(int kind, V3 param, V3 o)
{
    if (kind == kPlane) return dot(param, o) < 0.0f;
    if (kind == kEllipsoid) return len2(cdiv(o, param)) < 1.0f;
    V3 q = cdiv(o, param);
    return rmax(rmax(fabsf(q.x), fabsf(q.y)), fabsf(q.z)) < 1.0f;
}
// figure_normal and figure_inside for lanes that hold different kinds of object.  An ellipsoid's
// normal is ((o / r) + (d / r) t) / r and a box's (o + d t) / s; "inside" looks at o / r or o / s.
// The last division of the normal and the division of the origin are the same instructions for
// both kinds, so they are written once with per-lane operands; every lane evaluates exactly the
// expressions of its own kind.  `pow2` (the same in every lane): every ellipsoid of the scene has
// power-of-two radii, so their normals are exact products (see div_param) and only boxes divide.
RTX_DEV V3 figure_normal_inside(int kind, V3 param, const LocalRay &r, float t, bool want_inside, bool &inside,
                                bool pow2 = false)
{
    inside = false;
    if (kind == kPlane) {
        if (want_inside) inside = dot(param, r.o) < 0.0f;
        return param;
    }
    const bool ell = kind == kEllipsoid;
    if (pow2 && ell) {
        if (want_inside) inside = len2(cdiv(r.o, param)) < 1.0f;
        const V3 inv = pow2_inverse(param);
        return cmul(cmul(r.o, inv) + cmul(r.d, inv) * t, inv);
    }
    V3 u = v3(0.0f, 0.0f, 0.0f), x = r.o + r.d * t;
    if (ell || want_inside) u = cdiv(r.o, param);
    if (ell) x = u + cdiv(r.d, param) * t;
    V3 n = cdiv(x, param);
    if (want_inside) inside = ell ? len2(u) < 1.0f : rmax(rmax(fabsf(u.x), fabsf(u.y)), fabsf(u.z)) < 1.0f;
    if (!ell) {
        // keep the first component of largest magnitude, zero the other two
        float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
        int im = 0;
        float vm = ax;
        if (ay > vm) { vm = ay; im = 1; }
        if (az > vm) { vm = az; im = 2; }
        n = v3(im == 0 ? n.x : 0.0f, im == 1 ? n.y : 0.0f, im == 2 ? n.z : 0.0f);
    }
    return n;
}
// src/objects/geometry.rs:30-33
RTX_DEV V3 world_normal(const Geom &g, V3 n_local, V3 d_world)
{
    V3 n = normalize(rotate(g.rot, n_local));
    if (dot(n, d_world) > 0.0f) n = -n;
    return n;
}

#ifndef RTX_SCAN_SINGLE
#define RTX_SCAN_SINGLE 1
#endif
// One category of the scan: every lane tests the same object, `CAT` fixes the figure, whether
// the pose is the identity (callers pass plain rays only, see ray_is_plain) and the parameter
// shortcut.  Categories are visited out of object order, so among equal distances the lowest
// index is kept explicitly -- what "the first object reaching the minimum keeps it" comes to.
template <int CAT>
RTX_DEV void scan_category(const Table &tab, uint32_t begin, uint32_t end, V3 o, V3 d, float ray_length, int &best,
                           float &best_t)
{
    constexpr bool identity = CAT == kCatPlaneX || CAT == kCatPlaneY || CAT == kCatPlaneZ || CAT == kCatPlaneIdentity ||
                              CAT == kCatBoxIdentity || CAT == kCatEllipsoidIdentityPow2 || CAT == kCatEllipsoidIdentity;
    if (begin >= end) return;
    const int32_t *order = tab.order + kScanHeader;
    auto index_at = [&](uint32_t k) {
        const int i = order[k];
        return tab.in_lds ? i : __builtin_amdgcn_readfirstlane(i);   // a global table is read with scalar loads
    };
    auto test = [&](int i, const Geom &g) {
        const LocalRay lr = identity ? to_local_identity(g, o, d) : to_local(g, o, d);
        float t;
        bool hit;
        if constexpr (CAT == kCatPlaneX) hit = plane_axis_t(lr.o.x, lr.d.x, t);
        else if constexpr (CAT == kCatPlaneY) hit = plane_axis_t(lr.o.y, lr.d.y, t);
        else if constexpr (CAT == kCatPlaneZ) hit = plane_axis_t(lr.o.z, lr.d.z, t);
        else if constexpr (CAT == kCatPlaneIdentity || CAT == kCatPlane) hit = plane_t(g.param, lr, t);
        else if constexpr (CAT == kCatBoxIdentity || CAT == kCatBox) hit = box_t(g.param, lr, t);
        else hit = ellipsoid_t(g.param, lr, t, CAT == kCatEllipsoidIdentityPow2 || CAT == kCatEllipsoidPow2);
        hit = hit && (t * ray_length < kInf);
        if (hit && (t < best_t || (t == best_t && i < best))) {
            best_t = t;
            best = i;
        }
    };
#if RTX_SCAN_SINGLE
    if (end - begin == 1u) {
        // a category of one object (every category of a small room): no loop, nothing fetched ahead
        const int i = index_at(begin);
        test(i, load_geom(tab.geom, (uint32_t)i));
        return;
    }
#endif
    // Two loads ahead of the arithmetic: while object k is tested, the data of object k + 1 and
    // the index of object k + 2 are on their way, so the wave does not sit through an LDS round
    // trip (two, with the indirection) per object.  The list is padded with two valid entries.
    int i_next = index_at(begin);
    int i_next2 = index_at(begin + 1u);
    Geom g_next = load_geom(tab.geom, (uint32_t)i_next);
    for (uint32_t k = begin; k < end; ++k) {
        const int i = i_next;
        const Geom g = g_next;
        i_next = i_next2;
        g_next = load_geom(tab.geom, (uint32_t)i_next);
        i_next2 = index_at(k + 2u);
        test(i, g);
    }
}
// Two axis planes of one axis (the two walls of a room that face each other).  For a ray whose origin lies
// strictly between them -- by more than 1e-30 -- the plane behind the ray is certainly rejected: its distance
// -(o_k - p) / d_k is negative and cannot round to zero (|d_k| <= 1 up to rounding), or, for d_k = 0, minus
// infinity; and for d_k = 0 the plane ahead gives plus infinity, which the range filter drops.  So one
// division decides the pair: the plane ahead, chosen by the sign of d_k (for d_k = 0 either choice ends
// without a hit).  A lane whose origin is not strictly between the planes (a ray that leaves a wall at a
// grazing angle can start a rounding error behind it) tests both, the second in a branch that is rarely taken.
#ifndef RTX_PLANE_PAIRS
#define RTX_PLANE_PAIRS 1
#endif
template <int CAT>
RTX_DEV void scan_axis_plane_pair(const Table &tab, uint32_t begin, V3 o, V3 d, float ray_length, int &best, float &best_t)
{
    const int32_t *order = tab.order + kScanHeader;
    int ia = order[begin], ib = order[begin + 1u];
    if (!tab.in_lds) {
        ia = __builtin_amdgcn_readfirstlane(ia);
        ib = __builtin_amdgcn_readfirstlane(ib);
    }
    const float4 pa = tab.geom[3 * ia + 1], pb = tab.geom[3 * ib + 1];
    constexpr int k = CAT == kCatPlaneX ? 0 : (CAT == kCatPlaneY ? 1 : 2);
    const float ok = comp(o, k), dk = comp(d, k);
    const float na = ok - (k == 0 ? pa.x : (k == 1 ? pa.y : pa.z)), nb = ok - (k == 0 ? pb.x : (k == 1 ? pb.y : pb.z));
    // strictly between: the two offsets have opposite signs and neither is tiny
    const bool between = (na > 1e-30f && nb < -1e-30f) || (na < -1e-30f && nb > 1e-30f);
    // ahead of the ray: the plane whose offset has the sign opposite to d_k's
    const bool take_a = !between || ((na > 0.0f) != (dk > 0.0f));
    {
        const int i = take_a ? ia : ib;
        float t;
        bool hit = plane_axis_t(take_a ? na : nb, dk, t);
        hit = hit && (t * ray_length < kInf);
        if (hit && (t < best_t || (t == best_t && i < best))) {
            best_t = t;
            best = i;
        }
    }
    if (__ballot(!between) != 0ull) {
        if (!between) {
            float t;
            bool hit = plane_axis_t(nb, dk, t);
            hit = hit && (t * ray_length < kInf);
            if (hit && (t < best_t || (t == best_t && ib < best))) {
                best_t = t;
                best = ib;
            }
        }
    }
}

// PAIRS: bit k set = axis category k holds exactly two planes (two walls of a room that face each other), tested
// as a pair.  The choice is made once for the whole scan (scan_by_category) among three compiled forms -- no pairs,
// x and y, all three axes -- so the categories that follow are compiled three times: a choice at every axis, each
// followed by its own copy of the rest, made eight copies (100 KB of code per kernel), and one merged copy behind a
// choice per axis measured 0.5 % slower on the reference scene.
template <int CAT, int PAIRS>
RTX_DEV void scan_categories_from(const Table &tab, const uint32_t (&ends)[kScanHeader], uint32_t begin, V3 o, V3 d,
                                  float ray_length, int &best, float &best_t)
{
    if constexpr (CAT < kScanCats) {
        if constexpr (CAT <= kCatPlaneZ && ((PAIRS >> CAT) & 1) != 0)
            scan_axis_plane_pair<CAT>(tab, begin, o, d, ray_length, best, best_t);
        else
            scan_category<CAT>(tab, begin, ends[CAT], o, d, ray_length, best, best_t);
#ifdef RTX_X_REPEAT_CATS   // timing experiment: the categories of this bit mask run twice
        if constexpr (((RTX_X_REPEAT_CATS) >> CAT) & 1) {
            asm volatile("" : "+v"(o.x), "+v"(d.x));
            scan_category<CAT>(tab, begin, ends[CAT], o, d, ray_length, best, best_t);
        }
#endif
        scan_categories_from<CAT + 1, PAIRS>(tab, ends, ends[CAT], o, d, ray_length, best, best_t);
    }
}
RTX_DEV void scan_by_category(const Table &tab, V3 o, V3 d, float ray_length, int &best, float &best_t)
{
    // the category ends in one round trip (the list is 16-byte aligned).  (As kernel arguments instead -- scalar
    // registers, no load -- they stay live through the kernel and cost more in spilled scalar registers than the
    // load: reference scene 893.6 -> 927.8 ms at 4K; packed two to a word, six registers: 892.6 -> 894.8 ms.)
    const uint4 *header = reinterpret_cast<const uint4 *>(tab.order);
    uint32_t ends[kScanHeader];
#pragma unroll
    for (int q = 0; q < kScanHeader / 4; ++q) {
        const uint4 e = header[q];
        ends[4 * q + 0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
        ends[4 * q + 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
        ends[4 * q + 2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z);
        ends[4 * q + 3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.w);
    }
    static_assert(kCatPlaneX == 0 && kCatPlaneY == 1 && kCatPlaneZ == 2, "the axis categories come first");
    const bool xy = RTX_PLANE_PAIRS && ends[kCatPlaneX] == 2u && ends[kCatPlaneY] == 4u;
    if (xy && ends[kCatPlaneZ] == 6u) scan_categories_from<0, 7>(tab, ends, 0u, o, d, ray_length, best, best_t);
    else if (xy) scan_categories_from<0, 3>(tab, ends, 0u, o, d, ray_length, best, best_t);
    else scan_categories_from<0, 0>(tab, ends, 0u, o, d, ray_length, best, best_t);
}

// Closest hit over the whole table, linear scan in object order.  The first object
// reaching the minimum keeps it (strict <).  `i` is wave-uniform, so the table reads
// are broadcasts (LDS) or scalar loads (global).
template <bool UNIFORM_KIND>
RTX_DEV int closest_hit_scan(const Table &tab, V3 o, V3 d, float &best_t)
{
    float ray_length = len(d);
    int best = -1;
    best_t = kInf;
    // wave-uniform: every active lane's ray is plain
    const bool all_plain = UNIFORM_KIND && __ballot(!ray_is_plain(o, d)) == 0ull;
    if (UNIFORM_KIND && all_plain && tab.order != nullptr) {
        scan_by_category(tab, o, d, ray_length, best, best_t);
        return best;
    }
    for (uint32_t i = 0; i < tab.n_prims; ++i) {
        Geom g = load_geom(tab.geom, i);
        LocalRay lr;
        if (UNIFORM_KIND && all_plain && (__builtin_amdgcn_readfirstlane(g.flags) & kGeomIdentity))
            lr = to_local_identity(g, o, d);
        else
            lr = to_local(g, o, d);
        int kind = g.kind;
        if (UNIFORM_KIND) kind = __builtin_amdgcn_readfirstlane(g.kind | ((g.flags & kGeomPow2) << 8));
        float t;
        bool hit = figure_t(kind, g.param, lr, t);
        // src/trace.rs:113-119 with max_dist = +inf drops NaN and +inf distances
        hit = hit && (t * ray_length < kInf);
        if (hit && (best < 0 || t < best_t)) {
            best_t = t;
            best = (int)i;
        }
    }
    return best;
}

struct Hit {
    V3 n;        // unit, facing against the ray
    bool inside;
};
// Normal and inside flag of object `idx` for the ray that hit it at distance t.
// MIXED: the lanes of the calling kernel usually hold objects of different kinds (BVH scenes), which
// is what figure_normal_inside is for; kernels of small scenes keep the plain dispatch.
template <bool MIXED = false>
RTX_DEV Hit finish_hit(const Table &tab, int idx, V3 o, V3 d, float t, bool want_inside)
{
    Geom g = load_geom(tab.geom, (uint32_t)idx);
    LocalRay lr = to_local(g, o, d);
    Hit h;
    if constexpr (MIXED) {
        h.n = world_normal(g, figure_normal_inside(g.kind, g.param, lr, t, want_inside, h.inside, tab.ellipsoids_pow2), d);
    } else {
        h.n = world_normal(g, figure_normal(g.kind, g.param, lr, t, tab.ellipsoids_pow2), d);
        h.inside = want_inside ? figure_inside(g.kind, g.param, lr.o) : false;
    }
    return h;
}

// -------------------------------------------------------------------- lights
// Intersection of one light with a world ray as the light pdf needs it: the distance and
// |cos| between the ray and the surface normal there (src/random.rs:120).  The reference flips
// the normal against the ray first (src/objects/geometry.rs:31-33); negating a vector negates
// its dot product exactly, so the absolute value is taken of the unflipped one.
// UNIFORM: `g` is the same object in every lane (the light loop index is wave-uniform), which
// allows the identity-rotation shortcuts; otherwise every lane may hold a different light.
template <bool UNIFORM>
RTX_DEV bool light_intersect_cos(const Geom &g, V3 o, V3 d, float &t, float &cosv)
{
    bool ident = false, pow2 = false;
    if constexpr (UNIFORM) {
        const int flags = __builtin_amdgcn_readfirstlane(g.flags);
        ident = (flags & kGeomIdentity) && __ballot(!ray_is_plain(o, d)) == 0ull;
        pow2 = (flags & kGeomPow2) != 0;
    }
    LocalRay lr = ident ? to_local_identity(g, o, d) : to_local(g, o, d);
    if constexpr (UNIFORM) {
        if (!figure_t(g.kind | (pow2 ? kGeomPow2 << 8 : 0), g.param, lr, t)) return false;
    } else {
        if (!ellipsoid_or_box_t(g.kind, g.param, lr, t)) return false;   // lights are never planes
    }
    if (ident && __builtin_amdgcn_readfirstlane(g.kind) == kBox) {
        // Axis-aligned box: the local normal is v * e_k (k = first axis of largest |.|, the
        // other two components are +0).  The identity rotation returns it unchanged, and
        // normalising it gives exactly +-e_k: sqrt(fl(v*v)) = |v| in binary floating point when
        // v*v neither overflows nor underflows, and v / |v| = +-1.  Hence |n . d| = |d_k|.
        V3 q = div_param(lr.o + lr.d * t, g.param, pow2);
        float ax = fabsf(q.x), ay = fabsf(q.y), az = fabsf(q.z);
        int k = 0;
        float vm = ax;
        if (ay > vm) { vm = ay; k = 1; }
        if (az > vm) { vm = az; k = 2; }
        const bool tame = vm > 1e-18f && vm < 1e18f;
        if (__ballot(!tame) == 0ull) {
            cosv = fabsf(k == 0 ? d.x : (k == 1 ? d.y : d.z));
            return true;
        }
    }
    V3 nl;
    if constexpr (UNIFORM) {
        nl = figure_normal(g.kind, g.param, lr, t);
    } else {
        bool unused;
        nl = figure_normal_inside(g.kind, g.param, lr, t, false, unused);
    }
    V3 n = normalize(rotate(g.rot, nl));
    cosv = fabsf(dot(n, d));
    return true;
}
// src/objects/sample.rs:19-22 over :53-57 (box) and :66-74 (ellipsoid)
RTX_DEV float light_area_pdf(const Geom &g, float box_area_pdf, V3 point)
{
    // a box's pdf does not depend on the point: 1 / (8 (ab + bc + ac)), computed at upload
    if (g.kind == kBox) return box_area_pdf;
    V3 q = rotate_inv(g.rot, point - g.pos);
    V3 n = cdiv(q, g.param);
    n = cmul(n, n);
    V3 r = cmul(g.param, g.param);
    float denom = (n.x * r.y * r.z + r.x * n.y * r.z) + r.x * r.y * n.z;
    return 1.0f / (4.0f * kPi * __builtin_sqrtf(denom));
}
// src/random.rs:112-128
RTX_DEV float intersection_pdf(const Geom &g, float box_area_pdf, V3 o, V3 d, float t, float cosv, V3 initial_point)
{
    V3 obj_point = o + d * t;
    float dist = len2(initial_point - obj_point);
    float pdf = light_area_pdf(g, box_area_pdf, obj_point) * dist / cosv;
    return finite_f(pdf) ? pdf : 0.0f;
}
// src/random.rs:83-109.  Returns the sum over lights divided by their number.
//
// With many lights almost every (ray, light) pair is a miss.  A ray that passes a light's
// bounding sphere -- enlarged by the zone in which the rounding of the exact test can still
// report a hit -- at a distance cannot "hit" the light, and a miss adds nothing to the sum, so
// such lights are skipped: each lane first collects the lights whose sphere its ray touches in a bit mask
// (wave-uniform loop, ~20 instructions per light), then the lanes work through their own
// candidates, lowest index first, which keeps the order of the additions.
// MANY_LIGHTS selects the candidate-mask form at compile time (kernels are instantiated both
// ways and the launcher picks by light count), so few-light scenes do not carry its registers.
// `d` = the direction normalised once (Ray::new, src/ray.rs:11-16); the caller has it already.
template <bool STATS, bool MANY_LIGHTS>
RTX_DEV float light_pdf(const Table &tab, V3 p, V3 d, Counters<STATS> &cnt)
{
    if (tab.n_lights == 0) return 0.0f;
    float pdf = 0.0f;
    if constexpr (!MANY_LIGHTS) {
        // (written as a loop body, not as a function called from a loop: that form, and a loop-free copy for
        // scenes with one light, measured 2.4 % and 6 % slower on the reference scene)
        for (uint32_t li = 0; li < tab.n_lights; ++li) {
            const uint32_t lidx = (uint32_t)tab.lights[li];
            Geom g = load_geom(tab.geom, lidx);
            const float box_area_pdf = tab.mat[3 * lidx + 2].w;
            cnt.template add<C_LTEST_ELLIPSOID>(g.kind == kEllipsoid);
            cnt.template add<C_LTEST_BOX>(g.kind == kBox);
            float t1, t2, c1, c2;
            if (!light_intersect_cos<true>(g, p, d, t1, c1)) {
                cnt.template add<C_LPDF_0>();
                continue;
            }
            pdf += intersection_pdf(g, box_area_pdf, p, d, t1, c1, p);
            // Ray::new_shifted normalises the (already unit) direction once more
            V3 d2 = normalize(d);
            V3 o2 = (p + d * t1) + d2 * kRayEps;
            cnt.template add<C_LTEST_ELLIPSOID>(g.kind == kEllipsoid);
            cnt.template add<C_LTEST_BOX>(g.kind == kBox);
            if (!light_intersect_cos<true>(g, o2, d2, t2, c2)) {
                cnt.template add<C_LPDF_1>();
                continue;
            }
            pdf += intersection_pdf(g, box_area_pdf, o2, d2, t2, c2, p);
            cnt.template add<C_LPDF_2>();
        }
    } else {
        for (uint32_t chunk = 0; chunk < tab.n_lights; chunk += 64u) {
            const uint32_t n = tab.n_lights - chunk < 64u ? tab.n_lights - chunk : 64u;
            unsigned long long candidates = 0ull;
            for (uint32_t k = 0; k < n; ++k) {
                const float4 b = tab.light_bounds[2 * (chunk + k)];
                const float slack = tab.light_bounds[2 * (chunk + k) + 1].x;
                V3 oc = v3(b.x, b.y, b.z) - p;
                float along = dot(oc, d);
                float oc2 = dot(oc, oc);
                float dist2 = oc2 - along * along;
                // the zone in which the f32 intersection formula can report a hit grows with the
                // distance of the origin (rtx_api.hip: upload_scene)
                const float zone2 = b.w + slack * oc2;
                // written so that a NaN never skips a light
                bool skip = (dist2 > zone2) || (along < 0.0f && oc2 > zone2);
                if (!skip) candidates |= 1ull << k;
                if constexpr (STATS) {
                    // a skipped light is a light the reference tested and missed
                    int kind = __float_as_int(tab.geom[3 * tab.lights[chunk + k] + 1].w) & 0xff;
                    cnt.template add<C_LTEST_ELLIPSOID>(skip && kind == kEllipsoid);
                    cnt.template add<C_LTEST_BOX>(skip && kind == kBox);
                    cnt.template add<C_LPDF_0>(skip);
                }
            }
            while (candidates != 0ull) {
                const uint32_t k = (uint32_t)__builtin_ctzll(candidates);
                candidates &= candidates - 1ull;
                const uint32_t lidx = (uint32_t)tab.lights[chunk + k];
                Geom g = load_geom(tab.geom, lidx);
                const float box_area_pdf = tab.mat[3 * lidx + 2].w;
                cnt.template add<C_LTEST_ELLIPSOID>(g.kind == kEllipsoid);
                cnt.template add<C_LTEST_BOX>(g.kind == kBox);
                float t1, t2, c1, c2;
                if (!light_intersect_cos<false>(g, p, d, t1, c1)) {
                    cnt.template add<C_LPDF_0>();
                    continue;
                }
                pdf += intersection_pdf(g, box_area_pdf, p, d, t1, c1, p);
                V3 d2 = normalize(d);
                V3 o2 = (p + d * t1) + d2 * kRayEps;
                cnt.template add<C_LTEST_ELLIPSOID>(g.kind == kEllipsoid);
                cnt.template add<C_LTEST_BOX>(g.kind == kBox);
                if (!light_intersect_cos<false>(g, o2, d2, t2, c2)) {
                    cnt.template add<C_LPDF_1>();
                    continue;
                }
                pdf += intersection_pdf(g, box_area_pdf, o2, d2, t2, c2, p);
                cnt.template add<C_LPDF_2>();
            }
        }
    }
    // x / 1.0f is x: the division is skipped for a single light
    return tab.n_lights == 1 ? pdf : pdf / (float)tab.n_lights;
}
// src/random.rs:143-161
// d = the sampled direction as drawn, d_unit = normalize(d): the cosine pdf takes the former
// (src/random.rs:55-57), the light pdf builds a ray from it, which normalises it (src/random.rs:88).
template <bool STATS, bool MANY_LIGHTS>
RTX_DEV float mis_pdf(const Table &tab, V3 p, V3 n, V3 d, V3 d_unit, Counters<STATS> &cnt)
{
    float a = tab.n_lights == 0 ? 1.0f : 0.5f;
    float cos_pdf = rmax(dot(n, d), 0.0f) / kPi;
#if defined(RTX_X_REPEAT_PARTS) && ((RTX_X_REPEAT_PARTS) & 1)
    {
        V3 p2 = p;
        asm volatile("" : "+v"(p2.x), "+v"(d_unit.x));
        float extra = light_pdf<STATS, MANY_LIGHTS>(tab, p2, d_unit, cnt);
        asm volatile("" : : "v"(extra));
    }
#endif
    return cos_pdf * a + light_pdf<STATS, MANY_LIGHTS>(tab, p, d_unit, cnt) * (1.0f - a);
}

// ------------------------------------------------------------------ sampling
// src/random.rs:33-53 in two steps.  The cosine-weighted direction ends with normalising
// cross(xi, n), the light-sampled one (src/random.rs:76-80) with normalising the vector to the
// sampled point: the caller runs that one normalisation for both kinds of lanes together.
struct CosineDraw {
    float x, y, z;   // the point on the unit hemisphere
    V3 xi;           // first tangent
};
// returns cross(xi, n), whose normalisation is the second tangent
RTX_DEV V3 cosine_sample_begin(V3 n, uint32_t w_theta, uint32_t w_r, CosineDraw &c)
{
    float theta = gen_range(w_theta, 0.0f, 2.0f * kPi);
    float r = __builtin_sqrtf(gen_range(w_r, 0.0f, 1.0f));
    float st, ct;
    sincos_2pi(theta, st, ct);
    c.x = r * ct;
    c.y = r * st;
    c.z = __builtin_sqrtf(1.0f - c.x * c.x - c.y * c.y);

    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    float m = rmin(rmin(ax, ay), az);
    V3 xi = v3(ax > m ? 0.0f : 1.0f, ay > m ? 0.0f : 1.0f, az > m ? 0.0f : 1.0f);
    c.xi = normalize(xi - n * dot(xi, n));
    return cross(c.xi, n);
}
RTX_DEV V3 cosine_sample_end(const CosineDraw &c, V3 yi, V3 n)
{
    return v3((c.xi.x * c.x + yi.x * c.y) + n.x * c.z,
              (c.xi.y * c.x + yi.y * c.y) + n.y * c.z,
              (c.xi.z * c.x + yi.z * c.y) + n.z * c.z);
}
// src/objects/sample.rs:13-17 over :25-51 (box) and :60-64, :78-85 (ellipsoid).
// Word 0 of the draw block: bit 30 = the face sign; words 1..3: the floats.
// plain_boxes (the same in every lane): each lane's light has kGeomPlainBoxLight -- a box in the
// identity pose with moderate half-sizes.  Its surface point is finite and has no -0 component
// (u * 2a - a and +-b with b > 0 never give one), so the identity rotation would return it bit
// for bit (see ray_is_plain) and is skipped.
RTX_DEV V3 light_surface_sample(const Geom &g, uint4 w, bool plain_boxes)
{
    V3 point;
    if (g.kind == kBox) {
        float a = g.param.x, b = g.param.y, c = g.param.z;
        float area = (a * b + b * c) + a * c;
        float x = gen_range(w.y, 0.0f, area);
        // the face-picking chain never yields the x faces: its third branch repeats z
        bool y_face = !(x < a * b) && (x < a * b + a * c);
        float sgn = ((w.x >> 30) & 1u) == 0u ? -1.0f : 1.0f;
        // the picked axis carries +-size, the two others are redrawn in axis order from
        // words 2 and 3 (a zero half-size on the picked axis is rejected at scene set-up)
        float px = gen_range(w.z, -a, a);
        float py, pz;
        if (y_face) {
            py = sgn * b;
            pz = gen_range(w.w, -c, c);
        } else {
            py = gen_range(w.w, -b, b);
            pz = sgn * c;
        }
        point = v3(px, py, pz);
    } else {
        float phi = u24(w.y) * kPi;
        float z = u24(w.z) * 2.0f - 1.0f;
        float sp, cp;
        sincos_2pi(phi, sp, cp);
        float x = __builtin_sqrtf(1.0f - z * z) * cp;
        float y = __builtin_sqrtf(1.0f - z * z) * sp;
        point = cmul(v3(x, y, z), g.param);
    }
    if (plain_boxes) return point + g.pos;
    return rotate(g.rot, point) + g.pos;
}

// --------------------------------------------------------- specular helpers
RTX_DEV V3 reflect_dir(V3 d, V3 n) { return d - (n * 2.0f) * dot(d, n); }
RTX_DEV bool refract_dir(V3 d, V3 n, float eta, V3 &out)
{
    float cos1 = -dot(n, d);
    float sin2 = eta * __builtin_sqrtf(1.0f - cos1 * cos1);
    if (fabsf(sin2) > 1.0f) return false;
    float cos2 = __builtin_sqrtf(1.0f - sin2 * sin2);
    out = d * eta + n * (eta * cos1 - cos2);
    return true;
}
RTX_DEV float schlick(float eta, float cosv)
{
    float r0 = (eta - 1.0f) / (eta + 1.0f);
    r0 = r0 * r0;
    float x = 1.0f - cosv;
    float x2 = x * x;
    float x4 = x2 * x2;
    return r0 + (1.0f - r0) * (x * x4);
}

// What a kernel needs now and then but not in its inner loops -- the camera (once per sample), the
// background colour (a ray that leaves the scene), the constant draw words of RTX_RNG_CONST -- is copied to LDS at
// the start: as kernel arguments these numbers sit in scalar registers through the whole loop, where the compiler
// spills and reloads them around every surface interaction (reference scene at 4K: 893 -> 869 ms with the camera
// alone moved).  Layout in floats: camera position 0-2, axes 4-12, tangents 13-14, background 16-18, draw words 20-23,
// and the grid of the BVH's node boxes (rtx_bvh_traverse.hpp: bvh_grid): base 24-26, inverse step 28-30, far slope
// 31, step 32-34.
constexpr uint32_t kUniformFloats = 36u;
// bytes of the object table in LDS (geometry, materials, the by-category list, the light list) without and with the
// uniforms behind it
__host__ __device__ inline uint32_t table_only_lds_bytes(uint32_t n_prims, uint32_t n_lights)
{
    return (n_prims * 6u * (uint32_t)sizeof(float4) + (n_lights + kScanHeader + n_prims + 2u) * (uint32_t)sizeof(int32_t) + 15u) & ~15u;
}
__host__ __device__ inline uint32_t table_lds_bytes(uint32_t n_prims, uint32_t n_lights)
{
    return table_only_lds_bytes(n_prims, n_lights) + kUniformFloats * (uint32_t)sizeof(float);
}
// every thread of the workgroup calls it; the caller synchronises before the first read
RTX_DEV void stage_uniforms(const RenderParams &P, float *u)
{
    const uint32_t k = threadIdx.x;
    if (k < kUniformFloats) {
        float v = 0.0f;
        if (k < 3u) v = P.cam_pos[k];
        else if (k >= 4u && k < 13u) v = P.cam_axis[k - 4u];
        else if (k == 13u) v = P.tg_fov_x;
        else if (k == 14u) v = P.tg_fov_y;
        else if (k >= 16u && k < 19u) v = P.bg[k - 16u];
        else if (k >= 20u && k < 24u) v = __uint_as_float(P.const_words[k - 20u]);
        else if (k >= 24u && k < 27u) v = P.bvh.grid_base[k - 24u];
        else if (k >= 28u && k < 31u) v = P.bvh.grid_inv_step[k - 28u];
        else if (k == 31u) v = P.bvh.far_slope;
        else if (k >= 32u && k < 35u) v = P.bvh.grid_step[k - 32u];
        u[k] = v;
    }
}

// ------------------------------------------------------------ path stepping
struct Path {
    V3 o, d;      // current ray
    V3 T;         // throughput
    V3 L;         // radiance gathered so far by this sample
    uint32_t depth;
};

RTX_DEV uint4 draw_block(const RenderParams &P, uint32_t pixel, uint32_t sample, uint32_t slot)
{
    if (P.rng_const) return make_uint4(P.const_words[0], P.const_words[1], P.const_words[2], P.const_words[3]);
    return philox4x32_10(pixel, sample, slot, 0u, P.seed_lo, P.seed_hi);
}
// KEYS_PER_CALL: see philox4x32_10.  Measured per kernel: the kernels of few-light scenes gain about 1 % (reference
// scene at 4K 867.4 -> 857.6 ms, glass scene 183.7 -> 181.9 ms), the many-lights BVH kernel loses 3 % (10 000 objects
// 132.1 -> 136.5 ms, although its scalar spills go from 27 to none), so the callers pass !MANY_LIGHTS.
template <bool KEYS_PER_CALL = false>
RTX_DEV uint4 draw_block(const RenderParams &P, const Table &tab, uint32_t pixel, uint32_t sample, uint32_t slot)
{
    if (!tab.has_uniforms) return draw_block(P, pixel, sample, slot);
    if (P.rng_const) return *reinterpret_cast<const uint4 *>(tab.uniforms + 20);
    // (the wavefront kernels measured no gain or a loss: glass scene 326.8 -> 330.4 ms)
    if (KEYS_PER_CALL && tab.keys_per_call) return philox4x32_10<true>(pixel, sample, slot, 0u, P.seed_lo, P.seed_hi);
    return philox4x32_10<false>(pixel, sample, slot, 0u, P.seed_lo, P.seed_hi);
}

// src/main.rs:17-21 + src/camera.rs:15-22: pixel (i, j), j counted from the bottom row
RTX_DEV void camera_ray(const RenderParams &P, uint32_t i, uint32_t j, float du, float dv, V3 &o, V3 &d)
{
    float u = ((float)i + du) / (float)P.width * 2.0f - 1.0f;
    float v = ((float)j + dv) / (float)P.height * 2.0f - 1.0f;
    float dx = u * P.tg_fov_x, dy = v * P.tg_fov_y, dz = 1.0f;
    const float *m = P.cam_axis;
    V3 dir = v3((m[0] * dx + m[3] * dy) + m[6] * dz,
                (m[1] * dx + m[4] * dy) + m[7] * dz,
                (m[2] * dx + m[5] * dy) + m[8] * dz);
    o = v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    d = normalize(dir);
}
// The same with the camera read from its LDS copy when the kernel has one (Table::uniforms): the same numbers, the
// same operations.
RTX_DEV void camera_ray(const RenderParams &P, const Table &tab, uint32_t i, uint32_t j, float du, float dv, V3 &o, V3 &d)
{
    if (!tab.has_uniforms) {
        camera_ray(P, i, j, du, dv, o, d);
        return;
    }
    const float4 c0 = *reinterpret_cast<const float4 *>(tab.uniforms), c1 = *reinterpret_cast<const float4 *>(tab.uniforms + 4),
                 c2 = *reinterpret_cast<const float4 *>(tab.uniforms + 8), c3 = *reinterpret_cast<const float4 *>(tab.uniforms + 12);
    const float m[9] = {c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x};
    float u = ((float)i + du) / (float)P.width * 2.0f - 1.0f;
    float v = ((float)j + dv) / (float)P.height * 2.0f - 1.0f;
    float dx = u * c3.y, dy = v * c3.z, dz = 1.0f;
    V3 dir = v3((m[0] * dx + m[3] * dy) + m[6] * dz,
                (m[1] * dx + m[4] * dy) + m[7] * dz,
                (m[2] * dx + m[5] * dy) + m[8] * dz);
    o = v3(c0.x, c0.y, c0.z);
    d = normalize(dir);
}
RTX_DEV V3 background(const RenderParams &P, const Table &tab)
{
    if (!tab.has_uniforms) return v3(P.bg[0], P.bg[1], P.bg[2]);
    const float4 b = *reinterpret_cast<const float4 *>(tab.uniforms + 16);
    return v3(b.x, b.y, b.z);
}

// Emission of the object that was hit (src/trace.rs:23,70).  Returns true when the path
// ends here because its child ray would be cut before intersecting (src/trace.rs:12-14):
// the scatter at the last allowed depth multiplies a zero and is skipped.
RTX_DEV bool surface_emission(const RenderParams &P, const Table &tab, Path &p, int idx)
{
    float4 m1 = tab.mat[3 * idx + 1];
    p.L = p.L + cmul(p.T, v3(m1.x, m1.y, m1.z));
    return p.depth + 1 >= P.ray_depth;
}

// The material switch of src/trace.rs:25-68 at object `idx`, distance t along the path's
// ray: picks the next ray and updates the throughput.  Returns true when the path ends
// here (a rejected diffuse sample).  `material` is the object's material; kernels that
// run one material per launch pass it as a constant.
template <bool STATS, bool MANY_LIGHTS, bool MIXED = false>
RTX_DEV bool bounce(const RenderParams &P, const Table &tab, Path &p, int idx, float t, int material,
                    uint32_t pixel, uint32_t sample, Counters<STATS> &cnt)
{
    float4 m0 = tab.mat[3 * idx + 0], m1 = tab.mat[3 * idx + 1];
    V3 color = v3(m0.x, m0.y, m0.z);

#if defined(RTX_X_REPEAT_PARTS) && ((RTX_X_REPEAT_PARTS) & 8)
    {
        V3 o2 = p.o;
        asm volatile("" : "+v"(o2.x), "+v"(t));
        Hit h2 = finish_hit<MIXED>(tab, idx, o2, p.d, t, material == kDielectric);
        asm volatile("" : : "v"(h2.n.x), "v"(h2.n.y), "v"(h2.n.z));
    }
#endif
    Hit h = finish_hit<MIXED>(tab, idx, p.o, p.d, t, material == kDielectric);
    V3 point = p.o + p.d * t;
    V3 n = h.n;

    // Lanes of a wave may hold different materials, and what the three branches have in common is
    // written once for all of them: the draw block of this surface interaction (the diffuse sample
    // and the dielectric's coin take it from the same counter; a metallic surface draws nothing, a
    // dielectric one that cannot refract ignores it) and the normalisation of the new direction.
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (material != kMetallic) w = draw_block<!MANY_LIGHTS>(P, tab, pixel, sample, p.depth + 1);
    V3 dir;   // the new direction as the material gives it, before Ray::new_shifted normalises it
    if (material == kDiffuse) {
        // v = the vector whose normalisation both kinds of sample end with (see cosine_sample_begin)
        V3 v;
        CosineDraw cd;
        bool pick_cosine = tab.n_lights == 0 || (w.x >> 31) == 0u;
        if (pick_cosine) {
            cnt.template add<C_COSINE>();
            v = cosine_sample_begin(n, w.y, w.z, cd);
        } else {
            uint32_t li = __umulhi(w.x << 2, tab.n_lights);
            Geom lg = load_geom(tab.geom, (uint32_t)tab.lights[li]);
            cnt.template add<C_LSAMPLE_ELLIPSOID>(lg.kind == kEllipsoid);
            cnt.template add<C_LSAMPLE_BOX>(lg.kind == kBox);
            const bool plain_boxes = __ballot(!(lg.flags & kGeomPlainBoxLight)) == 0ull;
            v = light_surface_sample(lg, w, plain_boxes) - point;
        }
        const V3 u = normalize(v);
        dir = u;
        if (pick_cosine) dir = cosine_sample_end(cd, u, n);
        if (dot(dir, n) < 0.0f) return true;
        cnt.template add<C_PDF_EVALS>();
    } else if (material == kMetallic) {
        dir = reflect_dir(p.d, n);
        p.T = cmul(p.T, color);
        cnt.template add<C_METALLIC>();
    } else {
        float ior = m1.w;
        float eta = h.inside ? ior : 1.0f / ior;
        V3 rdir = reflect_dir(p.d, n);
        V3 fdir;
        bool can_refract = refract_dir(p.d, n, eta, fdir);
        float coeff = schlick(eta, -dot(p.d, n));
        const bool take_refracted = can_refract && u24(w.y) < 1.0f - coeff;
        if (take_refracted && !h.inside) p.T = cmul(p.T, color);
        dir = take_refracted ? fdir : rdir;
        cnt.template add<C_DIELECTRIC>();
    }
    // Ray::new inside the light pdf (diffuse) and Ray::new_shifted for the next segment
    // (src/trace.rs:43, 125, 141) normalise the same vector: once here
    const V3 nd = normalize(dir);
    if (material == kDiffuse) {
        float4 m2 = tab.mat[3 * idx + 2];
        V3 color_obj = v3(m2.x, m2.y, m2.z);  // color / pi, divided at upload
        float pdf = mis_pdf<STATS, MANY_LIGHTS>(tab, point, n, dir, nd, cnt);
        if (!finite_f(pdf) || pdf < 1e-6f) return true;
        float cosv = dot(n, nd);
        p.T = (cmul(p.T, color_obj) * cosv) / pdf;
        cnt.template add<C_DIFFUSE>();
    }
    p.o = point + nd * kRayEps;
    p.d = nd;
    // A throughput of exactly zero -- the path bounced off a black object, like the light of
    // the reference's scene, which has no COLOR -- stays zero, and every later term of the
    // radiance is T * (finite emission or background) = +-0, which changes nothing when added.
    // The reference goes on tracing such a path (src/trace.rs:43-48); ending it here gives the
    // same bits.  Only in scenes whose numbers are all finite and moderate (stop_on_zero, set
    // at upload): a non-finite factor later on would turn the zero into a NaN.
    if (P.stop_on_zero && p.T.x == 0.0f && p.T.y == 0.0f && p.T.z == 0.0f) return true;
    p.depth += 1;
    return false;
}

// A black diffuse or metallic surface ends the path (scenes with stop_on_zero only): the
// scatter there either rejects its sample or leaves T * 0 = +-0 -- T, the cosine and the pdf
// being finite -- and a path with zero throughput is over (see bounce).  Checked after the
// surface's emission has been added, before any of the scatter's work.
RTX_DEV bool black_surface_ends(const RenderParams &P, const Table &tab, const Path &p, int idx)
{
    if (!P.stop_on_zero) return false;
    const float4 m0 = tab.mat[3 * idx + 0];
    return __float_as_int(m0.w) != kDielectric && m0.x == 0.0f && m0.y == 0.0f && m0.z == 0.0f && finite_f(p.T.x) &&
           finite_f(p.T.y) && finite_f(p.T.z);
}

// One whole surface interaction (emission, depth cut, material switch); true = path ended.
template <bool STATS, bool MANY_LIGHTS, bool MIXED = false>
RTX_DEV bool scatter(const RenderParams &P, const Table &tab, Path &p, int idx, float t,
                     uint32_t pixel, uint32_t sample, Counters<STATS> &cnt)
{
    if (surface_emission(P, tab, p, idx)) return true;
    if (black_surface_ends(P, tab, p, idx)) return true;
    int material = __float_as_int(tab.mat[3 * idx + 0].w);
    return bounce<STATS, MANY_LIGHTS, MIXED>(P, tab, p, idx, t, material, pixel, sample, cnt);
}

}  // namespace rtx
