/*
 * rt_oracle.c -- scalar CPU restatement of the reference path tracer's
 * per-pixel trace loop.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see
 * rt_oracle.h).  Build: gcc -O2 -ffp-contract=off (see oracle/Makefile).
 *
 * Conventions taken from the reference's dependencies (nalgebra 0.32 / rand 0.8,
 * sources not available here; orders restated from their published kernels):
 *   dot(a,b)      = (ax*bx + ay*by) + az*bz
 *   length2(a)    = dot(a,a);  length = sqrt(length2)
 *   normalize(a)  = a / length(a)          (three divisions)
 *   cross(a,b)    = (ay*bz-az*by, az*bx-ax*bz, ax*by-ay*bx)
 *   M*v           = ((col0*v.x) + col1*v.y) + col2*v.z   per component
 *   q*v           = ((t*q.w) + cross(q.xyz,t)) + v,  t = cross(q.xyz,v)*2
 *   q.inverse()   = conjugate
 *   f32::min/max  ignore a NaN operand, -0 < +0; argmax = first strict maximum
 *   gen::<f32>()  = (u32 >> 8) * 2^-24;  gen_range(lo..hi) = ((u32 >> 9) * 2^-23) * (hi-lo) + lo
 *   gen_bool(0.5) = top bit of the draw is 0
 *
 * Random numbers.  The reference draws from one sequential ThreadRng; that
 * stream cannot be reproduced, so draws come from a counter-based generator:
 * one Philox4x32-10 block per (pixel, sample, slot), key = 64-bit seed, where
 * slot 0 feeds the pixel jitter (main.rs:17-18) and slot k >= 1 feeds whatever
 * the k-th surface hit of the path draws (trace.rs:35, trace.rs:90).  Within a
 * block: word 0 supplies the booleans and the light index, words 1..3 the
 * floats, in the order the reference draws them.
 *
 * sin/cos.  The reference calls the platform libm through f32::sin/cos.  To let
 * the device path be compared bit for bit, both sides evaluate one documented
 * polynomial (orc_sincos below; Cody-Waite reduction by pi/2 and the Cephes
 * single-precision minimax polynomials), accurate to about 1 ulp on [0, 2pi].
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.14159274101257324f /* std::f32::consts::PI */
#define ORC_EPS 1e-4f               /* ray.rs:3 */

typedef struct { float x, y, z; } v3;
typedef struct { v3 o, d; } ray_t;
typedef struct { float t; v3 n; int inside; } hit_t;

/* ------------------------------------------------------------------ vectors */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add3(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 div3(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 scale3(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 unscale3(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V(-a.x, -a.y, -a.z); }
/* ORC_ALT_*: other plausible operation orders of the dependency kernels, for measuring how much
 * hangs on the recalled ones (oracle/Makefile: alternatives; tests/test_order_fragility.py). */
#ifdef ORC_ALT_DOT
static inline float dot3(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
#else
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
#endif
static inline float length2_3(v3 a) { return dot3(a, a); }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
#ifdef ORC_ALT_NORMALIZE
static inline v3 normalize3(v3 a) { return scale3(a, 1.0f / length3(a)); }
#else
static inline v3 normalize3(v3 a) { return unscale3(a, length3(a)); }
#endif
static inline v3 cross3(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* f32::min / f32::max: a NaN operand is ignored.  For operands that compare equal (+0 and
 * -0) the language leaves the choice open; -0 is taken as the smaller, as the IEEE
 * minimumNumber / maximumNumber operations (and the GPU's v_min_f32 / v_max_f32) do. */
static inline float rmin(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? a : b;
    return (b < a) ? b : a;
}
static inline float rmax(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? b : a;
    return (b > a) ? b : a;
}
static inline int is_finite_f(float x) { return (x - x) == 0.0f; }

/* unit quaternion (x,y,z,w) times vector */
static inline v3 quat_rotate(const float q[4], v3 v)
{
    v3 u = V(q[0], q[1], q[2]);
    v3 t = scale3(cross3(u, v), 2.0f);
    v3 c = cross3(u, t);
#ifdef ORC_ALT_QUAT
    return add3(add3(v, scale3(t, q[3])), c);
#else
    return add3(add3(scale3(t, q[3]), c), v);
#endif
}
static inline v3 quat_rotate_inv(const float q[4], v3 v)
{
    float qi[4] = {-q[0], -q[1], -q[2], q[3]};
    return quat_rotate(qi, v);
}
/* column-major 3x3 times vector */
static inline v3 mat3_mul(const float m[9], v3 v)
{
#ifdef ORC_ALT_MATVEC
    return V(m[0] * v.x + (m[3] * v.y + m[6] * v.z),
             m[1] * v.x + (m[4] * v.y + m[7] * v.z),
             m[2] * v.x + (m[5] * v.y + m[8] * v.z));
#else
    return V((m[0] * v.x + m[3] * v.y) + m[6] * v.z,
             (m[1] * v.x + m[4] * v.y) + m[7] * v.z,
             (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
#endif
}

/* --------------------------------------------------------------------- rays */
/* ray.rs:11-16 */
static inline ray_t ray_new(v3 origin, v3 direction)
{
    ray_t r; r.o = origin; r.d = normalize3(direction); return r;
}
/* ray.rs:18-24 */
static inline ray_t ray_new_shifted(v3 origin, v3 direction)
{
    ray_t r; r.d = normalize3(direction); r.o = add3(origin, scale3(r.d, ORC_EPS)); return r;
}

/* ---------------------------------------------------------------------- rng */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    const orc_scene *scene;
    const orc_opts *opts;
    int n_lights;
    int32_t *lights;
    orc_stats *stats; /* may be NULL */
    int stop_on_zero; /* FORWARD order: end a path whose throughput is exactly zero */
} ctx_t;

static void draw_block(const ctx_t *c, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t w[4])
{
    if (c->opts->rng_mode == ORC_RNG_CONST) {
        memcpy(w, c->opts->const_words, 4 * sizeof(uint32_t));
        return;
    }
    uint32_t ctr[4] = {pixel, sample, slot, 0u};
    uint32_t key[2] = {(uint32_t)c->opts->seed, (uint32_t)(c->opts->seed >> 32)};
    orc_philox4x32_10(ctr, key, w);
}
static inline float u24(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-8f; }
static inline float u23(uint32_t w) { return (float)(w >> 9) * 1.1920928955078125e-7f; }
static inline float gen_range_f(uint32_t w, float lo, float hi) { return u23(w) * (hi - lo) + lo; }

/* ------------------------------------------------------------------ sin/cos */
void orc_sincos(float x, float *s_out, float *c_out)
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
    switch (k & 3) {
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

/* ------------------------------------------------------------ intersections */
/* geometry.rs:56-71 */
static int plane_intersect(v3 normal, ray_t r, hit_t *h)
{
    float t = -dot3(r.o, normal) / dot3(r.d, normal);
    int inside = dot3(normal, r.o) < 0.0f;
    if (t < 0.0f) return 0;
    h->t = t; h->n = normal; h->inside = inside;
    return 1;
}
/* geometry.rs:73-107 */
static int ellipsoid_intersect(v3 radii, ray_t r, hit_t *h)
{
    v3 u = div3(r.o, radii);
    v3 v = div3(r.d, radii);
    float a = length2_3(v);
    float b = dot3(u, v);
    float c = length2_3(u) - 1.0f;
    float det = b * b - a * c;
    if (det < 0.0f) return 0;
    float t1 = (-b + sqrtf(det)) / a;
    float t2 = (-b - sqrtf(det)) / a;
    float lo = rmin(t1, t2), hi = rmax(t1, t2);
    float t;
    if (lo > 0.0f) t = lo;
    else if (hi > 0.0f) t = hi;
    else return 0;
    h->t = t;
    h->inside = length2_3(u) < 1.0f;
    h->n = div3(add3(u, scale3(v, t)), radii);
    return 1;
}
/* geometry.rs:109-147 */
static int box_intersect(v3 sizes, ray_t r, hit_t *h)
{
    const float s[3] = {sizes.x, sizes.y, sizes.z};
    const float o[3] = {r.o.x, r.o.y, r.o.z};
    const float d[3] = {r.d.x, r.d.y, r.d.z};
    float l[3], g[3];
    for (int i = 0; i < 3; ++i) {
        float t1 = (s[i] - o[i]) / d[i];
        float t2 = (-s[i] - o[i]) / d[i];
        l[i] = rmin(t1, t2);
        g[i] = rmax(t1, t2);
    }
    float t1 = rmax(rmax(l[0], l[1]), l[2]);
    float t2 = rmin(rmin(g[0], g[1]), g[2]);
    float t;
    if (t1 > t2) return 0;
    else if (t1 >= 0.0f) t = t1;
    else if (t2 >= 0.0f) t = t2;
    else return 0;

    float n[3];
    for (int i = 0; i < 3; ++i) n[i] = (o[i] + t * d[i]) / s[i];
    int im = 0;
    float vm = fabsf(n[0]);
    for (int i = 1; i < 3; ++i) {
        float a = fabsf(n[i]);
        if (a > vm) { vm = a; im = i; }
    }
    n[(im + 1) % 3] = 0.0f;
    n[(im + 2) % 3] = 0.0f;
    h->t = t;
    h->inside = rmax(rmax(fabsf(o[0] / s[0]), fabsf(o[1] / s[1])), fabsf(o[2] / s[2])) < 1.0f;
    h->n = V(n[0], n[1], n[2]);
    return 1;
}
static int figure_intersect(uint32_t kind, v3 param, ray_t r, hit_t *h)
{
    switch (kind) {
    case ORC_PLANE: return plane_intersect(param, r, h);
    case ORC_ELLIPSOID: return ellipsoid_intersect(param, r, h);
    default: return box_intersect(param, r, h);
    }
}
/* geometry.rs:22-37 and :39-54 (identical bodies) */
static int positioned_intersect(const orc_prim *p, ray_t r, hit_t *h)
{
    ray_t local;
    local.o = quat_rotate_inv(p->rotation, sub3(r.o, ld3(p->position)));
    local.d = quat_rotate_inv(p->rotation, r.d);
    if (!figure_intersect(p->kind, ld3(p->param), local, h)) return 0;
    h->n = normalize3(quat_rotate(p->rotation, h->n));
    if (dot3(h->n, r.d) > 0.0f) h->n = neg3(h->n);
    return 1;
}
/* trace.rs:102-121 with max_dist = +inf */
static int closest_hit(const ctx_t *c, ray_t r, hit_t *best)
{
    const orc_scene *sc = c->scene;
    float ray_length = length3(r.d);
    int best_i = -1;
    for (uint32_t i = 0; i < sc->n_prims; ++i) {
        hit_t h;
        if (c->stats) c->stats->prim_tests[sc->prims[i].kind]++;
        if (!positioned_intersect(&sc->prims[i], r, &h)) continue;
        if (!(h.t * ray_length < INFINITY)) continue;
        if (best_i < 0 || h.t < best->t) { *best = h; best_i = (int)i; }
    }
    return best_i;
}

/* ------------------------------------------------------ light surface pdfs */
/* sample.rs:19-22 with :53-57 and :66-74 */
static float light_area_pdf(const orc_prim *p, v3 point)
{
    v3 q = quat_rotate_inv(p->rotation, sub3(point, ld3(p->position)));
    if (p->kind == ORC_BOX) {
        float a = p->param[0], b = p->param[1], c = p->param[2];
        float area = 8.0f * ((a * b + b * c) + a * c);
        return 1.0f / area;
    }
    v3 rad = ld3(p->param);
    v3 n = div3(q, rad);
    n = mul3(n, n);
    v3 r = mul3(rad, rad);
    float denom = (n.x * r.y * r.z + r.x * n.y * r.z) + r.x * r.y * n.z;
    return 1.0f / (4.0f * ORC_PI * sqrtf(denom));
}
/* random.rs:112-128 */
static float intersection_pdf(const orc_prim *p, ray_t r, const hit_t *h, v3 initial_point)
{
    v3 obj_point = add3(r.o, scale3(r.d, h->t));
    float dist = length2_3(sub3(initial_point, obj_point));
    float cosv = fabsf(dot3(r.d, h->n));
    float pdf = light_area_pdf(p, obj_point) * dist / cosv;
    if (!is_finite_f(pdf)) pdf = 0.0f;
    return pdf;
}
/* random.rs:83-109 */
static float light_pdf(const ctx_t *c, v3 p, v3 d)
{
    if (c->n_lights == 0) return 0.0f;
    ray_t r = ray_new(p, d);
    float pdf = 0.0f;
    for (int li = 0; li < c->n_lights; ++li) {
        const orc_prim *obj = &c->scene->prims[c->lights[li]];
        hit_t i1, i2;
        if (c->stats) c->stats->light_tests[obj->kind]++;
        if (!positioned_intersect(obj, r, &i1)) {
            if (c->stats) c->stats->light_pdf[0]++;
            continue;
        }
        pdf += intersection_pdf(obj, r, &i1, p);
        ray_t r2 = ray_new_shifted(add3(r.o, scale3(r.d, i1.t)), r.d);
        if (c->stats) c->stats->light_tests[obj->kind]++;
        if (!positioned_intersect(obj, r2, &i2)) {
            if (c->stats) c->stats->light_pdf[1]++;
            continue;
        }
        pdf += intersection_pdf(obj, r2, &i2, p);
        if (c->stats) c->stats->light_pdf[2]++;
    }
    pdf /= (float)c->n_lights;
    return pdf;
}
/* random.rs:55-57 */
static float cosine_pdf(v3 n, v3 d) { return rmax(dot3(n, d), 0.0f) / ORC_PI; }
/* random.rs:143-161 */
static float mis_pdf(const ctx_t *c, v3 p, v3 n, v3 d)
{
    float a = (c->n_lights == 0) ? 1.0f : 0.5f;
    return cosine_pdf(n, d) * a + light_pdf(c, p, d) * (1.0f - a);
}

/* ----------------------------------------------------------------- sampling */
/* random.rs:33-53 */
static v3 cosine_sample(v3 n, uint32_t w_theta, uint32_t w_r)
{
    float theta = gen_range_f(w_theta, 0.0f, 2.0f * ORC_PI);
    float r = sqrtf(gen_range_f(w_r, 0.0f, 1.0f));
    float st, ct;
    orc_sincos(theta, &st, &ct);
    float x = r * ct;
    float y = r * st;
    float z = sqrtf(1.0f - x * x - y * y);

    v3 z_image = n;
    float m = rmin(rmin(fabsf(n.x), fabsf(n.y)), fabsf(n.z));
    v3 x_image = V(fabsf(n.x) > m ? 0.0f : 1.0f, fabsf(n.y) > m ? 0.0f : 1.0f, fabsf(n.z) > m ? 0.0f : 1.0f);
    x_image = normalize3(sub3(x_image, scale3(n, dot3(x_image, z_image))));
    v3 y_image = normalize3(cross3(x_image, z_image));
    return V((x_image.x * x + y_image.x * y) + z_image.x * z,
             (x_image.y * x + y_image.y * y) + z_image.y * z,
             (x_image.z * x + y_image.z * y) + z_image.z * z);
}
/* sample.rs:13-17 over :25-51 (box) and :60-64 with :78-85 (ellipsoid) */
static v3 light_surface_sample(const orc_prim *p, const uint32_t w[4])
{
    v3 point;
    if (p->kind == ORC_BOX) {
        float s[3] = {p->param[0], p->param[1], p->param[2]};
        float a = s[0], b = s[1], c = s[2];
        float area = (a * b + b * c) + a * c;
        float x = gen_range_f(w[1], 0.0f, area);
        float q[3] = {0.0f, 0.0f, 0.0f};
        if (x < a * b) q[2] = 1.0f;
        else if (x < a * b + a * c) q[1] = 1.0f;
        else q[2] = 1.0f;
        if (((w[0] >> 30) & 1u) == 0u) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; }
        for (int i = 0; i < 3; ++i) q[i] = q[i] * s[i];
        /* sample.rs:44-48: every zero coordinate (-0 included) is redrawn, in axis
         * order; the first takes word 2 and the second word 3 of the block */
        int next = 2;
        for (int i = 0; i < 3; ++i) {
            if (q[i] == 0.0f) {
                q[i] = gen_range_f(w[next], -s[i], s[i]);
                if (next < 3) next++;
            }
        }
        point = V(q[0], q[1], q[2]);
    } else {
        float phi = u24(w[1]) * ORC_PI;
        float z = u24(w[2]) * 2.0f - 1.0f;
        float sp, cp;
        orc_sincos(phi, &sp, &cp);
        float x = sqrtf(1.0f - z * z) * cp;
        float y = sqrtf(1.0f - z * z) * sp;
        point = mul3(V(x, y, z), ld3(p->param));
    }
    return add3(quat_rotate(p->rotation, point), ld3(p->position));
}
/* random.rs:73-81 */
static v3 to_light_sample(ctx_t *c, v3 p, const uint32_t w[4])
{
    uint32_t idx = (uint32_t)(((uint64_t)(w[0] << 2) * (uint64_t)c->n_lights) >> 32);
    const orc_prim *obj = &c->scene->prims[c->lights[idx]];
    if (c->stats) c->stats->light_samples[obj->kind]++;
    v3 p_light = light_surface_sample(obj, w);
    return normalize3(sub3(p_light, p));
}
/* random.rs:135-141 */
static v3 mis_sample(ctx_t *c, v3 p, v3 n, const uint32_t w[4])
{
    int pick_cosine = (c->n_lights == 0) ? 1 : ((w[0] >> 31) == 0u);
    if (pick_cosine) {
        if (c->stats) c->stats->cosine_samples++;
        return cosine_sample(n, w[1], w[2]);
    }
    return to_light_sample(c, p, w);
}

/* ------------------------------------------------------- specular materials */
/* trace.rs:123-126 */
static ray_t reflected_ray(v3 dir, v3 point, v3 normal)
{
    v3 nd = sub3(dir, scale3(scale3(normal, 2.0f), dot3(dir, normal)));
    return ray_new_shifted(point, nd);
}
/* trace.rs:128-142 (the two assert! lines are not restated: they hold for unit inputs) */
static int refracted_ray(v3 dir, v3 point, v3 normal, float eta, ray_t *out)
{
    float cos1 = -dot3(normal, dir);
    float sin2 = eta * sqrtf(1.0f - cos1 * cos1);
    if (fabsf(sin2) > 1.0f) return 0;
    float cos2 = sqrtf(1.0f - sin2 * sin2);
    v3 nd = add3(scale3(dir, eta), scale3(normal, eta * cos1 - cos2));
    *out = ray_new_shifted(point, nd);
    return 1;
}
/* trace.rs:144-149; powi(5) = x * ((x*x)*(x*x)) */
static float schlick(float eta, float cosv)
{
    float r0 = (eta - 1.0f) / (eta + 1.0f);
    r0 = r0 * r0;
    float x = 1.0f - cosv;
    float x2 = x * x;
    float x4 = x2 * x2;
    return r0 + (1.0f - r0) * (x * x4);
}

/* --------------------------------------------------- reference-order tracer */
/* trace.rs:11-100 */
static v3 trace_reference(ctx_t *c, ray_t r, uint32_t depth, uint32_t pixel, uint32_t sample)
{
    const orc_scene *sc = c->scene;
    if (depth >= sc->ray_depth) return V(0, 0, 0);
    hit_t h;
    int idx = closest_hit(c, r, &h);
    if (idx < 0) return ld3(sc->background);
    const orc_prim *obj = &sc->prims[idx];
    v3 point = add3(r.o, scale3(r.d, h.t));
    v3 normal = h.n;
    v3 emitted = ld3(obj->emission);
    v3 color;
    uint32_t w[4];

    if (obj->material == ORC_DIFFUSE) {
        v3 color_obj = unscale3(ld3(obj->color), ORC_PI);
        draw_block(c, pixel, sample, depth + 1, w);
        v3 new_dir = mis_sample(c, point, normal, w);
        if (dot3(new_dir, normal) < 0.0f) {
            color = V(0, 0, 0);
        } else {
            float pdf = mis_pdf(c, point, normal, new_dir);
            if (!is_finite_f(pdf) || pdf < 1e-6f) {
                color = V(0, 0, 0);
            } else {
                ray_t nr = ray_new_shifted(point, new_dir);
                float cosv = dot3(normal, nr.d);
                v3 cin = trace_reference(c, nr, depth + 1, pixel, sample);
                color = unscale3(scale3(mul3(cin, color_obj), cosv), pdf);
            }
        }
    } else if (obj->material == ORC_METALLIC) {
        ray_t rr = reflected_ray(r.d, point, normal);
        v3 cin = trace_reference(c, rr, depth + 1, pixel, sample);
        color = mul3(cin, ld3(obj->color));
    } else {
        /* trace.rs:73-100 */
        float eta = h.inside ? obj->ior : 1.0f / obj->ior;
        ray_t rr = reflected_ray(r.d, point, normal);
        ray_t rf;
        int can_refract = refracted_ray(r.d, point, normal, eta, &rf);
        float coeff = schlick(eta, -dot3(r.d, normal));
        int take_refracted = 0;
        if (can_refract) {
            draw_block(c, pixel, sample, depth + 1, w);
            take_refracted = u24(w[1]) < 1.0f - coeff;
        }
        if (take_refracted) {
            color = trace_reference(c, rf, depth + 1, pixel, sample);
            if (!h.inside) color = mul3(color, ld3(obj->color));
        } else {
            color = trace_reference(c, rr, depth + 1, pixel, sample);
        }
    }
    return add3(color, emitted);
}

/* ----------------------------------------------------- forward-order tracer */
/* The same path as trace_reference, accumulated front to back.  The scatter
 * at the last allowed depth is skipped: its child ray is cut at trace.rs:12-14
 * and multiplies the scattered term by zero. */
static v3 trace_forward(ctx_t *c, ray_t r, uint32_t pixel, uint32_t sample)
{
    const orc_scene *sc = c->scene;
    orc_stats *st = c->stats;
    v3 L = V(0, 0, 0), T = V(1, 1, 1);
    for (uint32_t depth = 0; depth < sc->ray_depth; ++depth) {
        hit_t h;
        if (st) st->rays++;
        int idx = closest_hit(c, r, &h);
        if (idx < 0) { L = add3(L, mul3(T, ld3(sc->background))); break; }
        const orc_prim *obj = &sc->prims[idx];
        if (st) st->hits[obj->kind]++;
        L = add3(L, mul3(T, ld3(obj->emission)));
        if (depth + 1 >= sc->ray_depth) break;
        /* A black diffuse or metallic surface: the scatter either rejects its sample or leaves
         * T * 0 = +-0 (T, cos and pdf being finite), so the path is over either way. */
        if (c->stop_on_zero && obj->material != ORC_DIELECTRIC && obj->color[0] == 0.0f && obj->color[1] == 0.0f &&
            obj->color[2] == 0.0f && is_finite_f(T.x) && is_finite_f(T.y) && is_finite_f(T.z))
            break;
        v3 point = add3(r.o, scale3(r.d, h.t));
        v3 normal = h.n;
        uint32_t w[4];
        if (obj->material == ORC_DIFFUSE) {
            v3 color_obj = unscale3(ld3(obj->color), ORC_PI);
            draw_block(c, pixel, sample, depth + 1, w);
            v3 new_dir = mis_sample(c, point, normal, w);
            if (dot3(new_dir, normal) < 0.0f) break;
            if (st) st->pdf_evals++;
            float pdf = mis_pdf(c, point, normal, new_dir);
            if (!is_finite_f(pdf) || pdf < 1e-6f) break;
            ray_t nr = ray_new_shifted(point, new_dir);
            float cosv = dot3(normal, nr.d);
            T = unscale3(scale3(mul3(T, color_obj), cosv), pdf);
            r = nr;
            if (st) st->diffuse_bounces++;
        } else if (obj->material == ORC_METALLIC) {
            r = reflected_ray(r.d, point, normal);
            T = mul3(T, ld3(obj->color));
            if (st) st->metallic_bounces++;
        } else {
            float eta = h.inside ? obj->ior : 1.0f / obj->ior;
            ray_t rr = reflected_ray(r.d, point, normal);
            ray_t rf;
            int can_refract = refracted_ray(r.d, point, normal, eta, &rf);
            float coeff = schlick(eta, -dot3(r.d, normal));
            int take_refracted = 0;
            if (can_refract) {
                draw_block(c, pixel, sample, depth + 1, w);
                take_refracted = u24(w[1]) < 1.0f - coeff;
            }
            if (take_refracted) {
                if (!h.inside) T = mul3(T, ld3(obj->color));
                r = rf;
            } else {
                r = rr;
            }
            if (st) st->dielectric_bounces++;
        }
        /* every later term is T * (finite emission or background) = +-0: nothing can change L */
        if (c->stop_on_zero && T.x == 0.0f && T.y == 0.0f && T.z == 0.0f) break;
    }
    return L;
}

/* ------------------------------------------------------------------- camera */
/* camera.rs:15-22 (the assert! on |u|,|v| <= 1 is not restated) */
static ray_t camera_ray(const orc_scene *sc, float u, float v)
{
    v3 direction = V(u * sc->tg_fov_x, v * sc->tg_fov_y, 1.0f);
    direction = mat3_mul(sc->cam_axis, direction);
    return ray_new(ld3(sc->cam_pos), direction);
}

/* parser.rs:69-89: emissive ellipsoids and boxes, in object order */
static int collect_lights(const orc_scene *sc, int32_t *out, int cap)
{
    int n = 0;
    for (uint32_t i = 0; i < sc->n_prims; ++i) {
        const orc_prim *p = &sc->prims[i];
        if (length2_3(ld3(p->emission)) == 0.0f) continue;
        if (p->kind == ORC_PLANE) continue;
        if (out && n < cap) out[n] = (int32_t)i;
        n++;
    }
    return n;
}
static int tame(float x, float lo, float hi)
{
    float a = fabsf(x);
    return a >= lo && a <= hi; /* false for NaN */
}

int orc_scene_is_tame(const orc_scene *sc)
{
    if (!sc) return 0;
    for (int k = 0; k < 3; ++k)
        if (!tame(sc->background[k], 0.0f, 1e6f)) return 0;
    for (uint32_t i = 0; i < sc->n_prims; ++i) {
        const orc_prim *p = &sc->prims[i];
        float q2 = 0.0f;
        int nonzero = 0;
        for (int k = 0; k < 3; ++k) {
            if (!tame(p->emission[k], 0.0f, 1e6f) || !tame(p->color[k], 0.0f, 1e3f)) return 0;
            if (!tame(p->position[k], 0.0f, 1e6f)) return 0;
            if (p->kind == ORC_PLANE) {
                if (!tame(p->param[k], 0.0f, 1e6f)) return 0;
                nonzero |= fabsf(p->param[k]) >= 1e-6f;
            } else {
                if (!tame(p->param[k], 1e-4f, 1e6f)) return 0;
                nonzero = 1;
            }
        }
        if (!nonzero) return 0;
        for (int k = 0; k < 4; ++k) {
            if (!tame(p->rotation[k], 0.0f, 2.0f)) return 0;
            q2 += p->rotation[k] * p->rotation[k];
        }
        if (!(q2 >= 0.5f && q2 <= 2.0f)) return 0;
        if (p->material == ORC_DIELECTRIC && !tame(p->ior, 1e-3f, 1e3f)) return 0;
    }
    return 1;
}

static int ctx_init(ctx_t *c, const orc_scene *sc, const orc_opts *opts, orc_stats *stats)
{
    c->scene = sc; c->opts = opts; c->stats = stats;
    c->stop_on_zero = opts->order == ORC_ORDER_FORWARD && !opts->full_paths && orc_scene_is_tame(sc);
    c->n_lights = collect_lights(sc, NULL, 0);
    c->lights = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c->n_lights > 0 ? c->n_lights : 1));
    if (!c->lights) return -1;
    collect_lights(sc, c->lights, c->n_lights);
    return 0;
}

/* ------------------------------------------------------------- entry points */
/* main.rs:13-31.  Pixel (i, j) has j = 0 at the BOTTOM; it is stored in row
 * height-1-j (image.rs:26-29).  The draw-block pixel id is that storage index. */
int orc_render(const orc_scene *sc, const orc_opts *opts, float *rgb, orc_stats *stats)
{
    ctx_t c;
    if (!sc || !opts || !rgb) return -1;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (ctx_init(&c, sc, opts, opts->order == ORC_ORDER_FORWARD ? stats : NULL)) return -2;
    const uint32_t w = sc->width, h = sc->height;
    uint32_t i0 = 0, i1 = w;
    if (opts->col_count) {
        i0 = opts->col_first < w ? opts->col_first : w;
        i1 = opts->col_count < w - i0 ? i0 + opts->col_count : w;
    } else {
        memset(rgb, 0, sizeof(float) * 3u * (size_t)w * h);
    }

    if (opts->order == ORC_ORDER_REFERENCE) {
        for (uint32_t step = 0; step < sc->n_samples; ++step) {
            for (uint32_t i = i0; i < i1; ++i) {
                for (uint32_t j = 0; j < h; ++j) {
                    uint32_t pixel = (h - 1 - j) * w + i;
                    uint32_t blk[4];
                    draw_block(&c, pixel, step, 0, blk);
                    float du = u24(blk[0]), dv = u24(blk[1]);
                    float u = ((float)i + du) / (float)w * 2.0f - 1.0f;
                    float v = ((float)j + dv) / (float)h * 2.0f - 1.0f;
                    ray_t r = camera_ray(sc, u, v);
                    float *px = rgb + 3u * (size_t)pixel;
                    v3 old = ld3(px);
                    v3 color = trace_reference(&c, r, 0, pixel, step);
                    float sf = (float)step;
                    st3(px, unscale3(add3(scale3(old, sf), color), sf + 1.0f));
                }
            }
        }
    } else {
        uint32_t first = opts->sample_first;
        uint32_t count = opts->sample_count ? opts->sample_count : sc->n_samples;
        for (uint32_t i = i0; i < i1; ++i) {
            for (uint32_t j = 0; j < h; ++j) {
                uint32_t pixel = (h - 1 - j) * w + i;
                v3 sum = V(0, 0, 0);
                for (uint32_t s = first; s < first + count; ++s) {
                    uint32_t blk[4];
                    draw_block(&c, pixel, s, 0, blk);
                    float du = u24(blk[0]), dv = u24(blk[1]);
                    float u = ((float)i + du) / (float)w * 2.0f - 1.0f;
                    float v = ((float)j + dv) / (float)h * 2.0f - 1.0f;
                    ray_t r = camera_ray(sc, u, v);
                    if (c.stats) c.stats->samples++;
                    sum = add3(sum, trace_forward(&c, r, pixel, s));
                }
                if (!opts->output_sum) sum = unscale3(sum, (float)count);
                st3(rgb + 3u * (size_t)pixel, sum);
            }
        }
    }
    free(c.lights);
    return 0;
}

int orc_primary(const orc_scene *sc, int32_t *hit_id, float *t, float *normal, uint8_t *inside)
{
    ctx_t c;
    orc_opts opts;
    memset(&opts, 0, sizeof(opts));
    if (!sc || !hit_id) return -1;
    if (ctx_init(&c, sc, &opts, NULL)) return -2;
    const uint32_t w = sc->width, h = sc->height;
    for (uint32_t j = 0; j < h; ++j) {
        for (uint32_t i = 0; i < w; ++i) {
            uint32_t pixel = (h - 1 - j) * w + i;
            float u = ((float)i + 0.5f) / (float)w * 2.0f - 1.0f;
            float v = ((float)j + 0.5f) / (float)h * 2.0f - 1.0f;
            ray_t r = camera_ray(sc, u, v);
            hit_t hh;
            int idx = closest_hit(&c, r, &hh);
            hit_id[pixel] = idx;
            if (idx < 0) { hh.t = 0.0f; hh.n = V(0, 0, 0); hh.inside = 0; }
            if (t) t[pixel] = hh.t;
            if (normal) st3(normal + 3u * (size_t)pixel, hh.n);
            if (inside) inside[pixel] = (uint8_t)hh.inside;
        }
    }
    free(c.lights);
    return 0;
}

int orc_closest_hit(const orc_scene *sc, const float origin[3], const float dir[3],
                    float *t, float normal[3], int *inside)
{
    ctx_t c;
    orc_opts opts;
    memset(&opts, 0, sizeof(opts));
    c.scene = sc; c.opts = &opts; c.stats = NULL; c.n_lights = 0; c.lights = NULL;
    ray_t r; r.o = ld3(origin); r.d = ld3(dir);
    hit_t h;
    int idx = closest_hit(&c, r, &h);
    if (idx >= 0) {
        if (t) *t = h.t;
        if (normal) st3(normal, h.n);
        if (inside) *inside = h.inside;
    }
    return idx;
}

void orc_camera_ray(const orc_scene *sc, float u, float v, float origin[3], float dir[3])
{
    ray_t r = camera_ray(sc, u, v);
    st3(origin, r.o); st3(dir, r.d);
}

int orc_figure_intersect(uint32_t kind, const float param[3], const float origin[3], const float dir[3],
                         float *t, float normal[3], int *inside)
{
    ray_t r; r.o = ld3(origin); r.d = ld3(dir);
    hit_t h;
    if (!figure_intersect(kind, ld3(param), r, &h)) return 0;
    if (t) *t = h.t;
    if (normal) st3(normal, h.n);
    if (inside) *inside = h.inside;
    return 1;
}

void orc_reflect(const float dir[3], const float point[3], const float normal[3],
                 float out_origin[3], float out_dir[3])
{
    ray_t r = reflected_ray(ld3(dir), ld3(point), ld3(normal));
    st3(out_origin, r.o); st3(out_dir, r.d);
}
int orc_refract(const float dir[3], const float point[3], const float normal[3], float eta,
                float out_origin[3], float out_dir[3])
{
    ray_t r;
    if (!refracted_ray(ld3(dir), ld3(point), ld3(normal), eta, &r)) return 0;
    st3(out_origin, r.o); st3(out_dir, r.d);
    return 1;
}
float orc_schlick(float eta, float cosv) { return schlick(eta, cosv); }
void orc_cosine_sample(const float n[3], uint32_t w_theta, uint32_t w_r, float out[3])
{
    st3(out, cosine_sample(ld3(n), w_theta, w_r));
}
void orc_light_sample(const orc_scene *sc, uint32_t prim_index, const uint32_t w[4], float out_point[3])
{
    st3(out_point, light_surface_sample(&sc->prims[prim_index], w));
}
float orc_mis_pdf(const orc_scene *sc, const float p[3], const float n[3], const float d[3])
{
    ctx_t c;
    orc_opts opts;
    memset(&opts, 0, sizeof(opts));
    if (ctx_init(&c, sc, &opts, NULL)) return NAN;
    float r = mis_pdf(&c, ld3(p), ld3(n), ld3(d));
    free(c.lights);
    return r;
}
float orc_light_pdf(const orc_scene *sc, const float p[3], const float d[3])
{
    ctx_t c;
    orc_opts opts;
    memset(&opts, 0, sizeof(opts));
    if (ctx_init(&c, sc, &opts, NULL)) return NAN;
    float r = light_pdf(&c, ld3(p), ld3(d));
    free(c.lights);
    return r;
}
int orc_lights(const orc_scene *sc, int32_t *indices, int capacity)
{
    return collect_lights(sc, indices, capacity);
}

/* image.rs:60-83 then :51-57; clamp = simd_clamp(0, 1) */
void orc_color_correction(float *rgb, uint64_t n)
{
    const float A = 2.51f, B = 0.03f, C = 2.43f, D = 0.59f, E = 0.14f;
    for (uint64_t i = 0; i < n; ++i) {
        float x = rgb[i];
        float up = x * (A * x + B);
        float down = x * (C * x + D) + E;
        float y = up / down;
        y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
        rgb[i] = powf(y, 1.0f / 2.2f);
    }
}
/* image.rs:38-46: (255.0 * x).round() as u8 -- round half away from zero, saturating cast, NaN -> 0 */
void orc_to_bytes(const float *rgb, uint64_t n, uint8_t *out)
{
    for (uint64_t i = 0; i < n; ++i) {
        float v = roundf(255.0f * rgb[i]);
        out[i] = (v != v) ? 0 : (v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (uint8_t)v));
    }
}
