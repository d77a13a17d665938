// rtx_api.hip -- the C ABI of include/rtx.h: context, scene upload, launches.
//
// Replaces the call `render(&mut scene)` of src/main.rs:38 and the Scene hand-over of
// src/main.rs:37; the lights list is derived here by the rule of src/parser.rs:69-89.
#include "../../include/rtx.h"
#include "rtx_bvh.h"
#include "rtx_device.h"
#include "rtx_wavefront.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? RTX_ERR_NO_MEMORY : RTX_ERR_DEVICE, "%s: %s",  \
                        #expr, hipGetErrorString(e_));                                             \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (dev != prev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct rtx_ctx {
    rtx_opts opts{};
    int device = 0;
    // host copy of the scene
    bool have_scene = false;
    rtx_scene scene{};
    std::vector<rtx_prim> prims;
    std::vector<int32_t> lights;
    int src = rtx::kSrcLds;
    // device scene block
    void *d_scene = nullptr;
    size_t d_scene_bytes = 0;
    rtx::RenderParams params{};
    // scratch owned by the context
    float *d_image = nullptr;
    size_t d_image_floats = 0;
    uint8_t *d_bytes = nullptr;   // tone-mapped image
    size_t d_bytes_size = 0;
    // progressive rendering: d_image holds the sum of samples [0, progress_done)
    bool progress_active = false;
    uint32_t progress_done = 0;
    void *d_wavefront = nullptr;  // path state of the wavefront variant
    size_t d_wavefront_bytes = 0;
    uint32_t *d_tiles = nullptr;  // cost and dispatch order of the 8x8 tiles, n_tiles words each
    size_t d_tiles_words = 0;
    unsigned long long *d_stats = nullptr;
    rtx_stats stats{};
    bool stats_valid = false;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
};

namespace {

void free_scene(rtx_ctx *c)
{
    if (c->d_scene) (void)hipFree(c->d_scene);
    c->d_scene = nullptr;
    c->d_scene_bytes = 0;
    c->have_scene = false;
}

size_t align16(size_t x) { return (x + 15u) & ~(size_t)15u; }

// src/parser.rs:69-89
void derive_lights(const std::vector<rtx_prim> &prims, std::vector<int32_t> &out)
{
    out.clear();
    for (size_t i = 0; i < prims.size(); ++i) {
        const float *e = prims[i].emission;
        float l2 = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
        if (l2 == 0.0f) continue;
        if (prims[i].kind == RTX_PLANE) continue;
        out.push_back((int32_t)i);
    }
}

// rotation exactly (+0, +0, +0, 1) and a finite, moderate position: the device may skip
// the two rotations of such an object for plain rays (rtx_trace.hpp: ray_is_plain)
bool is_identity_pose(const rtx_prim &p)
{
    uint32_t q[4];
    std::memcpy(q, p.rotation, sizeof(q));
    if (q[0] != 0u || q[1] != 0u || q[2] != 0u || q[3] != 0x3f800000u) return false;
    for (int k = 0; k < 3; ++k)
        if (!(std::fabs(p.position[k]) < 1e30f)) return false;
    return true;
}

// every component of the parameter vector is 2^e with |e| <= 60: dividing by it and multiplying
// by its (exact) inverse round the same real number (rtx_trace.hpp: div_param)
bool is_pow2_param(const rtx_prim &p)
{
    if (p.kind == RTX_PLANE) return false;
    for (int k = 0; k < 3; ++k) {
        uint32_t b;
        std::memcpy(&b, &p.param[k], 4);
        const uint32_t e = b >> 23;  // sign bit included: negative values fail the range test
        if ((b & 0x7fffffu) != 0u || e < 127u - 60u || e > 127u + 60u) return false;
    }
    return true;
}

// Every number finite and moderate, no degenerate object: the condition under which a path
// with zero throughput may be ended (rtx_trace.hpp: bounce).  Magnitude limits keep every
// intermediate of the trace loop finite, so the zero cannot meet an inf or a NaN later on.
bool tame(float x, float lo, float hi)
{
    const float a = std::fabs(x);
    return a >= lo && a <= hi;  // false for NaN
}
bool scene_is_tame(const rtx_scene &s, const std::vector<rtx_prim> &prims)
{
    for (int k = 0; k < 3; ++k)
        if (!tame(s.background[k], 0.0f, 1e6f)) return false;
    for (const rtx_prim &p : prims) {
        float q2 = 0.0f;
        bool nonzero = false;
        for (int k = 0; k < 3; ++k) {
            if (!tame(p.emission[k], 0.0f, 1e6f) || !tame(p.color[k], 0.0f, 1e3f)) return false;
            if (!tame(p.position[k], 0.0f, 1e6f)) return false;
            if (p.kind == RTX_PLANE) {
                if (!tame(p.param[k], 0.0f, 1e6f)) return false;
                nonzero = nonzero || std::fabs(p.param[k]) >= 1e-6f;
            } else {
                if (!tame(p.param[k], 1e-4f, 1e6f)) return false;
                nonzero = true;
            }
        }
        if (!nonzero) return false;
        for (int k = 0; k < 4; ++k) {
            if (!tame(p.rotation[k], 0.0f, 2.0f)) return false;
            q2 += p.rotation[k] * p.rotation[k];
        }
        if (!(q2 >= 0.5f && q2 <= 2.0f)) return false;
        if (p.material == RTX_DIELECTRIC && !tame(p.ior, 1e-3f, 1e3f)) return false;
    }
    return true;
}

int validate_scene(const rtx_scene *s)
{
    if (!s) return fail(RTX_ERR_INVALID_ARG, "scene is null");
    if (s->width == 0 || s->height == 0) return fail(RTX_ERR_INVALID_SCENE, "image dimensions must be positive");
    if ((uint64_t)s->width * s->height > 0x7fffffffull / 3u)
        return fail(RTX_ERR_INVALID_SCENE, "image too large: %u x %u", s->width, s->height);
    if (s->n_prims > 0 && !s->prims) return fail(RTX_ERR_INVALID_ARG, "prims is null");
    if (s->n_prims > (1u << 24)) return fail(RTX_ERR_INVALID_SCENE, "too many objects: %u", s->n_prims);
    for (uint32_t i = 0; i < s->n_prims; ++i) {
        const rtx_prim &p = s->prims[i];
        if (p.kind > RTX_BOX) return fail(RTX_ERR_INVALID_SCENE, "object %u: unknown kind %u", i, p.kind);
        if (p.material > RTX_DIELECTRIC)
            return fail(RTX_ERR_INVALID_SCENE, "object %u: unknown material %u", i, p.material);
        const float *e = p.emission;
        bool emissive = ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) != 0.0f;
        if (emissive && p.kind == RTX_BOX) {
            // sampling this light draws from [-s, s) per axis and from [0, area): the
            // reference panics on an empty range
            for (int k = 0; k < 3; ++k)
                if (!(p.param[k] > 0.0f))
                    return fail(RTX_ERR_INVALID_SCENE, "object %u: emissive box needs positive half-sizes", i);
        }
    }
    return RTX_OK;
}

int upload_scene(rtx_ctx *c)
{
    const size_t n = c->prims.size();
    const size_t nl = c->lights.size();
    const size_t geom_bytes = align16(n * 3 * sizeof(float4));
    const size_t mat_bytes = align16(n * 3 * sizeof(float4));
    const size_t light_bytes = align16(nl * sizeof(int32_t));
    const size_t lbound_bytes = align16(nl * 2 * sizeof(float4));

    rtx::BvhHost bvh;
    const bool want_bvh = c->src == rtx::kSrcBvh;
    if (want_bvh) rtx::build_bvh(c->prims.data(), (uint32_t)n, bvh);
    const size_t node_bytes = align16(bvh.nodes.size() * sizeof(uint4));
    const size_t plane_bytes = align16(bvh.planes.size() * sizeof(int32_t));

    const size_t total = geom_bytes + mat_bytes + light_bytes + lbound_bytes + node_bytes + plane_bytes + 16;
    std::vector<unsigned char> host(total, 0);
    float4 *geom = reinterpret_cast<float4 *>(host.data());
    float4 *mat = reinterpret_cast<float4 *>(host.data() + geom_bytes);
    int32_t *lights = reinterpret_cast<int32_t *>(host.data() + geom_bytes + mat_bytes);
    for (size_t i = 0; i < n; ++i) {
        const rtx_prim &p = c->prims[i];
        int32_t kind = (int32_t)p.kind | (is_identity_pose(p) ? (rtx::kGeomIdentity << 8) : 0) |
                       (is_pow2_param(p) ? (rtx::kGeomPow2 << 8) : 0);
        int32_t material = (int32_t)p.material;
        float kind_bits, mat_bits;
        std::memcpy(&kind_bits, &kind, 4);
        std::memcpy(&mat_bits, &material, 4);
        geom[3 * i + 0] = make_float4(p.rotation[0], p.rotation[1], p.rotation[2], p.rotation[3]);
        geom[3 * i + 1] = make_float4(p.position[0], p.position[1], p.position[2], kind_bits);
        geom[3 * i + 2] = make_float4(p.param[0], p.param[1], p.param[2], 0.0f);
        mat[3 * i + 0] = make_float4(p.color[0], p.color[1], p.color[2], mat_bits);
        mat[3 * i + 1] = make_float4(p.emission[0], p.emission[1], p.emission[2], p.ior);
        // src/trace.rs:27 (color / PI) and src/objects/sample.rs:53-57 (box surface pdf) depend on
        // the object only: evaluated here once, with the same IEEE operations
        const float pi = 3.14159274101257324f;
        float area_pdf = 0.0f;
        if (p.kind == RTX_BOX) {
            float a = p.param[0], b = p.param[1], cc = p.param[2];
            float area = 8.0f * ((a * b + b * cc) + a * cc);
            area_pdf = 1.0f / area;
        }
        mat[3 * i + 2] = make_float4(p.color[0] / pi, p.color[1] / pi, p.color[2] / pi, area_pdf);
    }
    for (size_t i = 0; i < nl; ++i) lights[i] = c->lights[i];
    // Per light, what the many-lights pdf needs to skip lights a ray certainly "misses":
    //   [2k]   bounding sphere: centre = position, squared radius (farthest surface point, +2 %)
    //   [2k+1] x = slack: the sphere is taken larger by slack * |centre - origin|^2
    // "Misses" means: the f32 intersection formula the pdf evaluates reports no hit -- and far
    // from the light that formula reports hits for rays that pass well outside it.  Ellipsoid
    // (src/objects/geometry.rs:75-91): det = b^2 - a c is off by up to 8.5e-7 |u|^2 |v|^2 (three
    // roundings in each of a, b, c, one in each product), while its exact value is
    // |v|^2 (1 - q^2), q = distance of the line from the centre in the scaled space where the
    // ellipsoid is the unit sphere; so a hit can be reported while q^2 < 1 + 8.5e-7 |u|^2, and
    // |u| <= |centre - origin| / r_min: in world units the "hit zone" is the sphere of squared
    // radius r_max^2 + 8.5e-7 (r_max / r_min)^2 |centre - origin|^2.  Box (geometry.rs:114-130):
    // the slab distances are off by a few 1e-7 relative, the zone grows by about
    // 2e-6 |centre - origin|.  The slacks below are those bounds with a factor 2 to spare.
    const size_t off_lbound = geom_bytes + mat_bytes + light_bytes;
    float4 *lbound = reinterpret_cast<float4 *>(host.data() + off_lbound);
    for (size_t i = 0; i < nl; ++i) {
        const rtx_prim &p = c->prims[(size_t)c->lights[i]];
        double r = 0.0, rmin = INFINITY;
        for (int k = 0; k < 3; ++k) rmin = std::fmin(rmin, std::fabs((double)p.param[k]));
        // |A|, |A^-1| of the world -> local map: both 1 for a unit quaternion (rtx_bvh.h)
        double ainv[3][3], na = INFINITY, nai = INFINITY;
        if (!rtx::pose_matrices(p, ainv, na, nai)) na = nai = INFINITY;   // never skipped
        double slack;
        if (p.kind == RTX_BOX) {
            for (int k = 0; k < 3; ++k) r += (double)p.param[k] * p.param[k];
            r = std::sqrt(r);
            slack = 1e-5 * (na * nai) * (na * nai);
        } else {
            for (int k = 0; k < 3; ++k) r = std::fmax(r, std::fabs((double)p.param[k]));
            const double kappa = r / rmin * na * nai;  // inf for a degenerate ellipsoid: never skipped
            slack = 2e-6 * kappa * kappa + 2e-6;
        }
        r *= 1.02 * nai;
        double r2 = r * r + 1e-6;
        float r2f = (float)r2, slackf = (float)slack;
        if (!(r2f >= 0.0f) || !std::isfinite(r2f)) r2f = INFINITY;  // never skipped
        if (!(slackf >= 0.0f)) slackf = INFINITY;
        lbound[2 * i + 0] = make_float4(p.position[0], p.position[1], p.position[2], r2f);
        lbound[2 * i + 1] = make_float4(slackf, 0.0f, 0.0f, 0.0f);
    }
    size_t off_nodes = off_lbound + lbound_bytes;
    size_t off_planes = off_nodes + node_bytes;
    if (!bvh.nodes.empty()) std::memcpy(host.data() + off_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(uint4));
    if (!bvh.planes.empty())
        std::memcpy(host.data() + off_planes, bvh.planes.data(), bvh.planes.size() * sizeof(int32_t));

    if (c->d_scene_bytes < total) {
        if (c->d_scene) (void)hipFree(c->d_scene);
        c->d_scene = nullptr;
        c->d_scene_bytes = 0;
        HIP_TRY(hipMalloc(&c->d_scene, total));
        c->d_scene_bytes = total;
    }
    HIP_TRY(hipMemcpy(c->d_scene, host.data(), total, hipMemcpyHostToDevice));

    unsigned char *base = static_cast<unsigned char *>(c->d_scene);
    rtx::RenderParams &P = c->params;
    std::memset(&P, 0, sizeof(P));
    P.geom = reinterpret_cast<const float4 *>(base);
    P.mat = reinterpret_cast<const float4 *>(base + geom_bytes);
    P.lights = reinterpret_cast<const int32_t *>(base + geom_bytes + mat_bytes);
    P.light_bounds = reinterpret_cast<const float4 *>(base + off_lbound);
    P.n_prims = (uint32_t)n;
    P.n_lights = (uint32_t)nl;
    const rtx_scene &s = c->scene;
    P.width = s.width;
    P.height = s.height;
    P.ray_depth = s.ray_depth;
    std::memcpy(P.bg, s.background, sizeof(P.bg));
    std::memcpy(P.cam_pos, s.cam_pos, sizeof(P.cam_pos));
    std::memcpy(P.cam_axis, s.cam_axis, sizeof(P.cam_axis));
    P.tg_fov_x = s.tg_fov_x;
    P.tg_fov_y = s.tg_fov_y;
    P.seed_lo = (uint32_t)c->opts.seed;
    P.seed_hi = (uint32_t)(c->opts.seed >> 32);
    P.rng_const = c->opts.rng_mode == RTX_RNG_CONST ? 1u : 0u;
    P.stop_on_zero = scene_is_tame(c->scene, c->prims) ? 1u : 0u;
    std::memcpy(P.const_words, c->opts.const_words, sizeof(P.const_words));
    P.bvh.nodes = reinterpret_cast<const uint4 *>(base + off_nodes);
    std::memcpy(P.bvh.grid_base, bvh.grid_base, sizeof(P.bvh.grid_base));
    std::memcpy(P.bvh.grid_inv_step, bvh.grid_inv_step, sizeof(P.bvh.grid_inv_step));
    std::memcpy(P.bvh.grid_step, bvh.grid_step, sizeof(P.bvh.grid_step));
    P.bvh.far_slope = bvh.far_slope;
    P.bvh.planes = reinterpret_cast<const int32_t *>(base + off_planes);
    P.bvh.n_nodes = (uint32_t)(bvh.nodes.size() / 2);
    P.bvh.n_planes = (uint32_t)bvh.planes.size();
    P.bvh.stack_entries = (uint32_t)std::min(bvh.depth + 1, rtx::kBvhStackSize);
    {
        // objects outside the always-tested list are the leaves
        std::vector<char> listed(n, 0);
        for (int32_t i : bvh.planes) listed[(size_t)i] = 1;
        bool all = true;
        for (size_t i = 0; i < n; ++i) all = all && (listed[i] || is_identity_pose(c->prims[i]));
        P.bvh.leaves_identity = all ? 1u : 0u;
    }
    return RTX_OK;
}

int ensure_image(rtx_ctx *c, size_t floats)
{
    if (c->d_image_floats >= floats) return RTX_OK;
    if (c->d_image) (void)hipFree(c->d_image);
    c->d_image = nullptr;
    c->d_image_floats = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_image), floats * sizeof(float)));
    c->d_image_floats = floats;
    return RTX_OK;
}

int ensure_bytes(rtx_ctx *c, size_t n)
{
    if (c->d_bytes_size >= n) return RTX_OK;
    if (c->d_bytes) (void)hipFree(c->d_bytes);
    c->d_bytes = nullptr;
    c->d_bytes_size = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_bytes), n));
    c->d_bytes_size = n;
    return RTX_OK;
}

void stats_from_counters(const rtx_ctx *c, const unsigned long long *v, rtx_stats *s)
{
    std::memset(s, 0, sizeof(*s));
    s->samples = v[rtx::C_SAMPLES];
    s->rays = v[rtx::C_RAYS];
    s->hits[0] = v[rtx::C_HIT_PLANE];
    s->hits[1] = v[rtx::C_HIT_ELLIPSOID];
    s->hits[2] = v[rtx::C_HIT_BOX];
    s->cosine_samples = v[rtx::C_COSINE];
    s->light_samples[1] = v[rtx::C_LSAMPLE_ELLIPSOID];
    s->light_samples[2] = v[rtx::C_LSAMPLE_BOX];
    s->pdf_evals = v[rtx::C_PDF_EVALS];
    s->light_pdf[0] = v[rtx::C_LPDF_0];
    s->light_pdf[1] = v[rtx::C_LPDF_1];
    s->light_pdf[2] = v[rtx::C_LPDF_2];
    s->diffuse_bounces = v[rtx::C_DIFFUSE];
    s->metallic_bounces = v[rtx::C_METALLIC];
    s->dielectric_bounces = v[rtx::C_DIELECTRIC];
    s->light_tests[1] = v[rtx::C_LTEST_ELLIPSOID];
    s->light_tests[2] = v[rtx::C_LTEST_BOX];
    if (c->src != rtx::kSrcBvh) {
        // a brute-force closest-hit query tests every object once
        uint64_t per_kind[3] = {0, 0, 0};
        for (const rtx_prim &p : c->prims) per_kind[p.kind]++;
        for (int k = 0; k < 3; ++k) s->prim_tests[k] = s->rays * per_kind[k];
    } else {
        s->prim_tests[0] = v[rtx::C_PTEST_PLANE];
        s->prim_tests[1] = v[rtx::C_PTEST_ELLIPSOID];
        s->prim_tests[2] = v[rtx::C_PTEST_BOX];
        s->bvh_node_tests = v[rtx::C_NODE_TESTS];
    }
}

}  // namespace

extern "C" {

uint32_t rtx_abi_version(void) { return RTX_ABI_VERSION; }

const char *rtx_last_error(void) { return g_last_error.c_str(); }

int rtx_create(rtx_ctx **out, const rtx_opts *opts)
{
    if (!out || !opts) return fail(RTX_ERR_INVALID_ARG, "rtx_create: null argument");
    *out = nullptr;
    if (opts->mode > RTX_MODE_WAVEFRONT) return fail(RTX_ERR_INVALID_ARG, "unknown mode %u", opts->mode);
    if (opts->accel > RTX_ACCEL_BVH) return fail(RTX_ERR_INVALID_ARG, "unknown accel %u", opts->accel);
    if (opts->rng_mode > RTX_RNG_CONST) return fail(RTX_ERR_INVALID_ARG, "unknown rng_mode %u", opts->rng_mode);
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) return fail(RTX_ERR_DEVICE, "no HIP device visible");
    int dev = opts->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= n_dev) return fail(RTX_ERR_INVALID_ARG, "device %d out of range (%d visible)", dev, n_dev);
    rtx_ctx *c = new (std::nothrow) rtx_ctx();
    if (!c) return fail(RTX_ERR_NO_MEMORY, "out of host memory");
    c->opts = *opts;
    c->device = dev;
    DeviceGuard g(dev);
    if (!g.ok) { delete c; return fail(RTX_ERR_DEVICE, "cannot select device %d", dev); }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_stats), rtx::C_COUNT * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipEventCreate(&c->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_stop);
    if (e != hipSuccess) {
        rtx_destroy(c);
        return fail(RTX_ERR_DEVICE, "context set-up: %s", hipGetErrorString(e));
    }
    *out = c;
    return RTX_OK;
}

void rtx_destroy(rtx_ctx *c)
{
    if (!c) return;
    DeviceGuard g(c->device);
    free_scene(c);
    if (c->d_image) (void)hipFree(c->d_image);
    if (c->d_bytes) (void)hipFree(c->d_bytes);
    if (c->d_wavefront) (void)hipFree(c->d_wavefront);
    if (c->d_tiles) (void)hipFree(c->d_tiles);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    delete c;
}

int rtx_set_scene(rtx_ctx *c, const rtx_scene *s)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    int rc = validate_scene(s);
    if (rc != RTX_OK) return rc;
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", c->device);
    c->have_scene = false;
    c->scene = *s;
    c->prims.assign(s->prims, s->prims + s->n_prims);
    c->scene.prims = nullptr;
    derive_lights(c->prims, c->lights);
    switch (c->opts.accel) {
    case RTX_ACCEL_AUTO: {
        // planes are tested for every ray either way; with more than RTX_AUTO_BRUTE_MAX other
        // objects the traversal is faster than the scan (measured cross-over: 6 to 10)
        uint32_t bounded = 0;
        for (const rtx_prim &p : c->prims) bounded += p.kind != RTX_PLANE ? 1u : 0u;
        c->src = (bounded <= RTX_AUTO_BRUTE_MAX && s->n_prims <= RTX_MAX_LDS_PRIMS) ? rtx::kSrcLds : rtx::kSrcBvh;
        break;
    }
    case RTX_ACCEL_BRUTE_LDS:
        if (s->n_prims > RTX_MAX_LDS_PRIMS)
            return fail(RTX_ERR_UNSUPPORTED, "%u objects exceed the LDS table limit of %u", s->n_prims,
                        RTX_MAX_LDS_PRIMS);
        c->src = rtx::kSrcLds;
        break;
    case RTX_ACCEL_BRUTE_SCALAR: c->src = rtx::kSrcScalar; break;
    default: c->src = rtx::kSrcBvh; break;
    }
    rc = upload_scene(c);
    if (rc != RTX_OK) return rc;
    c->have_scene = true;
    return RTX_OK;
}

// cost-ordered dispatch of the megakernel's tiles: used from this many samples and tiles on
// (fewer tiles than two per wave slot of the GPU leave nothing to reorder)
constexpr uint32_t kProbeSamples = 8u, kOrderedMinSamples = 64u, kOrderedMinTiles = 8192u;

static int render_range(rtx_ctx *c, uint32_t sample_first, uint32_t sample_count, float *d_sum, void *hip_stream,
                        bool accumulate)
{
    if (!c || !d_sum) return fail(RTX_ERR_INVALID_ARG, "render: null argument");
    if (!c->have_scene) return fail(RTX_ERR_NO_SCENE, "rtx_set_scene has not been called");
    if ((uint64_t)sample_first + sample_count > 0xffffffffull)
        return fail(RTX_ERR_INVALID_ARG, "sample range overflows 32 bits");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", c->device);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    rtx::RenderParams P = c->params;
    P.sample_first = sample_first;
    P.sample_count = sample_count;
    P.out_sum = d_sum;
    P.accumulate = accumulate ? 1u : 0u;
    const bool stats = c->opts.collect_stats != 0;
    P.stats = stats ? c->d_stats : nullptr;
    c->stats_valid = false;
    if (stats) HIP_TRY(hipMemsetAsync(c->d_stats, 0, rtx::C_COUNT * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(c->ev_start, stream));
    if (c->opts.mode == RTX_MODE_WAVEFRONT) {
        rtx::WfState S;
        S.rec = nullptr;
        if (!rtx::wavefront_state_in_lds(P, c->src)) {
            const size_t need = rtx::wavefront_state_bytes(P.width, P.height);
            if (c->d_wavefront_bytes < need) {
                if (c->d_wavefront) (void)hipFree(c->d_wavefront);
                c->d_wavefront = nullptr;
                c->d_wavefront_bytes = 0;
                HIP_TRY(hipMalloc(&c->d_wavefront, need));
                c->d_wavefront_bytes = need;
            }
            rtx::wavefront_bind_state(c->d_wavefront, P.width, P.height, S);
        }
        HIP_TRY(rtx::launch_wavefront(P, S, c->src, stats, stream));
    } else {
        // Waves run one tile each and the launch ends when the last of them does; with tiles
        // handed out in image order the last ones started can be the long ones.  A first launch
        // of kProbeSamples samples records what each tile costs, the rest of the samples are
        // continued on top of its sums with the tiles dispatched longest first.  The split
        // changes no bits (the sums continue in sample order, as rtx_accumulate_samples does).
        const uint32_t n_tiles = ((P.width + 7u) / 8u) * ((P.height + 7u) / 8u);
        uint32_t probe = 0;
        if (!stats && sample_count >= kOrderedMinSamples && n_tiles >= kOrderedMinTiles) {
            probe = kProbeSamples;
            if (c->d_tiles_words < 2u * (size_t)n_tiles) {
                if (c->d_tiles) (void)hipFree(c->d_tiles);
                c->d_tiles = nullptr;
                c->d_tiles_words = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_tiles), 2u * (size_t)n_tiles * sizeof(uint32_t)));
                c->d_tiles_words = 2u * (size_t)n_tiles;
            }
        }
        if (probe) {
            uint32_t *cost = c->d_tiles, *order = c->d_tiles + n_tiles;
            HIP_TRY(hipMemsetAsync(cost, 0, n_tiles * sizeof(uint32_t), stream));
            rtx::RenderParams A = P;
            A.sample_count = probe;
            A.tile_cost = cost;
            HIP_TRY(rtx::launch_megakernel(A, c->src, false, stream));
            HIP_TRY(rtx::launch_tile_order(cost, order, n_tiles, stream));
            P.sample_first = sample_first + probe;
            P.sample_count = sample_count - probe;
            P.accumulate = 1u;
            P.tile_order = order;
        }
        HIP_TRY(rtx::launch_megakernel(P, c->src, stats, stream));
    }
    HIP_TRY(hipEventRecord(c->ev_stop, stream));
    c->timed = true;
    if (stats) {
        unsigned long long v[rtx::C_COUNT];
        HIP_TRY(hipMemcpyAsync(v, c->d_stats, sizeof(v), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        stats_from_counters(c, v, &c->stats);
        c->stats_valid = true;
    }
    return RTX_OK;
}

int rtx_render_samples(rtx_ctx *c, uint32_t sample_first, uint32_t sample_count, float *d_sum, void *hip_stream)
{
    return render_range(c, sample_first, sample_count, d_sum, hip_stream, false);
}

int rtx_accumulate_samples(rtx_ctx *c, uint32_t sample_first, uint32_t sample_count, float *d_sum, void *hip_stream)
{
    return render_range(c, sample_first, sample_count, d_sum, hip_stream, true);
}

int rtx_tonemap_bytes(rtx_ctx *c, const float *d_sum, uint64_t n_floats, float divisor, uint8_t *d_bytes,
                      void *hip_stream)
{
    if (!c || !d_sum || !d_bytes) return fail(RTX_ERR_INVALID_ARG, "rtx_tonemap_bytes: null argument");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", c->device);
    HIP_TRY(rtx::launch_tonemap(d_sum, n_floats, divisor, d_bytes, static_cast<hipStream_t>(hip_stream)));
    return RTX_OK;
}

int rtx_divide(rtx_ctx *c, float *d_rgb, uint64_t n_floats, float divisor, void *hip_stream)
{
    if (!c || !d_rgb) return fail(RTX_ERR_INVALID_ARG, "rtx_divide: null argument");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", c->device);
    HIP_TRY(rtx::launch_divide(d_rgb, n_floats, divisor, static_cast<hipStream_t>(hip_stream)));
    return RTX_OK;
}

int rtx_render(rtx_ctx *c, const rtx_scene *s, float *rgb)
{
    if (!c || !rgb) return fail(RTX_ERR_INVALID_ARG, "rtx_render: null argument");
    int rc = rtx_set_scene(c, s);
    if (rc != RTX_OK) return rc;
    DeviceGuard g(c->device);
    const size_t floats = (size_t)s->width * s->height * 3u;
    rc = ensure_image(c, floats);
    if (rc != RTX_OK) return rc;
    c->progress_active = false;
    if (s->n_samples == 0) {
        // src/main.rs:14: zero iterations leave the zero-initialised image (src/image.rs:17)
        std::memset(rgb, 0, floats * sizeof(float));
        return RTX_OK;
    }
    rc = rtx_render_samples(c, 0, s->n_samples, c->d_image, nullptr);
    if (rc != RTX_OK) return rc;
    rc = rtx_divide(c, c->d_image, floats, (float)s->n_samples, nullptr);
    if (rc != RTX_OK) return rc;
    HIP_TRY(hipMemcpy(rgb, c->d_image, floats * sizeof(float), hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_render_bytes(rtx_ctx *c, const rtx_scene *s, uint8_t *bytes)
{
    if (!c || !bytes) return fail(RTX_ERR_INVALID_ARG, "rtx_render_bytes: null argument");
    int rc = rtx_set_scene(c, s);
    if (rc != RTX_OK) return rc;
    DeviceGuard g(c->device);
    const size_t floats = (size_t)s->width * s->height * 3u;
    rc = ensure_image(c, floats);
    if (rc == RTX_OK) rc = ensure_bytes(c, floats);
    if (rc != RTX_OK) return rc;
    c->progress_active = false;
    // zero samples leave the zero image (src/main.rs:14, src/image.rs:17); 0 / 1 keeps it
    rc = rtx_render_samples(c, 0, s->n_samples, c->d_image, nullptr);
    if (rc != RTX_OK) return rc;
    const float divisor = s->n_samples == 0 ? 1.0f : (float)s->n_samples;
    rc = rtx_tonemap_bytes(c, c->d_image, floats, divisor, c->d_bytes, nullptr);
    if (rc != RTX_OK) return rc;
    HIP_TRY(hipMemcpy(bytes, c->d_bytes, floats, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_progress_begin(rtx_ctx *c, const rtx_scene *s, const float *sum, uint32_t samples_done)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    c->progress_active = false;
    if (!sum && samples_done != 0)
        return fail(RTX_ERR_INVALID_ARG, "rtx_progress_begin: %u samples done but no sum to resume from", samples_done);
    int rc = rtx_set_scene(c, s);
    if (rc != RTX_OK) return rc;
    DeviceGuard g(c->device);
    const size_t floats = (size_t)s->width * s->height * 3u;
    rc = ensure_image(c, floats);
    if (rc != RTX_OK) return rc;
    if (sum) HIP_TRY(hipMemcpy(c->d_image, sum, floats * sizeof(float), hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(c->d_image, 0, floats * sizeof(float)));
    c->progress_done = samples_done;
    c->progress_active = true;
    return RTX_OK;
}

int rtx_progress_step(rtx_ctx *c, uint32_t sample_count)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    if (!c->progress_active) return fail(RTX_ERR_NO_SCENE, "rtx_progress_begin has not been called");
    if ((uint64_t)c->progress_done + sample_count > 0xffffffffull)
        return fail(RTX_ERR_INVALID_ARG, "sample range overflows 32 bits");
    DeviceGuard g(c->device);
    int rc = rtx_accumulate_samples(c, c->progress_done, sample_count, c->d_image, nullptr);
    if (rc != RTX_OK) { c->progress_active = false; return rc; }
    HIP_TRY(hipStreamSynchronize(nullptr));
    c->progress_done += sample_count;
    return RTX_OK;
}

int rtx_progress_read(rtx_ctx *c, float *rgb_mean, float *sum, uint8_t *bytes, uint32_t *samples_done)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    if (!c->progress_active) return fail(RTX_ERR_NO_SCENE, "rtx_progress_begin has not been called");
    DeviceGuard g(c->device);
    const size_t floats = (size_t)c->scene.width * c->scene.height * 3u;
    if (samples_done) *samples_done = c->progress_done;
    if (sum) HIP_TRY(hipMemcpy(sum, c->d_image, floats * sizeof(float), hipMemcpyDeviceToHost));
    // the mean of zero samples is the zero image the reference starts from (src/image.rs:17)
    const float divisor = c->progress_done == 0 ? 1.0f : (float)c->progress_done;
    if (bytes) {
        int rc = ensure_bytes(c, floats);
        if (rc != RTX_OK) return rc;
        rc = rtx_tonemap_bytes(c, c->d_image, floats, divisor, c->d_bytes, nullptr);
        if (rc != RTX_OK) return rc;
        HIP_TRY(hipMemcpy(bytes, c->d_bytes, floats, hipMemcpyDeviceToHost));
    }
    if (rgb_mean) {
        // the sum stays on the device untouched: divide on the host copy, the same IEEE division
        HIP_TRY(hipMemcpy(rgb_mean, c->d_image, floats * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < floats; ++i) rgb_mean[i] = rgb_mean[i] / divisor;
    }
    return RTX_OK;
}

int rtx_primary(rtx_ctx *c, const rtx_scene *s, int32_t *hit_id, float *t, float *normal, uint8_t *inside)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    int rc = rtx_set_scene(c, s);
    if (rc != RTX_OK) return rc;
    DeviceGuard g(c->device);
    const size_t n = (size_t)s->width * s->height;
    // one device block: ids | t | normal | inside
    const size_t bytes = n * 4 + n * 4 + n * 12 + align16(n);
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    unsigned char *b = static_cast<unsigned char *>(d);
    rtx::PrimaryParams out;
    out.hit_id = reinterpret_cast<int32_t *>(b);
    out.t = reinterpret_cast<float *>(b + n * 4);
    out.normal = reinterpret_cast<float *>(b + n * 8);
    out.inside = reinterpret_cast<uint8_t *>(b + n * 20);
    hipError_t e = rtx::launch_primary(c->params, out, c->src, nullptr);
    if (e == hipSuccess && hit_id) e = hipMemcpy(hit_id, out.hit_id, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && t) e = hipMemcpy(t, out.t, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && normal) e = hipMemcpy(normal, out.normal, n * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && inside) e = hipMemcpy(inside, out.inside, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RTX_ERR_DEVICE, "primary pass: %s", hipGetErrorString(e));
    return RTX_OK;
}

int rtx_get_stats(rtx_ctx *c, rtx_stats *out)
{
    if (!c || !out) return fail(RTX_ERR_INVALID_ARG, "rtx_get_stats: null argument");
    if (!c->stats_valid) return fail(RTX_ERR_UNSUPPORTED, "no statistics: create the context with collect_stats = 1");
    *out = c->stats;
    return RTX_OK;
}

int rtx_last_kernel_ms(rtx_ctx *c, float *ms)
{
    if (!c || !ms) return fail(RTX_ERR_INVALID_ARG, "rtx_last_kernel_ms: null argument");
    if (!c->timed) return fail(RTX_ERR_NO_SCENE, "nothing has been rendered yet");
    DeviceGuard g(c->device);
    HIP_TRY(hipEventSynchronize(c->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return RTX_OK;
}

int rtx_get_lights(rtx_ctx *c, int32_t *indices, int32_t capacity)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    if (!c->have_scene) return fail(RTX_ERR_NO_SCENE, "rtx_set_scene has not been called");
    int32_t n = (int32_t)c->lights.size();
    for (int32_t i = 0; i < n && i < capacity && indices; ++i) indices[i] = c->lights[i];
    return n;
}

}  // extern "C"
