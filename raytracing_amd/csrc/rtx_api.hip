// rtx_api.hip -- the C ABI of include/rtx.h: context, scene upload, launches, multi-GPU split.
//
// Replaces the call `render(&mut scene)` of src/main.rs:38 and the Scene hand-over of
// src/main.rs:37; the lights list is derived here by the rule of src/parser.rs:69-89.
#include "../../include/rtx.h"
#include "rtx_bvh.h"
#include "rtx_device.h"
#include "rtx_wavefront.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

// Tile cost and dispatch order tables for rtx_render_samples_static, the entry point that may not
// allocate: one copy per GPU, held until the process ends.  Host code never uses this name as an
// address (it is a host-side stand-in): rtx_create asks the runtime for the device address, once
// per context with the context's device current, and everything else goes through that pointer.
static __device__ __align__(256) uint32_t scratch[3u * RTX_STATIC_MAX_TILES + 64u];

namespace {

// fixed storage: reporting an error must not allocate (it may be reporting an allocation failure)
thread_local char g_last_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? RTX_ERR_NO_MEMORY : RTX_ERR_DEVICE, "%s: %s",  \
                        #expr, hipGetErrorString(e_));                                             \
    } while (0)

// No exception may cross the C boundary: every entry point that can allocate host memory runs its
// body through this.
template <class F>
int guarded(const char *what, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(RTX_ERR_NO_MEMORY, "%s: out of host memory", what);
    } catch (...) {
        return fail(RTX_ERR_DEVICE, "%s: unexpected exception", what);
    }
}

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

// ------------------------------------------------------------------ RCCL, loaded on demand
// Only a context created with RTX_REDUCE_RCCL touches librccl.so; the symbols are resolved at run
// time so that librtx.so itself depends on the HIP runtime alone (a process that already holds an
// RCCL -- PyTorch brings its own -- keeps using that one: dlopen finds it by its soname).
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclFloat32 = 7, kNcclSum = 0;   // ncclDataType_t / ncclRedOp_t values of nccl.h
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return RTX_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(RTX_ERR_UNSUPPORTED, "RTX_REDUCE_RCCL: cannot load librccl.so (%s)", dlerror());
    Rccl r;
    r.handle = h;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
        return fail(RTX_ERR_UNSUPPORTED, "RTX_REDUCE_RCCL: librccl.so lacks an expected symbol");
    g_rccl = r;
    return RTX_OK;
}
#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        int e_ = (expr);                                                                           \
        if (e_ != 0) return fail(RTX_ERR_DEVICE, "%s: %s", #expr, g_rccl.GetErrorString(e_));      \
    } while (0)

// ------------------------------------------------------------------ multi-GPU slice reduce
// GPU g of G owns floats [first, first + count) of the image: it adds that slice of every GPU's
// sum buffer in device order (so the result does not depend on which GPU does the adding), reading
// the other GPUs' buffers over peer access, and optionally divides by the sample count.
struct ReduceSources {
    const float *src[RTX_MAX_DEVICES];
    int n;
};
__global__ void __launch_bounds__(256) k_reduce_slices(ReduceSources S, uint64_t first, uint64_t count, float divisor,
                                                       int divide, float *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    // `first` is a multiple of 4 floats and the buffers are 256-byte aligned: 16-byte accesses
    const uint64_t n4 = count / 4u;
    for (uint64_t i = tid; i < n4; i += stride) {
        float4 acc = reinterpret_cast<const float4 *>(S.src[0] + first)[i];
        for (int d = 1; d < S.n; ++d) {
            const float4 v = reinterpret_cast<const float4 *>(S.src[d] + first)[i];
            acc.x = acc.x + v.x;
            acc.y = acc.y + v.y;
            acc.z = acc.z + v.z;
            acc.w = acc.w + v.w;
        }
        if (divide) {
            acc.x = acc.x / divisor;
            acc.y = acc.y / divisor;
            acc.z = acc.z / divisor;
            acc.w = acc.w / divisor;
        }
        reinterpret_cast<float4 *>(out)[i] = acc;
    }
    for (uint64_t i = 4u * n4 + tid; i < count; i += stride) {
        float acc = S.src[0][first + i];
        for (int d = 1; d < S.n; ++d) acc = acc + S.src[d][first + i];
        out[i] = divide ? acc / divisor : acc;
    }
}

// Every path is cut before its first intersection when ray_depth is 0 (src/trace.rs:12-14): each
// sample adds (0, 0, 0) to its pixel.  (-0) + 0 = +0, so a continued sum is not simply kept.
__global__ void __launch_bounds__(256) k_add_zero_samples(float *sum, uint64_t n, int accumulate, int any_samples)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float s = accumulate ? sum[i] : 0.0f;
        sum[i] = any_samples ? s + 0.0f : s;
    }
}

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
    void *d_primary = nullptr;    // outputs of the primary pass
    size_t d_primary_bytes = 0;
    // progressive rendering: d_image holds the sum of samples [0, progress_done)
    bool progress_active = false;
    uint32_t progress_done = 0;
    void *d_wavefront = nullptr;  // path state of the wavefront variant
    size_t d_wavefront_bytes = 0;
    uint32_t *d_tiles = nullptr;  // cost and dispatch order of the 8x8 tiles, n_tiles words each
    size_t d_tiles_words = 0;
    // The order table of d_tiles is still the one measured for the scene, frame size and share in use: a scene is
    // measured once (the probe launch of render_range), and later renders of it are dispatched by that order.
    bool order_valid = false;
    uint32_t order_units = 0;
    uint32_t *static_tiles = nullptr;   // device address of this GPU's `scratch`
    uint32_t pool_limit = rtx::kPoolEntriesMax;   // cap of the megakernel's hand-over pool (RTX_POOL_ENTRIES)
    // this context's share of every frame (rtx_set_share; the workers of a multi-GPU context: worker g of G)
    uint32_t share_index = 0, share_count = 1;
    unsigned long long *d_stats = nullptr;
    rtx_stats stats{};
    bool stats_valid = false;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    // A multi-GPU context is its own worker 0 and owns one more context per further GPU.  Every
    // worker renders its share of the frame's work units into its d_image on its own stream.
    std::vector<rtx_ctx *> peers;
    hipStream_t stream = nullptr;   // workers of a multi-GPU context: their own stream
    hipEvent_t ev_done = nullptr;   // this worker's share of a render is complete
    hipEvent_t ev_out = nullptr;    // this worker's part of the image has been delivered
    float *d_reduced = nullptr;     // P2P: this worker's slice of the summed image; RCCL: the whole of it
    size_t d_reduced_floats = 0;
    void *comm = nullptr;           // RCCL communicator rank of this worker
};

namespace {

// the workers of a context: itself, then its peers
std::vector<rtx_ctx *> workers_of(rtx_ctx *c)
{
    std::vector<rtx_ctx *> w;
    w.reserve(1 + c->peers.size());
    w.push_back(c);
    for (rtx_ctx *p : c->peers) w.push_back(p);
    return w;
}

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

// a plane in the identity pose whose normal is +-1 along one axis and +-0 along the two others:
// 1 + that axis, or 0 (rtx_trace.hpp: plane_axis_t)
int axis_plane(const rtx_prim &p)
{
    if (p.kind != RTX_PLANE || !is_identity_pose(p)) return 0;
    int axis = -1;
    for (int k = 0; k < 3; ++k) {
        if (p.param[k] == 0.0f) continue;
        if (std::fabs(p.param[k]) != 1.0f || axis >= 0) return 0;
        axis = k;
    }
    return axis + 1;
}

// a box in the identity pose with half-sizes in (0, 1e30): points sampled on its surface skip the
// rotation (rtx_trace.hpp: light_surface_sample)
bool is_plain_box(const rtx_prim &p)
{
    if (p.kind != RTX_BOX || !is_identity_pose(p)) return false;
    for (int k = 0; k < 3; ++k)
        if (!(p.param[k] > 0.0f && p.param[k] < 1e30f)) return false;
    return true;
}

int scan_category_of(const rtx_prim &p)
{
    const bool identity = is_identity_pose(p), pow2 = is_pow2_param(p);
    if (p.kind == RTX_PLANE) {
        const int axis = axis_plane(p);
        if (axis != 0) return rtx::kCatPlaneX + (axis - 1);
        return identity ? rtx::kCatPlaneIdentity : rtx::kCatPlane;
    }
    if (p.kind == RTX_BOX) return identity ? rtx::kCatBoxIdentity : rtx::kCatBox;
    if (identity) return pow2 ? rtx::kCatEllipsoidIdentityPow2 : rtx::kCatEllipsoidIdentity;
    return pow2 ? rtx::kCatEllipsoidPow2 : rtx::kCatEllipsoid;
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

// The device scene block as the host builds it, once per rtx_set_scene; every worker of the
// context uploads the same bytes and points its kernel parameters into its own copy.
struct SceneBlock {
    std::vector<unsigned char> bytes;
    size_t off_mat = 0, off_lights = 0, off_lbound = 0, off_nodes = 0, off_planes = 0, off_order = 0;
    rtx::RenderParams params{};   // everything but the device pointers
};

void build_scene_block(const rtx_ctx *c, SceneBlock &blk)
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

    const size_t order_bytes = align16((rtx::kScanHeader + n + 2) * sizeof(int32_t));   // two entries of padding (zeros: a valid index)

    const size_t total = geom_bytes + mat_bytes + light_bytes + lbound_bytes + node_bytes + plane_bytes + order_bytes + 16;
    blk.bytes.assign(total, 0);
    std::vector<unsigned char> &host = blk.bytes;
    float4 *geom = reinterpret_cast<float4 *>(host.data());
    float4 *mat = reinterpret_cast<float4 *>(host.data() + geom_bytes);
    int32_t *lights = reinterpret_cast<int32_t *>(host.data() + geom_bytes + mat_bytes);
    for (size_t i = 0; i < n; ++i) {
        const rtx_prim &p = c->prims[i];
        int32_t kind = (int32_t)p.kind | (is_identity_pose(p) ? (rtx::kGeomIdentity << 8) : 0) |
                       (is_pow2_param(p) ? (rtx::kGeomPow2 << 8) : 0) |
                       (is_plain_box(p) ? (rtx::kGeomPlainBoxLight << 8) : 0);
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
    const size_t off_nodes = off_lbound + lbound_bytes;
    const size_t off_planes = off_nodes + node_bytes;
    if (!bvh.nodes.empty()) std::memcpy(host.data() + off_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(uint4));
    if (!bvh.planes.empty())
        std::memcpy(host.data() + off_planes, bvh.planes.data(), bvh.planes.size() * sizeof(int32_t));
    // the objects by scan category (rtx_device.h: kCat*)
    const size_t off_order = off_planes + plane_bytes;
    {
        int32_t *order = reinterpret_cast<int32_t *>(host.data() + off_order);
        size_t filled = 0;
        for (int cat = 0; cat < rtx::kScanCats; ++cat) {
            // a BVH scene scans the objects outside the tree only (rtx_bvh_traverse.hpp: bvh_always_tested)
            for (size_t i = 0; i < n; ++i) {
                if (want_bvh && !std::binary_search(bvh.planes.begin(), bvh.planes.end(), (int32_t)i)) continue;
                if (scan_category_of(c->prims[i]) == cat) order[rtx::kScanHeader + filled++] = (int32_t)i;
            }
            order[cat] = (int32_t)filled;
        }
    }
    blk.off_order = off_order;
    blk.off_mat = geom_bytes;
    blk.off_lights = geom_bytes + mat_bytes;
    blk.off_lbound = off_lbound;
    blk.off_nodes = off_nodes;
    blk.off_planes = off_planes;

    rtx::RenderParams &P = blk.params;
    std::memset(&P, 0, sizeof(P));
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
    P.ellipsoids_pow2 = 1u;
    for (const rtx_prim &p : c->prims)
        if (p.kind == RTX_ELLIPSOID && !is_pow2_param(p)) P.ellipsoids_pow2 = 0u;
    {
        // Scenes like the reference's own (assets/scene.txt): a few diffuse objects and lights without a
        // COLOR line, i.e. black ones, at which every path that reaches them ends (see stop_on_zero).
        // There a third of the closest hits are not followed by a scatter, which is what the two-pixel
        // scan kernel is for; where nearly every hit scatters it measured 3 to 24 % slower than the
        // one-pixel kernel.
        bool fits = P.stop_on_zero != 0u && n <= 16u && nl > 0u;
        for (const rtx_prim &p : c->prims) fits = fits && p.material == RTX_DIFFUSE;
        for (int32_t li : c->lights) {
            const rtx_prim &p = c->prims[(size_t)li];
#ifndef RTX_X_TWO_PIXELS_ANY_LIGHT   // timing experiment: the two-pixel kernel also where lights have a colour
            fits = fits && p.color[0] == 0.0f && p.color[1] == 0.0f && p.color[2] == 0.0f;
#endif
        }
        P.paths_end_at_lights = fits ? 1u : 0u;
    }
    std::memcpy(P.const_words, c->opts.const_words, sizeof(P.const_words));
    std::memcpy(P.bvh.grid_base, bvh.grid_base, sizeof(P.bvh.grid_base));
    std::memcpy(P.bvh.grid_inv_step, bvh.grid_inv_step, sizeof(P.bvh.grid_inv_step));
    std::memcpy(P.bvh.grid_step, bvh.grid_step, sizeof(P.bvh.grid_step));
    P.bvh.far_slope = bvh.far_slope;
    P.bvh.n_nodes = (uint32_t)(bvh.nodes.size() / rtx::kBvhNodeVec);
    P.bvh.n_planes = (uint32_t)bvh.planes.size();
    P.bvh.stack_entries = (uint32_t)std::min(bvh.depth + 1, rtx::kBvhStackSize);
    // Measured on scenes of 33 to 40 000 objects: small trees run faster when the object tests wait for
    // twice as many lanes as still descend, trees of a few hundred objects and more when they do not wait
    // (60 objects: 90.3 against 92.1 ms; 200: 101.1 / 100.4; 10 000: 156.8 / 151.6; 40 000: 152.4 / 144.1)
    P.bvh.leaf_cost = P.bvh.n_nodes >= 128u ? 1u : 2u;
    // The smaller the tree, the more a wave gains from going on until few lanes traverse: its traversals are short
    // and its surface interactions, which then run fuller, are most of the work.  Measured (ms per frame for 1 - 6
    // sixteenths): glass scene, 27 nodes, 191.6 189.4 191.6 198.1 201.5 209.2; 300 objects - 90.9 91.9 94.6; 1 000 objects
    // 106.8 104.7 104.6 106.7 110.1 113.9; 3 000 - 127.5 128.4 126.6; 10 000 147.0 139.0 135.1 133.1 132.7 133.1;
    // 20 000 - 132.5 126.5 126.1; 100 000 165.3 160.6 161.4 160.6 163.5 160.8.
    P.bvh.exit_sixteenths = P.bvh.n_nodes < 2048u ? 2u : 4u;
    // tuning aids (like RTX_POOL_ENTRIES): either weight from the environment.  Object-test weights of 1 - 4 with the
    // thresholds above: glass scene 192.5 / 189.7 / 189.2 / 189.0 ms, 60 objects 78.4 / 78.5 / 78.9 / 79.2, 1 000
    // objects 105.0 / 107.7 / 110.3 / 112.7, 10 000 objects 134.1 / 138.7 / 142.4 / 146.8 -- the rule above stands.
    if (const char *e = std::getenv("RTX_LEAF_COST")) P.bvh.leaf_cost = (uint32_t)std::min(64l, std::max(1l, std::strtol(e, nullptr, 10)));
    if (const char *e = std::getenv("RTX_EXIT_SIXTEENTHS")) P.bvh.exit_sixteenths = (uint32_t)std::min(16l, std::max(1l, std::strtol(e, nullptr, 10)));
}

// one worker's copy of the block; the worker's device must be current
int upload_scene_block(rtx_ctx *w, const SceneBlock &blk)
{
    // a render of this worker may still be reading the block that is about to be overwritten
    HIP_TRY(hipDeviceSynchronize());
    const size_t total = blk.bytes.size();
    if (w->d_scene_bytes < total) {
        if (w->d_scene) (void)hipFree(w->d_scene);
        w->d_scene = nullptr;
        w->d_scene_bytes = 0;
        HIP_TRY(hipMalloc(&w->d_scene, total));
        w->d_scene_bytes = total;
    }
    HIP_TRY(hipMemcpy(w->d_scene, blk.bytes.data(), total, hipMemcpyHostToDevice));
    unsigned char *base = static_cast<unsigned char *>(w->d_scene);
    rtx::RenderParams &P = w->params;
    P = blk.params;
    P.geom = reinterpret_cast<const float4 *>(base);
    P.mat = reinterpret_cast<const float4 *>(base + blk.off_mat);
    P.lights = reinterpret_cast<const int32_t *>(base + blk.off_lights);
    P.light_bounds = reinterpret_cast<const float4 *>(base + blk.off_lbound);
    P.scan_order = reinterpret_cast<const int32_t *>(base + blk.off_order);
    P.bvh.nodes = reinterpret_cast<const uint4 *>(base + blk.off_nodes);
    P.bvh.planes = reinterpret_cast<const int32_t *>(base + blk.off_planes);
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

int ensure_reduced(rtx_ctx *c, size_t floats)
{
    if (c->d_reduced_floats >= floats) return RTX_OK;
    if (c->d_reduced) (void)hipFree(c->d_reduced);
    c->d_reduced = nullptr;
    c->d_reduced_floats = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_reduced), floats * sizeof(float)));
    c->d_reduced_floats = floats;
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

void add_stats(rtx_stats &a, const rtx_stats &b)
{
    static_assert(sizeof(rtx_stats) % sizeof(uint64_t) == 0, "rtx_stats is an array of 64-bit counters");
    uint64_t *pa = reinterpret_cast<uint64_t *>(&a);
    const uint64_t *pb = reinterpret_cast<const uint64_t *>(&b);
    for (size_t i = 0; i < sizeof(rtx_stats) / sizeof(uint64_t); ++i) pa[i] += pb[i];
}

// contiguous, balanced cut of `count` items into `parts`: the first count % parts get one more
void split_range(uint64_t count, uint32_t parts, uint32_t part, uint64_t &first, uint64_t &len)
{
    const uint64_t base = count / parts, extra = count % parts;
    first = part * base + std::min<uint64_t>(part, extra);
    len = base + (part < extra ? 1u : 0u);
}
// the slice of the image worker g of G reduces: boundaries at multiples of 4 floats
void image_slice(uint64_t floats, uint32_t G, uint32_t g, uint64_t &first, uint64_t &len)
{
    const uint64_t quads = (floats + 3u) / 4u;
    uint64_t qf, ql;
    split_range(quads, G, g, qf, ql);
    first = std::min(floats, qf * 4u);
    len = std::min(floats, (qf + ql) * 4u) - first;
}

int create_worker(rtx_ctx **out, const rtx_opts *opts, int dev, bool own_stream)
{
    rtx_ctx *c = new (std::nothrow) rtx_ctx();
    if (!c) return fail(RTX_ERR_NO_MEMORY, "out of host memory");
    c->opts = *opts;
    c->opts.devices = nullptr;
    c->device = dev;
    // diagnostic knob (tests, measurements), read once: a smaller hand-over pool, 0 = no hand-over
    if (const char *e = getenv("RTX_POOL_ENTRIES")) {
        char *end = nullptr;
        const unsigned long v = strtoul(e, &end, 10);
        if (end != e && *end == '\0') c->pool_limit = (uint32_t)std::min<unsigned long>(v, rtx::kPoolEntriesMax);
    }
    DeviceGuard g(dev);
    if (!g.ok) { delete c; return fail(RTX_ERR_DEVICE, "cannot select device %d", dev); }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_stats), rtx::C_COUNT * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipEventCreate(&c->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_stop);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
    if (e == hipSuccess && own_stream) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    // this GPU's copy of the static tables (the device is current): the address every later use takes
    if (e == hipSuccess) e = hipGetSymbolAddress(reinterpret_cast<void **>(&c->static_tiles), HIP_SYMBOL(scratch));
    if (e != hipSuccess) {
        rtx_destroy(c);
        return fail(RTX_ERR_DEVICE, "context set-up on device %d: %s", dev, hipGetErrorString(e));
    }
    *out = c;
    return RTX_OK;
}

void destroy_worker(rtx_ctx *c)
{
    DeviceGuard g(c->device);
    (void)hipDeviceSynchronize();
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    free_scene(c);
    if (c->d_image) (void)hipFree(c->d_image);
    if (c->d_bytes) (void)hipFree(c->d_bytes);
    if (c->d_primary) (void)hipFree(c->d_primary);
    if (c->d_reduced) (void)hipFree(c->d_reduced);
    if (c->d_wavefront) (void)hipFree(c->d_wavefront);
    if (c->d_tiles) (void)hipFree(c->d_tiles);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // namespace

extern "C" {

uint32_t rtx_abi_version(void) { return RTX_ABI_VERSION; }

const char *rtx_last_error(void) { return g_last_error; }

int rtx_create(rtx_ctx **out, const rtx_opts *opts)
{
    if (!out || !opts) return fail(RTX_ERR_INVALID_ARG, "rtx_create: null argument");
    *out = nullptr;
    return guarded("rtx_create", [&]() -> int {
        if (opts->mode > RTX_MODE_WAVEFRONT) return fail(RTX_ERR_INVALID_ARG, "unknown mode %u", opts->mode);
        if (opts->accel > RTX_ACCEL_BVH) return fail(RTX_ERR_INVALID_ARG, "unknown accel %u", opts->accel);
        if (opts->rng_mode > RTX_RNG_CONST) return fail(RTX_ERR_INVALID_ARG, "unknown rng_mode %u", opts->rng_mode);
        if (opts->reduce > RTX_REDUCE_RCCL) return fail(RTX_ERR_INVALID_ARG, "unknown reduce %u", opts->reduce);
        if (opts->n_devices < 0 || opts->n_devices > RTX_MAX_DEVICES)
            return fail(RTX_ERR_INVALID_ARG, "n_devices %d out of range (at most %d)", opts->n_devices, RTX_MAX_DEVICES);
        int n_dev = 0;
        HIP_TRY(hipGetDeviceCount(&n_dev));
        if (n_dev <= 0) return fail(RTX_ERR_DEVICE, "no HIP device visible");
        const bool multi = opts->n_devices > 1 || (opts->n_devices == 1 && opts->reduce == RTX_REDUCE_RCCL);
        std::vector<int> devs;
        if (opts->n_devices <= 1 && !(opts->n_devices == 1 && opts->devices)) {
            int dev = opts->device;
            if (dev < 0) HIP_TRY(hipGetDevice(&dev));
            devs.push_back(dev);
        } else {
            for (int i = 0; i < opts->n_devices; ++i) devs.push_back(opts->devices ? opts->devices[i] : i);
        }
        for (int d : devs)
            if (d < 0 || d >= n_dev) return fail(RTX_ERR_INVALID_ARG, "device %d out of range (%d visible)", d, n_dev);
        if (multi && opts->reduce == RTX_REDUCE_RCCL) {
            for (size_t i = 0; i < devs.size(); ++i)
                for (size_t j = 0; j < i; ++j)
                    if (devs[i] == devs[j])
                        return fail(RTX_ERR_INVALID_ARG, "RTX_REDUCE_RCCL needs distinct devices (%d is listed twice)", devs[i]);
            int rc = load_rccl();
            if (rc != RTX_OK) return rc;
        }
        rtx_ctx *c = nullptr;
        int rc = create_worker(&c, opts, devs[0], multi);
        if (rc != RTX_OK) return rc;
        for (size_t i = 1; i < devs.size(); ++i) {
            rtx_ctx *p = nullptr;
            rc = create_worker(&p, opts, devs[i], true);
            if (rc != RTX_OK) { rtx_destroy(c); return rc; }
            c->peers.push_back(p);
        }
        {
            std::vector<rtx_ctx *> ws = workers_of(c);
            for (size_t i = 0; i < ws.size(); ++i) {
                ws[i]->share_index = (uint32_t)i;
                ws[i]->share_count = (uint32_t)ws.size();
            }
        }
        if (devs.size() > 1 && opts->reduce == RTX_REDUCE_P2P) {
            // every GPU reads its slice from every other GPU's buffer
            for (size_t i = 0; i < devs.size(); ++i) {
                DeviceGuard g(devs[i]);
                for (size_t j = 0; j < devs.size(); ++j) {
                    if (devs[i] == devs[j]) continue;
                    int can = 0;
                    hipError_t e = hipDeviceCanAccessPeer(&can, devs[i], devs[j]);
                    if (e == hipSuccess && can) {
                        e = hipDeviceEnablePeerAccess(devs[j], 0);
                        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
                    }
                    if (e != hipSuccess || !can) {
                        rtx_destroy(c);
                        return fail(RTX_ERR_UNSUPPORTED, "device %d cannot read device %d's memory (no peer access); use RTX_REDUCE_RCCL",
                                    devs[i], devs[j]);
                    }
                }
            }
        }
        if (multi && opts->reduce == RTX_REDUCE_RCCL) {
            std::vector<void *> comms(devs.size(), nullptr);
            int e = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
            if (e != 0) {
                rtx_destroy(c);
                return fail(RTX_ERR_DEVICE, "ncclCommInitAll: %s", g_rccl.GetErrorString(e));
            }
            std::vector<rtx_ctx *> ws = workers_of(c);
            for (size_t i = 0; i < ws.size(); ++i) ws[i]->comm = comms[i];
        }
        *out = c;
        return RTX_OK;
    });
}

void rtx_destroy(rtx_ctx *c)
{
    if (!c) return;
    for (rtx_ctx *p : c->peers) destroy_worker(p);
    c->peers.clear();
    destroy_worker(c);
}

int rtx_set_scene(rtx_ctx *c, const rtx_scene *s)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    int rc = validate_scene(s);
    if (rc != RTX_OK) return rc;
    return guarded("rtx_set_scene", [&]() -> int {
        std::vector<rtx_ctx *> ws = workers_of(c);
        for (rtx_ctx *w : ws) {
            w->have_scene = false;
            w->order_valid = false;   // another scene costs its tiles differently
        }
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
        SceneBlock blk;
        build_scene_block(c, blk);
        for (rtx_ctx *w : ws) {
            DeviceGuard g(w->device);
            if (!g.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", w->device);
            if (w != c) {
                // what the launch and statistics code reads from a worker's host copy
                w->scene = c->scene;
                w->src = c->src;
                w->prims = c->prims;
                w->lights = c->lights;
            }
            rc = upload_scene_block(w, blk);
            if (rc != RTX_OK) return rc;
        }
        for (rtx_ctx *w : ws) w->have_scene = true;
        return RTX_OK;
    });
}

}  // extern "C"

namespace {

// cost-ordered dispatch of the megakernel's tiles: used from this many samples and tiles on
// (fewer tiles than two per wave slot of the GPU leave nothing to reorder)
#ifndef RTX_PROBE_SAMPLES
#define RTX_PROBE_SAMPLES 8u
#endif
constexpr uint32_t kProbeSamples = RTX_PROBE_SAMPLES, kOrderedMinSamples = 64u, kOrderedMinTiles = 8192u;
#ifndef RTX_CHUNK_MIN
#define RTX_CHUNK_MIN 16u
#endif
#ifndef RTX_CHUNK_MIN_ONE
#define RTX_CHUNK_MIN_ONE 33u   // 64 samples stay one item (10 000 objects, 64 samples: two chunks of 32 139.3 ms, one item 132.9 ms)
#endif
#ifndef RTX_CHUNK_MAX
#define RTX_CHUNK_MAX 6u
#endif
// work queue: fewest samples of a chunk (two-pixel kernel, one-pixel kernels), most chunks (1: no queue)
constexpr uint32_t kChunkMin = RTX_CHUNK_MIN, kChunkMinOnePixel = RTX_CHUNK_MIN_ONE, kChunkMax = RTX_CHUNK_MAX;
#ifndef RTX_ONE_PIXEL_PLAN
#define RTX_ONE_PIXEL_PLAN 0   // experiment: 1 = the one-pixel kernels take the two-pixel kernel's launch plan (no cost
                               // order, only the end of the image order in chunks).  Measured: 10 000 objects 145.8 against
                               // 140.6 ms, glass scene 210.8 against 203.5 ms (their units differ too much in cost to go
                               // without the order), reference scene with a coloured light 87.9 against 88.8 ms
#endif
#ifndef RTX_QUEUE_TAIL
#define RTX_QUEUE_TAIL 8192u
#endif
#ifndef RTX_WHOLE_ONE
#define RTX_WHOLE_ONE 0   // experiment: 1 = the one-pixel kernels too keep the first units of their (cost) order whole.
                          // Measured: glass scene 199.5 against 197.7 ms, reference scene with a coloured light 85.9 against 86.3 ms
#endif
#ifndef RTX_REUSE_ORDER
#define RTX_REUSE_ORDER 1   // 0: every call measures the scene again (a probe launch per call)
#endif
constexpr uint32_t kQueueTail = RTX_QUEUE_TAIL;   // units at the end of the order that are cut into chunks
// the launch that follows the probe renders sample_count - kProbeSamples samples: at least one
static_assert(kProbeSamples >= 1u && kProbeSamples < kOrderedMinSamples, "the probe must leave samples for the ordered launch");

// how the kernel this context launches for its scene cuts the frame into work units
rtx::UnitShape unit_shape_of(const rtx_ctx *c)
{
    if (c->opts.mode == RTX_MODE_WAVEFRONT) return rtx::UnitShape{0u, rtx::wavefront_group_tiles(c->params, c->src)};
    return rtx::megakernel_unit_shape(c->params, c->src);
}

// `no_alloc`: the caller may not allocate device memory -- the tile tables are this GPU's static
// ones, and whatever would need a fresh block is refused.
int render_range(rtx_ctx *c, uint32_t sample_first, uint32_t sample_count, float *d_sum, void *hip_stream,
                 bool accumulate, bool no_alloc = false)
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
    P.share_index = c->share_index;
    P.share_count = c->share_count;
    const bool stats = c->opts.collect_stats != 0;
    P.stats = stats ? c->d_stats : nullptr;
    c->stats_valid = false;
    const uint32_t n_tiles = ((P.width + 7u) / 8u) * ((P.height + 7u) / 8u);
    // A call that may sit under HIP graph capture records no timing events (an event recorded into a
    // capturing stream cannot be waited for or read) and reads no counters back (a copy to the host
    // followed by a wait is illegal there): rtx_last_kernel_ms and rtx_get_stats do not cover it.
    if (no_alloc && stats)
        return fail(RTX_ERR_UNSUPPORTED, "rtx_render_samples_static: not available on a context with collect_stats = 1");
    const bool timed = !no_alloc;
    c->timed = false;
    if (P.ray_depth == 0u) {
        // src/trace.rs:12-14: the camera ray itself is cut, nothing is intersected
        if (timed) HIP_TRY(hipEventRecord(c->ev_start, stream));
        const uint64_t n = (uint64_t)P.width * P.height * 3u;
        hipLaunchKernelGGL(k_add_zero_samples, dim3((uint32_t)std::min<uint64_t>((n + 255u) / 256u, 4096u)), dim3(256), 0,
                           stream, d_sum, n, accumulate ? 1 : 0, sample_count > 0u ? 1 : 0);
        HIP_TRY(hipGetLastError());
        if (timed) HIP_TRY(hipEventRecord(c->ev_stop, stream));
        c->timed = timed;
        if (stats) {
            std::memset(&c->stats, 0, sizeof(c->stats));
            c->stats.samples = (uint64_t)P.width * P.height * sample_count;
            c->stats_valid = true;
        }
        return RTX_OK;
    }
    if (no_alloc) {
        if (c->opts.mode == RTX_MODE_WAVEFRONT && !rtx::wavefront_state_in_lds(P, c->src))
            return fail(RTX_ERR_UNSUPPORTED,
                        "rtx_render_samples_static: the wavefront variant keeps its path state in device memory for scenes with a deep BVH");
        if (n_tiles > RTX_STATIC_MAX_TILES)
            return fail(RTX_ERR_UNSUPPORTED, "rtx_render_samples_static: %u x %u has %u tiles of 8 x 8 pixels, the static tables hold %u",
                        P.width, P.height, n_tiles, (unsigned)RTX_STATIC_MAX_TILES);
    }
    if (stats) HIP_TRY(hipMemsetAsync(c->d_stats, 0, rtx::C_COUNT * sizeof(unsigned long long), stream));
    if (timed) HIP_TRY(hipEventRecord(c->ev_start, stream));
    // A share of the frame: the kernels write this share's pixels only, and a fresh sum is zero everywhere
    // else, so that the shares of a job add up to the frame.  (A kernel, not a memset: see the tile tables.)
    if (P.share_count > 1u && !accumulate)
        HIP_TRY(rtx::launch_zero_words(reinterpret_cast<uint32_t *>(d_sum), (uint64_t)P.width * P.height * 3u, stream));
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
        // A wave runs one unit (a tile or two) and the launch ends when the last of them does; with
        // units handed out in image order the last ones started can be the long ones.  Two remedies,
        // neither of which changes a bit (a pixel's sums continue in sample order, as
        // rtx_accumulate_samples does): a first launch of kProbeSamples samples that records what each
        // unit costs, so that the rest can be dispatched longest first; and the sample range cut into
        // chunks that waves draw from a queue, so that the last items to run are short.
        uint32_t probe = 0;
        uint32_t *tiles = nullptr;
        if (!stats && sample_count >= kOrderedMinSamples && rtx::share_units(n_tiles, P.share_index, P.share_count) >= kOrderedMinTiles) {
            probe = kProbeSamples;
            if (no_alloc) {
                tiles = c->static_tiles;
            } else {
                // cost | order | work queue (a ticket counter and one word per unit)
                const size_t words = 3u * (size_t)n_tiles + 64u;
                if (c->d_tiles_words < words) {
                    if (c->d_tiles) (void)hipFree(c->d_tiles);
                    c->d_tiles = nullptr;
                    c->d_tiles_words = 0;
                    c->order_valid = false;
                    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_tiles), words * sizeof(uint32_t)));
                    c->d_tiles_words = words;
                }
                tiles = c->d_tiles;
            }
        }
        // The rest of the launch plan depends on the kernel (measured, ms per frame, queue alone / cost
        // order + queue / cost order alone): the two-pixel scan kernel takes the work queue alone (reference
        // scene at 1920x1080x256: 61.1 / 61.8 / 65.0; at 3840x2160x1024: 938 / 939 / 946); the one-pixel kernels take the
        // cost order and, with chunks of 32 samples or more, the queue as well (glass scene, 256 samples: 210 / 202 /
        // 208; 10 000 objects, 64 samples, two chunks of 28: 148 / 145 / 140 -- short chunks cost more in
        // lanes that wait at each chunk's end than the shorter end of the launch returns).
        const bool two_pixel = probe != 0u && (RTX_ONE_PIXEL_PLAN == 1 || rtx::megakernel_two_pixels(P, c->src));
        if (two_pixel) probe = 0u;
        const uint32_t chunk_min = two_pixel ? kChunkMin : kChunkMinOnePixel;
        if (probe && RTX_REUSE_ORDER && !no_alloc && c->order_valid && c->order_units == rtx::megakernel_units(P, c->src)) {
            // The scene has been measured (an earlier call on this context, same frame size and share): its units
            // go out in that order again, all samples in one launch -- no probe, whose launch costs every unit a
            // wave's start and end (as much as 2.8 samples of each pixel) and has an end of its own.  The order
            // only decides which wave starts when, so a cost table that is a call old changes no bit.
            probe = 0u;
            P.tile_order = tiles + n_tiles;
        }
        if (probe) {
            // the launch deals out work units (8 x 8 tiles or pairs of them): at most n_tiles
            const uint32_t n_units = rtx::megakernel_units(P, c->src);
            uint32_t *cost = tiles, *order = tiles + n_tiles;
            // the tables keep the last call's contents: costs are maxima and must start from zero
            HIP_TRY(rtx::launch_zero_words(cost, n_tiles, stream));
            rtx::RenderParams A = P;
            A.sample_count = probe;
            A.tile_cost = cost;
            HIP_TRY(rtx::launch_megakernel(A, c->src, false, c->pool_limit, stream));
            HIP_TRY(rtx::launch_tile_order(cost, order, n_units, stream));
            P.sample_first = sample_first + probe;
            P.sample_count = sample_count - probe;
            P.accumulate = 1u;
            P.tile_order = order;
            if (!no_alloc) {
                c->order_valid = true;
                c->order_units = n_units;
            }
        }
        if (tiles) {
            // the samples in chunks -- halves of what is left, down to chunk_min samples -- drawn from a
            // queue, so that the launch ends with short work items instead of whole units (rtx_device.h)
            uint32_t chunks = 1u;
            while (chunks < kChunkMax && (P.sample_count >> chunks) >= chunk_min) ++chunks;
            if (chunks > 1u) {
                // Two-pixel kernel: only the last kQueueTail units of the order are cut into chunks -- twice what
                // the GPU runs at once, so that their work outlasts the whole units still running when the first of
                // them starts -- because every item costs a wave's start and end, as much as 2.8 samples of each of
                // its pixels (reference scene, ms per frame, all units chunked / the last 16 384 / 8 192 / 6 144 /
                // 4 096: 1080p x 256 61.3 / 61.3 / 60.5 / 60.3 / 63.9; 4K x 1024 938.5 / 922.3 / 920.2 / 920.4 / 934.1;
                // 4K x 128 122.6 / 120.5 / 120.1 / - / 121.4).  One-pixel kernels: every unit (their order is by
                // cost, and the glass scene measured 202.7 ms that way against 204.3 ms).
                const uint32_t n_units = rtx::megakernel_units(P, c->src);
                const uint32_t n_own = rtx::share_units(n_units, P.share_index, P.share_count);
                P.queue = tiles + 2u * (size_t)n_tiles;
                P.n_chunks = chunks;
                P.n_whole = (two_pixel || RTX_WHOLE_ONE) && n_own > kQueueTail ? n_own - kQueueTail : 0u;
                HIP_TRY(rtx::launch_zero_words(P.queue, 1u + (uint64_t)n_units, stream));
            }
        }
        HIP_TRY(rtx::launch_megakernel(P, c->src, stats, c->pool_limit, stream));
    }
    if (timed) HIP_TRY(hipEventRecord(c->ev_stop, stream));
    c->timed = timed;
    if (stats) {
        unsigned long long v[rtx::C_COUNT];
        HIP_TRY(hipMemcpyAsync(v, c->d_stats, sizeof(v), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        stats_from_counters(c, v, &c->stats);
        c->stats_valid = true;
    }
    return RTX_OK;
}

// ---- a multi-GPU context: the frame's work units dealt out to the workers, sums added on the devices ----

// Worker g renders samples [first, first + count) of its share of the frame's work units onto
// (accumulate) or into its own d_image, on its own stream; ev_done marks the end of its share.
// Returns without waiting.
int render_shares(rtx_ctx *c, uint32_t first, uint32_t count, bool accumulate)
{
    std::vector<rtx_ctx *> ws = workers_of(c);
    const uint32_t G = (uint32_t)ws.size();
    const size_t floats = (size_t)c->scene.width * c->scene.height * 3u;
    for (uint32_t g = 0; g < G; ++g) {
        rtx_ctx *w = ws[g];
        DeviceGuard dg(w->device);
        if (!dg.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", w->device);
        int rc = ensure_image(w, floats);
        if (rc != RTX_OK) return rc;
        rc = render_range(w, first, count, w->d_image, w->stream, accumulate);
        if (rc != RTX_OK) return rc;
        HIP_TRY(hipEventRecord(w->ev_done, w->stream));
    }
    return RTX_OK;
}

// Where the added-up image of a multi-GPU context goes.  Any subset may be asked for at once: the
// sums are added once whatever is delivered.  Host pointers are the caller's memory; the copies into
// it are asynchronous on each GPU's stream, which lets them run side by side when that memory is
// pinned (rtx_alloc_pinned) -- into pageable memory the runtime stages them one after the other.
struct Delivery {
    float *host_sum = nullptr;     // the plain sum (the state a progressive render saves)
    float *host_mean = nullptr;    // sum / divisor
    uint8_t *host_bytes = nullptr; // sum / divisor, tone-mapped
    float *dev_mean = nullptr;     // sum / divisor in a device buffer on the context's first GPU
};

// Adds the workers' d_image buffers and delivers the result.  The workers' own sums stay as they
// are, so a progressive render can go on.  Returns when everything asked for has arrived.
int reduce_deliver(rtx_ctx *c, float divisor, const Delivery &D)
{
    std::vector<rtx_ctx *> ws = workers_of(c);
    const uint32_t G = (uint32_t)ws.size();
    const uint64_t floats = (uint64_t)c->scene.width * c->scene.height * 3u;
    if (D.host_sum && D.dev_mean) return fail(RTX_ERR_INVALID_ARG, "the plain sum and a device image cannot be delivered together");
    // the device divides unless the plain sum has to leave it: then the mean is that sum divided on
    // the host, the same IEEE division
    const bool dev_divides = !D.host_sum && (D.host_mean || D.dev_mean);
    float *host_floats = D.host_sum ? D.host_sum : D.host_mean;
    if (c->opts.reduce == RTX_REDUCE_RCCL) {
        for (rtx_ctx *w : ws) {
            DeviceGuard dg(w->device);
            int rc = ensure_reduced(w, floats);
            if (rc != RTX_OK) return rc;
        }
        // one communicator rank per GPU, all driven by this thread: the calls form one group
        RCCL_TRY(g_rccl.GroupStart());
        for (rtx_ctx *w : ws) {
            DeviceGuard dg(w->device);
            int e = g_rccl.AllReduce(w->d_image, w->d_reduced, floats, kNcclFloat32, kNcclSum, w->comm, w->stream);
            if (e != 0) {
                (void)g_rccl.GroupEnd();
                return fail(RTX_ERR_DEVICE, "ncclAllReduce: %s", g_rccl.GetErrorString(e));
            }
        }
        RCCL_TRY(g_rccl.GroupEnd());
        // every GPU now holds the sum; GPU 0 finishes it and sends it on
        DeviceGuard dg(c->device);
        if (D.host_bytes) {
            int rc = ensure_bytes(c, floats);
            if (rc != RTX_OK) return rc;
            HIP_TRY(rtx::launch_tonemap(c->d_reduced, floats, divisor, c->d_bytes, c->stream));
            HIP_TRY(hipMemcpyAsync(D.host_bytes, c->d_bytes, floats, hipMemcpyDeviceToHost, c->stream));
        }
        if (dev_divides) HIP_TRY(rtx::launch_divide(c->d_reduced, floats, divisor, c->stream));
        if (host_floats) HIP_TRY(hipMemcpyAsync(host_floats, c->d_reduced, floats * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        if (D.dev_mean)
            HIP_TRY(hipMemcpyAsync(D.dev_mean, c->d_reduced, floats * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else {
        ReduceSources S;
        std::memset(&S, 0, sizeof(S));
        S.n = (int)G;
        for (uint32_t g = 0; g < G; ++g) S.src[g] = ws[g]->d_image;
        for (uint32_t g = 0; g < G; ++g) {
            rtx_ctx *w = ws[g];
            uint64_t first, len;
            image_slice(floats, G, g, first, len);
            if (len == 0) continue;
            DeviceGuard dg(w->device);
            if (!dg.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", w->device);
            // a device image alone: every GPU writes its slice straight into it (over peer access)
            const bool direct = D.dev_mean && !host_floats && !D.host_bytes;
            int rc = direct ? RTX_OK : ensure_reduced(w, len);
            if (rc != RTX_OK) return rc;
            // the slice is complete once every worker's share is
            for (rtx_ctx *o : ws)
                if (o != w) HIP_TRY(hipStreamWaitEvent(w->stream, o->ev_done, 0));
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((len / 4u + 255u) / 256u + 1u, 2048u);
            const bool div_now = dev_divides && !D.host_bytes;
            float *out = direct ? D.dev_mean + first : w->d_reduced;
            hipLaunchKernelGGL(k_reduce_slices, dim3(blocks), dim3(256), 0, w->stream, S, first, len, divisor, div_now ? 1 : 0, out);
            HIP_TRY(hipGetLastError());
            if (D.host_bytes) {
                rc = ensure_bytes(w, len);
                if (rc != RTX_OK) return rc;
                HIP_TRY(rtx::launch_tonemap(w->d_reduced, len, divisor, w->d_bytes, w->stream));
                HIP_TRY(hipMemcpyAsync(D.host_bytes + first, w->d_bytes, len, hipMemcpyDeviceToHost, w->stream));
                if (dev_divides) HIP_TRY(rtx::launch_divide(w->d_reduced, len, divisor, w->stream));
            }
            if (host_floats)
                HIP_TRY(hipMemcpyAsync(host_floats + first, w->d_reduced, len * sizeof(float), hipMemcpyDeviceToHost, w->stream));
            if (D.dev_mean && !direct)
                HIP_TRY(hipMemcpyAsync(D.dev_mean + first, w->d_reduced, len * sizeof(float), hipMemcpyDeviceToDevice, w->stream));
        }
    }
    // one wait for all of it: the first GPU's stream waits for the others' (a blocking wait per stream
    // costs the host a wake-up each -- four workers, 1920 x 1080 into pinned memory: 3.1 ms beyond the
    // kernels with a wait per stream, 2.0 ms with one)
    for (rtx_ctx *w : ws) {
        if (w == c) continue;
        {
            DeviceGuard dg(w->device);
            HIP_TRY(hipEventRecord(w->ev_out, w->stream));
        }
        DeviceGuard dg(c->device);
        HIP_TRY(hipStreamWaitEvent(c->stream, w->ev_out, 0));
    }
    {
        DeviceGuard dg(c->device);
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (D.host_sum && D.host_mean)
        for (uint64_t i = 0; i < floats; ++i) D.host_mean[i] = D.host_sum[i] / divisor;
    return RTX_OK;
}

bool is_multi(const rtx_ctx *c) { return c->stream != nullptr; }

// what a multi-GPU render leaves for rtx_get_stats: the workers' counters added up
void gather_stats(rtx_ctx *c)
{
    if (!c->opts.collect_stats) return;
    rtx_stats total;
    std::memset(&total, 0, sizeof(total));
    bool valid = true;
    for (rtx_ctx *w : workers_of(c)) {
        valid = valid && w->stats_valid;
        if (w->stats_valid) add_stats(total, w->stats);
    }
    c->stats = total;
    c->stats_valid = valid;
}

}  // namespace

extern "C" {

int rtx_render_samples(rtx_ctx *c, uint32_t sample_first, uint32_t sample_count, float *d_sum, void *hip_stream)
{
    return render_range(c, sample_first, sample_count, d_sum, hip_stream, false);
}

int rtx_accumulate_samples(rtx_ctx *c, uint32_t sample_first, uint32_t sample_count, float *d_sum, void *hip_stream)
{
    return render_range(c, sample_first, sample_count, d_sum, hip_stream, true);
}

int rtx_set_share(rtx_ctx *c, uint32_t share_index, uint32_t share_count)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    if (share_count == 0u || share_index >= share_count)
        return fail(RTX_ERR_INVALID_ARG, "rtx_set_share: share %u of %u", share_index, share_count);
    if (is_multi(c)) return fail(RTX_ERR_UNSUPPORTED, "rtx_set_share: a multi-GPU context deals the frame out to its own GPUs");
    c->share_index = share_index;
    c->share_count = share_count;
    c->order_valid = false;
    c->progress_active = false;   // a progressive render's sums belong to the share they were started with
    return RTX_OK;
}

int rtx_render_samples_static(rtx_ctx *c, uint32_t sample_first, uint32_t sample_count, float *d_sum, void *hip_stream)
{
    return render_range(c, sample_first, sample_count, d_sum, hip_stream, false, true);
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
    return guarded("rtx_render", [&]() -> int {
        DeviceGuard g(c->device);
        const size_t floats = (size_t)s->width * s->height * 3u;
        c->progress_active = false;
        if (s->n_samples == 0) {
            // src/main.rs:14: zero iterations leave the zero-initialised image (src/image.rs:17)
            std::memset(rgb, 0, floats * sizeof(float));
            return RTX_OK;
        }
        if (is_multi(c)) {
            // one call from the host does everything (src/main.rs:38): split, render, add, divide
            rc = render_shares(c, 0, s->n_samples, false);
            if (rc == RTX_OK) {
                Delivery D;
                D.host_mean = rgb;
                rc = reduce_deliver(c, (float)s->n_samples, D);
            }
            gather_stats(c);
            return rc;
        }
        rc = ensure_image(c, floats);
        if (rc != RTX_OK) return rc;
        rc = rtx_render_samples(c, 0, s->n_samples, c->d_image, nullptr);
        if (rc != RTX_OK) return rc;
        rc = rtx_divide(c, c->d_image, floats, (float)s->n_samples, nullptr);
        if (rc != RTX_OK) return rc;
        HIP_TRY(hipMemcpy(rgb, c->d_image, floats * sizeof(float), hipMemcpyDeviceToHost));
        return RTX_OK;
    });
}

int rtx_render_device(rtx_ctx *c, uint32_t n_samples, float *d_rgb)
{
    if (!c || !d_rgb) return fail(RTX_ERR_INVALID_ARG, "rtx_render_device: null argument");
    if (!c->have_scene) return fail(RTX_ERR_NO_SCENE, "rtx_set_scene has not been called");
    return guarded("rtx_render_device", [&]() -> int {
        DeviceGuard g(c->device);
        if (!g.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", c->device);
        const size_t floats = (size_t)c->scene.width * c->scene.height * 3u;
        c->progress_active = false;
        hipStream_t own = c->stream;   // a multi-GPU context's first worker has its own stream
        if (n_samples == 0) {
            // src/main.rs:14: zero iterations leave the zero-initialised image (src/image.rs:17)
            HIP_TRY(hipMemsetAsync(d_rgb, 0, floats * sizeof(float), own));
            HIP_TRY(hipStreamSynchronize(own));
            return RTX_OK;
        }
        int rc;
        if (is_multi(c)) {
            rc = render_shares(c, 0, n_samples, false);
            if (rc == RTX_OK) {
                Delivery D;
                D.dev_mean = d_rgb;
                rc = reduce_deliver(c, (float)n_samples, D);
            }
            gather_stats(c);
            return rc;
        }
        rc = render_range(c, 0, n_samples, d_rgb, nullptr, false);
        if (rc == RTX_OK) rc = rtx_divide(c, d_rgb, floats, (float)n_samples, nullptr);
        if (rc == RTX_OK) HIP_TRY(hipStreamSynchronize(nullptr));
        return rc;
    });
}

int rtx_render_bytes(rtx_ctx *c, const rtx_scene *s, uint8_t *bytes)
{
    if (!c || !bytes) return fail(RTX_ERR_INVALID_ARG, "rtx_render_bytes: null argument");
    int rc = rtx_set_scene(c, s);
    if (rc != RTX_OK) return rc;
    return guarded("rtx_render_bytes", [&]() -> int {
        DeviceGuard g(c->device);
        const size_t floats = (size_t)s->width * s->height * 3u;
        c->progress_active = false;
        // zero samples leave the zero image (src/main.rs:14, src/image.rs:17); 0 / 1 keeps it
        const float divisor = s->n_samples == 0 ? 1.0f : (float)s->n_samples;
        if (is_multi(c)) {
            rc = render_shares(c, 0, s->n_samples, false);
            if (rc == RTX_OK) {
                Delivery D;
                D.host_bytes = bytes;
                rc = reduce_deliver(c, divisor, D);
            }
            gather_stats(c);
            return rc;
        }
        rc = ensure_image(c, floats);
        if (rc == RTX_OK) rc = ensure_bytes(c, floats);
        if (rc != RTX_OK) return rc;
        rc = rtx_render_samples(c, 0, s->n_samples, c->d_image, nullptr);
        if (rc != RTX_OK) return rc;
        rc = rtx_tonemap_bytes(c, c->d_image, floats, divisor, c->d_bytes, nullptr);
        if (rc != RTX_OK) return rc;
        HIP_TRY(hipMemcpy(bytes, c->d_bytes, floats, hipMemcpyDeviceToHost));
        return RTX_OK;
    });
}

int rtx_progress_begin(rtx_ctx *c, const rtx_scene *s, const float *sum, uint32_t samples_done)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    c->progress_active = false;
    if (!sum && samples_done != 0)
        return fail(RTX_ERR_INVALID_ARG, "rtx_progress_begin: %u samples done but no sum to resume from", samples_done);
    int rc = rtx_set_scene(c, s);
    if (rc != RTX_OK) return rc;
    return guarded("rtx_progress_begin", [&]() -> int {
        const size_t floats = (size_t)s->width * s->height * 3u;
        // A saved sum continues, pixel by pixel, on the GPU that owns the pixel: every worker gets the sums of
        // its own share of the frame and zeros elsewhere.  (Added up again they give the saved sum back, but for
        // the sign of a saved -0.)
        std::vector<rtx_ctx *> ws = workers_of(c);
        std::vector<float> part;
        for (rtx_ctx *w : ws) {
            DeviceGuard g(w->device);
            rc = ensure_image(w, floats);
            if (rc != RTX_OK) return rc;
            if (sum && ws.size() == 1) {
                HIP_TRY(hipMemcpy(w->d_image, sum, floats * sizeof(float), hipMemcpyHostToDevice));
            } else if (sum) {
                const rtx::UnitShape shape = unit_shape_of(w);
                part.assign(floats, 0.0f);
                for (uint32_t y = 0; y < s->height; ++y)
                    for (uint32_t x = 0; x < s->width; ++x) {
                        if (rtx::unit_of_pixel(shape, s->width, x, y) % w->share_count != w->share_index) continue;
                        const size_t at = 3u * ((size_t)y * s->width + x);
                        part[at] = sum[at];
                        part[at + 1] = sum[at + 1];
                        part[at + 2] = sum[at + 2];
                    }
                HIP_TRY(hipMemcpy(w->d_image, part.data(), floats * sizeof(float), hipMemcpyHostToDevice));
            } else {
                HIP_TRY(hipMemset(w->d_image, 0, floats * sizeof(float)));
            }
            HIP_TRY(hipDeviceSynchronize());
        }
        c->progress_done = samples_done;
        c->progress_active = true;
        return RTX_OK;
    });
}

int rtx_progress_step(rtx_ctx *c, uint32_t sample_count)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    if (!c->progress_active) return fail(RTX_ERR_NO_SCENE, "rtx_progress_begin has not been called");
    if ((uint64_t)c->progress_done + sample_count > 0xffffffffull)
        return fail(RTX_ERR_INVALID_ARG, "sample range overflows 32 bits");
    return guarded("rtx_progress_step", [&]() -> int {
        int rc;
        if (is_multi(c)) {
            rc = render_shares(c, c->progress_done, sample_count, true);
            if (rc == RTX_OK) {
                for (rtx_ctx *w : workers_of(c)) {
                    DeviceGuard g(w->device);
                    HIP_TRY(hipStreamSynchronize(w->stream));
                }
            }
            gather_stats(c);
        } else {
            DeviceGuard g(c->device);
            rc = rtx_accumulate_samples(c, c->progress_done, sample_count, c->d_image, nullptr);
            if (rc == RTX_OK) HIP_TRY(hipStreamSynchronize(nullptr));
        }
        if (rc != RTX_OK) { c->progress_active = false; return rc; }
        c->progress_done += sample_count;
        return RTX_OK;
    });
}

int rtx_progress_read(rtx_ctx *c, float *rgb_mean, float *sum, uint8_t *bytes, uint32_t *samples_done)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    if (!c->progress_active) return fail(RTX_ERR_NO_SCENE, "rtx_progress_begin has not been called");
    return guarded("rtx_progress_read", [&]() -> int {
        DeviceGuard g(c->device);
        const size_t floats = (size_t)c->scene.width * c->scene.height * 3u;
        if (samples_done) *samples_done = c->progress_done;
        // the mean of zero samples is the zero image the reference starts from (src/image.rs:17)
        const float divisor = c->progress_done == 0 ? 1.0f : (float)c->progress_done;
        if (is_multi(c)) {
            // the state to save is the added-up sum: a resume puts it on the first GPU
            if (!sum && !rgb_mean && !bytes) return RTX_OK;
            Delivery D;
            D.host_sum = sum;
            D.host_mean = rgb_mean;
            D.host_bytes = bytes;
            return reduce_deliver(c, divisor, D);   // one pass over the GPUs' sums whatever is asked for
        }
        if (sum) HIP_TRY(hipMemcpy(sum, c->d_image, floats * sizeof(float), hipMemcpyDeviceToHost));
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
    });
}

int rtx_primary(rtx_ctx *c, const rtx_scene *s, int32_t *hit_id, float *t, float *normal, uint8_t *inside)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    int rc = rtx_set_scene(c, s);
    if (rc != RTX_OK) return rc;
    return guarded("rtx_primary", [&]() -> int {
    DeviceGuard g(c->device);   // a multi-GPU context runs this pass on its first GPU
    if (!g.ok) return fail(RTX_ERR_DEVICE, "cannot select device %d", c->device);
    const size_t n = (size_t)s->width * s->height;
    // one device block, kept by the context: ids | t | normal | inside
    const size_t bytes = n * 4 + n * 4 + n * 12 + align16(n);
    if (c->d_primary_bytes < bytes) {
        if (c->d_primary) (void)hipFree(c->d_primary);
        c->d_primary = nullptr;
        c->d_primary_bytes = 0;
        HIP_TRY(hipMalloc(&c->d_primary, bytes));
        c->d_primary_bytes = bytes;
    }
    unsigned char *b = static_cast<unsigned char *>(c->d_primary);
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
    if (e != hipSuccess) return fail(RTX_ERR_DEVICE, "primary pass: %s", hipGetErrorString(e));
    return RTX_OK;
    });
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
    // a multi-GPU context: its GPUs run side by side, the render takes as long as the slowest
    float worst = 0.0f;
    for (size_t i = 0; i <= c->peers.size(); ++i) {
        rtx_ctx *w = i == 0 ? c : c->peers[i - 1];
        if (!w->timed) continue;
        DeviceGuard g(w->device);
        float one = 0.0f;
        HIP_TRY(hipEventSynchronize(w->ev_stop));
        HIP_TRY(hipEventElapsedTime(&one, w->ev_start, w->ev_stop));
        worst = std::max(worst, one);
    }
    *ms = worst;
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

int rtx_alloc_pinned(void **out, uint64_t bytes)
{
    if (!out) return fail(RTX_ERR_INVALID_ARG, "rtx_alloc_pinned: null argument");
    *out = nullptr;
    if (bytes == 0) return RTX_OK;
    // portable: page-locked for every GPU of the node, so that all of a context's GPUs can copy into it
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return RTX_OK;
}

void rtx_free_pinned(void *p)
{
    if (p) (void)hipHostFree(p);
}

int rtx_device_count(rtx_ctx *c)
{
    if (!c) return fail(RTX_ERR_INVALID_ARG, "ctx is null");
    return (int)(1 + c->peers.size());
}

}  // extern "C"
