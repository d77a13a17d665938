// rtx_wavefront.hip -- wavefront variant of the trace loop (gfx950): one launch, workgroup-
// local queues in LDS, path state in HBM.
//
// The same per-pixel computation as k_megakernel, cut into stages that each run with
// (almost) every lane busy.  A workgroup of 512 threads owns 64 tiles of 8x8 pixels (4096
// pixels) from start to finish, so no path ever crosses a workgroup: queue counters and the
// queues themselves live in LDS, there are no global atomics, no inter-kernel hand-offs and
// no host round trips.  A pixel has at most one path in flight, so its samples are still
// accumulated in sample order and the image is bit-identical to the megakernel's.
//
// Path state is one 64-byte record per pixel in HBM (rtx_wavefront.h), indexed by SLOT =
// workgroup * 4096 + workgroup-local pixel number: a stage reads and writes whole cache lines
// whatever order the compacted queue hands the paths out in (a structure-of-arrays layout
// measured 2 KB of HBM traffic per path segment and ran at the HBM roof).  Only the radiance
// sum (the output buffer) is indexed by pixel.  Queues hold 16-bit local numbers.
// One iteration of the in-kernel loop advances every live path of the workgroup by one
// segment:
//   generate  fresh queue  -> primary ray of the pixel's next sample -> active queue
//   extend    active queue -> closest hit; a miss or the depth cut finishes the path here
//                             (-> next fresh queue); others -> the queue of the hit's material
//   shade<M>  queue M      -> one material at a time: scatter; continuing paths -> next
//                             active queue, rejected diffuse samples -> next fresh queue
// Appending to a queue is a wave64 compaction: ballot of the predicate, count of set bits
// below the lane (v_mbcnt), ONE LDS atomic per wave for the base offset.
#include "rtx_bvh_traverse.hpp"
#include "rtx_trace.hpp"
#include "rtx_wavefront.h"

namespace rtx {

namespace {

constexpr uint32_t kWfThreads = 512;
constexpr uint32_t kWfTilesPerGroup = 64;                       // 8x8 pixel tiles
constexpr uint32_t kWfPixelsPerGroup = kWfTilesPerGroup * 64u;  // 4096: fits 16-bit queue entries

enum : int { Q_ACTIVE0 = 0, Q_ACTIVE1, Q_FRESH0, Q_FRESH1, Q_DIFFUSE, Q_METALLIC, Q_DIELECTRIC, Q_COUNT };

struct WfLds {
    uint16_t queue[Q_COUNT][kWfPixelsPerGroup];
    uint32_t count[Q_COUNT];
};

extern __shared__ float4 rtx_smem[];

// wave64 compaction into an LDS queue: lanes with `pred` append `value`
RTX_DEV void wf_push(WfLds &L, int q, bool pred, uint32_t value)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int leader = __builtin_ctzll(mask);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&L.count[q], (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (pred) L.queue[q][base + below] = (uint16_t)value;
}

// workgroup-local pixel number -> pixel; tiles of a workgroup are consecutive in tile order
RTX_DEV bool wf_pixel(const RenderParams &P, uint32_t local, uint32_t &px, uint32_t &py)
{
    const uint32_t tile = blockIdx.x * kWfTilesPerGroup + (local >> 6);
    const uint32_t in_tile = local & 63u;
    const uint32_t tiles_x = (P.width + 7u) >> 3;
    px = (tile % tiles_x) * 8u + (in_tile & 7u);
    py = (tile / tiles_x) * 8u + (in_tile >> 3);
    return px < P.width && py < P.height;
}

struct WfPath {
    Path p;
    float hit_t;
    int hit_idx;
    uint32_t sample;
};
RTX_DEV WfPath wf_load(const WfState &S, uint32_t slot)
{
    const float4 *r = S.rec + 4ull * slot;
    float4 a = r[0], b = r[1], c = r[2], e = r[3];
    WfPath w;
    w.p.o = v3(a.x, a.y, a.z);
    w.p.d = v3(b.x, b.y, b.z);
    w.p.T = v3(c.x, c.y, c.z);
    w.p.L = v3(e.x, e.y, e.z);
    w.hit_t = a.w;
    w.hit_idx = __float_as_int(b.w);
    w.p.depth = __float_as_uint(c.w);
    w.sample = __float_as_uint(e.w);
    return w;
}
RTX_DEV void wf_store_ray(const WfState &S, uint32_t slot, const Path &p, float hit_t, int hit_idx)
{
    float4 *r = S.rec + 4ull * slot;
    r[0] = make_float4(p.o.x, p.o.y, p.o.z, hit_t);
    r[1] = make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(hit_idx));
}
RTX_DEV void wf_store_thr(const WfState &S, uint32_t slot, const Path &p)
{
    S.rec[4ull * slot + 2] = make_float4(p.T.x, p.T.y, p.T.z, __uint_as_float(p.depth));
}
RTX_DEV void wf_store_rad(const WfState &S, uint32_t slot, V3 L, uint32_t sample)
{
    S.rec[4ull * slot + 3] = make_float4(L.x, L.y, L.z, __uint_as_float(sample));
}
// the path of pixel `pixel` is complete: add it to the pixel's sum, move to the next sample
RTX_DEV bool wf_finish_path(const RenderParams &P, const WfState &S, uint32_t slot, uint32_t pixel, V3 L,
                            uint32_t sample)
{
    float *out = P.out_sum + 3ull * pixel;
    out[0] = out[0] + L.x;
    out[1] = out[1] + L.y;
    out[2] = out[2] + L.z;
    wf_store_rad(S, slot, v3(0.0f, 0.0f, 0.0f), sample + 1u);
    return sample + 1u < P.sample_first + P.sample_count;  // more samples to do
}

// one material's queue: scatter (src/trace.rs:25-68 for that material)
template <int MATERIAL, bool STATS, bool MANY_LIGHTS>
RTX_DEV void wf_shade(const RenderParams &P, const WfState &S, WfLds &L, const Table &tab, uint32_t nxt,
                      Counters<STATS> &cnt)
{
    const uint32_t count = L.count[Q_DIFFUSE + MATERIAL];
    for (uint32_t base = 0; base < count; base += kWfThreads) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < count;
        uint32_t local = 0;
        bool go_on = false, refresh = false;
        if (valid) {
            local = L.queue[Q_DIFFUSE + MATERIAL][i];
            uint32_t px, py;
            wf_pixel(P, local, px, py);
            const uint32_t pixel = py * P.width + px;
            const uint32_t slot = blockIdx.x * kWfPixelsPerGroup + local;
            WfPath w = wf_load(S, slot);
            const bool ended = bounce<STATS, MANY_LIGHTS>(P, tab, w.p, w.hit_idx, w.hit_t, MATERIAL, pixel, w.sample, cnt);
            if (ended) {
                refresh = wf_finish_path(P, S, slot, pixel, w.p.L, w.sample);
            } else {
                wf_store_ray(S, slot, w.p, 0.0f, -1);
                wf_store_thr(S, slot, w.p);
                go_on = true;
            }
        }
        wf_push(L, Q_ACTIVE0 + nxt, go_on, local);
        wf_push(L, Q_FRESH0 + nxt, refresh, local);
    }
}

template <int SRC, bool STATS, bool MANY_LIGHTS>
__global__ void __launch_bounds__(kWfThreads) k_wavefront(const RenderParams P, const WfState S)
{
    // dynamic LDS: [queues + counters][object table when SRC == kSrcLds]
    WfLds &L = *reinterpret_cast<WfLds *>(rtx_smem);
    Table tab;
    tab.n_prims = P.n_prims;
    tab.n_lights = P.n_lights;
    tab.light_bounds = P.light_bounds;
    if constexpr (SRC == kSrcLds) {
        float4 *tbl = rtx_smem + (sizeof(WfLds) + 15u) / 16u;
        const uint32_t ng = 3u * P.n_prims, nm = 3u * P.n_prims;
        for (uint32_t i = threadIdx.x; i < ng; i += blockDim.x) tbl[i] = P.geom[i];
        for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) tbl[ng + i] = P.mat[i];
        int32_t *sl = reinterpret_cast<int32_t *>(tbl + ng + nm);
        for (uint32_t i = threadIdx.x; i < P.n_lights; i += blockDim.x) sl[i] = P.lights[i];
        tab.geom = tbl;
        tab.mat = tbl + ng;
        tab.lights = sl;
    } else {
        tab.geom = P.geom;
        tab.mat = P.mat;
        tab.lights = P.lights;
    }
    if (threadIdx.x < Q_COUNT) L.count[threadIdx.x] = 0u;
    __syncthreads();

    const V3 bg = v3(P.bg[0], P.bg[1], P.bg[2]);
    Counters<STATS> cnt;
    cnt.clear();

    // every pixel of the workgroup starts in the fresh queue with a zero sum
    for (uint32_t base = 0; base < kWfPixelsPerGroup; base += kWfThreads) {
        const uint32_t local = base + threadIdx.x;
        uint32_t px, py;
        const bool valid = wf_pixel(P, local, px, py);
        if (valid) {
            const uint32_t pixel = py * P.width + px;
            wf_store_rad(S, blockIdx.x * kWfPixelsPerGroup + local, v3(0.0f, 0.0f, 0.0f), P.sample_first);
            if (!P.accumulate) {
                P.out_sum[3ull * pixel + 0] = 0.0f;
                P.out_sum[3ull * pixel + 1] = 0.0f;
                P.out_sum[3ull * pixel + 2] = 0.0f;
            }
        }
        wf_push(L, Q_FRESH0, valid && P.sample_count > 0u, local);
    }
    __syncthreads();

    for (uint32_t it = 0;; ++it) {
        const uint32_t cur = it & 1u, nxt = cur ^ 1u;
        if (threadIdx.x == 0) {
            L.count[Q_ACTIVE0 + nxt] = 0u;
            L.count[Q_FRESH0 + nxt] = 0u;
            L.count[Q_DIFFUSE] = 0u;
            L.count[Q_METALLIC] = 0u;
            L.count[Q_DIELECTRIC] = 0u;
        }
        __syncthreads();

        // ---- generate: src/main.rs:17-21
        {
            const uint32_t count = L.count[Q_FRESH0 + cur];
            for (uint32_t base = 0; base < count; base += kWfThreads) {
                const uint32_t i = base + threadIdx.x;
                const bool valid = i < count;
                uint32_t local = 0;
                if (valid) {
                    local = L.queue[Q_FRESH0 + cur][i];
                    uint32_t px, py;
                    wf_pixel(P, local, px, py);
                    const uint32_t pixel = py * P.width + px;
                    const uint32_t slot = blockIdx.x * kWfPixelsPerGroup + local;
                    // the finished path left (0, 0, 0 | next sample index) in the radiance record
                    const uint32_t sample = __float_as_uint(S.rec[4ull * slot + 3].w);
                    uint4 w = draw_block(P, pixel, sample, 0u);
                    Path p;
                    camera_ray(P, px, P.height - 1u - py, u24(w.x), u24(w.y), p.o, p.d);
                    p.T = v3(1.0f, 1.0f, 1.0f);
                    p.depth = 0u;
                    wf_store_ray(S, slot, p, 0.0f, -1);
                    wf_store_thr(S, slot, p);
                    cnt.template add<C_SAMPLES>();
                }
                wf_push(L, Q_ACTIVE0 + cur, valid, local);
            }
        }
        __syncthreads();

        // ---- extend: src/trace.rs:12-23
        {
            const uint32_t count = L.count[Q_ACTIVE0 + cur];
            for (uint32_t base = 0; base < count; base += kWfThreads) {
                const uint32_t i = base + threadIdx.x;
                const bool valid = i < count;
                uint32_t local = 0;
                int material = -1;
                bool refresh = false;
                if (valid) {
                    local = L.queue[Q_ACTIVE0 + cur][i];
                    uint32_t px, py;
                    wf_pixel(P, local, px, py);
                    const uint32_t pixel = py * P.width + px;
                    const uint32_t slot = blockIdx.x * kWfPixelsPerGroup + local;
                    WfPath w = wf_load(S, slot);
                    Path &p = w.p;
                    cnt.template add<C_RAYS>();
                    float t;
                    int idx;
                    if constexpr (SRC == kSrcBvh) idx = closest_hit_bvh<STATS, (int)kWfThreads>(P.bvh, tab, p.o, p.d, t, cnt);
                    else idx = closest_hit_scan<true>(tab, p.o, p.d, t);
                    bool done;
                    if (idx < 0) {
                        p.L = p.L + cmul(p.T, bg);
                        done = true;
                    } else {
                        if constexpr (STATS) {
                            int kind = __float_as_int(tab.geom[3 * idx + 1].w) & 0xff;
                            cnt.template add<C_HIT_PLANE>(kind == kPlane);
                            cnt.template add<C_HIT_ELLIPSOID>(kind == kEllipsoid);
                            cnt.template add<C_HIT_BOX>(kind == kBox);
                        }
                        done = surface_emission(P, tab, p, idx) || black_surface_ends(P, tab, p, idx);
                    }
                    if (done) {
                        refresh = wf_finish_path(P, S, slot, pixel, p.L, w.sample);
                    } else {
                        wf_store_rad(S, slot, p.L, w.sample);
                        wf_store_ray(S, slot, p, t, idx);
                        material = __float_as_int(tab.mat[3 * idx + 0].w);
                    }
                }
                wf_push(L, Q_FRESH0 + nxt, refresh, local);
                wf_push(L, Q_DIFFUSE, material == kDiffuse, local);
                wf_push(L, Q_METALLIC, material == kMetallic, local);
                wf_push(L, Q_DIELECTRIC, material == kDielectric, local);
            }
        }
        __syncthreads();

        // ---- shade, one material at a time
        wf_shade<kDiffuse, STATS, MANY_LIGHTS>(P, S, L, tab, nxt, cnt);
        wf_shade<kMetallic, STATS, MANY_LIGHTS>(P, S, L, tab, nxt, cnt);
        wf_shade<kDielectric, STATS, MANY_LIGHTS>(P, S, L, tab, nxt, cnt);
        __syncthreads();

        if (L.count[Q_ACTIVE0 + nxt] + L.count[Q_FRESH0 + nxt] == 0u) break;
        __syncthreads();  // the counters are reset at the top of the next iteration
    }

    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < C_COUNT; ++i)
            if (cnt.v[i]) atomicAdd(&P.stats[i], (unsigned long long)cnt.v[i]);
    }
}

}  // namespace

static uint32_t wavefront_slots(uint32_t width, uint32_t height)
{
    const uint32_t tiles = ((width + 7u) / 8u) * ((height + 7u) / 8u);
    return (tiles + kWfTilesPerGroup - 1u) / kWfTilesPerGroup * kWfPixelsPerGroup;
}

size_t wavefront_state_bytes(uint32_t width, uint32_t height)
{
    return (size_t)wavefront_slots(width, height) * 4u * sizeof(float4);
}

void wavefront_bind_state(void *block, uint32_t, uint32_t, WfState &S)
{
    S.rec = static_cast<float4 *>(block);
}

template <int SRC, bool STATS, bool MANY_LIGHTS>
static hipError_t launch_variant(const RenderParams &P, const WfState &S, dim3 grid, size_t lds, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wavefront<SRC, STATS, MANY_LIGHTS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_wavefront<SRC, STATS, MANY_LIGHTS>), grid, dim3(kWfThreads), lds, stream, P, S);
    return hipGetLastError();
}
template <int SRC, bool STATS>
static hipError_t launch_one(const RenderParams &P, const WfState &S, dim3 grid, size_t lds, hipStream_t stream)
{
    return P.n_lights > kFewLights ? launch_variant<SRC, STATS, true>(P, S, grid, lds, stream)
                                   : launch_variant<SRC, STATS, false>(P, S, grid, lds, stream);
}

hipError_t launch_wavefront(const RenderParams &P, const WfState &S, int src, bool stats, hipStream_t stream)
{
    const uint32_t tiles = ((P.width + 7u) / 8u) * ((P.height + 7u) / 8u);
    const dim3 grid((tiles + kWfTilesPerGroup - 1u) / kWfTilesPerGroup);
    size_t lds = (sizeof(WfLds) + 15u) / 16u * 16u;
    if (src == kSrcLds) lds += scene_lds_bytes(P.n_prims, P.n_lights);
    if (src == kSrcLds) return stats ? launch_one<kSrcLds, true>(P, S, grid, lds, stream)
                                     : launch_one<kSrcLds, false>(P, S, grid, lds, stream);
    if (src == kSrcScalar) return stats ? launch_one<kSrcScalar, true>(P, S, grid, lds, stream)
                                        : launch_one<kSrcScalar, false>(P, S, grid, lds, stream);
    return stats ? launch_one<kSrcBvh, true>(P, S, grid, lds, stream)
                 : launch_one<kSrcBvh, false>(P, S, grid, lds, stream);
}

}  // namespace rtx
