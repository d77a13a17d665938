// rtx_wavefront.hip -- wavefront variant of the trace loop (gfx950): one launch, workgroup-
// local queues in LDS, path state in LDS (small scenes) or in HBM.
//
// The same per-pixel computation as k_megakernel, cut into stages that each run with
// (almost) every lane busy.  A workgroup owns a set of 8x8 pixel tiles from start to finish,
// so no path ever crosses a workgroup: queue counters and the queues themselves live in LDS,
// there are no global atomics, no inter-kernel hand-offs and no host round trips.  A pixel has
// at most one path in flight, so its samples are still accumulated in sample order and the
// image is bit-identical to the megakernel's.
//
// Path state is one 64-byte record per pixel (rtx_wavefront.h), indexed by the workgroup-local
// pixel number.  Two layouts of the workgroup (WfShape):
//   * LDS state: 512 threads own 16 tiles (1024 pixels); the records (64 KB) sit next to the
//     queues in LDS, two workgroups per CU.  With the state in HBM the kernel ran at the memory
//     roof (about 400 bytes of traffic per path segment; the compacted queues hand the paths
//     out in any order, so every stage re-reads and re-writes scattered records), while its
//     instruction count is a fifth below the megakernel's.  Used for the linear scan (the BVH
//     needs the LDS for its traversal stacks); an object table that does not fit beside the
//     records is read from device memory through scalar loads instead.
//   * HBM state: 512 threads own 64 tiles (4096 pixels), records in device memory at
//     workgroup * 4096 + local, a whole cache line per access (a structure-of-arrays layout
//     measured 2 KB of traffic per path segment).
// Only the radiance sum (the output buffer) is indexed by pixel.  Queues hold 16-bit local
// numbers.  One iteration of the in-kernel loop advances every live path of the workgroup by
// one segment:
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

// shape of a workgroup: threads, 8x8 tiles and pixels it owns (pixel numbers fit 16-bit queue entries)
template <bool LDS_STATE>
struct WfShape {
    static constexpr uint32_t kThreads = 512u;
    static constexpr uint32_t kTiles = LDS_STATE ? 16u : 64u;
    static constexpr uint32_t kPixels = kTiles * 64u;
    static constexpr uint32_t kPlane = LDS_STATE ? kPixels : 0u;   // see wf_at
    // the start-up loop walks the pixels in steps of kThreads and must not run past them
    static_assert(kPixels % kThreads == 0 && kPixels <= 65536u, "a workgroup's pixels: a multiple of its threads, 16-bit numbers");
};

enum : int { Q_ACTIVE0 = 0, Q_ACTIVE1, Q_FRESH0, Q_FRESH1, Q_DIFFUSE, Q_METALLIC, Q_DIELECTRIC, Q_COUNT };

template <uint32_t PIXELS>
struct WfLds {
    uint16_t queue[Q_COUNT][PIXELS];
    uint32_t count[Q_COUNT];
};

extern __shared__ float4 rtx_smem[];

// wave64 compaction into an LDS queue: lanes with `pred` append `value`
template <uint32_t PIXELS>
RTX_DEV void wf_push(WfLds<PIXELS> &L, int q, bool pred, uint32_t value)
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
template <uint32_t TILES>
RTX_DEV bool wf_pixel(const RenderParams &P, uint32_t local, uint32_t &px, uint32_t &py)
{
    const uint32_t tile = blockIdx.x * TILES + (local >> 6);
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
// `rec` = the workgroup's records, `local` = workgroup-local pixel number.  PLANE = 0: device
// memory, the four float4 of a path are consecutive (one cache line).  PLANE = n: LDS, part k
// of all n paths forms one array -- with consecutive parts a wave's 16-byte reads fall on four
// of the sixteen bank groups whatever the paths are (measured: 57 % of LDS cycles were bank
// conflicts), with one array per part they spread over all of them.
template <uint32_t PLANE>
RTX_DEV uint32_t wf_at(uint32_t local, uint32_t k) { return PLANE ? k * PLANE + local : 4u * local + k; }

template <uint32_t PLANE>
RTX_DEV WfPath wf_load(const float4 *rec, uint32_t local)
{
    float4 a = rec[wf_at<PLANE>(local, 0)], b = rec[wf_at<PLANE>(local, 1)], c = rec[wf_at<PLANE>(local, 2)],
           e = rec[wf_at<PLANE>(local, 3)];
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
template <uint32_t PLANE>
RTX_DEV void wf_store_ray(float4 *rec, uint32_t local, const Path &p, float hit_t, int hit_idx)
{
    rec[wf_at<PLANE>(local, 0)] = make_float4(p.o.x, p.o.y, p.o.z, hit_t);
    rec[wf_at<PLANE>(local, 1)] = make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(hit_idx));
}
template <uint32_t PLANE>
RTX_DEV void wf_store_thr(float4 *rec, uint32_t local, const Path &p)
{
    rec[wf_at<PLANE>(local, 2)] = make_float4(p.T.x, p.T.y, p.T.z, __uint_as_float(p.depth));
}
template <uint32_t PLANE>
RTX_DEV void wf_store_rad(float4 *rec, uint32_t local, V3 L, uint32_t sample)
{
    rec[wf_at<PLANE>(local, 3)] = make_float4(L.x, L.y, L.z, __uint_as_float(sample));
}
// the path of pixel `pixel` is complete: add it to the pixel's sum, move to the next sample
template <uint32_t PLANE>
RTX_DEV bool wf_finish_path(const RenderParams &P, float4 *rec, uint32_t local, uint32_t pixel, V3 L,
                            uint32_t sample)
{
    float *out = P.out_sum + 3ull * pixel;
    out[0] = out[0] + L.x;
    out[1] = out[1] + L.y;
    out[2] = out[2] + L.z;
    wf_store_rad<PLANE>(rec, local, v3(0.0f, 0.0f, 0.0f), sample + 1u);
    return sample + 1u < P.sample_first + P.sample_count;  // more samples to do
}

// one material's queue: scatter (src/trace.rs:25-68 for that material)
template <int MATERIAL, bool STATS, bool MANY_LIGHTS, class SHAPE>
RTX_DEV void wf_shade(const RenderParams &P, float4 *rec, WfLds<SHAPE::kPixels> &L, const Table &tab, uint32_t nxt,
                      Counters<STATS> &cnt)
{
    const uint32_t count = L.count[Q_DIFFUSE + MATERIAL];
    for (uint32_t base = 0; base < count; base += SHAPE::kThreads) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < count;
        uint32_t local = 0;
        bool go_on = false, refresh = false;
        if (valid) {
            local = L.queue[Q_DIFFUSE + MATERIAL][i];
            uint32_t px, py;
            wf_pixel<SHAPE::kTiles>(P, local, px, py);
            const uint32_t pixel = py * P.width + px;
            WfPath w = wf_load<SHAPE::kPlane>(rec, local);
            const bool ended = bounce<STATS, MANY_LIGHTS>(P, tab, w.p, w.hit_idx, w.hit_t, MATERIAL, pixel, w.sample, cnt);
            if (ended) {
                refresh = wf_finish_path<SHAPE::kPlane>(P, rec, local, pixel, w.p.L, w.sample);
            } else {
                wf_store_ray<SHAPE::kPlane>(rec, local, w.p, 0.0f, -1);
                wf_store_thr<SHAPE::kPlane>(rec, local, w.p);
                go_on = true;
            }
        }
        wf_push(L, Q_ACTIVE0 + nxt, go_on, local);
        wf_push(L, Q_FRESH0 + nxt, refresh, local);
    }
}

template <int SRC, bool STATS, bool MANY_LIGHTS, bool LDS_STATE>
__global__ void __launch_bounds__(WfShape<LDS_STATE>::kThreads) k_wavefront(const RenderParams P, const WfState S)
{
    using SHAPE = WfShape<LDS_STATE>;
    constexpr uint32_t kWfThreads = SHAPE::kThreads, kWfPixelsPerGroup = SHAPE::kPixels;
    // dynamic LDS: [queues + counters][records when LDS_STATE][object table when SRC == kSrcLds]
    WfLds<kWfPixelsPerGroup> &L = *reinterpret_cast<WfLds<kWfPixelsPerGroup> *>(rtx_smem);
    constexpr uint32_t kQueueVec = (sizeof(WfLds<kWfPixelsPerGroup>) + 15u) / 16u;
    constexpr uint32_t kRecVec = LDS_STATE ? 4u * kWfPixelsPerGroup : 0u;
    float4 *rec;
    if constexpr (LDS_STATE) rec = rtx_smem + kQueueVec;
    else rec = S.rec + 4ull * blockIdx.x * kWfPixelsPerGroup;
    Table tab;
    tab.n_prims = P.n_prims;
    tab.n_lights = P.n_lights;
    tab.light_bounds = P.light_bounds;
    if constexpr (SRC == kSrcLds) {
        float4 *tbl = rtx_smem + kQueueVec + kRecVec;
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
        const bool valid = wf_pixel<SHAPE::kTiles>(P, local, px, py);
        if (valid) {
            const uint32_t pixel = py * P.width + px;
            wf_store_rad<SHAPE::kPlane>(rec, local, v3(0.0f, 0.0f, 0.0f), P.sample_first);
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
                    wf_pixel<SHAPE::kTiles>(P, local, px, py);
                    const uint32_t pixel = py * P.width + px;
                    // the finished path left (0, 0, 0 | next sample index) in the radiance record
                    const uint32_t sample = __float_as_uint(rec[wf_at<SHAPE::kPlane>(local, 3)].w);
                    uint4 w = draw_block(P, pixel, sample, 0u);
                    Path p;
                    camera_ray(P, px, P.height - 1u - py, u24(w.x), u24(w.y), p.o, p.d);
                    p.T = v3(1.0f, 1.0f, 1.0f);
                    p.depth = 0u;
                    wf_store_ray<SHAPE::kPlane>(rec, local, p, 0.0f, -1);
                    wf_store_thr<SHAPE::kPlane>(rec, local, p);
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
                    wf_pixel<SHAPE::kTiles>(P, local, px, py);
                    const uint32_t pixel = py * P.width + px;
                    WfPath w = wf_load<SHAPE::kPlane>(rec, local);
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
                        refresh = wf_finish_path<SHAPE::kPlane>(P, rec, local, pixel, p.L, w.sample);
                    } else {
                        wf_store_rad<SHAPE::kPlane>(rec, local, p.L, w.sample);
                        wf_store_ray<SHAPE::kPlane>(rec, local, p, t, idx);
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
        wf_shade<kDiffuse, STATS, MANY_LIGHTS, SHAPE>(P, rec, L, tab, nxt, cnt);
        wf_shade<kMetallic, STATS, MANY_LIGHTS, SHAPE>(P, rec, L, tab, nxt, cnt);
        wf_shade<kDielectric, STATS, MANY_LIGHTS, SHAPE>(P, rec, L, tab, nxt, cnt);
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

// the HBM-state shape decides the size of the state block (the LDS-state shape needs none)
static uint32_t wavefront_slots(uint32_t width, uint32_t height)
{
    const uint32_t tiles = ((width + 7u) / 8u) * ((height + 7u) / 8u);
    return (tiles + WfShape<false>::kTiles - 1u) / WfShape<false>::kTiles * WfShape<false>::kPixels;
}

size_t wavefront_state_bytes(uint32_t width, uint32_t height)
{
    return (size_t)wavefront_slots(width, height) * 4u * sizeof(float4);
}

void wavefront_bind_state(void *block, uint32_t, uint32_t, WfState &S)
{
    S.rec = static_cast<float4 *>(block);
}

// LDS a workgroup of the given shape needs: queues, records (LDS state), object table (kSrcLds)
template <bool LDS_STATE>
static size_t wavefront_lds_bytes(const RenderParams &P, int src)
{
    size_t lds = (sizeof(WfLds<WfShape<LDS_STATE>::kPixels>) + 15u) / 16u * 16u;
    if (LDS_STATE) lds += (size_t)WfShape<true>::kPixels * 4u * sizeof(float4);
    if (src == kSrcLds) lds += scene_lds_bytes(P.n_prims, P.n_lights);
    return lds;
}
// two workgroups per CU (160 KB of LDS on gfx950): while one waits at a barrier or runs out of
// paths near its end, the other keeps the SIMDs busy (one workgroup of 1024 threads and 2048
// pixels per CU measured 93.7 ms on the reference scene at 1080p, two of 512 / 1024: 88.2 ms)
constexpr size_t kMaxLdsPerWorkgroup = 80u * 1024u;

bool wavefront_state_in_lds(const RenderParams &, int src)
{
    return src != kSrcBvh;
}

template <int SRC, bool STATS, bool MANY_LIGHTS, bool LDS_STATE>
static hipError_t launch_variant(const RenderParams &P, const WfState &S, hipStream_t stream)
{
    using SHAPE = WfShape<LDS_STATE>;
    const uint32_t tiles = ((P.width + 7u) / 8u) * ((P.height + 7u) / 8u);
    const dim3 grid((tiles + SHAPE::kTiles - 1u) / SHAPE::kTiles);
    const size_t lds = wavefront_lds_bytes<LDS_STATE>(P, SRC);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wavefront<SRC, STATS, MANY_LIGHTS, LDS_STATE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_wavefront<SRC, STATS, MANY_LIGHTS, LDS_STATE>), grid, dim3(SHAPE::kThreads), lds, stream, P, S);
    return hipGetLastError();
}
template <int SRC, bool STATS, bool LDS_STATE>
static hipError_t launch_one(const RenderParams &P, const WfState &S, hipStream_t stream)
{
    return P.n_lights > kFewLights ? launch_variant<SRC, STATS, true, LDS_STATE>(P, S, stream)
                                   : launch_variant<SRC, STATS, false, LDS_STATE>(P, S, stream);
}
template <int SRC, bool LDS_STATE>
static hipError_t launch_src(const RenderParams &P, const WfState &S, bool stats, hipStream_t stream)
{
    return stats ? launch_one<SRC, true, LDS_STATE>(P, S, stream) : launch_one<SRC, false, LDS_STATE>(P, S, stream);
}

hipError_t launch_wavefront(const RenderParams &P, const WfState &S, int src, bool stats, hipStream_t stream)
{
    if (src == kSrcBvh) return launch_src<kSrcBvh, false>(P, S, stats, stream);
    if (src == kSrcLds && wavefront_lds_bytes<true>(P, kSrcLds) <= kMaxLdsPerWorkgroup)
        return launch_src<kSrcLds, true>(P, S, stats, stream);
    return launch_src<kSrcScalar, true>(P, S, stats, stream);
}

}  // namespace rtx
