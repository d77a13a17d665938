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
// pixel number, in LDS or in device memory; the shapes of a workgroup are listed at WfShapeT below.
// With the state in device memory and 512-thread workgroups the linear-scan kernel ran at the memory
// roof (about 400 bytes of traffic per path segment; the compacted queues hand the paths out in any
// order, so every stage re-reads and re-writes scattered records), while its instruction count is a
// fifth below the megakernel's: the linear scan keeps its records in LDS (an object table that does
// not fit beside them is read from device memory through scalar loads instead); BVH scenes do so
// when the traversal stacks leave room, else a whole cache line per record access (a
// structure-of-arrays layout measured 2 KB of traffic per path segment).
// Only the radiance sum (the output buffer) is indexed by pixel.  Queues hold 16-bit local
// numbers.  One iteration of the in-kernel loop advances every live path of the workgroup by
// one segment:
//   generate  fresh queue  -> primary ray of the pixel's next sample -> active queue
//             (BVH scenes; the linear scan has no such stage: its extend stage takes the fresh
//             queue behind the active one and makes those camera rays itself)
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

// Shape of a workgroup: threads, 8x8 tiles and pixels it owns (pixel numbers fit 16-bit queue
// entries), and where the path records live.  Three shapes are in use:
//   ShapeScan     512 threads, 16 tiles, records in LDS: the linear scan, two workgroups per CU;
//   ShapeBvhLds    64 threads,  2 tiles, records in LDS: BVH scenes with a shallow tree.  A
//                 workgroup is ONE wave: its barriers cost nothing, and no wave ever waits for
//                 another one's longest traversal;
//   ShapeBvhHbm    64 threads,  4 tiles, records in device memory: deep trees, whose traversal
//                 stacks leave no room for records in LDS.
// Measured on 10 000 objects (ms per frame of 64 spp) with the records in device memory:
// 512 threads x 64 tiles 418, 256 x 32 375, 256 x 16 287, 128 x 8 286, 64 x 8 379 (LDS limits
// the waves per CU), 64 x 4 267.
template <uint32_t THREADS, uint32_t TILES, bool LDS_STATE>
struct WfShapeT {
    static constexpr uint32_t kThreads = THREADS;
    static constexpr uint32_t kTiles = TILES;
    static constexpr bool kLdsState = LDS_STATE;
    static constexpr uint32_t kPixels = kTiles * 64u;
    static constexpr uint32_t kPlane = LDS_STATE ? kPixels : 0u;   // see wf_at
    // the start-up loop walks the pixels in steps of kThreads and must not run past them
    static_assert(kPixels % kThreads == 0 && kPixels <= 65536u, "a workgroup's pixels: a multiple of its threads, 16-bit numbers");
};
#ifndef RTX_WF_SCAN_THREADS
#define RTX_WF_SCAN_THREADS 512u
#endif
#ifndef RTX_WF_SCAN_TILES
#define RTX_WF_SCAN_TILES 16u
#endif
#ifndef RTX_WF_BVH_LDS_TILES
#define RTX_WF_BVH_LDS_TILES 2u
#endif
#ifndef RTX_WF_BVH_HBM_TILES
#define RTX_WF_BVH_HBM_TILES 4u
#endif
using ShapeScan = WfShapeT<RTX_WF_SCAN_THREADS, RTX_WF_SCAN_TILES, true>;
using ShapeBvhLds = WfShapeT<64u, RTX_WF_BVH_LDS_TILES, true>;
using ShapeBvhHbm = WfShapeT<64u, RTX_WF_BVH_HBM_TILES, false>;

enum : int { Q_ACTIVE0 = 0, Q_ACTIVE1, Q_FRESH0, Q_FRESH1, Q_DIFFUSE, Q_METALLIC, Q_DIELECTRIC, Q_COUNT };

template <uint32_t PIXELS>
struct WfLds {
    uint16_t queue[Q_COUNT][PIXELS];
    uint32_t count[Q_COUNT];
    uint32_t next;   // BVH extend stage: rays of the active queue handed out so far
    uint32_t mcount[2][3];   // linear scan: the material queues' counters, one set per iteration parity
};

extern __shared__ float4 rtx_smem[];

// Build with -DRTX_WF_DEBUG to count, per launch, the rounds of every stage and the lanes they
// served (rtx_wf_debug_read): how full the waves run where.
#ifdef RTX_WF_DEBUG
__device__ unsigned long long g_wf_dbg[16];
#define WF_DBG(i, n) dbg[i] += (n)
#else
#define WF_DBG(i, n) ((void)0)
#endif
enum { D_NODE_ROUNDS = 0, D_NODE_LANES, D_LEAF_ROUNDS, D_LEAF_LANES, D_SERVE_ROUNDS, D_SERVE_DONE, D_SERVE_FETCHED,
       D_GEN_ROUNDS, D_GEN_LANES, D_SHADE_ROUNDS, D_SHADE_LANES, D_ITERATIONS, D_COUNT };

// wave64 compaction into an LDS queue: lanes with `pred` append `value`
// (`counter`: the queue's fill count, L.count[q] unless the caller keeps it elsewhere)
template <uint32_t PIXELS>
RTX_DEV void wf_push(WfLds<PIXELS> &L, int q, bool pred, uint32_t value, uint32_t *counter = nullptr)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    if (counter == nullptr) counter = &L.count[q];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int leader = __builtin_ctzll(mask);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (pred) L.queue[q][base + below] = (uint16_t)value;
}

// workgroup-local pixel number -> pixel; tiles of a workgroup are consecutive in tile order
template <uint32_t TILES>
RTX_DEV bool wf_pixel(const RenderParams &P, uint32_t local, uint32_t &px, uint32_t &py)
{
    // the workgroup's unit: units are dealt out round-robin to the shares of a multi-GPU job
    const uint32_t tile = (P.share_index + blockIdx.x * P.share_count) * TILES + (local >> 6);
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
template <int MATERIAL, bool STATS, bool MANY_LIGHTS, class SHAPE, bool MIXED>
RTX_DEV void wf_shade(const RenderParams &P, float4 *rec, WfLds<SHAPE::kPixels> &L, const Table &tab, uint32_t nxt,
                      uint32_t count, Counters<STATS> &cnt, uint32_t *dbg)
{
    for (uint32_t base = 0; base < count; base += SHAPE::kThreads) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < count;
        WF_DBG(D_SHADE_ROUNDS, 1u);
        WF_DBG(D_SHADE_LANES, (uint32_t)__popcll(__ballot(valid)));
        uint32_t local = 0;
        bool go_on = false, refresh = false;
        if (valid) {
            local = L.queue[Q_DIFFUSE + MATERIAL][i];
            uint32_t px, py;
            wf_pixel<SHAPE::kTiles>(P, local, px, py);
            const uint32_t pixel = py * P.width + px;
            WfPath w = wf_load<SHAPE::kPlane>(rec, local);
            const bool ended = bounce<STATS, MANY_LIGHTS, MIXED>(P, tab, w.p, w.hit_idx, w.hit_t, MATERIAL, pixel, w.sample, cnt);
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

// Rough costs, in instructions, of the three things a wave of the BVH extend stage can do next;
// only their ratios matter (they weigh how many lanes each would serve).
#ifndef RTX_WF_COST_NODE
#define RTX_WF_COST_NODE 1u      // a batch of node steps
#endif
#ifndef RTX_WF_COST_LEAF
#define RTX_WF_COST_LEAF 2u      // one object test
#endif
#ifndef RTX_WF_COST_SERVE
#define RTX_WF_COST_SERVE 8u     // write back finished rays, fetch new ones, test the plane list
#endif
#ifndef RTX_WF_NODE_BATCH
#define RTX_WF_NODE_BATCH 3
#endif
// The extend stage (src/trace.rs:12-23) for scenes with a BVH.  A traversal visits anything from
// a handful to several hundred nodes, so dealing the active queue out in fixed portions leaves
// most lanes of a wave waiting for its longest ray (measured on 10 000 objects: 46 node visits
// per ray on average, 219 loop iterations per wave).  Here every lane fetches its next ray from
// the queue when its current one is done (the waves of the workgroup share one counter in LDS),
// and in every round the wave does what serves most lanes per instruction spent:
//   * node steps for the lanes at an inner node,
//   * the (kind-dependent, expensive) object test for the lanes holding a leaf,
//   * service: finished rays are written back and queued by material, free lanes get their next
//     ray, which starts with the always-tested list (planes).
// Which lane traces which ray changes nothing: a ray's result goes to its own record.
template <bool STATS, class SHAPE>
RTX_DEV void wf_extend_bvh(const RenderParams &P, float4 *rec, WfLds<SHAPE::kPixels> &L, const Table &tab, uint32_t cur_q,
                           uint32_t nxt, int32_t *stacks, Counters<STATS> &cnt, uint32_t *dbg)
{
    constexpr int BLOCK = (int)SHAPE::kThreads;
    constexpr uint32_t PLANE = SHAPE::kPlane;
    constexpr uint32_t kNoRay = 0xffffffffu;
    // one column per lane, P.bvh.stack_entries deep (the tree's depth + 1), in dynamic LDS
    int32_t *stack = stacks + threadIdx.x;
    const BvhView &B = P.bvh;
    const uint32_t count = L.count[Q_ACTIVE0 + cur_q];

    uint32_t local = kNoRay;      // the lane's ray (workgroup-local pixel number) or none
    int32_t cur = kBvhEmpty;      // traversal position; kBvhEmpty with a ray = finished, to be written back
    int sp = 0, best = -1;
    float best_t = kInf, ray_length = 0.0f;
    V3 o = v3(0.0f, 0.0f, 0.0f), d = o;
    BvhRay R;
    R.inv = o;
    R.ox = R.oy = R.oz = F2{0.0f, 0.0f};
    R.far = false;
    bool plain = true;
    bool exhausted = false;       // wave-uniform: the queue has been handed out completely
    for (;;) {
        const bool at_leaf = cur < 0 && cur != kBvhEmpty;
        const uint32_t n_node = (uint32_t)__popcll(__ballot(cur >= 0));
        const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
        // lanes a service round would serve: finished rays, and free lanes while rays are left
        const uint32_t n_done = (uint32_t)__popcll(__ballot(local != kNoRay && cur == kBvhEmpty));
        const uint32_t n_serve = exhausted ? n_done : 64u - n_node - n_leaf;
        if (n_node + n_leaf + n_serve == 0u) break;
        // lanes served per unit of cost, compared by cross-multiplication
        const uint32_t w_node = n_node * (RTX_WF_COST_LEAF * RTX_WF_COST_SERVE);
        const uint32_t w_leaf = n_leaf * (RTX_WF_COST_NODE * RTX_WF_COST_SERVE);
        const uint32_t w_serve = n_serve * (RTX_WF_COST_NODE * RTX_WF_COST_LEAF);
        if (w_node >= w_leaf && w_node >= w_serve && n_node != 0u) {
#pragma unroll
            for (int k = 0; k < RTX_WF_NODE_BATCH; ++k) {
                WF_DBG(D_NODE_ROUNDS, 1u);
                WF_DBG(D_NODE_LANES, (uint32_t)__popcll(__ballot(cur >= 0)));
                if (cur >= 0) bvh_node_step<STATS, BLOCK>(B, R, best_t, cur, sp, stack, cnt);
            }
        } else if (w_leaf >= w_serve && n_leaf != 0u) {
            const bool leaves_plain = __ballot(at_leaf && !plain) == 0ull;
            WF_DBG(D_LEAF_ROUNDS, 1u);
            WF_DBG(D_LEAF_LANES, n_leaf);
            if (at_leaf) {
                bvh_try_object<STATS>(tab, ~cur, o, d, ray_length, best, best_t, cnt, leaves_plain);
                if (sp > 0) {
                    sp--;
                    cur = stack[sp * BLOCK];
                } else {
                    cur = kBvhEmpty;
                }
            }
        } else {
            // ---- write back the finished rays
            WF_DBG(D_SERVE_ROUNDS, 1u);
            WF_DBG(D_SERVE_DONE, n_done);
            int material = -1;
            bool refresh = false;
            const uint32_t mine = local;
            if (local != kNoRay && cur == kBvhEmpty) {
                uint32_t px, py;
                wf_pixel<SHAPE::kTiles>(P, local, px, py);
                const uint32_t pixel = py * P.width + px;
                const float4 c = rec[wf_at<PLANE>(local, 2)], e = rec[wf_at<PLANE>(local, 3)];
                Path p;
                p.o = o;
                p.d = d;
                p.T = v3(c.x, c.y, c.z);
                p.depth = __float_as_uint(c.w);
                p.L = v3(e.x, e.y, e.z);
                const uint32_t sample = __float_as_uint(e.w);
                cnt.template add<C_RAYS>();
                bool done;
                if (best < 0) {
                    p.L = p.L + cmul(p.T, background(P, tab));
                    done = true;
                } else {
                    if constexpr (STATS) {
                        int kind = __float_as_int(tab.geom[3 * best + 1].w) & 0xff;
                        cnt.template add<C_HIT_PLANE>(kind == kPlane);
                        cnt.template add<C_HIT_ELLIPSOID>(kind == kEllipsoid);
                        cnt.template add<C_HIT_BOX>(kind == kBox);
                    }
                    done = surface_emission(P, tab, p, best) || black_surface_ends(P, tab, p, best);
                }
                if (done) {
                    refresh = wf_finish_path<PLANE>(P, rec, local, pixel, p.L, sample);
                } else {
                    wf_store_rad<PLANE>(rec, local, p.L, sample);
                    wf_store_ray<PLANE>(rec, local, p, best_t, best);
                    material = __float_as_int(tab.mat[3 * best + 0].w);
                }
                local = kNoRay;
            }
            wf_push(L, Q_FRESH0 + nxt, refresh, mine);
            wf_push(L, Q_DIFFUSE, material == kDiffuse, mine);
            wf_push(L, Q_METALLIC, material == kMetallic, mine);
            wf_push(L, Q_DIELECTRIC, material == kDielectric, mine);
            // ---- hand the free lanes their next rays
            if (!exhausted) {
                const bool want = local == kNoRay;
                const unsigned long long mask = __ballot(want);   // not empty: this round serves at least one lane
                const uint32_t n = (uint32_t)__popcll(mask);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                const int leader = __builtin_ctzll(mask);
                uint32_t base = 0;
                if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(&L.next, n);
                base = (uint32_t)__shfl((int)base, leader, 64);
                exhausted = base + n >= count;
                const bool fresh = want && base + rank < count;
                WF_DBG(D_SERVE_FETCHED, (uint32_t)__popcll(__ballot(fresh)));
                bool fresh_plain = true;
                if (fresh) {
                    local = L.queue[Q_ACTIVE0 + cur_q][base + rank];
                    const float4 a = rec[wf_at<PLANE>(local, 0)], b = rec[wf_at<PLANE>(local, 1)];
                    o = v3(a.x, a.y, a.z);
                    d = v3(b.x, b.y, b.z);
                    fresh_plain = ray_is_plain(o, d);
                }
                // the shortcuts of the always-tested list need every ray of this batch to be plain
                const bool all_plain = __ballot(!fresh_plain) == 0ull;
                if (fresh) {
                    plain = fresh_plain;
                    ray_length = len(d);
                    best = -1;
                    best_t = kInf;
                    bvh_always_tested<STATS>(B, tab, o, d, ray_length, all_plain, best, best_t, cnt);
                    R = bvh_ray_setup(bvh_grid(B, tab), o, d);
                    sp = 0;
                    cur = B.n_nodes ? 0 : kBvhEmpty;
                }
            }
        }
    }
}

template <int SRC, bool STATS, bool MANY_LIGHTS, class SHAPE>
__global__ void __launch_bounds__(SHAPE::kThreads) k_wavefront(const RenderParams P, const WfState S)
{
    constexpr bool LDS_STATE = SHAPE::kLdsState;
    constexpr uint32_t kWfThreads = SHAPE::kThreads, kWfPixelsPerGroup = SHAPE::kPixels;
    // dynamic LDS: [queues + counters][records when LDS_STATE][object table when SRC == kSrcLds | traversal stacks when SRC == kSrcBvh]
    WfLds<kWfPixelsPerGroup> &L = *reinterpret_cast<WfLds<kWfPixelsPerGroup> *>(rtx_smem);
    constexpr uint32_t kQueueVec = (sizeof(WfLds<kWfPixelsPerGroup>) + 15u) / 16u;
    constexpr uint32_t kRecVec = LDS_STATE ? 4u * kWfPixelsPerGroup : 0u;
    float4 *rec;
    if constexpr (LDS_STATE) rec = rtx_smem + kQueueVec;
    else rec = S.rec + 4ull * blockIdx.x * kWfPixelsPerGroup;
    Table tab;
    {
        // the uniforms (rtx_trace.hpp: stage_uniforms) behind the object table or the traversal stacks
        uint32_t before = (kQueueVec + kRecVec) * (uint32_t)sizeof(float4);
        if constexpr (SRC == kSrcLds) before += table_only_lds_bytes(P.n_prims, P.n_lights);
        if constexpr (SRC == kSrcBvh) before += P.bvh.stack_entries * SHAPE::kThreads * (uint32_t)sizeof(int32_t);
        float *u = reinterpret_cast<float *>(reinterpret_cast<char *>(rtx_smem) + before);
        stage_uniforms(P, u);   // the barrier below covers it
        tab.uniforms = u;
        tab.has_uniforms = true;
        tab.keys_per_call = false;
    }
    tab.n_prims = P.n_prims;
    tab.n_lights = P.n_lights;
    tab.ellipsoids_pow2 = P.ellipsoids_pow2 != 0u;
    tab.light_bounds = P.light_bounds;
    tab.order = P.scan_order;
    tab.in_lds = SRC == kSrcLds;
    if constexpr (SRC == kSrcLds) {
        float4 *tbl = rtx_smem + kQueueVec + kRecVec;
        const uint32_t ng = 3u * P.n_prims, nm = 3u * P.n_prims;
        for (uint32_t i = threadIdx.x; i < ng; i += blockDim.x) tbl[i] = P.geom[i];
        for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) tbl[ng + i] = P.mat[i];
        int32_t *so = reinterpret_cast<int32_t *>(tbl + ng + nm);
        const uint32_t no = kScanHeader + P.n_prims + 2u;
        for (uint32_t i = threadIdx.x; i < no; i += blockDim.x) so[i] = P.scan_order[i];
        int32_t *sl = so + no;
        for (uint32_t i = threadIdx.x; i < P.n_lights; i += blockDim.x) sl[i] = P.lights[i];
        tab.geom = tbl;
        tab.mat = tbl + ng;
        tab.lights = sl;
        tab.order = so;
    } else {
        tab.geom = P.geom;
        tab.mat = P.mat;
        tab.lights = P.lights;
    }
    if (threadIdx.x < Q_COUNT) L.count[threadIdx.x] = 0u;
    if (threadIdx.x < 6u) L.mcount[threadIdx.x / 3u][threadIdx.x % 3u] = 0u;
    if (threadIdx.x == 0) L.next = 0u;
    __syncthreads();

    Counters<STATS> cnt;
    cnt.clear();
    uint32_t dbg[D_COUNT] = {};
    (void)dbg;

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
        // Barriers: one after each stage.  Queue counters are reset by thread 0 behind the barrier
        // that ends their last use and ahead of a barrier that precedes their next one: the two
        // queues this iteration consumes (they are the next iteration's targets) after the extend
        // stage; the material queues and the fetch counter after the shade stage (BVH scenes), or,
        // where extend follows shade without a barrier (linear scan), one set of material counters
        // per iteration parity, reset while the other set is in use.
        const uint32_t cur = it & 1u, nxt = cur ^ 1u;
        // Linear scan: no generate stage.  The extend stage takes the fresh queue behind the active
        // one and generates those paths' camera rays itself: the ray stays in registers, the waves
        // that do it are full but for one, and an iteration has two barriers instead of three.
        constexpr bool kMerged = SRC != kSrcBvh;
        uint32_t *const mcount = L.mcount[cur];

        // ---- generate: src/main.rs:17-21
        if constexpr (!kMerged) {
            const uint32_t count = L.count[Q_FRESH0 + cur];
            for (uint32_t base = 0; base < count; base += kWfThreads) {
                const uint32_t i = base + threadIdx.x;
                const bool valid = i < count;
                WF_DBG(D_GEN_ROUNDS, 1u);
                WF_DBG(D_GEN_LANES, (uint32_t)__popcll(__ballot(valid)));
                uint32_t local = 0;
                if (valid) {
                    local = L.queue[Q_FRESH0 + cur][i];
                    uint32_t px, py;
                    wf_pixel<SHAPE::kTiles>(P, local, px, py);
                    const uint32_t pixel = py * P.width + px;
                    // the finished path left (0, 0, 0 | next sample index) in the radiance record
                    const uint32_t sample = __float_as_uint(rec[wf_at<SHAPE::kPlane>(local, 3)].w);
                    uint4 w = draw_block(P, tab, pixel, sample, 0u);
                    Path p;
                    camera_ray(P, tab, px, P.height - 1u - py, u24(w.x), u24(w.y), p.o, p.d);
                    p.T = v3(1.0f, 1.0f, 1.0f);
                    p.depth = 0u;
                    wf_store_ray<SHAPE::kPlane>(rec, local, p, 0.0f, -1);
                    wf_store_thr<SHAPE::kPlane>(rec, local, p);
                    cnt.template add<C_SAMPLES>();
                }
                wf_push(L, Q_ACTIVE0 + cur, valid, local);
            }
            __syncthreads();
        }

        // ---- extend: src/trace.rs:12-23
        if constexpr (SRC == kSrcBvh) {
            wf_extend_bvh<STATS, SHAPE>(P, rec, L, tab, cur, nxt, reinterpret_cast<int32_t *>(rtx_smem + kQueueVec + kRecVec), cnt, dbg);
        } else {
            const uint32_t n_active = L.count[Q_ACTIVE0 + cur], count = n_active + L.count[Q_FRESH0 + cur];
            for (uint32_t base = 0; base < count; base += kWfThreads) {
                const uint32_t i = base + threadIdx.x;
                const bool valid = i < count;
                const bool fresh = valid && i >= n_active;
                uint32_t local = 0;
                int material = -1;
                bool refresh = false;
                WfPath w;
                w.p.o = w.p.d = w.p.T = w.p.L = v3(0.0f, 0.0f, 0.0f);
                w.p.depth = 0u;
                w.hit_t = 0.0f;
                w.hit_idx = -1;
                w.sample = 0u;
                uint32_t px = 0, py = 0;
                if (valid) {
                    local = fresh ? L.queue[Q_FRESH0 + cur][i - n_active] : L.queue[Q_ACTIVE0 + cur][i];
                    wf_pixel<SHAPE::kTiles>(P, local, px, py);
                }
                if (__ballot(fresh) != 0ull) {
                    WF_DBG(D_GEN_ROUNDS, 1u);
                    WF_DBG(D_GEN_LANES, (uint32_t)__popcll(__ballot(fresh)));
                    if (fresh) {
                        // src/main.rs:17-21; the finished path left (0, 0, 0 | next sample index) in the radiance record
                        w.sample = __float_as_uint(rec[wf_at<SHAPE::kPlane>(local, 3)].w);
                        uint4 r = draw_block(P, tab, py * P.width + px, w.sample, 0u);
                        camera_ray(P, tab, px, P.height - 1u - py, u24(r.x), u24(r.y), w.p.o, w.p.d);
                        w.p.T = v3(1.0f, 1.0f, 1.0f);
                        cnt.template add<C_SAMPLES>();
                    }
                }
                if (valid) {
                    const uint32_t pixel = py * P.width + px;
                    if (!fresh) w = wf_load<SHAPE::kPlane>(rec, local);
                    Path &p = w.p;
                    cnt.template add<C_RAYS>();
                    float t;
                    int idx;
                    idx = closest_hit_scan<true>(tab, p.o, p.d, t);
                    bool done;
                    if (idx < 0) {
                        p.L = p.L + cmul(p.T, background(P, tab));
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
                        if (fresh) wf_store_thr<SHAPE::kPlane>(rec, local, p);
                        material = __float_as_int(tab.mat[3 * idx + 0].w);
                    }
                }
                wf_push(L, Q_FRESH0 + nxt, refresh, local);
                wf_push(L, Q_DIFFUSE, material == kDiffuse, local, &mcount[kDiffuse]);
                wf_push(L, Q_METALLIC, material == kMetallic, local, &mcount[kMetallic]);
                wf_push(L, Q_DIELECTRIC, material == kDielectric, local, &mcount[kDielectric]);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            L.count[Q_ACTIVE0 + cur] = 0u;
            L.count[Q_FRESH0 + cur] = 0u;
            if constexpr (kMerged) L.mcount[nxt][0] = L.mcount[nxt][1] = L.mcount[nxt][2] = 0u;
        }

        // ---- shade, one material at a time
        wf_shade<kDiffuse, STATS, MANY_LIGHTS, SHAPE, SRC == kSrcBvh>(P, rec, L, tab, nxt, kMerged ? mcount[kDiffuse] : L.count[Q_DIFFUSE], cnt, dbg);
        wf_shade<kMetallic, STATS, MANY_LIGHTS, SHAPE, SRC == kSrcBvh>(P, rec, L, tab, nxt, kMerged ? mcount[kMetallic] : L.count[Q_METALLIC], cnt, dbg);
        wf_shade<kDielectric, STATS, MANY_LIGHTS, SHAPE, SRC == kSrcBvh>(P, rec, L, tab, nxt, kMerged ? mcount[kDielectric] : L.count[Q_DIELECTRIC], cnt, dbg);
        __syncthreads();

        WF_DBG(D_ITERATIONS, 1u);
        if (L.count[Q_ACTIVE0 + nxt] + L.count[Q_FRESH0 + nxt] == 0u) break;
        if (!kMerged && threadIdx.x == 0) {
            L.count[Q_DIFFUSE] = 0u;
            L.count[Q_METALLIC] = 0u;
            L.count[Q_DIELECTRIC] = 0u;
            L.next = 0u;
        }
    }

    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < C_COUNT; ++i)
            if (cnt.v[i]) atomicAdd(&P.stats[i], (unsigned long long)cnt.v[i]);
    }
#ifdef RTX_WF_DEBUG
    if ((threadIdx.x & 63u) == 0u)
        for (int i = 0; i < D_COUNT; ++i) atomicAdd(&g_wf_dbg[i], (unsigned long long)dbg[i]);
#endif
}

}  // namespace

#ifdef RTX_WF_DEBUG
extern "C" int rtx_wf_debug_read(unsigned long long *out, int reset)
{
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wf_dbg), sizeof(zero)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_wf_dbg), zero, sizeof(zero)) != hipSuccess) return -1;
    return D_COUNT;
}
#endif

// the HBM-state shape decides the size of the state block (the LDS-state shapes need none)
static uint32_t wavefront_slots(uint32_t width, uint32_t height)
{
    const uint32_t tiles = ((width + 7u) / 8u) * ((height + 7u) / 8u);
    return (tiles + ShapeBvhHbm::kTiles - 1u) / ShapeBvhHbm::kTiles * ShapeBvhHbm::kPixels;
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
// or traversal stacks (kSrcBvh)
template <class SHAPE>
static size_t wavefront_lds_bytes(const RenderParams &P, int src)
{
    size_t lds = (sizeof(WfLds<SHAPE::kPixels>) + 15u) / 16u * 16u;
    if (SHAPE::kLdsState) lds += (size_t)SHAPE::kPixels * 4u * sizeof(float4);
    if (src == kSrcLds) lds += scene_lds_bytes(P.n_prims, P.n_lights);   // the uniforms included
    if (src == kSrcBvh) lds += (size_t)P.bvh.stack_entries * SHAPE::kThreads * sizeof(int32_t);
    if (src != kSrcLds) lds += kUniformFloats * sizeof(float);
    return lds;
}
// Linear scan: two workgroups per CU (160 KB of LDS on gfx950): while one waits at a barrier or
// runs out of paths near its end, the other keeps the SIMDs busy (one workgroup of 1024 threads
// and 2048 pixels per CU measured 93.7 ms on the reference scene at 1080p, two of 512 / 1024: 88.2 ms)
constexpr size_t kMaxLdsScan = 80u * 1024u;
// BVH, records in LDS: one-wave workgroups, at least 12 of them per CU (3 waves per SIMD)
#ifndef RTX_WF_BVH_LDS_MAX
#define RTX_WF_BVH_LDS_MAX (160u * 1024u / 12u)
#endif

bool wavefront_state_in_lds(const RenderParams &P, int src)
{
    return src != kSrcBvh || wavefront_lds_bytes<ShapeBvhLds>(P, kSrcBvh) <= RTX_WF_BVH_LDS_MAX;
}

template <int SRC, bool STATS, bool MANY_LIGHTS, class SHAPE>
static hipError_t launch_variant(const RenderParams &P, const WfState &S, hipStream_t stream)
{
    const uint32_t tiles = ((P.width + 7u) / 8u) * ((P.height + 7u) / 8u);
    const uint32_t groups = share_units((tiles + SHAPE::kTiles - 1u) / SHAPE::kTiles, P.share_index, P.share_count);
    if (groups == 0u) return hipSuccess;
    const dim3 grid(groups);
    const size_t lds = wavefront_lds_bytes<SHAPE>(P, SRC);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wavefront<SRC, STATS, MANY_LIGHTS, SHAPE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_wavefront<SRC, STATS, MANY_LIGHTS, SHAPE>), grid, dim3(SHAPE::kThreads), lds, stream, P, S);
    return hipGetLastError();
}
template <int SRC, bool STATS, class SHAPE>
static hipError_t launch_one(const RenderParams &P, const WfState &S, hipStream_t stream)
{
    return P.n_lights > kFewLights ? launch_variant<SRC, STATS, true, SHAPE>(P, S, stream)
                                   : launch_variant<SRC, STATS, false, SHAPE>(P, S, stream);
}
template <int SRC, class SHAPE>
static hipError_t launch_src(const RenderParams &P, const WfState &S, bool stats, hipStream_t stream)
{
    return stats ? launch_one<SRC, true, SHAPE>(P, S, stream) : launch_one<SRC, false, SHAPE>(P, S, stream);
}

uint32_t wavefront_group_tiles(const RenderParams &P, int src)
{
    if (src == kSrcBvh) return wavefront_state_in_lds(P, kSrcBvh) ? ShapeBvhLds::kTiles : ShapeBvhHbm::kTiles;
    return ShapeScan::kTiles;
}

hipError_t launch_wavefront(const RenderParams &P, const WfState &S, int src, bool stats, hipStream_t stream)
{
    if (src == kSrcBvh) {
        if (wavefront_state_in_lds(P, kSrcBvh)) return launch_src<kSrcBvh, ShapeBvhLds>(P, S, stats, stream);
        return launch_src<kSrcBvh, ShapeBvhHbm>(P, S, stats, stream);
    }
    if (src == kSrcLds && wavefront_lds_bytes<ShapeScan>(P, kSrcLds) <= kMaxLdsScan)
        return launch_src<kSrcLds, ShapeScan>(P, S, stats, stream);
    return launch_src<kSrcScalar, ShapeScan>(P, S, stats, stream);
}

}  // namespace rtx
