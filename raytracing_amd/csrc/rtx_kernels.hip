// rtx_kernels.hip -- megakernel path tracers and the deterministic primary pass (gfx950).
//
// Work decomposition (replaces the serial loop nest of src/main.rs:13-31):
//   * one lane per pixel; a wave64 owns an 8x8 pixel tile, a 256-thread workgroup four tiles
//     (k_megakernel); for small all-diffuse scenes with black lights a lane owns a pixel of each
//     of two tiles, one of them parked in LDS (k_megakernel_scan2);
//   * the sample loop (main.rs:14) and the bounce recursion (trace.rs:11-71, one child per
//     call, so a loop) are flattened into ONE loop per lane whose round is a path segment:
//     closest hit, then scatter.  A lane whose path ends starts its next sample in the same
//     round, so the closest-hit scan -- a third of the work -- runs with a full exec mask; in
//     BVH scenes a lane's traversal can pause while other lanes scatter and go on in the next
//     round (k_megakernel, kSrcBvh);
//   * the object index of the scan is wave-uniform: table reads are LDS broadcasts
//     (kSrcLds: table staged once per workgroup) or scalar loads (kSrcScalar);
//   * each lane keeps its pixel's radiance sum in registers and stores 12 bytes once.
#include "rtx_trace.hpp"
#include "rtx_bvh_traverse.hpp"

#include <algorithm>
#include <cstdlib>

namespace rtx {

extern __shared__ float4 rtx_smem[];

size_t scene_lds_bytes(uint32_t n_prims, uint32_t n_lights) { return table_lds_bytes(n_prims, n_lights); }
// dynamic LDS every kernel of this file starts with: the object table or the traversal stacks
template <int SRC>
__host__ __device__ inline uint32_t fixed_lds_bytes(const RenderParams &P)
{
    if (SRC == kSrcLds) return table_lds_bytes(P.n_prims, P.n_lights);
    if (SRC == kSrcBvh) return P.bvh.stack_entries * 256u * (uint32_t)sizeof(int32_t) + kUniformFloats * (uint32_t)sizeof(float);
    return kUniformFloats * (uint32_t)sizeof(float);
}
static uint32_t fixed_lds_bytes(const RenderParams &P, int src)
{
    return src == kSrcLds ? fixed_lds_bytes<kSrcLds>(P) : src == kSrcBvh ? fixed_lds_bytes<kSrcBvh>(P) : fixed_lds_bytes<kSrcScalar>(P);
}

// copies the object table into LDS; every thread of the workgroup must call it
RTX_DEV Table stage_table_lds(const RenderParams &P)
{
    const uint32_t ng = 3u * P.n_prims, nm = 3u * P.n_prims;
    for (uint32_t i = threadIdx.x; i < ng; i += blockDim.x) rtx_smem[i] = P.geom[i];
    for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) rtx_smem[ng + i] = P.mat[i];
    int32_t *so = reinterpret_cast<int32_t *>(rtx_smem + ng + nm);
    const uint32_t no = kScanHeader + P.n_prims + 2u;
    for (uint32_t i = threadIdx.x; i < no; i += blockDim.x) so[i] = P.scan_order[i];
    int32_t *sl = so + no;
    for (uint32_t i = threadIdx.x; i < P.n_lights; i += blockDim.x) sl[i] = P.lights[i];
    stage_uniforms(P, reinterpret_cast<float *>(reinterpret_cast<char *>(rtx_smem) + table_only_lds_bytes(P.n_prims, P.n_lights)));
    __syncthreads();
    Table t;
    t.geom = rtx_smem;
    t.mat = rtx_smem + ng;
    t.lights = sl;
    t.order = so;
    t.uniforms = reinterpret_cast<const float *>(reinterpret_cast<const char *>(rtx_smem) + table_only_lds_bytes(P.n_prims, P.n_lights));
    t.has_uniforms = true;
    t.keys_per_call = true;
    t.in_lds = true;
    t.light_bounds = P.light_bounds;
    t.n_prims = P.n_prims;
    t.n_lights = P.n_lights;
    t.ellipsoids_pow2 = P.ellipsoids_pow2 != 0u;
    return t;
}
// STAGE: the caller is a megakernel, whose dynamic LDS has room for the uniforms behind the traversal stacks
// (fixed_lds_bytes); every thread of the workgroup must call it then
template <int SRC, bool STAGE>
RTX_DEV Table global_table(const RenderParams &P)
{
    Table t;
    t.uniforms = nullptr;
    t.has_uniforms = STAGE;
    t.keys_per_call = STAGE;
    if constexpr (STAGE) {
        float *u = reinterpret_cast<float *>(reinterpret_cast<char *>(rtx_smem) + fixed_lds_bytes<SRC>(P)) - kUniformFloats;
        stage_uniforms(P, u);
        __syncthreads();
        t.uniforms = u;
    }
    t.geom = P.geom;
    t.mat = P.mat;
    t.lights = P.lights;
    t.order = P.scan_order;
    t.in_lds = false;
    t.light_bounds = P.light_bounds;
    t.n_prims = P.n_prims;
    t.n_lights = P.n_lights;
    t.ellipsoids_pow2 = P.ellipsoids_pow2 != 0u;
    return t;
}

// wave -> 8x8 tile, lane -> pixel inside it; rows counted from the top of the image
RTX_DEV bool lane_pixel(const RenderParams &P, uint32_t &px, uint32_t &py, uint32_t &tile)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles_x = (P.width + 7u) >> 3;
    tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (P.tile_order) {
        if (tile >= tiles_x * ((P.height + 7u) >> 3)) return false;
        tile = P.tile_order[tile];
    }
    px = (tile % tiles_x) * 8u + (lane & 7u);
    py = (tile / tiles_x) * 8u + (lane >> 3);
    return px < P.width && py < P.height;
}
RTX_DEV bool lane_pixel(const RenderParams &P, uint32_t &px, uint32_t &py)
{
    uint32_t tile;
    return lane_pixel(P, px, py, tile);
}

template <int SRC, bool STATS>
RTX_DEV int closest_hit(const RenderParams &P, const Table &tab, V3 o, V3 d, float &t, Counters<STATS> &cnt)
{
    // BVH kernels of this file: the traversal stacks are the first part of the dynamic LDS
    if constexpr (SRC == kSrcBvh)
        return closest_hit_bvh<STATS, 256, true>(P.bvh, tab, o, d, t, cnt, reinterpret_cast<int32_t *>(rtx_smem));
    else return closest_hit_scan<true>(tab, o, d, t);
}

// BVH scenes: the wave leaves its traversal loop when fewer than BvhView::exit_sixteenths / 16 of its active
// lanes are still traversing; an object test is worth LEAF_COST node steps (default: BvhView::leaf_cost); both
// are chosen by the size of the tree at upload; node steps come in batches.
#ifndef RTX_MK_LEAF_COST
#define RTX_MK_LEAF_COST P.bvh.leaf_cost
#endif
#ifndef RTX_MK_NODE_BATCH
#define RTX_MK_NODE_BATCH 2
#endif
// The diffuse scatter of a hit is postponed until NUM / 64 of the wave's active lanes wait for one
// (0: the scatter is one call for all materials, as soon as the hit is known), separately for the
// linear scan and the BVH kernels.
#ifndef RTX_MK_DIFFUSE_SCAN
#define RTX_MK_DIFFUSE_SCAN 1u
#endif
#ifndef RTX_MK_DIFFUSE_BVH
#define RTX_MK_DIFFUSE_BVH 0u
#endif
// 1: linear-scan scenes run k_megakernel_scan2 (two pixels per lane), 0: k_megakernel
#ifndef RTX_MK_TWO
#define RTX_MK_TWO 1
#endif
#ifndef RTX_MEGAKERNEL_ATTR
#define RTX_MEGAKERNEL_ATTR __attribute__((amdgpu_waves_per_eu(4)))  // at most 128 VGPRs: 4 waves per SIMD measured fastest for the scan and the BVH kernels alike
#endif
// Build with -DRTX_MK_DEBUG to count, per launch, the rounds of the megakernels and the lanes each
// step served (rtx_mk_debug_read; tools/mk_debug.py).
#ifdef RTX_MK_DEBUG
__device__ unsigned long long g_mk_dbg[16];
#define MK_DBG(i, n) dbg[i] += (n)
#else
#define MK_DBG(i, n) ((void)0)
#endif
enum { M_ROUNDS = 0, M_SCAN_ROUNDS, M_SCAN_LANES, M_SCATTER_ROUNDS, M_SCATTER_LANES, M_SWAP_ROUNDS, M_SWAP_LANES, M_FRESH_LANES,
       M_NODE_ROUNDS, M_NODE_LANES, M_LEAF_ROUNDS, M_LEAF_LANES, M_START_ROUNDS, M_START_LANES, M_COUNT };
// largest value of `v` over the wave's 64 lanes (every lane must be active)
RTX_DEV uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m, 64));
    return v;
}

// A wave owns the 64 pixels of a tile and every lane the samples of one of them, added to the
// pixel's sum in index order.  Lanes do not finish together (path lengths are random: over 256
// samples the slowest of 64 lanes traces ~8 % more rays than the mean), so a lane that has run
// out of samples takes over the *last* unstarted sample of the lane that has the most of them
// left, and leaves its radiance in an entry of the wave's pool in LDS (P.pool_entries float4 per
// wave: x, y, z, and in w the link to the entry taken from the same owner before it).  The
// owner walks its list after its own samples -- the entry taken last holds the lowest sample
// index -- so the pixel's sum sees the same additions in the same order as without the hand-over.
// RECORD: the launch also leaves in P.tile_cost[tile] (zero before the launch) the largest number
// of rays one lane of the tile's wave traced -- what the wave's running time follows.
constexpr uint32_t kNoEntry = 0x3ffu;   // end of a list; pool entries are numbered below it

// A wave's work item when the launch has a work queue (RenderParams::queue): one wave per item, every wave
// drawing a ticket.  The first n_whole units of the order are items of their own (all samples); the
// others are cut into chunks of the sample range, and their tickets go chunk by chunk -- all those units'
// first chunks before all second ones -- a unit's chunks continuing its pixels' sums one after the other,
// so a chunk waits for the one before it.  Returns false for a wave beyond the last item.
// the i-th unit of the launch: by the order table, else the i-th of this share's units (RenderParams)
RTX_DEV uint32_t launch_unit(const RenderParams &P, uint32_t i)
{
    return P.tile_order ? P.tile_order[i] : P.share_index + i * P.share_count;
}
RTX_DEV bool queue_take_item(const RenderParams &P, uint32_t slot, uint32_t &unit, uint32_t &chunk,
                             uint32_t &sample_first, uint32_t &sample_count, bool &accumulate)
{
    const uint32_t n_tail = P.n_units - P.n_whole;
    if (slot >= P.n_whole + n_tail * P.n_chunks) return false;
    uint32_t ticket = 0;
    if ((threadIdx.x & 63u) == 0u) ticket = atomicAdd(P.queue, 1u);
    ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
    if (ticket < P.n_whole) {
        // the first units of the order are items of their own, all samples
        chunk = 0;
        unit = launch_unit(P, ticket);
        sample_first = P.sample_first;
        sample_count = P.sample_count;
        return true;
    }
    chunk = (ticket - P.n_whole) / n_tail;
    unit = launch_unit(P, P.n_whole + (ticket - P.n_whole) - chunk * n_tail);
    // chunk k of n samples: [n - (n >> k), n - (n >> (k + 1))), the last one up to n
    const uint32_t lo = P.sample_count - (P.sample_count >> chunk);
    const uint32_t hi = chunk + 1u == P.n_chunks ? P.sample_count : P.sample_count - (P.sample_count >> (chunk + 1u));
    sample_first = P.sample_first + lo;
    sample_count = hi - lo;
    if (chunk != 0u) {
        // The unit's previous chunk has a lower ticket, so a wave that is running (or through) holds it:
        // wait for its sums.  (Bounded, so that a mistake here ends in wrong pixels, not in a hung GPU.)
        accumulate = true;
        const uint32_t *done = P.queue + 1u + unit;
        for (uint32_t spin = 0; spin < (1u << 24) && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunk; ++spin)
            __builtin_amdgcn_s_sleep(16);
        __threadfence();   // acquire: nothing below reads the sums from a stale cache line
    }
    return true;
}
// the item's sums are written: whichever wave takes the unit's next chunk may read them
RTX_DEV void queue_item_done(const RenderParams &P, uint32_t unit, uint32_t chunk)
{
    __threadfence();
    if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(P.queue + 1u + unit, chunk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int SRC, bool STATS, bool MANY_LIGHTS, bool RECORD = false, bool QUEUE = false>
__global__ void __launch_bounds__(256) RTX_MEGAKERNEL_ATTR k_megakernel(const RenderParams P)
{
    Table tab = (SRC == kSrcLds) ? stage_table_lds(P) : global_table<SRC, true>(P);
    const uint32_t tiles_x = (P.width + 7u) >> 3;
    uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t sample_first = P.sample_first, sample_count = P.sample_count, chunk = 0;
    bool accumulate = P.accumulate != 0u;
    if constexpr (QUEUE) {
        if (!queue_take_item(P, tile, tile, chunk, sample_first, sample_count, accumulate)) return;
    } else {
        if (tile >= P.n_units) return;   // the whole wave
        tile = launch_unit(P, tile);
    }
    const uint32_t tile_x = (tile % tiles_x) * 8u, tile_y = (tile / tiles_x) * 8u;
    const bool in_image = tile_x + (threadIdx.x & 7u) < P.width && tile_y + ((threadIdx.x & 63u) >> 3) < P.height;
    // the wave's pool, behind the object table (scan) or the traversal stacks (BVH)
    float4 *pool = rtx_smem + fixed_lds_bytes<SRC>(P) / 16u + (threadIdx.x >> 6) * P.pool_entries;
    uint32_t pool_used = 0;   // wave-uniform

    uint32_t rays = 0;
    Counters<STATS> cnt;
    cnt.clear();
    V3 sum = v3(0.0f, 0.0f, 0.0f);
    if (in_image && accumulate) {
        const float *out = P.out_sum + 3ull * ((tile_y + ((threadIdx.x & 63u) >> 3)) * P.width + tile_x + (threadIdx.x & 7u));
        sum = v3(out[0], out[1], out[2]);
    }
    const uint32_t last = sample_first + sample_count;
    uint32_t next = in_image ? sample_first : last;   // own samples [next, end) are not started yet
    uint32_t end = last;                                // own samples [end, last) were handed over
    uint32_t head = kNoEntry;                           // entry of own sample `end`, the list's first
    Path p;
    p.o = p.d = p.T = p.L = v3(0.0f, 0.0f, 0.0f);
    p.depth = 0;
    uint32_t s = 0;                                     // the sample being traced,
    uint32_t job = threadIdx.x & 63u;                   // whose it is (bits 0-5) and its entry (6-15)
    bool active = false;
    // BVH scenes: a lane's traversal can be suspended while other lanes of the wave scatter (below)
    int32_t cur = kBvhEmpty;                            // traversal position; kBvhEmpty = none in progress
    int sp = 0, best = -1;
    float best_t = kInf;
    bool trav = false;                                  // the current ray's closest hit is still being searched
    constexpr uint32_t kDiffuseNum = SRC == kSrcBvh ? RTX_MK_DIFFUSE_BVH : RTX_MK_DIFFUSE_SCAN;
    static_assert(kDiffuseNum <= 64u, "a threshold above all lanes would stall the wave");
    bool wait_diffuse = false;                          // hit known (hit_idx, hit_t), diffuse scatter not yet done
    int hit_idx = -1;
    float hit_t = 0.0f;
    auto finish_sample = [&]() {
        if ((job & 63u) == (threadIdx.x & 63u)) {
            sum = sum + p.L;
        } else {
            float4 *e = pool + (job >> 6);
            e->x = p.L.x;
            e->y = p.L.y;
            e->z = p.L.z;
        }
        active = false;
    };
    uint32_t dbg[M_COUNT] = {};
    (void)dbg;
    for (;;) {
        MK_DBG(M_ROUNDS, 1u);
        const uint32_t lane = threadIdx.x & 63u;
        bool fresh = false;
        if (!active && next < end) {
            job = lane;
            s = next++;
            active = fresh = true;
        }
        if (pool_used < P.pool_entries) {
            unsigned long long idle = __ballot(!active);
            while (idle != 0ull && pool_used < P.pool_entries) {
                // every lane with unstarted samples is active here (it would have taken one above)
                const uint32_t key = end - next;
                const uint32_t most = wave_max(key);
                if (most == 0u) break;
                const int giver = __builtin_ctzll(__ballot(key == most));
                const int taker = __builtin_ctzll(idle);
                const uint32_t taken = (uint32_t)__builtin_amdgcn_readlane((int)end, giver) - 1u;
                if ((int)lane == giver) {
                    pool[pool_used].w = __uint_as_float(head);
                    head = pool_used;
                    end = taken;
                }
                if ((int)lane == taker) {
                    job = (uint32_t)giver | (pool_used << 6);
                    s = taken;
                    active = fresh = true;
                }
                pool_used += 1u;
                idle &= idle - 1ull;
            }
        }
        if (__ballot(active) == 0ull) break;
        const uint32_t px = tile_x + (job & 7u), py = tile_y + ((job & 63u) >> 3);
        const uint32_t pixel = py * P.width + px;
        if (active && fresh) {
            uint4 w = draw_block<!MANY_LIGHTS>(P, tab, pixel, s, 0u);
            camera_ray(P, tab, px, P.height - 1u - py, u24(w.x), u24(w.y), p.o, p.d);
            p.T = v3(1.0f, 1.0f, 1.0f);
            p.L = v3(0.0f, 0.0f, 0.0f);
            p.depth = 0;
            cnt.template add<C_SAMPLES>();
        }
        float t = 0.0f;
        int idx = -1;
        // a lane whose diffuse scatter is pending (below) has its hit already
        const bool need_hit = active && !wait_diffuse;
        bool hit_known = need_hit;    // the lane's closest hit is known: it goes on to the scatter
        if constexpr (SRC != kSrcBvh) {
            if (need_hit) {
                cnt.template add<C_RAYS>();
                if constexpr (RECORD) rays += 1u;
                idx = closest_hit_scan<true>(tab, p.o, p.d, t);
            }
        } else {
            // Traversals take anything from a handful to several hundred node visits, and a wave
            // that waits for its longest one runs its node loop a fifth full (measured on 10 000
            // objects: 46 visits per ray on average, 219 loop iterations per wave).  So the wave
            // stops traversing as soon as fewer than a threshold of its lanes still do: the lanes
            // that are through scatter and come back with their next ray, the others keep their
            // position (cur, the stack in LDS, the best hit so far) and go on in the next round
            // together with them.
            int32_t *stack = reinterpret_cast<int32_t *>(rtx_smem) + threadIdx.x;
            const bool plain = ray_is_plain(p.o, p.d);
            const float ray_length = len(p.d);
            const bool starting = need_hit && !trav;
            const bool all_plain = __ballot(starting && !plain) == 0ull;
#ifdef RTX_MK_DEBUG
            {
                const uint32_t n_st = (uint32_t)__popcll(__ballot(starting));
                MK_DBG(M_START_ROUNDS, n_st ? 1u : 0u);
                MK_DBG(M_START_LANES, n_st);
            }
#endif
            if (starting) {
                cnt.template add<C_RAYS>();
                if constexpr (RECORD) rays += 1u;
                best = -1;
                best_t = kInf;
                bvh_always_tested<STATS>(P.bvh, tab, p.o, p.d, ray_length, all_plain, best, best_t, cnt);
#if defined(RTX_X_REPEAT_PARTS) && ((RTX_X_REPEAT_PARTS) & 4)
                {
                    V3 o2 = p.o;
                    asm volatile("" : "+v"(o2.x));
                    bvh_always_tested<STATS>(P.bvh, tab, o2, p.d, ray_length, all_plain, best, best_t, cnt);
                }
#endif
                sp = 0;
                cur = P.bvh.n_nodes ? 0 : kBvhEmpty;
                trav = true;
            }
#if defined(RTX_X_REPEAT_PARTS) && ((RTX_X_REPEAT_PARTS) & 16)
            {
                V3 o2 = p.o;
                asm volatile("" : "+v"(o2.x));
                const BvhRay R2 = bvh_ray_setup(bvh_grid(P.bvh, tab), o2, p.d);
                asm volatile("" : : "v"(R2.inv.x), "v"(R2.inv.y), "v"(R2.inv.z), "v"(R2.ox.x), "v"(R2.oy.y), "v"(R2.oz.x), "v"(R2.ox.y), "v"(R2.oy.x), "v"(R2.oz.y));
            }
#endif
            const BvhRay R = bvh_ray_setup(bvh_grid(P.bvh, tab), p.o, p.d);
            const uint32_t n_active = (uint32_t)__popcll(__ballot(active));
            const uint32_t threshold = max(1u, n_active * P.bvh.exit_sixteenths >> 4);
            for (;;) {
                const bool at_leaf = cur < 0 && cur != kBvhEmpty;
                const uint32_t n_node = (uint32_t)__popcll(__ballot(cur >= 0));
                const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
                if (n_node + n_leaf < threshold) break;
                // node steps are cheap, the kind-dependent object test is not: it waits until it
                // serves enough lanes, or nothing else is left to do
                if (n_node * RTX_MK_LEAF_COST >= n_leaf) {
#pragma unroll
                    for (int k = 0; k < RTX_MK_NODE_BATCH; ++k) {
                        MK_DBG(M_NODE_ROUNDS, 1u);
                        MK_DBG(M_NODE_LANES, (uint32_t)__popcll(__ballot(cur >= 0)));
                        if (cur >= 0) bvh_node_step<STATS, 256>(P.bvh, R, best_t, cur, sp, stack, cnt);
                    }
                } else {
                    MK_DBG(M_LEAF_ROUNDS, 1u);
                    MK_DBG(M_LEAF_LANES, n_leaf);
                    const bool leaves_plain = __ballot(at_leaf && !plain) == 0ull;
                    if (at_leaf) {
#if defined(RTX_X_REPEAT_PARTS) && ((RTX_X_REPEAT_PARTS) & 2)
                        {
                            V3 o2 = p.o;
                            asm volatile("" : "+v"(o2.x));
                            bvh_try_object<STATS>(tab, ~cur, o2, p.d, ray_length, best, best_t, cnt, leaves_plain);
                        }
#endif
                        bvh_try_object<STATS>(tab, ~cur, p.o, p.d, ray_length, best, best_t, cnt, leaves_plain);
                        if (sp > 0) {
                            sp--;
                            cur = stack[sp * 256];
                        } else {
                            cur = kBvhEmpty;
                        }
                    }
                }
            }
            hit_known = need_hit && cur == kBvhEmpty;
            if (hit_known) {
                trav = false;
                idx = best;
                t = best_t;
            }
        }
#ifdef RTX_MK_DEBUG
        {
            const uint32_t n_hk = (uint32_t)__popcll(__ballot(hit_known));
            MK_DBG(M_SCATTER_ROUNDS, n_hk ? 1u : 0u);
            MK_DBG(M_SCATTER_LANES, n_hk);
        }
#endif
        if (hit_known) {
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
                if constexpr (kDiffuseNum == 0u) {
                    done = scatter<STATS, MANY_LIGHTS, SRC == kSrcBvh>(P, tab, p, idx, t, pixel, s, cnt);
                } else {
                    // emission, depth cut and the cheap materials at once; the diffuse scatter --
                    // direction sampling and the light pdf, most of a surface interaction's cost --
                    // waits until enough lanes of the wave want it
                    done = surface_emission(P, tab, p, idx) || black_surface_ends(P, tab, p, idx);
                    if (!done) {
                        const int material = __float_as_int(tab.mat[3 * idx + 0].w);
                        if (material == kDiffuse) {
                            wait_diffuse = true;
                            hit_idx = idx;
                            hit_t = t;
                        } else {
                            done = bounce<STATS, MANY_LIGHTS, SRC == kSrcBvh>(P, tab, p, idx, t, material, pixel, s, cnt);
                        }
                    }
                }
            }
            if (done) finish_sample();
        }
        if constexpr (kDiffuseNum != 0u) {
            const uint32_t n_wait = (uint32_t)__popcll(__ballot(wait_diffuse));
            const uint32_t n_act = (uint32_t)__popcll(__ballot(active));
            // every active lane waiting always satisfies this (kDiffuseNum <= 64): the wave cannot stall
            if (n_wait != 0u && n_wait * 64u >= n_act * kDiffuseNum) {
                if (wait_diffuse) {
                    wait_diffuse = false;
                    if (bounce<STATS, MANY_LIGHTS, SRC == kSrcBvh>(P, tab, p, hit_idx, hit_t, kDiffuse, pixel, s, cnt)) finish_sample();
                }
            }
        }
    }
    if (in_image) {
        V3 total = sum;
        for (uint32_t e = head; e != kNoEntry;) {   // samples end .. last-1 in this order
            const float4 v = pool[e];
            total = total + v3(v.x, v.y, v.z);
            e = __float_as_uint(v.w);
        }
        float *out = P.out_sum + 3ull * ((tile_y + ((threadIdx.x & 63u) >> 3)) * P.width + tile_x + (threadIdx.x & 7u));
        out[0] = total.x;
        out[1] = total.y;
        out[2] = total.z;
    }
    if constexpr (RECORD) atomicMax(&P.tile_cost[tile], rays);
    if constexpr (QUEUE) queue_item_done(P, tile, chunk);
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < C_COUNT; ++i)
            if (cnt.v[i]) atomicAdd(&P.stats[i], (unsigned long long)cnt.v[i]);
    }
#ifdef RTX_MK_DEBUG
    if ((threadIdx.x & 63u) == 0u)
        for (int i = 0; i < M_COUNT; ++i) atomicAdd(&g_mk_dbg[i], (unsigned long long)dbg[i]);
#endif
}

// ---------------------------------------------------------------------------------------------
// Linear-scan scenes: two pixels per lane.
//
// A closest-hit scan runs with every lane busy, but only about two thirds of the hits go on to a
// scatter (the others end their path: a light, a black surface, the depth limit), so a scatter
// that follows every scan runs a third empty before its own branches thin it further -- and the
// scatter is two thirds of the work.  Here a wave owns TWO 8x8 tiles side by side and a lane one
// pixel of each: one pixel's state ("context": path, sample counters, sum) is in registers, the
// other's is parked in LDS (6 x 16 bytes per lane, one ds_read/write_b128 each way per part).
// Every round scans for the context that needs a scan -- swapping the two where the parked one
// does and the current one does not -- and the scatter runs only when at least
// RTX_MK2_SCATTER_MIN lanes have a context waiting for one (or nothing else can be done), each
// lane bringing in whichever of its contexts waits.  A pixel's samples are still traced one after
// the other and added in order, so the image keeps its bits.
//
// The hand-over of samples between lanes works as in k_megakernel, between the contexts that are
// in registers at that moment; `job` names the owner of the sample in flight: lane (bits 0-5),
// tile half (bit 6), pool entry (bits 7-16).
#ifndef RTX_MK2_SCATTER_MIN
#define RTX_MK2_SCATTER_MIN 50u
#endif
constexpr uint32_t kParkVec = 6u;   // float4 per parked context
// threads per workgroup of k_megakernel_scan2 (a multiple of 64)
#ifndef RTX_MK2_BLOCK
#define RTX_MK2_BLOCK 256
#endif
constexpr uint32_t kScan2Block = RTX_MK2_BLOCK;
enum : uint32_t { kPhaseEmpty = 0u,    // no sample in flight, none of its own left to start
                  kPhaseIdle = 1u,     // no sample in flight, own samples left
                  kPhaseScan = 2u,     // a ray whose closest hit is wanted
                  kPhaseScatter = 3u };// hit known (hit_idx, hit_t), diffuse scatter wanted
struct PixelContext {
    Path p;
    V3 sum;
    uint32_t s, job, next, end, head;
    int hit_idx;
    float hit_t;
    uint32_t phase;   // bits 0-1 phase, bit 2 tile half, bit 3 pixel inside the image
};
RTX_DEV void park_store(float4 *park, const PixelContext &c)
{
    park[0 * kScan2Block] = make_float4(c.p.o.x, c.p.o.y, c.p.o.z, c.p.T.x);
    park[1 * kScan2Block] = make_float4(c.p.d.x, c.p.d.y, c.p.d.z, c.p.T.y);
    park[2 * kScan2Block] = make_float4(c.p.L.x, c.p.L.y, c.p.L.z, c.p.T.z);
    park[3 * kScan2Block] = make_float4(c.sum.x, c.sum.y, c.sum.z, c.hit_t);
    park[4 * kScan2Block] = make_float4(__uint_as_float(c.s), __uint_as_float(c.job), __uint_as_float(c.next), __uint_as_float(c.end));
    park[5 * kScan2Block] = make_float4(__uint_as_float(c.head), __uint_as_float(c.p.depth), __int_as_float(c.hit_idx), __uint_as_float(c.phase));
}
RTX_DEV PixelContext park_load(const float4 *park)
{
    const float4 a = park[0 * kScan2Block], b = park[1 * kScan2Block], c = park[2 * kScan2Block], d = park[3 * kScan2Block], e = park[4 * kScan2Block], f = park[5 * kScan2Block];
    PixelContext x;
    x.p.o = v3(a.x, a.y, a.z);
    x.p.d = v3(b.x, b.y, b.z);
    x.p.L = v3(c.x, c.y, c.z);
    x.p.T = v3(a.w, b.w, c.w);
    x.sum = v3(d.x, d.y, d.z);
    x.hit_t = d.w;
    x.s = __float_as_uint(e.x);
    x.job = __float_as_uint(e.y);
    x.next = __float_as_uint(e.z);
    x.end = __float_as_uint(e.w);
    x.head = __float_as_uint(f.x);
    x.p.depth = __float_as_uint(f.y);
    x.hit_idx = __float_as_int(f.z);
    x.phase = __float_as_uint(f.w);
    return x;
}

__host__ __device__ inline uint32_t scan_units_x(uint32_t width) { return (width + 15u) >> 4; }
__host__ __device__ inline uint32_t scan_units(uint32_t width, uint32_t height) { return scan_units_x(width) * ((height + 7u) >> 3); }

template <int SRC, bool STATS, bool MANY_LIGHTS, bool RECORD = false, bool QUEUE = false>
__global__ void __launch_bounds__(kScan2Block) RTX_MEGAKERNEL_ATTR k_megakernel_scan2(const RenderParams P)
{
    Table tab = (SRC == kSrcLds) ? stage_table_lds(P) : global_table<SRC, true>(P);
    const uint32_t lane = threadIdx.x & 63u;
    // LDS: [object table][parked contexts of the workgroup][the waves' pools]
    float4 *park = rtx_smem + fixed_lds_bytes<SRC>(P) / 16u + threadIdx.x;
    float4 *pool = rtx_smem + fixed_lds_bytes<SRC>(P) / 16u + kParkVec * kScan2Block + (threadIdx.x >> 6) * P.pool_entries;
    Counters<STATS> cnt;
    cnt.clear();
    uint32_t dbg[M_COUNT] = {};
    (void)dbg;
    // The wave's work item.  Without a queue: the unit of its launch slot, all samples.  With a queue
    // (P.queue, see RenderParams) the launch has one wave per (chunk of the sample range, unit) and every
    // wave draws a ticket: tickets go chunk by chunk -- all units' first chunks before all second ones --
    // and a unit's chunks continue its pixels' sums one after the other.
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t unit, sample_first, sample_count, chunk = 0;
    bool accumulate = P.accumulate != 0u;
    if constexpr (QUEUE) {
        if (!queue_take_item(P, slot, unit, chunk, sample_first, sample_count, accumulate)) return;   // the whole wave
    } else {
        if (slot >= P.n_units) return;   // the whole wave
        unit = launch_unit(P, slot);
        sample_first = P.sample_first;
        sample_count = P.sample_count;
    }
    const uint32_t unit_x = (unit % scan_units_x(P.width)) * 16u + (lane & 7u), unit_y = (unit / scan_units_x(P.width)) * 8u + (lane >> 3);
    uint32_t pool_used = 0;   // wave-uniform

    uint32_t rays = 0;
    const uint32_t last = sample_first + sample_count;
    auto fresh_context = [&](uint32_t half) {
        PixelContext c;
        c.p.o = c.p.d = c.p.T = c.p.L = v3(0.0f, 0.0f, 0.0f);
        c.p.depth = 0;
        c.sum = v3(0.0f, 0.0f, 0.0f);
        const uint32_t px = unit_x + 8u * half;
        const bool in_image = px < P.width && unit_y < P.height;
        if (in_image && accumulate) {
            const float *out = P.out_sum + 3ull * (unit_y * P.width + px);
            c.sum = v3(out[0], out[1], out[2]);
        }
        c.s = 0;
        c.job = lane | (half << 6);
        c.next = in_image ? sample_first : last;     // own samples [next, end) are not started yet
        c.end = last;                                // own samples [end, last) were handed over
        c.head = kNoEntry;                           // entry of own sample `end`, the list's first
        c.hit_idx = -1;
        c.hit_t = 0.0f;
        c.phase = (c.next < c.end ? kPhaseIdle : kPhaseEmpty) | (half << 2) | (in_image ? 8u : 0u);
        return c;
    };
    PixelContext c = fresh_context(1u);
    park_store(park, c);
    uint32_t parked_phase = c.phase & 3u;   // what the context in LDS waits for
    c = fresh_context(0u);

    auto set_phase = [&](uint32_t ph) { c.phase = (c.phase & ~3u) | ph; };
    auto swap_contexts = [&]() {
        const PixelContext other = park_load(park);
        park_store(park, c);
        parked_phase = c.phase & 3u;
        c = other;
    };
    // the sample in flight is complete: its radiance goes to its owner's sum or pool entry
    auto finish_sample = [&]() {
        if ((c.job & 127u) == (lane | (((c.phase >> 2) & 1u) << 6))) {
            c.sum = c.sum + c.p.L;
        } else {
            float4 *e = pool + (c.job >> 7);
            e->x = c.p.L.x;
            e->y = c.p.L.y;
            e->z = c.p.L.z;
        }
        set_phase(c.next < c.end ? kPhaseIdle : kPhaseEmpty);
    };
    auto scannable = [](uint32_t ph) { return ph == kPhaseIdle || ph == kPhaseScan; };

    for (;;) {
        MK_DBG(M_ROUNDS, 1u);
        // the context that can use a scan comes into the registers
        {
            const bool sw = !scannable(c.phase & 3u) && scannable(parked_phase);
#ifdef RTX_MK_DEBUG
            const uint32_t n_sw = (uint32_t)__popcll(__ballot(sw));
            MK_DBG(M_SWAP_ROUNDS, n_sw ? 1u : 0u);
            MK_DBG(M_SWAP_LANES, n_sw);
#endif
            if (sw) swap_contexts();
        }
        bool fresh = false;
        if ((c.phase & 3u) == kPhaseIdle) {
            c.job = lane | (((c.phase >> 2) & 1u) << 6);
            c.s = c.next++;
            set_phase(kPhaseScan);
            fresh = true;
        }
        if (pool_used < P.pool_entries) {
            // lanes with nothing to scan in either context take over a sample of the context (among
            // those in registers) that has the most left
            unsigned long long idle = __ballot((c.phase & 3u) == kPhaseEmpty);
            while (idle != 0ull && pool_used < P.pool_entries) {
                const uint32_t key = c.end - c.next;
                const uint32_t most = wave_max(key);
                if (most == 0u) break;
                const int giver = __builtin_ctzll(__ballot(key == most));
                const int taker = __builtin_ctzll(idle);
                const uint32_t taken = (uint32_t)__builtin_amdgcn_readlane((int)c.end, giver) - 1u;
                const uint32_t giver_half = ((uint32_t)__builtin_amdgcn_readlane((int)c.phase, giver) >> 2) & 1u;
                if ((int)lane == giver) {
                    pool[pool_used].w = __uint_as_float(c.head);
                    c.head = pool_used;
                    c.end = taken;
                    if ((c.phase & 3u) == kPhaseIdle && c.next >= c.end) set_phase(kPhaseEmpty);
                }
                if ((int)lane == taker) {
                    c.job = (uint32_t)giver | (giver_half << 6) | (pool_used << 7);
                    c.s = taken;
                    set_phase(kPhaseScan);
                    fresh = true;
                }
                pool_used += 1u;
                idle &= idle - 1ull;
            }
        }
        if (__ballot((c.phase & 3u) != kPhaseEmpty || parked_phase != kPhaseEmpty) == 0ull) break;
        const uint32_t px = unit_x - (lane & 7u) + (c.job & 7u) + 8u * ((c.job >> 6) & 1u);
        const uint32_t py = unit_y - (lane >> 3) + ((c.job & 63u) >> 3);
        const uint32_t pixel = py * P.width + px;
        if (fresh) {
#if defined(RTX_X_REPEAT_PARTS) && ((RTX_X_REPEAT_PARTS) & 32)
            {
                uint32_t s2 = c.s;
                asm volatile("" : "+v"(s2));
                uint4 w2 = draw_block<!MANY_LIGHTS>(P, tab, pixel, s2, 0u);
                V3 o2, d2;
                camera_ray(P, tab, px, P.height - 1u - py, u24(w2.x), u24(w2.y), o2, d2);
                asm volatile("" : : "v"(d2.x), "v"(d2.y), "v"(d2.z));
            }
#endif
            uint4 w = draw_block<!MANY_LIGHTS>(P, tab, pixel, c.s, 0u);
            camera_ray(P, tab, px, P.height - 1u - py, u24(w.x), u24(w.y), c.p.o, c.p.d);
            c.p.T = v3(1.0f, 1.0f, 1.0f);
            c.p.L = v3(0.0f, 0.0f, 0.0f);
            c.p.depth = 0;
            cnt.template add<C_SAMPLES>();
        }
        const bool scanning = (c.phase & 3u) == kPhaseScan;
#ifdef RTX_MK_DEBUG
        {
            const uint32_t n_sc = (uint32_t)__popcll(__ballot(scanning));
            MK_DBG(M_SCAN_ROUNDS, n_sc ? 1u : 0u);
            MK_DBG(M_SCAN_LANES, n_sc);
            MK_DBG(M_FRESH_LANES, (uint32_t)__popcll(__ballot(fresh)));
        }
#endif
        if (scanning) {
            cnt.template add<C_RAYS>();
            if constexpr (RECORD) rays += 1u;
            float t;
            // BVH scenes (small trees only, see two_pixels_per_lane): the traversal runs to its end here
            const int idx = closest_hit<SRC, STATS>(P, tab, c.p.o, c.p.d, t, cnt);
            bool done;
            if (idx < 0) {
                c.p.L = c.p.L + cmul(c.p.T, background(P, tab));
                done = true;
            } else {
                if constexpr (STATS) {
                    int kind = __float_as_int(tab.geom[3 * idx + 1].w) & 0xff;
                    cnt.template add<C_HIT_PLANE>(kind == kPlane);
                    cnt.template add<C_HIT_ELLIPSOID>(kind == kEllipsoid);
                    cnt.template add<C_HIT_BOX>(kind == kBox);
                }
                // emission, depth cut and the cheap materials at once; the diffuse scatter waits
                done = surface_emission(P, tab, c.p, idx) || black_surface_ends(P, tab, c.p, idx);
                if (!done) {
                    const int material = __float_as_int(tab.mat[3 * idx + 0].w);
                    if (material == kDiffuse) {
                        c.hit_idx = idx;
                        c.hit_t = t;
                        set_phase(kPhaseScatter);
                    } else {
                        done = bounce<STATS, MANY_LIGHTS, SRC == kSrcBvh>(P, tab, c.p, idx, t, material, pixel, c.s, cnt);
                    }
                }
            }
            if (done) finish_sample();
        }
        // the scatter: when enough lanes have a context that waits for it, or no lane could scan next
        const uint32_t n_wait = (uint32_t)__popcll(__ballot((c.phase & 3u) == kPhaseScatter || parked_phase == kPhaseScatter));
        const bool can_scan = __ballot(scannable(c.phase & 3u) || scannable(parked_phase)) != 0ull;
        if (n_wait >= RTX_MK2_SCATTER_MIN || (n_wait != 0u && !can_scan)) {
            {
                const bool sw = (c.phase & 3u) != kPhaseScatter && parked_phase == kPhaseScatter;
#ifdef RTX_MK_DEBUG
                const uint32_t n_sw = (uint32_t)__popcll(__ballot(sw));
                MK_DBG(M_SWAP_ROUNDS, n_sw ? 1u : 0u);
                MK_DBG(M_SWAP_LANES, n_sw);
#endif
                if (sw) swap_contexts();
            }
            MK_DBG(M_SCATTER_ROUNDS, 1u);
            MK_DBG(M_SCATTER_LANES, (uint32_t)__popcll(__ballot((c.phase & 3u) == kPhaseScatter)));
            if ((c.phase & 3u) == kPhaseScatter) {
                const uint32_t qx = unit_x - (lane & 7u) + (c.job & 7u) + 8u * ((c.job >> 6) & 1u);
                const uint32_t qy = unit_y - (lane >> 3) + ((c.job & 63u) >> 3);
                set_phase(kPhaseScan);
                if (bounce<STATS, MANY_LIGHTS, SRC == kSrcBvh>(P, tab, c.p, c.hit_idx, c.hit_t, kDiffuse, qy * P.width + qx, c.s, cnt)) finish_sample();
            }
        }
    }
    // both pixels of the lane: own samples, then the handed-over ones in sample order
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (c.phase & 8u) {
            V3 total = c.sum;
            for (uint32_t e = c.head; e != kNoEntry;) {   // samples end .. last-1 in this order
                const float4 v = pool[e];
                total = total + v3(v.x, v.y, v.z);
                e = __float_as_uint(v.w);
            }
            float *out = P.out_sum + 3ull * (unit_y * P.width + unit_x + 8u * ((c.phase >> 2) & 1u));
            out[0] = total.x;
            out[1] = total.y;
            out[2] = total.z;
        }
        if (k == 0) c = park_load(park);
    }
    if constexpr (RECORD) atomicMax(&P.tile_cost[unit], rays);
    if constexpr (QUEUE) queue_item_done(P, unit, chunk);
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < C_COUNT; ++i)
            if (cnt.v[i]) atomicAdd(&P.stats[i], (unsigned long long)cnt.v[i]);
    }
#ifdef RTX_MK_DEBUG
    if (lane == 0u)
        for (int i = 0; i < M_COUNT; ++i) atomicAdd(&g_mk_dbg[i], (unsigned long long)dbg[i]);
#endif
}

template <int SRC>
__global__ void __launch_bounds__(256) k_primary(const RenderParams P, const PrimaryParams out)
{
    Table tab = (SRC == kSrcLds) ? stage_table_lds(P) : global_table<SRC, false>(P);
    uint32_t px, py;
    if (!lane_pixel(P, px, py)) return;
    const uint32_t pixel = py * P.width + px;
    V3 o, d;
    camera_ray(P, px, P.height - 1u - py, 0.5f, 0.5f, o, d);
    float t;
    Counters<false> cnt;
    int idx = closest_hit<SRC, false>(P, tab, o, d, t, cnt);
    Hit h;
    h.n = v3(0.0f, 0.0f, 0.0f);
    h.inside = false;
    if (idx >= 0) h = finish_hit(tab, idx, o, d, t, true);
    else t = 0.0f;
    if (out.hit_id) out.hit_id[pixel] = idx;
    if (out.t) out.t[pixel] = t;
    if (out.normal) {
        out.normal[3ull * pixel + 0] = h.n.x;
        out.normal[3ull * pixel + 1] = h.n.y;
        out.normal[3ull * pixel + 2] = h.n.z;
    }
    if (out.inside) out.inside[pixel] = h.inside ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_divide(float *d, uint64_t n, float divisor)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        d[i] = d[i] / divisor;
}

// The output stage of src/image.rs:65-79 (ACES curve, saturate), :60-63 (gamma 1/2.2) and :44
// (byte conversion) applied to sum / divisor, the same float operations in the same order.
// powf is the device library's, not the host libm's: a byte can differ by one from the host
// path where 255 * x^(1/2.2) falls within an ulp of a rounding boundary.
__global__ void __launch_bounds__(256) k_tonemap(const float *sum, uint64_t n, float divisor, uint8_t *bytes)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const float inv_gamma = 1.0f / 2.2f;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = sum[i] / divisor;
        const float up = x * (a * x + b);
        const float down = x * (c * x + d) + e;
        float y = up / down;
        y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
        const float v = roundf(255.0f * powf(y, inv_gamma));
        bytes[i] = (v != v) ? 0 : (v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (uint8_t)v));
    }
}

// Clears the tile tables before a launch that uses them.  (A kernel rather than hipMemsetAsync: inside a
// captured graph the memset nodes of these calls were seen to clear the tables on the first replay only.)
__global__ void __launch_bounds__(256) k_zero_words(uint32_t *p, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = 0u;
}

// Counting sort of the tiles by decreasing cost, one workgroup.  Tiles of equal cost land in
// the order their atomics happen to be served: the order table only decides which wave starts
// when, never what it computes.
__global__ void __launch_bounds__(1024) k_tile_order(const uint32_t *cost, uint32_t *order, uint32_t n_tiles)
{
    __shared__ uint32_t start[kTileCostMax + 1u];
    for (uint32_t b = threadIdx.x; b <= kTileCostMax; b += blockDim.x) start[b] = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_tiles; i += blockDim.x) atomicAdd(&start[min(cost[i], kTileCostMax)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t at = 0;
        for (uint32_t b = kTileCostMax + 1u; b-- > 0u;) {
            const uint32_t n = start[b];
            start[b] = at;
            at += n;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_tiles; i += blockDim.x)
        order[atomicAdd(&start[min(cost[i], kTileCostMax)], 1u)] = i;
}

static dim3 tile_grid(const RenderParams &p)
{
    uint32_t tiles = ((p.width + 7u) / 8u) * ((p.height + 7u) / 8u);
    return dim3((tiles + 3u) / 4u);
}

template <int SRC>
static void launch_megakernel_src(const RenderParams &p, bool stats, dim3 grid, size_t lds, hipStream_t stream)
{
    const dim3 block(256);
    const bool many = p.n_lights > kFewLights;
    if (stats) {
        if (many) hipLaunchKernelGGL((k_megakernel<SRC, true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((k_megakernel<SRC, true, false>), grid, block, lds, stream, p);
    } else if (p.tile_cost) {
        if (many) hipLaunchKernelGGL((k_megakernel<SRC, false, true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((k_megakernel<SRC, false, false, true>), grid, block, lds, stream, p);
    } else if (p.queue) {
        if (many) hipLaunchKernelGGL((k_megakernel<SRC, false, true, false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((k_megakernel<SRC, false, false, false, true>), grid, block, lds, stream, p);
    } else {
        if (many) hipLaunchKernelGGL((k_megakernel<SRC, false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((k_megakernel<SRC, false, false>), grid, block, lds, stream, p);
    }
}
template <int SRC>
static void launch_megakernel_scan2(const RenderParams &p, bool stats, dim3 grid, size_t lds, hipStream_t stream)
{
    const dim3 block(kScan2Block);
    const bool many = p.n_lights > kFewLights;
    if (stats) {
        if (many) hipLaunchKernelGGL((k_megakernel_scan2<SRC, true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((k_megakernel_scan2<SRC, true, false>), grid, block, lds, stream, p);
    } else if (p.tile_cost) {
        if (many) hipLaunchKernelGGL((k_megakernel_scan2<SRC, false, true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((k_megakernel_scan2<SRC, false, false, true>), grid, block, lds, stream, p);
    } else if (p.queue) {
        if (many) hipLaunchKernelGGL((k_megakernel_scan2<SRC, false, true, false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((k_megakernel_scan2<SRC, false, false, false, true>), grid, block, lds, stream, p);
    } else {
        if (many) hipLaunchKernelGGL((k_megakernel_scan2<SRC, false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((k_megakernel_scan2<SRC, false, false>), grid, block, lds, stream, p);
    }
}

// The two-pixel kernel parks one context per lane in LDS (24 KB per workgroup): used for the scenes it
// pays on (paths_end_at_lights, set at upload) when that and the object table fit a workgroup's 40 KB
// share of the CU, so that four workgroups still do.
static bool two_pixels_per_lane(const RenderParams &p, int src)
{
    if (!RTX_MK_TWO || fixed_lds_bytes(p, src) + kParkVec * kScan2Block * sizeof(float4) > 40u * 1024u * kScan2Block / 256u) return false;
#ifdef RTX_X_TWO_PIXELS_BVH   // timing experiment: small trees through the two-pixel kernel, traversals run to their end.
    // Measured on the glass scene (27 nodes), diffuse scatter held back until 40 / 44 / 50 / 56 lanes wait for it:
    // 236 / 248 / 261 / 290 ms against 202.5 ms for the one-pixel kernel with suspendable traversals -- not used.
    if (src == kSrcBvh) return p.bvh.n_nodes <= 128u && p.stop_on_zero != 0u;
#endif
    return src != kSrcBvh && p.paths_end_at_lights != 0u;
}
// work units a megakernel launch deals out to its waves (and the tile cost / order tables index):
// 8x8 tiles, or pairs of them (16x8 pixels) for the two-pixel kernel
uint32_t megakernel_units(const RenderParams &p, int src)
{
    if (two_pixels_per_lane(p, src)) return scan_units(p.width, p.height);
    return ((p.width + 7u) / 8u) * ((p.height + 7u) / 8u);
}

bool megakernel_two_pixels(const RenderParams &p, int src) { return two_pixels_per_lane(p, src); }
UnitShape megakernel_unit_shape(const RenderParams &p, int src)
{
    return two_pixels_per_lane(p, src) ? UnitShape{1u, 1u} : UnitShape{0u, 1u};
}

hipError_t launch_megakernel(const RenderParams &p0, int src, bool stats, uint32_t pool_limit, hipStream_t stream)
{
    RenderParams p = p0;
    const bool two = two_pixels_per_lane(p, src);
    if (stats) p.queue = nullptr;
    const uint32_t waves = two ? kScan2Block / 64u : 4u;   // per workgroup
    p.n_units = share_units(megakernel_units(p, src), p.share_index, p.share_count);
    if (p.n_units == 0u) return hipSuccess;   // more shares than units: nothing of the frame is this one's
    if (p.queue && p.n_whole > p.n_units) p.n_whole = p.n_units;
    // a wave per work item
    const uint32_t slots = p.queue ? p.n_whole + (p.n_units - p.n_whole) * p.n_chunks : p.n_units;
    const dim3 grid((slots + waves - 1u) / waves);
    // sixteen waves share a CU's 160 KB of LDS: what the table (scan) or the traversal
    // stacks (BVH), and the parked pixel contexts of the two-pixel kernel, leave of a workgroup's
    // 10 KB per wave goes to its waves' hand-over pools
    const size_t fixed = fixed_lds_bytes(p, src) + (two ? kParkVec * kScan2Block * sizeof(float4) : 0u);
    const size_t budget = 10u * 1024u * waves;
    p.pool_entries = fixed < budget ? (uint32_t)std::min<size_t>(kPoolEntriesMax, (budget - fixed) / (waves * sizeof(float4))) : 0u;
    p.pool_entries = std::min(p.pool_entries, pool_limit);
    const size_t lds = fixed + (size_t)p.pool_entries * waves * sizeof(float4);
    if (two) {
        if (src == kSrcLds) launch_megakernel_scan2<kSrcLds>(p, stats, grid, lds, stream);
#ifdef RTX_X_TWO_PIXELS_BVH
        else if (src == kSrcBvh) launch_megakernel_scan2<kSrcBvh>(p, stats, grid, lds, stream);
#endif
        else launch_megakernel_scan2<kSrcScalar>(p, stats, grid, lds, stream);
    } else if (src == kSrcLds) launch_megakernel_src<kSrcLds>(p, stats, grid, lds, stream);
    else if (src == kSrcScalar) launch_megakernel_src<kSrcScalar>(p, stats, grid, lds, stream);
    else launch_megakernel_src<kSrcBvh>(p, stats, grid, lds, stream);
    return hipGetLastError();
}

hipError_t launch_zero_words(uint32_t *p, uint64_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_zero_words, dim3((uint32_t)std::min<uint64_t>((n + 255u) / 256u, 1024u)), dim3(256), 0, stream, p, n);
    return hipGetLastError();
}

hipError_t launch_tile_order(const uint32_t *cost, uint32_t *order, uint32_t n_tiles, hipStream_t stream)
{
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, stream, cost, order, n_tiles);
    return hipGetLastError();
}

hipError_t launch_primary(const RenderParams &p, const PrimaryParams &out, int src, hipStream_t stream)
{
    dim3 grid = tile_grid(p), block(256);
    size_t lds = fixed_lds_bytes(p, src);
    if (src == kSrcLds) hipLaunchKernelGGL((k_primary<kSrcLds>), grid, block, lds, stream, p, out);
    else if (src == kSrcScalar) hipLaunchKernelGGL((k_primary<kSrcScalar>), grid, block, lds, stream, p, out);
    else hipLaunchKernelGGL((k_primary<kSrcBvh>), grid, block, lds, stream, p, out);
    return hipGetLastError();
}

hipError_t launch_divide(float *d, uint64_t n, float divisor, hipStream_t stream)
{
    uint64_t blocks = (n + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_divide, dim3((uint32_t)blocks), dim3(256), 0, stream, d, n, divisor);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float *sum, uint64_t n, float divisor, uint8_t *bytes, hipStream_t stream)
{
    uint64_t blocks = (n + 255u) / 256u;
    if (blocks > 4096u) blocks = 4096u;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tonemap, dim3((uint32_t)blocks), dim3(256), 0, stream, sum, n, divisor, bytes);
    return hipGetLastError();
}

}  // namespace rtx

#ifdef RTX_MK_DEBUG
extern "C" int rtx_mk_debug_read(unsigned long long *out, int reset)
{
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_mk_dbg), sizeof(zero)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_mk_dbg), zero, sizeof(zero)) != hipSuccess) return -1;
    return rtx::M_COUNT;
}
#endif
