// rtx_bvh_traverse.hpp -- closest hit through the host-built BVH (node layout: rtx_bvh.h).
//
// Same result as the linear scan of src/trace.rs:102-121: smallest distance, ties to the
// lowest object index.  Any traversal order gives that result because (a) the update rule
// is lexicographic in (t, index) and (b) a node is skipped only when its entry distance is
// strictly greater than the best distance so far.  Unbounded objects (planes) are tested
// first, in a wave-uniform loop.  Each lane keeps its stack in LDS, one column per lane.
#pragma once
#include "rtx_bvh.h"
#include "rtx_trace.hpp"

namespace rtx {

template <bool STATS>
RTX_DEV void bvh_try_object(const Table &tab, int pi, V3 o, V3 d, float ray_length, int &best, float &best_t,
                            Counters<STATS> &cnt)
{
    Geom g = load_geom(tab.geom, (uint32_t)pi);
    cnt.template add<C_PTEST_PLANE>(g.kind == kPlane);
    cnt.template add<C_PTEST_ELLIPSOID>(g.kind == kEllipsoid);
    cnt.template add<C_PTEST_BOX>(g.kind == kBox);
    LocalRay lr = to_local(g, o, d);
    float t;
    bool hit = figure_t(g.kind, g.param, lr, t);
    hit = hit && (t * ray_length < kInf);
    if (hit && (best < 0 || t < best_t || (t == best_t && pi < best))) {
        best_t = t;
        best = pi;
    }
}

// entry distance of the ray into a box, or "miss"; conservative (boxes are padded)
RTX_DEV bool slab_test(V3 lo, V3 hi, V3 o, V3 inv, float t_limit, float &t_near)
{
    float x0 = (lo.x - o.x) * inv.x, x1 = (hi.x - o.x) * inv.x;
    float y0 = (lo.y - o.y) * inv.y, y1 = (hi.y - o.y) * inv.y;
    float z0 = (lo.z - o.z) * inv.z, z1 = (hi.z - o.z) * inv.z;
    // fminf / fmaxf ignore a NaN operand (0 * inf when the origin lies on a slab plane)
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    t_near = tn;
    return tn <= tf && !(tn > t_limit);
}

// BLOCK = threads per workgroup of the calling kernel (the stack array is sized for it).
//
// The loop alternates two phases so that the expensive, kind-dependent object test is not
// interleaved with the cheap node steps: every lane first descends until it holds a leaf (or
// has finished), then all lanes holding a leaf test their object together.
template <bool STATS, int BLOCK = 256>
RTX_DEV int closest_hit_bvh(const BvhView &B, const Table &tab, V3 o, V3 d, float &best_t, Counters<STATS> &cnt)
{
    __shared__ int32_t s_stack[kBvhStackSize * BLOCK];
    int32_t *stack = s_stack + threadIdx.x;

    const float ray_length = len(d);
    int best = -1;
    best_t = kInf;
    for (uint32_t i = 0; i < B.n_planes; ++i)
        bvh_try_object<STATS>(tab, B.planes[i], o, d, ray_length, best, best_t, cnt);
    if (B.n_nodes == 0) return best;

    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int sp = 0;
    int32_t cur = 0;  // >= 0 inner node, < 0 leaf of object ~cur, kBvhEmpty = finished
    for (;;) {
        while (cur >= 0) {
            const float4 *nd = B.nodes + 4 * (size_t)cur;
            float4 f0 = nd[0], f1 = nd[1], f2 = nd[2], f3 = nd[3];
            const int32_t r0 = __float_as_int(f3.x), r1 = __float_as_int(f3.y);
            float d0, d1;
            bool h0 = r0 != kBvhEmpty && slab_test(v3(f0.x, f0.y, f0.z), v3(f0.w, f1.x, f1.y), o, inv, best_t, d0);
            bool h1 = r1 != kBvhEmpty && slab_test(v3(f1.z, f1.w, f2.x), v3(f2.y, f2.z, f2.w), o, inv, best_t, d1);
            if constexpr (STATS) cnt.v[C_NODE_TESTS] += 2;
            if (h0 && h1) {
                const bool first0 = d0 <= d1;
                stack[sp * BLOCK] = first0 ? r1 : r0;   // depth <= kBvhMaxDepth < kBvhStackSize
                sp++;
                cur = first0 ? r0 : r1;
            } else if (h0) {
                cur = r0;
            } else if (h1) {
                cur = r1;
            } else if (sp > 0) {
                sp--;
                cur = stack[sp * BLOCK];
            } else {
                cur = kBvhEmpty;
            }
        }
        if (__ballot(cur != kBvhEmpty) == 0ull) break;
        if (cur != kBvhEmpty) {
            bvh_try_object<STATS>(tab, ~cur, o, d, ray_length, best, best_t, cnt);
            if (sp > 0) {
                sp--;
                cur = stack[sp * BLOCK];
            } else {
                cur = kBvhEmpty;
            }
        }
    }
    return best;
}

}  // namespace rtx
