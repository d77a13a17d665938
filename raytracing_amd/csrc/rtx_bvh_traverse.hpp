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

// BLOCK = threads per workgroup of the calling kernel (the stack array is sized for it)
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
    int node = 0;
    while (true) {
        const float4 *nd = B.nodes + 4 * (size_t)node;
        float4 f0 = nd[0], f1 = nd[1], f2 = nd[2], f3 = nd[3];
        int c0 = __float_as_int(f3.x), n0 = __float_as_int(f3.y);
        int c1 = __float_as_int(f3.z), n1 = __float_as_int(f3.w);
        float d0, d1;
        bool h0 = (n0 > 0 || c0 >= 0) && slab_test(v3(f0.x, f0.y, f0.z), v3(f0.w, f1.x, f1.y), o, inv, best_t, d0);
        bool h1 = (n1 > 0 || c1 >= 0) && slab_test(v3(f1.z, f1.w, f2.x), v3(f2.y, f2.z, f2.w), o, inv, best_t, d1);
        if constexpr (STATS) cnt.v[C_NODE_TESTS] += 2;
        if (h0 && n0 > 0) {
            for (int k = 0; k < n0; ++k)
                bvh_try_object<STATS>(tab, B.prim_index[c0 + k], o, d, ray_length, best, best_t, cnt);
            h0 = false;
        }
        if (h1 && n1 > 0) {
            // the first leaf may have shortened the ray
            if (!(d1 > best_t))
                for (int k = 0; k < n1; ++k)
                    bvh_try_object<STATS>(tab, B.prim_index[c1 + k], o, d, ray_length, best, best_t, cnt);
            h1 = false;
        }
        if (h0 && h1) {
            bool first0 = d0 <= d1;
            int near_c = first0 ? c0 : c1, far_c = first0 ? c1 : c0;
            if (sp < kBvhStackSize) {
                stack[sp * BLOCK] = far_c;
                sp++;
            }
            node = near_c;
        } else if (h0) {
            node = c0;
        } else if (h1) {
            node = c1;
        } else {
            if (sp == 0) break;
            sp--;
            node = stack[sp * BLOCK];
        }
    }
    return best;
}

}  // namespace rtx
