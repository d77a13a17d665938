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

// UNIFORM: every lane tests the same object (the list of unbounded objects), which allows the
// shortcuts of the linear scan: no rotations for an identity pose when every lane's ray is
// plain (all_plain), the kind read once per wave.
template <bool STATS, bool UNIFORM = false>
RTX_DEV void bvh_try_object(const Table &tab, int pi, V3 o, V3 d, float ray_length, int &best, float &best_t,
                            Counters<STATS> &cnt, bool all_plain = false)
{
    Geom g = load_geom(tab.geom, (uint32_t)pi);
    cnt.template add<C_PTEST_PLANE>(g.kind == kPlane);
    cnt.template add<C_PTEST_ELLIPSOID>(g.kind == kEllipsoid);
    cnt.template add<C_PTEST_BOX>(g.kind == kBox);
    LocalRay lr;
    int kind = g.kind;
    if constexpr (UNIFORM) {
        if (all_plain && (__builtin_amdgcn_readfirstlane(g.flags) & kGeomIdentity)) lr = to_local_identity(g, o, d);
        else lr = to_local(g, o, d);
        kind = __builtin_amdgcn_readfirstlane(g.kind | ((g.flags & kGeomPow2) << 8));
    } else {
        // leaves: every lane holds another object; `all_plain` = every calling lane's ray is plain.
        // The rotations are skipped when every object under test has the identity pose.
        if (all_plain && __ballot(!(g.flags & kGeomIdentity)) == 0ull) lr = to_local_identity(g, o, d);
        else lr = to_local(g, o, d);
    }
    float t;
    bool hit;
    // a leaf never holds a plane (unbounded objects are on the always-tested list)
    if constexpr (UNIFORM) hit = figure_t(kind, g.param, lr, t);
    else hit = ellipsoid_or_box_t(kind, g.param, lr, t);
    hit = hit && (t * ray_length < kInf);
    if (hit && (best < 0 || t < best_t || (t == best_t && pi < best))) {
        best_t = t;
        best = pi;
    }
}

// The objects every ray is tested against (unbounded ones: planes), before the traversal.  For
// plain rays they go category by category like the linear scan's objects (scan_by_category: in a
// BVH scene the category list holds exactly these objects); the event-counting builds keep the
// object-by-object loop, which counts.
template <bool STATS>
RTX_DEV void bvh_always_tested(const BvhView &B, const Table &tab, V3 o, V3 d, float ray_length, bool all_plain,
                               int &best, float &best_t, Counters<STATS> &cnt)
{
    if constexpr (!STATS) {
        if (all_plain && tab.order != nullptr) {
            scan_by_category(tab, o, d, ray_length, best, best_t);
            return;
        }
    }
    for (uint32_t i = 0; i < B.n_planes; ++i)
        bvh_try_object<STATS, true>(tab, B.planes[i], o, d, ray_length, best, best_t, cnt, all_plain);
}

// A ray prepared for the node tests: its reciprocal direction, the negated slab offsets on the
// node grid per axis as (low plane, high plane) pairs (see slab_test), and whether its margin is
// so large that it takes every node.
struct BvhRay {
    V3 inv;
    F2 ox, oy, oz;
    bool far;
};
// Entry distance of the ray into a node box, or "miss".  Box corners are grid coordinates
// (integers up to 65535, exact in f32); inv is the reciprocal of the ray direction in grid units,
// oi_lo = (origin + margin) * inv and oi_hi = (origin - margin) * inv with the origin in grid
// units (BvhRay holds them negated), so q * inv - oi_lo is the world-space distance at which the ray crosses the low plane
// q moved outwards by the ray's margin, and likewise for the high planes (numerator and
// denominator carry the same scale).  One fused multiply-add per plane.  Conservative: the
// builder rounds outwards with a grid unit to spare, and the margin (closest_hit_bvh) covers
// the rounding of the origin and the growth of the objects' hit zones for distant origins.
// For a direction component of exactly 0 the products are inf - inf = NaN, which fminf / fmaxf
// ignore: that axis then never culls (more nodes visited by such a ray, the same result); the
// caller passes NaN offsets for an axis whose products could overflow, to the same effect.
RTX_DEV bool slab_test(F2 px, F2 py, F2 pz, const BvhRay &R, float t_limit, float &t_near)
{
#ifdef RTX_BVH_PACKED_SLABS
    // the low and the high plane of an axis in one packed FMA (v_pk_fma_f32): the same IEEE operations
    // as two scalar ones; measured no faster even with the pairs side by side in the node (10 000
    // objects: 157.4 ms against 156.5, the wavefront variant 249 against 244), so not the default
    const F2 ix = {R.inv.x, R.inv.x}, iy = {R.inv.y, R.inv.y}, iz = {R.inv.z, R.inv.z};
    const F2 rx = __builtin_elementwise_fma(px, ix, R.ox), ry = __builtin_elementwise_fma(py, iy, R.oy),
             rz = __builtin_elementwise_fma(pz, iz, R.oz);
    const float x0 = rx.x, x1 = rx.y, y0 = ry.x, y1 = ry.y, z0 = rz.x, z1 = rz.y;
#else
    float x0 = __builtin_fmaf(px.x, R.inv.x, R.ox.x), x1 = __builtin_fmaf(px.y, R.inv.x, R.ox.y);
    float y0 = __builtin_fmaf(py.x, R.inv.y, R.oy.x), y1 = __builtin_fmaf(py.y, R.inv.y, R.oy.y);
    float z0 = __builtin_fmaf(pz.x, R.inv.z, R.oz.x), z1 = __builtin_fmaf(pz.y, R.inv.z, R.oz.y);
#endif
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    t_near = tn;
    return tn <= tf && !(tn > t_limit);
}
// a ray whose margin reaches this many grid units takes every node
constexpr float kBvhMaxMargin = 16384.0f;

// The ray on the node grid, and its margin.  The node boxes bound the zone in which the
// object tests can report a hit for origins inside the grid; the zone grows by at most
// far_slope per world unit the origin lies outside it (rtx_bvh.cpp: world_bounds), and the
// f32 value of the origin in grid units is off by up to 2^-23 of its magnitude.
// The grid of the node boxes as the ray set-up reads it: from the kernel's arguments, or from the kernel's LDS copy
// of them (Table::uniforms, floats 24-34: base, -, inverse step, far slope, step) so that ten scalar registers are
// not held through the traversal loop for something read once per round.
struct BvhGrid {
    V3 base, inv_step, step;
    float far_slope;
};
RTX_DEV BvhGrid bvh_grid(const BvhView &B)
{
    BvhGrid G;
    G.base = v3(B.grid_base[0], B.grid_base[1], B.grid_base[2]);
    G.inv_step = v3(B.grid_inv_step[0], B.grid_inv_step[1], B.grid_inv_step[2]);
    G.step = v3(B.grid_step[0], B.grid_step[1], B.grid_step[2]);
    G.far_slope = B.far_slope;
    return G;
}
RTX_DEV BvhGrid bvh_grid(const BvhView &B, const Table &tab)
{
    if (!tab.has_uniforms) return bvh_grid(B);
    const float4 a = *reinterpret_cast<const float4 *>(tab.uniforms + 24), b = *reinterpret_cast<const float4 *>(tab.uniforms + 28),
                 c = *reinterpret_cast<const float4 *>(tab.uniforms + 32);
    BvhGrid G;
    G.base = v3(a.x, a.y, a.z);
    G.inv_step = v3(b.x, b.y, b.z);
    G.far_slope = b.w;
    G.step = v3(c.x, c.y, c.z);
    return G;
}
RTX_DEV BvhRay bvh_ray_setup(const BvhGrid &B, V3 o, V3 d)
{
    BvhRay R;
    const V3 gs = B.inv_step;
    const V3 og = cmul(o - B.base, gs);
    const V3 dg = cmul(d, gs);
    R.inv = cdiv(v3(1.0f, 1.0f, 1.0f), dg);
    const float gmax = (float)kBvhGridMax;
    const V3 outside = v3(fmaxf(fmaxf(-og.x, og.x - gmax), 0.0f), fmaxf(fmaxf(-og.y, og.y - gmax), 0.0f),
                          fmaxf(fmaxf(-og.z, og.z - gmax), 0.0f));                      // grid units
    const float outside_world = (outside.x * B.step.x + outside.y * B.step.y) + outside.z * B.step.z;
    const float grow = B.far_slope * outside_world * 1.01f;                             // world units
    const V3 margin = v3(grow * gs.x + 4e-7f * (fabsf(og.x) + gmax), grow * gs.y + 4e-7f * (fabsf(og.y) + gmax),
                         grow * gs.z + 4e-7f * (fabsf(og.z) + gmax));                   // grid units
    V3 oi_lo = cmul(og + margin, R.inv);
    V3 oi_hi = cmul(og - margin, R.inv);
    // q * inv - oi must not overflow in either product (then it would no longer be the distance
    // to plane q): an axis on which it could -- a direction almost parallel to the planes, on a
    // grid stretched by a far-away object -- gets NaN offsets and never culls
    const float kNaN = __int_as_float(0x7fc00000);
    if (!(fabsf(R.inv.x) * (fabsf(og.x) + margin.x + gmax) < 3e38f)) oi_lo.x = oi_hi.x = kNaN;
    if (!(fabsf(R.inv.y) * (fabsf(og.y) + margin.y + gmax) < 3e38f)) oi_lo.y = oi_hi.y = kNaN;
    if (!(fabsf(R.inv.z) * (fabsf(og.z) + margin.z + gmax) < 3e38f)) oi_lo.z = oi_hi.z = kNaN;
    R.ox = F2{-oi_lo.x, -oi_hi.x};
    R.oy = F2{-oi_lo.y, -oi_hi.y};
    R.oz = F2{-oi_lo.z, -oi_hi.z};
    // written so that a NaN counts as far
    R.far = !(fmaxf(fmaxf(margin.x, margin.y), margin.z) < kBvhMaxMargin);
    return R;
}

// One node visit: tests both children of inner node `cur` against the ray, descends into the
// nearer one that is hit (postponing the other on the lane's stack, one column of `stack` with
// stride BLOCK) or pops the stack.  cur >= 0 inner node, < 0 leaf of object ~cur, kBvhEmpty =
// nothing left.
template <bool STATS, int BLOCK>
RTX_DEV void bvh_node_step(const BvhView &B, const BvhRay &R, float best_t, int32_t &cur, int &sp, int32_t *stack,
                           Counters<STATS> &cnt)
{
    float d0 = 0.0f, d1 = 0.0f;
    // a 32-bit byte offset from the (wave-uniform) base: the loads take it as is, no 64-bit address arithmetic
    const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(B.nodes) +
                                                        (uint32_t)cur * (uint32_t)(kBvhNodeVec * sizeof(uint4)));
    const float4 a = nd[0], b = nd[1], c = nd[2], e = nd[3];
    const int32_t r0 = __float_as_int(b.z), r1 = __float_as_int(b.w);
    bool h0 = slab_test(F2{a.x, a.y}, F2{a.z, a.w}, F2{b.x, b.y}, R, best_t, d0);
    bool h1 = slab_test(F2{c.x, c.y}, F2{c.z, c.w}, F2{e.x, e.y}, R, best_t, d1);
    h0 = (h0 || R.far) && r0 != kBvhEmpty;
    h1 = (h1 || R.far) && r1 != kBvhEmpty;
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

// BLOCK = threads per workgroup of the calling kernel (the stack array is sized for it).
// DYN_STACK: the caller provides the stacks (B.stack_entries x BLOCK words of LDS: a tree of
// depth d never holds more than d postponed nodes) instead of the static kBvhStackSize x BLOCK.
//
// The loop alternates two phases so that the expensive, kind-dependent object test is not
// interleaved with the cheap node steps: every lane first descends until it holds a leaf (or
// has finished), then all lanes holding a leaf test their object together.
template <bool STATS, int BLOCK = 256, bool DYN_STACK = false>
RTX_DEV int closest_hit_bvh(const BvhView &B, const Table &tab, V3 o, V3 d, float &best_t, Counters<STATS> &cnt,
                            int32_t *dyn_stack = nullptr)
{
    int32_t *stack;
    if constexpr (DYN_STACK) {
        stack = dyn_stack + threadIdx.x;
    } else {
        __shared__ int32_t s_stack[kBvhStackSize * BLOCK];
        stack = s_stack + threadIdx.x;
    }

    const float ray_length = len(d);
    int best = -1;
    best_t = kInf;
    const bool all_plain = __ballot(!ray_is_plain(o, d)) == 0ull;
    bvh_always_tested<STATS>(B, tab, o, d, ray_length, all_plain, best, best_t, cnt);
    if (B.n_nodes == 0) return best;
    const bool leaves_plain = all_plain;   // wave-uniform

    const BvhRay R = bvh_ray_setup(bvh_grid(B, tab), o, d);
    int sp = 0;
    int32_t cur = 0;  // >= 0 inner node, < 0 leaf of object ~cur, kBvhEmpty = finished
    for (;;) {
        // node phase: every lane descends until it holds a leaf or has finished
        while (cur >= 0) bvh_node_step<STATS, BLOCK>(B, R, best_t, cur, sp, stack, cnt);
        if (__ballot(cur != kBvhEmpty) == 0ull) break;
        // leaf phase
        if (cur < 0 && cur != kBvhEmpty) {
            bvh_try_object<STATS>(tab, ~cur, o, d, ray_length, best, best_t, cnt, leaves_plain);
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
