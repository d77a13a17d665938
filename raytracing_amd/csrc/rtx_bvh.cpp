// rtx_bvh.cpp -- BVH construction on the host: binned surface-area heuristic, binary tree,
// one object per leaf, depth bounded by kBvhMaxDepth (balanced median splits take over where
// a subtree could not fit otherwise), nodes emitted breadth-first.
#include "rtx_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <queue>

#ifndef RTX_BVH_SWEEP_MAX
#define RTX_BVH_SWEEP_MAX 65536
#endif
#ifndef RTX_BVH_OPT_PASSES
#define RTX_BVH_OPT_PASSES 3
#endif

namespace rtx {
namespace {

struct Box3 {
    double lo[3], hi[3];
    void clear()
    {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<double>::infinity();
            hi[k] = -std::numeric_limits<double>::infinity();
        }
    }
    void grow(const Box3 &b)
    {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    double half_area() const
    {
        double e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (e[0] < 0 || e[1] < 0 || e[2] < 0) return 0.0;
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }
};

struct Ref {
    Box3 box;
    double c[3];
    int32_t prim;
};

// Relative error bound of det = b^2 - a c in the ellipsoid test (src/objects/geometry.rs:75-91)
// in units of |u|^2 |v|^2: three roundings in each of a, b, c and one in each product give
// 8.5e-7; 1e-6 is used.
constexpr double kDetEps = 1e-6;
// the zone in which the box test (geometry.rs:114-130) can report a hit grows by about 2e-6 of
// the distance between the ray origin and the box; twice that is used
constexpr double kBoxSlope = 4e-6;

// World bounds of a rotated ellipsoid / box, or false when they cannot be trusted.
//
// The bounds are not those of the geometric object but of the zone in which the reference's f32
// test can report a hit for a ray whose origin is at most `reach` away from the object's centre
// (every origin inside the node grid is).  For an ellipsoid the exact value of det is
// |v|^2 (1 - q^2), q = distance of the line from the centre in the scaled space where the
// ellipsoid is the unit sphere, and |u| <= reach / r_min, so hits are confined to
// q^2 < 1 + kDetEps (reach / r_min)^2: the ellipsoid scaled by the square root of that.  For
// origins farther away the zone grows by at most `slope` per unit of extra distance (the
// traversal adds that margin per ray).
bool world_bounds(const rtx_prim &p, double reach, Box3 &out, double &slope)
{
    double ainv[3][3], na, nai;
    if (!pose_matrices(p, ainv, na, nai)) return false;
    double maxc = 0.0, maxe = 0.0, rmin = std::numeric_limits<double>::infinity(), rmax = 0.0;
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(p.position[k]) || !std::isfinite(p.param[k])) return false;
        maxc = std::max(maxc, std::fabs((double)p.position[k]));
        rmin = std::min(rmin, std::fabs((double)p.param[k]));
        rmax = std::max(rmax, std::fabs((double)p.param[k]));
    }
    // distances in local space are at most |A| times those in world space, and local growth
    // shows in world space multiplied by at most |A^-1| (both 1 for a unit quaternion)
    const double reach_local = reach * na;
    double grow = 1.0, extra = 0.0;
    if (p.kind == RTX_ELLIPSOID) {
        if (!(rmin > 0.0)) return false;   // a flat ellipsoid: tested by every ray
        grow = std::sqrt(1.0 + kDetEps * (reach_local / rmin) * (reach_local / rmin));
        slope = std::sqrt(kDetEps) * (rmax / rmin) * na * nai;
        if (!std::isfinite(grow) || grow > 1e3) return false;
    } else {
        extra = kBoxSlope * reach_local;
        slope = kBoxSlope * na * nai;
    }
    double e[3];
    for (int k = 0; k < 3; ++k) {
        double s = 0.0;
        for (int j = 0; j < 3; ++j) {
            double a = std::fabs(ainv[k][j]) * (std::fabs((double)p.param[j]) * grow + extra);
            s += (p.kind == RTX_ELLIPSOID) ? a * a : a;
        }
        e[k] = (p.kind == RTX_ELLIPSOID ? std::sqrt(s) : s);
        maxe = std::max(maxe, e[k]);
    }
    // padding: f32 rounding of the positions involved
    double pad = 1e-4 * (1.0 + maxc + maxe);
    for (int k = 0; k < 3; ++k) {
        out.lo[k] = (double)p.position[k] - e[k] * (1.0 + 1e-5) - pad;
        out.hi[k] = (double)p.position[k] + e[k] * (1.0 + 1e-5) + pad;
        if (!std::isfinite(out.lo[k]) || !std::isfinite(out.hi[k])) return false;
    }
    return true;
}

struct BuildNode {
    Box3 box[2];
    int32_t ref[2];  // >= 0: index into the build-node array; < 0: leaf of object ~ref
};

int ceil_log2(size_t n)
{
    int k = 0;
    while (((size_t)1 << k) < n) ++k;
    return k;
}

constexpr size_t kSweepMax = RTX_BVH_SWEEP_MAX;

struct Builder {
    std::vector<Ref> refs;
    std::vector<BuildNode> nodes;
    int depth = 0;

    // returns the reference describing refs[b, e): a leaf or a new inner node
    int32_t make(size_t b, size_t e, int level)
    {
        const size_t n = e - b;
        if (n == 1) {
            depth = std::max(depth, level);
            return ~refs[b].prim;
        }
        // a balanced split needs ceil(log2 n) more levels: use it when nothing else fits
        const bool must_balance = level + ceil_log2(n) >= kBvhMaxDepth;
        size_t mid = must_balance ? split_median(b, e) : split(b, e);
        int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        BuildNode tmp;
        tmp.box[0].clear();
        tmp.box[1].clear();
        for (size_t i = b; i < mid; ++i) tmp.box[0].grow(refs[i].box);
        for (size_t i = mid; i < e; ++i) tmp.box[1].grow(refs[i].box);
        tmp.ref[0] = make(b, mid, level + 1);
        tmp.ref[1] = make(mid, e, level + 1);
        nodes[(size_t)me] = tmp;
        return me;
    }

    size_t split_median(size_t b, size_t e)
    {
        size_t mid = b + (e - b) / 2;
        std::nth_element(refs.begin() + (ptrdiff_t)b, refs.begin() + (ptrdiff_t)mid, refs.begin() + (ptrdiff_t)e,
                         [](const Ref &x, const Ref &y) { return x.prim < y.prim; });
        return mid;
    }

    // Exact surface-area heuristic: every partition of the centres' order along every axis is priced.
    // O(n log n) per call, used below kSweepMax objects.
    size_t split_sweep(size_t b, size_t e)
    {
        const size_t n = e - b;
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1;
        size_t best_k = 0;
        std::vector<double> right(n);
        for (int axis = 0; axis < 3; ++axis) {
            std::sort(refs.begin() + (ptrdiff_t)b, refs.begin() + (ptrdiff_t)e, [axis](const Ref &x, const Ref &y) {
                return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.prim < y.prim);
            });
            Box3 acc;
            acc.clear();
            for (size_t i = n - 1; i > 0; --i) {
                acc.grow(refs[b + i].box);
                right[i] = acc.half_area();
            }
            acc.clear();
            for (size_t i = 0; i + 1 < n; ++i) {
                acc.grow(refs[b + i].box);
                const double cost = acc.half_area() * (double)(i + 1) + right[i + 1] * (double)(n - i - 1);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_k = i + 1; }
            }
        }
        if (best_axis < 0) return split_median(b, e);
        if (best_axis != 2)
            std::sort(refs.begin() + (ptrdiff_t)b, refs.begin() + (ptrdiff_t)e, [best_axis](const Ref &x, const Ref &y) {
                return x.c[best_axis] < y.c[best_axis] || (x.c[best_axis] == y.c[best_axis] && x.prim < y.prim);
            });
        return b + best_k;
    }

    size_t split(size_t b, size_t e)
    {
        if (e - b <= kSweepMax) return split_sweep(b, e);
        Box3 cb;
        cb.clear();
        for (size_t i = b; i < e; ++i)
            for (int k = 0; k < 3; ++k) {
                cb.lo[k] = std::min(cb.lo[k], refs[i].c[k]);
                cb.hi[k] = std::max(cb.hi[k], refs[i].c[k]);
            }
        constexpr int kBins = 16;
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            double ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.0)) continue;
            Box3 bb[kBins];
            size_t cnt[kBins];
            for (int i = 0; i < kBins; ++i) { bb[i].clear(); cnt[i] = 0; }
            for (size_t i = b; i < e; ++i) {
                int bi = std::min(kBins - 1, (int)((refs[i].c[axis] - cb.lo[axis]) / ext * kBins));
                bb[bi].grow(refs[i].box);
                cnt[bi]++;
            }
            double right_area[kBins];
            size_t right_cnt[kBins];
            Box3 acc;
            acc.clear();
            size_t c = 0;
            for (int i = kBins - 1; i > 0; --i) {
                acc.grow(bb[i]);
                c += cnt[i];
                right_area[i] = acc.half_area();
                right_cnt[i] = c;
            }
            acc.clear();
            c = 0;
            for (int i = 0; i < kBins - 1; ++i) {
                acc.grow(bb[i]);
                c += cnt[i];
                if (c == 0 || right_cnt[i + 1] == 0) continue;
                double cost = acc.half_area() * (double)c + right_area[i + 1] * (double)right_cnt[i + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = i; }
            }
        }
        size_t mid;
        if (best_axis >= 0) {
            double ext = cb.hi[best_axis] - cb.lo[best_axis];
            double lo = cb.lo[best_axis];
            auto it = std::partition(refs.begin() + (ptrdiff_t)b, refs.begin() + (ptrdiff_t)e, [&](const Ref &r) {
                int bi = std::min(kBins - 1, (int)((r.c[best_axis] - lo) / ext * kBins));
                return bi <= best_bin;
            });
            mid = (size_t)(it - refs.begin());
        } else {
            mid = b;
        }
        // all centres coincide (or binning failed): split by object index
        if (mid == b || mid == e) mid = split_median(b, e);
        return mid;
    }
};

// The tree as the optimiser below wants it: every node with its own box and a link to its parent.
struct TreeNode {
    Box3 box;
    int32_t parent = -1;
    int32_t child[2] = {-1, -1};   // inner node: both >= 0
    int32_t prim = -1;             // leaf: the object
    bool leaf() const { return child[0] < 0; }
};

// Insertion-based optimisation of a finished tree (after Bittner, Hapala and Havran, "Fast insertion-based
// optimization of bounding volume hierarchies", 2013): a subtree is taken out -- its parent disappears, its
// sibling moves up -- and put back as the sibling of whichever node lowers the summed surface area of the inner
// nodes most.  The search walks up from the old position and, at every level, down into the subtree that hangs
// off the path, with the area saved so far as its budget (branch and bound).  The tree's leaves, and with them
// every result of a traversal, stay what they are; only the number of node visits changes.
struct TreeOptimizer {
    std::vector<TreeNode> &t;
    int32_t root;
    std::vector<std::pair<double, int32_t>> stack;

    int32_t other_child(int32_t parent, int32_t c) const { return t[(size_t)parent].child[0] == c ? t[(size_t)parent].child[1] : t[(size_t)parent].child[0]; }
    void refit_upwards(int32_t i)
    {
        for (; i >= 0; i = t[(size_t)i].parent) {
            Box3 b = t[(size_t)t[(size_t)i].child[0]].box;
            b.grow(t[(size_t)t[(size_t)i].child[1]].box);
            t[(size_t)i].box = b;
        }
    }
    // best new sibling for node `in` and the area it saves (<= 0: leave it where it is)
    std::pair<int32_t, double> find(int32_t in)
    {
        const Box3 in_box = t[(size_t)in].box;
        const double in_area = in_box.half_area();
        const int32_t parent = t[(size_t)in].parent;
        double budget = t[(size_t)parent].box.half_area();   // the parent disappears
        int32_t sibling = other_child(parent, in), pivot = parent;
        Box3 pivot_box = t[(size_t)sibling].box;                // the pivot's box once `in` is gone
        int32_t best = -1;
        double best_gain = 0.0;
        for (;;) {
            stack.clear();
            stack.emplace_back(budget, sibling);
            while (!stack.empty()) {
                const double b = stack.back().first;
                const int32_t dst = stack.back().second;
                stack.pop_back();
                if (b - in_area <= best_gain) continue;       // even a perfect fit cannot beat the best
                Box3 m = t[(size_t)dst].box;
                m.grow(in_box);
                const double gain = b - m.half_area();          // the new inner node costs its area
                if (gain > best_gain) { best = dst; best_gain = gain; }
                if (!t[(size_t)dst].leaf()) {
                    // below dst: dst grows to m
                    const double below = gain + t[(size_t)dst].box.half_area();
                    stack.emplace_back(below, t[(size_t)dst].child[0]);
                    stack.emplace_back(below, t[(size_t)dst].child[1]);
                }
            }
            if (pivot == root) break;
            if (pivot != parent) {
                pivot_box.grow(t[(size_t)sibling].box);
                budget += t[(size_t)pivot].box.half_area() - pivot_box.half_area();   // the pivot shrinks
            }
            sibling = other_child(t[(size_t)pivot].parent, pivot);
            pivot = t[(size_t)pivot].parent;
        }
        return {best, best_gain};
    }
    void move(int32_t in, int32_t out)
    {
        const int32_t parent = t[(size_t)in].parent, up = t[(size_t)parent].parent;
        const int32_t sibling = other_child(parent, in);
        // the sibling takes the parent's place
        t[(size_t)up].child[t[(size_t)up].child[0] == parent ? 0 : 1] = sibling;
        t[(size_t)sibling].parent = up;
        // the freed node becomes the parent of `out` and `in`, where `out` was
        const int32_t above = t[(size_t)out].parent;
        t[(size_t)above].child[t[(size_t)above].child[0] == out ? 0 : 1] = parent;
        t[(size_t)parent].parent = above;
        t[(size_t)parent].child[0] = out;
        t[(size_t)parent].child[1] = in;
        t[(size_t)out].parent = parent;
        t[(size_t)in].parent = parent;
        refit_upwards(up);
        refit_upwards(parent);
    }
    // One pass over all nodes, largest first.  Returns the area saved.
    double pass()
    {
        std::vector<std::pair<double, int32_t>> order;
        for (size_t i = 0; i < t.size(); ++i)
            if ((int32_t)i != root && t[i].parent != root) order.emplace_back(-t[i].box.half_area(), (int32_t)i);
        std::sort(order.begin(), order.end());
        double saved = 0.0;
        for (const auto &o : order) {
            const int32_t in = o.second;
            if (t[(size_t)in].parent == root) continue;   // moved there meanwhile
            const auto r = find(in);
            if (r.first < 0 || !(r.second > 1e-9 * t[(size_t)root].box.half_area())) continue;
            move(in, r.first);
            saved += r.second;
        }
        return saved;
    }
    int depth_below(int32_t i) const
    {
        // iterative: the tree may be deep
        int deepest = 0;
        std::vector<std::pair<int32_t, int>> todo{{i, 0}};
        while (!todo.empty()) {
            const auto [n, d] = todo.back();
            todo.pop_back();
            if (t[(size_t)n].leaf()) { deepest = std::max(deepest, d); continue; }
            todo.emplace_back(t[(size_t)n].child[0], d + 1);
            todo.emplace_back(t[(size_t)n].child[1], d + 1);
        }
        return deepest;
    }
};

// Grid coordinate of world coordinate x, rounded outwards with one unit to spare: the
// traversal moves the ray onto the grid in f32 (error far below one unit for origins within
// kBvhFarOrigin units of the grid, see rtx_bvh_traverse.hpp), so a ray that enters the true
// box always enters the stored one.
uint32_t grid_lo(double x, double base, double inv_step)
{
    double g = std::floor((x - base) * inv_step) - 1.0;
    return (uint32_t)std::min((double)kBvhGridMax, std::max(0.0, g));
}
uint32_t grid_hi(double x, double base, double inv_step)
{
    double g = std::ceil((x - base) * inv_step) + 1.0;
    return (uint32_t)std::min((double)kBvhGridMax, std::max(0.0, g));
}

}  // namespace

static double spectral_norm(const double m[3][3])
{
    // power iteration on m^T m; the result is scaled up a little, so it is an upper bound in practice
    double v[3] = {0.57, 0.58, 0.59}, lambda = 0.0;
    for (int it = 0; it < 60; ++it) {
        double mv[3] = {0, 0, 0}, w[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) mv[i] += m[i][j] * v[j];
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) w[j] += m[i][j] * mv[i];
        lambda = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if (!(lambda > 0.0) || !std::isfinite(lambda)) return lambda;
        for (int j = 0; j < 3; ++j) v[j] = w[j] / lambda;
    }
    return std::sqrt(lambda) * 1.01;
}

bool pose_matrices(const rtx_prim &p, double a_inv[3][3], double &norm_a, double &norm_a_inv)
{
    const double ux = -(double)p.rotation[0], uy = -(double)p.rotation[1], uz = -(double)p.rotation[2];
    const double w = (double)p.rotation[3];
    double a[3][3];
    for (int j = 0; j < 3; ++j) {
        const double v[3] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
        const double t[3] = {2 * (uy * v[2] - uz * v[1]), 2 * (uz * v[0] - ux * v[2]), 2 * (ux * v[1] - uy * v[0])};
        a[0][j] = v[0] + w * t[0] + (uy * t[2] - uz * t[1]);
        a[1][j] = v[1] + w * t[1] + (uz * t[0] - ux * t[2]);
        a[2][j] = v[2] + w * t[2] + (ux * t[1] - uy * t[0]);
    }
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (!std::isfinite(det) || std::fabs(det) < 1e-12) return false;
    a_inv[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det;
    a_inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
    a_inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
    a_inv[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det;
    a_inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det;
    a_inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
    a_inv[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det;
    a_inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
    a_inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
    norm_a = spectral_norm(a);
    norm_a_inv = spectral_norm(a_inv);
    return std::isfinite(norm_a) && std::isfinite(norm_a_inv) && norm_a > 0.0 && norm_a_inv > 0.0;
}

void build_bvh(const rtx_prim *prims, uint32_t n, BvhHost &out)
{
    out.nodes.clear();
    out.planes.clear();
    out.depth = 0;
    Builder bld;
    // `reach` = the largest distance between a ray origin inside the node grid and an object
    // centre = the grid's diagonal, which itself depends on the boxes: iterate from the
    // geometric bounds (reach 0) until it stops growing
    double reach = 0.0;
    for (int pass = 0; pass < 32; ++pass) {
        Box3 all;
        all.clear();
        for (uint32_t i = 0; i < n; ++i) {
            Box3 b;
            double slope;
            if (prims[i].kind != RTX_PLANE && world_bounds(prims[i], reach, b, slope)) all.grow(b);
        }
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) d2 += all.hi[k] >= all.lo[k] ? (all.hi[k] - all.lo[k]) * (all.hi[k] - all.lo[k]) : 0.0;
        const double diagonal = std::sqrt(d2) * 1.05;   // the grid is a little larger than the union
        if (diagonal <= reach) break;                   // the boxes built with `reach` fit a grid within reach
        reach = diagonal * 1.25;
        // not converging (needle-thin ellipsoids whose zone grows faster than the scene): with an
        // infinite reach such objects fail world_bounds and join the always-tested list
        if (pass == 30) reach = std::numeric_limits<double>::infinity();
    }
    out.far_slope = (float)kBoxSlope;
    for (uint32_t i = 0; i < n; ++i) {
        Ref r;
        r.prim = (int32_t)i;
        double slope = 0.0;
        if (prims[i].kind == RTX_PLANE || !world_bounds(prims[i], reach, r.box, slope)) {
            out.planes.push_back((int32_t)i);
            continue;
        }
        out.far_slope = std::max(out.far_slope, (float)(slope * 1.001));
        for (int k = 0; k < 3; ++k) r.c[k] = 0.5 * (r.box.lo[k] + r.box.hi[k]);
        bld.refs.push_back(r);
    }
    if (bld.refs.empty()) return;

    int32_t root = bld.make(0, bld.refs.size(), 0);
    if (root < 0) {
        // a single object: one node whose first child is that leaf
        BuildNode nd;
        nd.box[0] = bld.refs[0].box;
        nd.box[1].clear();
        nd.ref[0] = root;
        nd.ref[1] = kBvhEmpty;
        bld.nodes.push_back(nd);
        root = (int32_t)bld.nodes.size() - 1;
    }
    // (up to a quarter of a million objects: the passes cost about 15 microseconds per object)
    if (RTX_BVH_OPT_PASSES > 0 && bld.refs.size() >= 8 && bld.refs.size() <= 262144) {
        // the same tree with parent links, optimised, and written back if it still fits the depth limit
        std::vector<TreeNode> t;
        t.reserve(2 * bld.refs.size());
        std::vector<Box3> prim_box(n);
        for (const Ref &r : bld.refs) prim_box[(size_t)r.prim] = r.box;
        struct Item { int32_t ref, parent; int slot; };
        std::vector<Item> todo{{root, -1, 0}};
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            const int32_t me = (int32_t)t.size();
            t.emplace_back();
            t[(size_t)me].parent = it.parent;
            if (it.parent >= 0) t[(size_t)it.parent].child[it.slot] = me;
            if (it.ref < 0) {
                t[(size_t)me].prim = ~it.ref;
                t[(size_t)me].box = prim_box[(size_t)~it.ref];
            } else {
                const BuildNode &nd = bld.nodes[(size_t)it.ref];
                t[(size_t)me].box = nd.box[0];
                if (nd.ref[1] != kBvhEmpty) t[(size_t)me].box.grow(nd.box[1]);
                todo.push_back({nd.ref[0], me, 0});
                todo.push_back({nd.ref[1], me, 1});
            }
        }
        TreeOptimizer opt{t, 0, {}};
        const double root_area = t[0].box.half_area();
        for (int pass = 0; pass < RTX_BVH_OPT_PASSES; ++pass)
            if (!(opt.pass() > 1e-4 * root_area)) break;
        const int depth = opt.depth_below(0);
        if (depth < kBvhMaxDepth) {
            // back into build nodes, the root first
            std::vector<BuildNode> nodes;
            std::vector<int32_t> index(t.size(), -1);
            for (size_t i = 0; i < t.size(); ++i)
                if (!t[i].leaf()) { index[i] = (int32_t)nodes.size(); nodes.emplace_back(); }
            for (size_t i = 0; i < t.size(); ++i) {
                if (t[i].leaf()) continue;
                BuildNode &nd = nodes[(size_t)index[i]];
                for (int c = 0; c < 2; ++c) {
                    const TreeNode &ch = t[(size_t)t[i].child[c]];
                    nd.box[c] = ch.box;
                    nd.ref[c] = ch.leaf() ? ~ch.prim : index[(size_t)t[i].child[c]];
                }
            }
            bld.nodes.swap(nodes);
            bld.depth = depth;
            root = 0;
        }
    }
    // breadth-first renumbering
    std::vector<int32_t> order, new_index(bld.nodes.size(), -1);
    std::queue<int32_t> q;
    q.push(root);
    while (!q.empty()) {
        int32_t i = q.front();
        q.pop();
        new_index[(size_t)i] = (int32_t)order.size();
        order.push_back(i);
        for (int c = 0; c < 2; ++c)
            if (bld.nodes[(size_t)i].ref[c] >= 0) q.push(bld.nodes[(size_t)i].ref[c]);
    }
    // the grid: all boxes, two units of slack at either end so that the outward rounding
    // never has to clamp
    Box3 all;
    all.clear();
    for (const Ref &r : bld.refs) all.grow(r.box);
    double base[3], inv_step[3];
    for (int a = 0; a < 3; ++a) {
        double ext = std::max(all.hi[a] - all.lo[a], 1e-30);
        float step = (float)(ext / (double)(kBvhGridMax - 8));
        float fbase = (float)(all.lo[a] - 4.0 * (double)step);
        // f32 values are what the device sees; make sure they still cover the range
        while ((all.hi[a] - (double)fbase) / (double)step > (double)(kBvhGridMax - 3)) step = std::nextafterf(step, INFINITY);
        float finv = 1.0f / step;
        out.grid_base[a] = fbase;
        out.grid_inv_step[a] = finv;
        out.grid_step[a] = step;
        base[a] = (double)fbase;
        inv_step[a] = (double)finv;
    }
    out.nodes.resize(order.size() * kBvhNodeVec);
    for (size_t k = 0; k < order.size(); ++k) {
        const BuildNode &nd = bld.nodes[(size_t)order[k]];
        uint32_t lo[2][3] = {{0, 0, 0}, {0, 0, 0}}, hi[2][3] = {{0, 0, 0}, {0, 0, 0}};
        int32_t ref[2];
        for (int c = 0; c < 2; ++c) {
            ref[c] = nd.ref[c] >= 0 ? new_index[(size_t)nd.ref[c]] : nd.ref[c];
            if (ref[c] == kBvhEmpty) continue;
            for (int a = 0; a < 3; ++a) {
                lo[c][a] = grid_lo(nd.box[c].lo[a], base[a], inv_step[a]);
                hi[c][a] = grid_hi(nd.box[c].hi[a], base[a], inv_step[a]);
            }
        }
        auto fb = [](uint32_t q) { float f = (float)q; uint32_t b; std::memcpy(&b, &f, 4); return b; };
        out.nodes[4 * k + 0] = make_uint4(fb(lo[0][0]), fb(hi[0][0]), fb(lo[0][1]), fb(hi[0][1]));
        out.nodes[4 * k + 1] = make_uint4(fb(lo[0][2]), fb(hi[0][2]), (uint32_t)ref[0], (uint32_t)ref[1]);
        out.nodes[4 * k + 2] = make_uint4(fb(lo[1][0]), fb(hi[1][0]), fb(lo[1][1]), fb(hi[1][1]));
        out.nodes[4 * k + 3] = make_uint4(fb(lo[1][2]), fb(hi[1][2]), 0u, 0u);
    }
    out.depth = bld.depth;
}

}  // namespace rtx
