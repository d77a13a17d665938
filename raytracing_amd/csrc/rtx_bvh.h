// rtx_bvh.h -- host-built bounding volume hierarchy over the bounded objects.
//
// The reference scans every object for every ray (src/trace.rs:109-120).  For scenes of
// thousands of objects the scan is replaced by a binary BVH over the ellipsoids and boxes;
// planes are unbounded and stay in a list that every ray tests.  The traversal returns
// exactly what the scan returns: the smallest distance, ties going to the lowest object
// index.  A leaf box bounds not the geometric object but the zone in which the reference's f32
// object test can report a hit (rtx_bvh.cpp: world_bounds): far from an ellipsoid that test
// "hits" rays passing well outside it, and the scan would return those hits.
//
// Node layout (64 bytes, 4 x 16 bytes; both children's boxes live in the parent so one visit
// decides both).  Box corners are coordinates on a 16-bit grid that spans all bounded objects --
// world x = grid_base + q / grid_inv_step -- stored as the f32 values of those integers: the
// traversal is bound by its arithmetic, and unpacking 16-bit corners cost twelve conversions
// per visit (32-byte nodes: 10 000 objects 160 ms per frame, 64-byte nodes 156 ms).  The grid
// only has to be conservative, which the builder ensures by rounding outwards with one grid
// unit to spare.
//   v[0] = lo0.x, hi0.x, lo0.y, hi0.y     v[1] = lo0.z, hi0.z, ref0, ref1
//   v[2] = lo1.x, hi1.x, lo1.y, hi1.y     v[3] = lo1.z, hi1.z, -, -
//   ref (int bits) >= 0 -> inner node index; ref < 0 -> leaf holding object ~ref;
//   kBvhEmpty -> no child
// Every leaf holds exactly one object (an object test costs about five node tests, so
// larger leaves lose).  Nodes are stored in breadth-first order, so the first K nodes are
// the top levels.  The tree is at most kBvhMaxDepth deep: the builder switches from the
// surface-area heuristic to balanced median splits where a subtree could otherwise exceed
// it, so the traversal stack (one entry per level) cannot overflow.
#pragma once
#include "../../include/rtx.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace rtx {

constexpr int kBvhMaxDepth = 30;
constexpr int kBvhStackSize = 32;
constexpr int32_t kBvhEmpty = INT32_MIN;
constexpr int kBvhGridMax = 65535;      // largest grid coordinate
constexpr int kBvhNodeVec = 4;          // 16-byte words per node

struct BvhHost {
    std::vector<uint4> nodes;        // kBvhNodeVec per node
    float grid_base[3] = {0, 0, 0};  // world coordinate of grid coordinate 0, per axis
    float grid_inv_step[3] = {1, 1, 1};  // grid units per world unit
    float grid_step[3] = {1, 1, 1};      // world units per grid unit
    // Node boxes bound the zone in which the object tests can report a hit for ray origins
    // inside the grid; for an origin x world units outside it the zone is larger by at most
    // far_slope * x, which the traversal adds per ray.
    float far_slope = 0.0f;
    std::vector<int32_t> planes;     // unbounded (or unboundable) objects, tested by every ray
    int depth = 0;
};

void build_bvh(const rtx_prim *prims, uint32_t n, BvhHost &out);

// The linear map the object tests apply to (world point - position) and to world directions
// (src/objects/geometry.rs:24-27): v -> v + w t + u x t, t = 2 u x v with (u, w) the conjugate
// of the stored quaternion.  It is a rotation only for a unit quaternion; for any other it is a
// general matrix A, and the object in world space is A^-1 applied to the local shape.
// Returns false if A is singular or not finite; otherwise A^-1 and the spectral norms of both.
bool pose_matrices(const rtx_prim &p, double a_inv[3][3], double &norm_a, double &norm_a_inv);

}  // namespace rtx
