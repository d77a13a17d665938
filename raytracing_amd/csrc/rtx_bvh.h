// rtx_bvh.h -- host-built bounding volume hierarchy over the bounded objects.
//
// The reference scans every object for every ray (src/trace.rs:109-120).  For scenes of
// thousands of objects the scan is replaced by a binary BVH over the ellipsoids and boxes;
// planes are unbounded and stay in a list that every ray tests.  The traversal returns
// exactly what the scan returns: the smallest distance, ties going to the lowest object
// index.  Node boxes are padded so that rounding in the exact object test can never report
// a hit outside its node.
//
// Node layout (64 bytes, 4 x float4; both children's boxes live in the parent so one
// fetch decides both):
//   f[0] = lo0.x lo0.y lo0.z hi0.x
//   f[1] = hi0.y hi0.z lo1.x lo1.y
//   f[2] = lo1.z hi1.x hi1.y hi1.z
//   f[3] = c0 n0 c1 n1 (int bits): n > 0 -> leaf slots [c, c+n) of prim_index;
//                                  n = 0 -> inner node c (c >= 0) or empty (c < 0)
// Nodes are stored in breadth-first order, so the first K nodes are the top levels.
#pragma once
#include "../../include/rtx.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace rtx {

constexpr int kBvhMaxDepth = 28;   // the traversal stack holds one entry per level
constexpr int kBvhStackSize = 32;
constexpr int kBvhLeafSize = 4;

struct BvhHost {
    std::vector<float4> nodes;       // 4 float4 per node
    std::vector<int32_t> prim_index; // leaf slot -> object index
    std::vector<int32_t> planes;     // unbounded (or unboundable) objects, tested by every ray
    int depth = 0;
};

void build_bvh(const rtx_prim *prims, uint32_t n, BvhHost &out);

}  // namespace rtx
