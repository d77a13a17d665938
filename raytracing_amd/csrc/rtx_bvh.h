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
//   f[3] = ref0 ref1 - - (int bits): ref >= 0 -> inner node index; ref < 0 -> leaf holding
//                                    object ~ref; kBvhEmpty -> no child
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

struct BvhHost {
    std::vector<float4> nodes;       // 4 float4 per node
    std::vector<int32_t> planes;     // unbounded (or unboundable) objects, tested by every ray
    int depth = 0;
};

void build_bvh(const rtx_prim *prims, uint32_t n, BvhHost &out);

}  // namespace rtx
