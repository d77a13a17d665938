// rtx_bvh.h -- host-built bounding volume hierarchy over the bounded objects.
#pragma once
#include "../../include/rtx.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace rtx {

struct BvhHost {
    std::vector<float4> nodes;       // 2 float4 per node
    std::vector<int32_t> prim_index; // leaf slot -> object index
    std::vector<int32_t> planes;     // unbounded objects, tested for every ray
};

void build_bvh(const rtx_prim *prims, uint32_t n, BvhHost &out);

}  // namespace rtx
