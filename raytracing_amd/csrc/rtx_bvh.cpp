// rtx_bvh.cpp -- BVH construction (host).
#include "rtx_bvh.h"

namespace rtx {

void build_bvh(const rtx_prim *, uint32_t, BvhHost &out)
{
    out.nodes.clear();
    out.prim_index.clear();
    out.planes.clear();
}

}  // namespace rtx
