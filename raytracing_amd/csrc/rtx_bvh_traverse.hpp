// rtx_bvh_traverse.hpp -- closest hit through the host-built BVH (placeholder: linear scan
// until the traversal lands; the BVH source is rejected by rtx_set_scene meanwhile).
#pragma once
#include "rtx_trace.hpp"
namespace rtx {
RTX_DEV int closest_hit_bvh(const BvhView &, const Table &tab, V3 o, V3 d, float &t)
{
    return closest_hit_scan<false>(tab, o, d, t);
}
}  // namespace rtx
