// rtx_wavefront.h -- path state of the wavefront variant (see rtx_wavefront.hip).
#pragma once
#include "rtx_device.h"

namespace rtx {

// Path state: one 64-byte record per pixel a workgroup owns, in LDS when the workgroup's
// queues, records and object table or traversal stacks fit there (wavefront_state_in_lds), else in
// device memory at slot = workgroup * (pixels per workgroup) + workgroup-local pixel number, a whole
// cache line per path:
//   rec[4*slot+0] = ray origin xyz      | distance of the current segment's closest hit
//   rec[4*slot+1] = ray direction xyz   | object index of that hit (int bits)
//   rec[4*slot+2] = throughput xyz      | depth (int bits)
//   rec[4*slot+3] = radiance so far xyz | global index of the sample in flight (int bits)
struct WfState {
    float4 *rec;
};

// true: the launch keeps the records in LDS and needs no state block
bool wavefront_state_in_lds(const RenderParams &P, int src);
size_t wavefront_state_bytes(uint32_t width, uint32_t height);
void wavefront_bind_state(void *block, uint32_t width, uint32_t height, WfState &S);
// Renders samples [P.sample_first, +P.sample_count) into P.out_sum: one asynchronous launch.
hipError_t launch_wavefront(const RenderParams &P, const WfState &S, int src, bool stats, hipStream_t stream);
// 8x8 tiles (consecutive in row-major order) a workgroup of the variant chosen for (P, src) owns: its work unit
uint32_t wavefront_group_tiles(const RenderParams &P, int src);

}  // namespace rtx
