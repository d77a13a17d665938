// rtx_wavefront.h -- path state of the wavefront variant (see rtx_wavefront.hip).
#pragma once
#include "rtx_device.h"

namespace rtx {

// Structure-of-arrays path state in device memory, one entry per slot (workgroup * 4096 +
// workgroup-local pixel number).
struct WfState {
    float *ray[6];     // origin xyz, direction xyz
    float *thr[3];     // throughput
    float *rad[3];     // radiance gathered by the current sample
    uint32_t *depth;
    uint32_t *sample;  // global index of the sample the pixel is working on
    float *hit_t;      // closest hit of the current segment
    int32_t *hit_idx;
};

size_t wavefront_state_bytes(uint32_t width, uint32_t height);
void wavefront_bind_state(void *block, uint32_t width, uint32_t height, WfState &S);
// Renders samples [P.sample_first, +P.sample_count) into P.out_sum: one asynchronous launch.
hipError_t launch_wavefront(const RenderParams &P, const WfState &S, int src, bool stats, hipStream_t stream);

}  // namespace rtx
