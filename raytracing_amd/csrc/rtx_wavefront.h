// rtx_wavefront.h -- path state and queues of the wavefront variant (see rtx_wavefront.hip).
#pragma once
#include "rtx_device.h"

namespace rtx {

// Structure-of-arrays path state, one entry per pixel, plus the index queues.
struct WfState {
    float *ray[6];     // origin xyz, direction xyz
    float *thr[3];     // throughput
    float *rad[3];     // radiance gathered by the current sample
    uint32_t *depth;
    uint32_t *sample;  // global index of the sample the pixel is working on
    float *hit_t;      // closest hit of the current segment
    int32_t *hit_idx;
    uint32_t *queue[7];
    uint32_t *counts;  // queue lengths [7] + live-path count
};

size_t wavefront_state_bytes(uint32_t n_pixels);
void wavefront_bind_state(void *block, uint32_t n_pixels, WfState &S);
// Renders samples [P.sample_first, +P.sample_count) into P.out_sum; blocks until done.
hipError_t launch_wavefront(const RenderParams &P, const WfState &S, int src, bool stats, hipStream_t stream,
                            uint32_t *iterations_out);

}  // namespace rtx
