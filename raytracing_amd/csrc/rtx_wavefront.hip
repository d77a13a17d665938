// rtx_wavefront.hip -- wavefront variant of the trace loop (gfx950).
//
// The same per-pixel computation as k_megakernel, cut into stages that each run with
// (almost) every lane busy.  Path state lives in HBM as structure-of-arrays indexed by
// pixel: a pixel has at most one path in flight, so its samples are still accumulated in
// sample order and the result is bit-identical to the megakernel's.
//
// Queues hold pixel indices.  One iteration advances every live path by one segment:
//   prepare   reset the counters this iteration fills
//   generate  fresh queue  -> primary ray of the pixel's next sample -> active queue
//   extend    active queue -> closest hit; miss and depth cut finish the path here
//                             (-> next fresh queue); others -> queue of the hit's material
//   shade<M>  queue M      -> one material per launch: scatter; continuing paths -> next
//                             active queue, rejected diffuse samples -> next fresh queue
// Appending to a queue is a wave64 compaction: ballot of the predicate, prefix popcount
// below the lane, ONE atomicAdd per wave for the base offset.
// Stage kernels run on a fixed grid and stride over their queue in wave-sized chunks, so
// the host never needs a queue length to launch them; it reads the live-path count back
// once per batch of iterations to know when to stop.
#include "rtx_bvh_traverse.hpp"
#include "rtx_trace.hpp"
#include "rtx_wavefront.h"

namespace rtx {

extern __shared__ float4 rtx_smem[];

namespace {

enum : int { Q_ACTIVE0 = 0, Q_ACTIVE1, Q_FRESH0, Q_FRESH1, Q_DIFFUSE, Q_METALLIC, Q_DIELECTRIC, Q_COUNT };

RTX_DEV Table wf_table(const RenderParams &P, bool lds)
{
    Table t;
    t.n_prims = P.n_prims;
    t.n_lights = P.n_lights;
    if (lds) {
        const uint32_t ng = 3u * P.n_prims, nm = 3u * P.n_prims;
        for (uint32_t i = threadIdx.x; i < ng; i += blockDim.x) rtx_smem[i] = P.geom[i];
        for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) rtx_smem[ng + i] = P.mat[i];
        int32_t *sl = reinterpret_cast<int32_t *>(rtx_smem + ng + nm);
        for (uint32_t i = threadIdx.x; i < P.n_lights; i += blockDim.x) sl[i] = P.lights[i];
        __syncthreads();
        t.geom = rtx_smem;
        t.mat = rtx_smem + ng;
        t.lights = sl;
    } else {
        t.geom = P.geom;
        t.mat = P.mat;
        t.lights = P.lights;
    }
    return t;
}

// wave64 compaction: lanes with `pred` append `value` to queue q
RTX_DEV void wf_push(const WfState &S, int q, bool pred, uint32_t value)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int leader = __builtin_ctzll(mask);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&S.counts[q], (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (pred) S.queue[q][base + below] = value;
}

RTX_DEV Path wf_load_path(const WfState &S, uint32_t px)
{
    Path p;
    p.o = v3(S.ray[0][px], S.ray[1][px], S.ray[2][px]);
    p.d = v3(S.ray[3][px], S.ray[4][px], S.ray[5][px]);
    p.T = v3(S.thr[0][px], S.thr[1][px], S.thr[2][px]);
    p.L = v3(S.rad[0][px], S.rad[1][px], S.rad[2][px]);
    p.depth = S.depth[px];
    return p;
}
RTX_DEV void wf_store_ray(const WfState &S, uint32_t px, const Path &p)
{
    S.ray[0][px] = p.o.x; S.ray[1][px] = p.o.y; S.ray[2][px] = p.o.z;
    S.ray[3][px] = p.d.x; S.ray[4][px] = p.d.y; S.ray[5][px] = p.d.z;
}
RTX_DEV void wf_store_thr(const WfState &S, uint32_t px, const Path &p)
{
    S.thr[0][px] = p.T.x; S.thr[1][px] = p.T.y; S.thr[2][px] = p.T.z;
}
RTX_DEV void wf_store_rad(const WfState &S, uint32_t px, const Path &p)
{
    S.rad[0][px] = p.L.x; S.rad[1][px] = p.L.y; S.rad[2][px] = p.L.z;
}
// the path of pixel px is complete: add it to the pixel's sum, move to the next sample
RTX_DEV bool wf_finish_path(const RenderParams &P, const WfState &S, uint32_t px, V3 L)
{
    float *out = P.out_sum + 3ull * px;
    out[0] = out[0] + L.x;
    out[1] = out[1] + L.y;
    out[2] = out[2] + L.z;
    uint32_t s = S.sample[px] + 1u;
    S.sample[px] = s;
    return s < P.sample_first + P.sample_count;  // more samples to do
}

template <bool STATS>
RTX_DEV void wf_flush(const RenderParams &P, const Counters<STATS> &cnt)
{
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < C_COUNT; ++i)
            if (cnt.v[i]) atomicAdd(&P.stats[i], (unsigned long long)cnt.v[i]);
    }
}

#define WF_FOR_EACH_CHUNK(count_expr)                                                          \
    const uint32_t wf_count = (count_expr);                                                    \
    const uint32_t wf_lane = threadIdx.x & 63u;                                                \
    const uint32_t wf_wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                     \
    const uint32_t wf_nwaves = (gridDim.x * blockDim.x) >> 6;                                  \
    for (uint32_t wf_base = wf_wave * 64u; wf_base < wf_count; wf_base += wf_nwaves * 64u)

__global__ void __launch_bounds__(256) k_wf_init(const RenderParams P, const WfState S)
{
    const uint32_t n = P.width * P.height;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        S.sample[i] = P.sample_first;
        S.queue[Q_FRESH0][i] = i;
        P.out_sum[3ull * i + 0] = 0.0f;
        P.out_sum[3ull * i + 1] = 0.0f;
        P.out_sum[3ull * i + 2] = 0.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x < Q_COUNT)
        S.counts[threadIdx.x] = (threadIdx.x == Q_FRESH0 && P.sample_count > 0) ? n : 0u;
}

// iteration `it`: resets the queues this iteration fills, publishes the live-path count
__global__ void k_wf_prepare(const WfState S, uint32_t it)
{
    const uint32_t cur = it & 1u, nxt = cur ^ 1u;
    if (threadIdx.x == 0) {
        S.counts[Q_ACTIVE0 + nxt] = 0u;
        S.counts[Q_FRESH0 + nxt] = 0u;
        S.counts[Q_DIFFUSE] = 0u;
        S.counts[Q_METALLIC] = 0u;
        S.counts[Q_DIELECTRIC] = 0u;
        S.counts[Q_COUNT] = S.counts[Q_ACTIVE0 + cur] + S.counts[Q_FRESH0 + cur];  // live paths
    }
}

template <bool STATS>
__global__ void __launch_bounds__(256) k_wf_generate(const RenderParams P, const WfState S, uint32_t it)
{
    const uint32_t cur = it & 1u;
    Counters<STATS> cnt;
    cnt.clear();
    WF_FOR_EACH_CHUNK(S.counts[Q_FRESH0 + cur])
    {
        const uint32_t i = wf_base + wf_lane;
        const bool valid = i < wf_count;
        uint32_t px = 0;
        if (valid) {
            px = S.queue[Q_FRESH0 + cur][i];
            const uint32_t pi = px % P.width, py = px / P.width;
            const uint32_t s = S.sample[px];
            uint4 w = draw_block(P, px, s, 0u);
            Path p;
            camera_ray(P, pi, P.height - 1u - py, u24(w.x), u24(w.y), p.o, p.d);
            p.T = v3(1.0f, 1.0f, 1.0f);
            p.L = v3(0.0f, 0.0f, 0.0f);
            wf_store_ray(S, px, p);
            wf_store_thr(S, px, p);
            wf_store_rad(S, px, p);
            S.depth[px] = 0u;
            cnt.template add<C_SAMPLES>();
        }
        wf_push(S, Q_ACTIVE0 + cur, valid, px);
    }
    wf_flush<STATS>(P, cnt);
}

template <int SRC, bool STATS>
__global__ void __launch_bounds__(256) k_wf_extend(const RenderParams P, const WfState S, uint32_t it)
{
    Table tab = wf_table(P, SRC == kSrcLds);
    const uint32_t cur = it & 1u, nxt = cur ^ 1u;
    const V3 bg = v3(P.bg[0], P.bg[1], P.bg[2]);
    Counters<STATS> cnt;
    cnt.clear();
    WF_FOR_EACH_CHUNK(S.counts[Q_ACTIVE0 + cur])
    {
        const uint32_t i = wf_base + wf_lane;
        const bool valid = i < wf_count;
        uint32_t px = 0;
        int material = -1;
        bool refresh = false;
        if (valid) {
            px = S.queue[Q_ACTIVE0 + cur][i];
            Path p = wf_load_path(S, px);
            cnt.template add<C_RAYS>();
            float t;
            int idx;
            if constexpr (SRC == kSrcBvh) idx = closest_hit_bvh<STATS>(P.bvh, tab, p.o, p.d, t, cnt);
            else idx = closest_hit_scan<true>(tab, p.o, p.d, t);
            bool done;
            if (idx < 0) {
                p.L = p.L + cmul(p.T, bg);
                done = true;
            } else {
                if constexpr (STATS) {
                    int kind = __float_as_int(tab.geom[3 * idx + 1].w) & 0xff;
                    cnt.template add<C_HIT_PLANE>(kind == kPlane);
                    cnt.template add<C_HIT_ELLIPSOID>(kind == kEllipsoid);
                    cnt.template add<C_HIT_BOX>(kind == kBox);
                }
                done = surface_emission(P, tab, p, idx);
            }
            if (done) {
                refresh = wf_finish_path(P, S, px, p.L);
            } else {
                wf_store_rad(S, px, p);
                S.hit_t[px] = t;
                S.hit_idx[px] = idx;
                material = __float_as_int(tab.mat[3 * idx + 0].w);
            }
        }
        wf_push(S, Q_FRESH0 + nxt, refresh, px);
        wf_push(S, Q_DIFFUSE, material == kDiffuse, px);
        wf_push(S, Q_METALLIC, material == kMetallic, px);
        wf_push(S, Q_DIELECTRIC, material == kDielectric, px);
    }
    wf_flush<STATS>(P, cnt);
}

template <int SRC, int MATERIAL, bool STATS>
__global__ void __launch_bounds__(256) k_wf_shade(const RenderParams P, const WfState S, uint32_t it)
{
    Table tab = wf_table(P, SRC == kSrcLds);
    const uint32_t nxt = (it & 1u) ^ 1u;
    Counters<STATS> cnt;
    cnt.clear();
    WF_FOR_EACH_CHUNK(S.counts[Q_DIFFUSE + MATERIAL])
    {
        const uint32_t i = wf_base + wf_lane;
        const bool valid = i < wf_count;
        uint32_t px = 0;
        bool go_on = false, refresh = false;
        if (valid) {
            px = S.queue[Q_DIFFUSE + MATERIAL][i];
            Path p = wf_load_path(S, px);
            const bool ended = bounce<STATS>(P, tab, p, S.hit_idx[px], S.hit_t[px], MATERIAL, px, S.sample[px], cnt);
            if (ended) {
                refresh = wf_finish_path(P, S, px, p.L);
            } else {
                wf_store_ray(S, px, p);
                wf_store_thr(S, px, p);
                S.depth[px] = p.depth;
                go_on = true;
            }
        }
        wf_push(S, Q_ACTIVE0 + nxt, go_on, px);
        wf_push(S, Q_FRESH0 + nxt, refresh, px);
    }
    wf_flush<STATS>(P, cnt);
}

template <int SRC, bool STATS>
hipError_t run_iteration(const RenderParams &P, const WfState &S, uint32_t it, dim3 grid, size_t lds, hipStream_t st)
{
    const dim3 block(256);
    hipLaunchKernelGGL(k_wf_prepare, dim3(1), dim3(64), 0, st, S, it);
    hipLaunchKernelGGL((k_wf_generate<STATS>), grid, block, 0, st, P, S, it);
    hipLaunchKernelGGL((k_wf_extend<SRC, STATS>), grid, block, lds, st, P, S, it);
    hipLaunchKernelGGL((k_wf_shade<SRC, kDiffuse, STATS>), grid, block, lds, st, P, S, it);
    hipLaunchKernelGGL((k_wf_shade<SRC, kMetallic, STATS>), grid, block, lds, st, P, S, it);
    hipLaunchKernelGGL((k_wf_shade<SRC, kDielectric, STATS>), grid, block, lds, st, P, S, it);
    return hipGetLastError();
}

}  // namespace

size_t wavefront_state_bytes(uint32_t n_pixels)
{
    // 12 float arrays + depth + sample + hit_t + hit_idx + Q_COUNT queues, then the counters
    return (size_t)n_pixels * 4u * (12u + 4u + (size_t)Q_COUNT) + 256u;
}

void wavefront_bind_state(void *block, uint32_t n_pixels, WfState &S)
{
    uint32_t *base = static_cast<uint32_t *>(block);
    size_t k = 0;
    auto next = [&]() { uint32_t *p = base + k * (size_t)n_pixels; ++k; return p; };
    for (int i = 0; i < 6; ++i) S.ray[i] = reinterpret_cast<float *>(next());
    for (int i = 0; i < 3; ++i) S.thr[i] = reinterpret_cast<float *>(next());
    for (int i = 0; i < 3; ++i) S.rad[i] = reinterpret_cast<float *>(next());
    S.depth = next();
    S.sample = next();
    S.hit_t = reinterpret_cast<float *>(next());
    S.hit_idx = reinterpret_cast<int32_t *>(next());
    for (int i = 0; i < Q_COUNT; ++i) S.queue[i] = next();
    S.counts = next();
}

hipError_t launch_wavefront(const RenderParams &P, const WfState &S, int src, bool stats, hipStream_t stream,
                            uint32_t *iterations_out)
{
    const uint32_t n = P.width * P.height;
    const size_t lds = src == kSrcLds ? scene_lds_bytes(P.n_prims, P.n_lights) : 0;
    // fixed grid: stage kernels stride over their queues
    uint32_t blocks = (n + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    const dim3 grid(blocks);
    hipLaunchKernelGGL(k_wf_init, grid, dim3(256), 0, stream, P, S);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    // a pixel advances one segment per iteration: at most samples * (depth + 1) iterations
    const uint64_t max_iters = (uint64_t)P.sample_count * ((uint64_t)P.ray_depth + 1u) + 2u;
    const uint32_t batch = 32;
    uint32_t it = 0;
    uint32_t live = P.sample_count > 0 ? n : 0u;
    while (live != 0u && it < max_iters) {
        for (uint32_t k = 0; k < batch; ++k, ++it) {
            if (src == kSrcLds) e = stats ? run_iteration<kSrcLds, true>(P, S, it, grid, lds, stream)
                                          : run_iteration<kSrcLds, false>(P, S, it, grid, lds, stream);
            else if (src == kSrcScalar) e = stats ? run_iteration<kSrcScalar, true>(P, S, it, grid, lds, stream)
                                                  : run_iteration<kSrcScalar, false>(P, S, it, grid, lds, stream);
            else e = stats ? run_iteration<kSrcBvh, true>(P, S, it, grid, lds, stream)
                           : run_iteration<kSrcBvh, false>(P, S, it, grid, lds, stream);
            if (e != hipSuccess) return e;
        }
        // live-path count as published by the last prepare of the batch, plus what that
        // iteration produced: read the queue counters themselves
        uint32_t counts[Q_COUNT + 1];
        e = hipMemcpyAsync(counts, S.counts, sizeof(counts), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        const uint32_t nxt = it & 1u;  // the queues the last iteration filled
        live = counts[Q_ACTIVE0 + nxt] + counts[Q_FRESH0 + nxt];
    }
    if (iterations_out) *iterations_out = it;
    return hipSuccess;
}

}  // namespace rtx
