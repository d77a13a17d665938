/*
 * rtx.h -- C ABI of librtx, the MI355X (gfx950) implementation of the reference
 * path tracer's per-pixel trace loop.
 *
 * The reference has no plugin or FFI interface.  The seam this library replaces
 * is the Rust function `fn render(scene: &mut Scene)` (src/main.rs:13-31), which
 * `main` calls once between `parse_scene` and `color_correction`
 * (src/main.rs:37-41): input = the parsed Scene (src/parser.rs:12-24), output =
 * linear mean radiance per pixel in Image.data (src/image.rs:6-29).
 *
 * All structs are plain data, little-endian, IEEE binary32.  The caller owns
 * every pointer it passes; the library copies what it needs during the call and
 * keeps no caller pointer.  No function aborts or throws: each returns
 * RTX_OK (0) or a negative rtx_status, and rtx_last_error() gives a message for
 * the calling thread's last failure.  One rtx_ctx belongs to one host thread at
 * a time.  It drives one GPU, or -- rtx_opts.n_devices > 1 -- several GPUs of one
 * node from that one thread: the frame is dealt out to them in small blocks of pixels and
 * put together on the devices (see rtx_opts).  Renders of one context share device scratch
 * (path state, tile order): give them one stream, or wait for one to end before the
 * next starts on another stream; rtx_set_scene waits for the context's device(s) to
 * go idle before it overwrites the scene a render may still be reading.
 */
#ifndef RTX_H
#define RTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RTX_OK = 0,
    RTX_ERR_INVALID_ARG = -1,   /* null pointer, bad enum, bad range */
    RTX_ERR_INVALID_SCENE = -2, /* a scene the reference would panic on (see rtx_set_scene) */
    RTX_ERR_DEVICE = -3,        /* a HIP call failed */
    RTX_ERR_NO_SCENE = -4,      /* render entry point called before rtx_set_scene */
    RTX_ERR_UNSUPPORTED = -5,   /* combination of options not available */
    RTX_ERR_NO_MEMORY = -6
} rtx_status;

/* FigureType, src/parser.rs:43-47 / src/objects/figures.rs:4-17 */
typedef enum { RTX_PLANE = 0, RTX_ELLIPSOID = 1, RTX_BOX = 2 } rtx_kind;
/* Material, src/objects/object.rs:5-9 */
typedef enum { RTX_DIFFUSE = 0, RTX_METALLIC = 1, RTX_DIELECTRIC = 2 } rtx_material;

/* One Object<PositionedFigure<_>>: src/objects/object.rs:11-17 and
 * src/objects/figures.rs:19-23.  Defaults as in object.rs:19-27 / figures.rs:25-33:
 * position 0, rotation (0,0,0,1), color 0, emission 0, diffuse. */
typedef struct rtx_prim {
    uint32_t kind;       /* rtx_kind */
    float param[3];      /* plane normal (not normalised) | ellipsoid radii | box half-sizes */
    float position[3];
    float rotation[4];   /* unit quaternion x, y, z, w */
    float color[3];
    float emission[3];
    uint32_t material;   /* rtx_material */
    float ior;           /* used when material == RTX_DIELECTRIC */
} rtx_prim;

/* Scene, src/parser.rs:12-24, with Camera (src/camera.rs:6-12) flattened.
 * cam_axis = the columns right, up, forward (src/parser.rs:64);
 * tg_fov_x = tan(fov_x / 2), tg_fov_y = (height / width) * tg_fov_x (src/parser.rs:53-55).
 * The lights list of the Scene (src/parser.rs:69-89) is derived inside the library. */
typedef struct rtx_scene {
    uint32_t width, height;
    uint32_t ray_depth;
    uint32_t n_samples;
    float background[3];
    float cam_pos[3];
    float cam_axis[9];
    float tg_fov_x, tg_fov_y;
    uint32_t n_prims;
    const rtx_prim *prims;
} rtx_scene;

typedef enum {
    RTX_MODE_MEGAKERNEL = 0, /* one lane per pixel, sample and bounce loops in-kernel */
    RTX_MODE_WAVEFRONT = 1   /* staged kernels over compacted ray queues */
} rtx_mode;
typedef enum {
    RTX_ACCEL_AUTO = 0,      /* linear scan up to RTX_AUTO_BRUTE_MAX ellipsoids + boxes, BVH above */
    RTX_ACCEL_BRUTE_LDS = 1, /* linear scan, object table staged in LDS */
    RTX_ACCEL_BRUTE_SCALAR = 2, /* linear scan, object table read through the scalar cache */
    RTX_ACCEL_BVH = 3        /* host-built BVH over ellipsoids and boxes + plane list */
} rtx_accel;
typedef enum {
    RTX_RNG_PHILOX = 0, /* Philox4x32-10 block per (pixel, sample, slot), key = seed */
    RTX_RNG_CONST = 1   /* every draw block returns const_words (zero-randomness shading) */
} rtx_rng;

#define RTX_MAX_LDS_PRIMS 512u   /* largest object table RTX_ACCEL_BRUTE_LDS accepts */
#define RTX_AUTO_BRUTE_MAX 6u    /* RTX_ACCEL_AUTO: most non-plane objects the linear scan is chosen for */

/* How the per-GPU radiance sums of a multi-GPU context are added. */
typedef enum {
    RTX_REDUCE_P2P = 0,  /* hand-written slice reduce over peer access (xGMI): GPU g adds slice g of
                            every GPU's buffer in device order, divides, and sends its slice to the host */
    RTX_REDUCE_RCCL = 1  /* ncclAllReduce(sum) from librccl.so, one communicator rank per GPU in this
                            process; GPU 0 divides and sends the image to the host */
} rtx_reduce;

#define RTX_MAX_DEVICES 16

typedef struct rtx_opts {
    uint64_t seed;
    uint32_t mode;           /* rtx_mode */
    uint32_t accel;          /* rtx_accel */
    uint32_t rng_mode;       /* rtx_rng */
    uint32_t const_words[4];
    int32_t device;          /* HIP device ordinal; -1 = the current device (used when n_devices <= 1) */
    uint32_t collect_stats;  /* 1 = count path events (slower kernels; see rtx_get_stats) */
    /* Several GPUs behind one context.  The frame is cut into the kernels' work units (blocks of
     * 8 x 8 or 16 x 8 pixels, or a few consecutive ones), dealt out round-robin: GPU g renders every
     * sample step (src/main.rs:14) of units g, g + n_devices, ... into its own sum buffer, which is
     * zero elsewhere, on its own stream, and the buffers are added as `reduce` says -- all inside
     * rtx_render, rtx_render_device, rtx_render_bytes and rtx_progress_*, which keep their
     * signatures.  A pixel's samples are traced and added on one GPU, in order, and adding zeros
     * changes nothing: the image has the bits of a one-GPU render whatever n_devices is.
     * n_devices 0 or 1: one GPU, `device`.  devices = NULL: ordinals 0 .. n_devices-1.  An ordinal
     * may repeat (several shares on one GPU; RTX_REDUCE_P2P only).  Entry points that take DEVICE
     * pointers (rtx_render_samples, rtx_accumulate_samples, rtx_divide, rtx_tonemap_bytes) and
     * rtx_primary run on devices[0] alone. */
    int32_t n_devices;
    uint32_t reduce;         /* rtx_reduce */
    const int32_t *devices;  /* n_devices HIP ordinals, copied by rtx_create */
} rtx_opts;

/* Path events of the most recent render call, counted on the device when
 * collect_stats = 1.  Used to turn a run into algorithmic FLOPs. */
typedef struct rtx_stats {
    uint64_t samples;          /* (pixel, step) iterations, src/main.rs:14-28 */
    uint64_t rays;             /* closest-hit queries, src/trace.rs:16 */
    uint64_t hits[3];          /* closest hits by rtx_kind */
    uint64_t cosine_samples;   /* src/random.rs:33-53 */
    uint64_t light_samples[3]; /* src/random.rs:73-81 by light kind */
    uint64_t pdf_evals;        /* src/random.rs:143-152 */
    uint64_t light_pdf[3];     /* per (pdf eval, light): 0 / 1 / 2 intersections found */
    uint64_t diffuse_bounces;
    uint64_t metallic_bounces;
    uint64_t dielectric_bounces;
    uint64_t prim_tests[3];    /* object intersect calls in closest-hit queries by kind */
    uint64_t light_tests[3];   /* light intersect calls in light-pdf evaluation by kind */
    uint64_t bvh_node_tests;   /* node-box tests of the BVH traversal (0 for the linear scan) */
} rtx_stats;

typedef struct rtx_ctx rtx_ctx;

/* Context life cycle.  No counterpart in the reference (its state is the Scene). */
int rtx_create(rtx_ctx **out, const rtx_opts *opts);
void rtx_destroy(rtx_ctx *ctx);

/* Validate, copy and upload a scene; derive the lights (src/parser.rs:69-89) and,
 * when the accel option asks for it, build the BVH.  Replaces the hand-over of
 * `Scene` from parse_scene to render (src/main.rs:37-38).
 * RTX_ERR_INVALID_SCENE where the reference would panic: width/height 0, an
 * emissive box with a half-size <= 0 (src/objects/sample.rs:30,46 gen_range on an empty
 * range).
 * The first render of 64 or more samples of a large frame after this call may spend a short extra
 * launch on measuring what each block of the frame costs, so that this and later renders of the scene
 * can start the expensive blocks first; that changes when a pixel is rendered, never its value. */
int rtx_set_scene(rtx_ctx *ctx, const rtx_scene *scene);

/* Drop-in for `render(&mut scene)` (src/main.rs:13-31): renders scene->n_samples
 * samples per pixel and writes the mean linear radiance to rgb (host memory,
 * width*height*3 floats, row 0 = TOP row as src/image.rs:21-29 stores it). */
int rtx_render(rtx_ctx *ctx, const rtx_scene *scene, float *rgb);

/* The sample loop of src/main.rs:14 restricted to global sample indices
 * [sample_first, sample_first + sample_count) of the scene given to
 * rtx_set_scene.  Writes (not adds) the per-pixel SUM of radiance to d_sum, a
 * DEVICE buffer of width*height*3 floats, asynchronously on hip_stream
 * (a hipStream_t; NULL = the null stream).  Used to cut the loop into steps, and -- with
 * rtx_set_share -- to spread a frame over the GPUs of several processes. */
int rtx_render_samples(rtx_ctx *ctx, uint32_t sample_first, uint32_t sample_count,
                       float *d_sum, void *hip_stream);

/* One process per GPU: share `share_index` of `share_count` of every frame.  No counterpart in
 * the reference (its loop runs in one process).  From this call on rtx_render_samples and
 * rtx_render_samples_static write the sums of this share's pixels -- the kernels' work units
 * share_index, share_index + share_count, ... (blocks of 8 x 8 or 16 x 8 pixels, or a few consecutive
 * ones, depending on the kernel chosen for the scene) -- and ZERO everywhere else, and
 * rtx_accumulate_samples continues this share's pixels and leaves the others as they are.  The
 * share_count buffers of contexts with the same options and scene, added up (an all-reduce), are
 * the frame, with the bits of one context rendering it alone: every pixel's samples are traced
 * and added by one share, in order.  rtx_get_stats then counts this share's events, and the
 * counters of all shares add up to the one-context numbers.  (0, 1) = the whole frame, the
 * default.  RTX_ERR_UNSUPPORTED on a multi-GPU context, which deals the frame out itself. */
int rtx_set_share(rtx_ctx *ctx, uint32_t share_index, uint32_t share_count);

/* rtx_render_samples for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the
 * reference.  Same arguments, same output bits as rtx_render_samples, but no device or pinned
 * memory is allocated or freed on any path: the launch tables of the megakernel (tile costs,
 * dispatch order, work queue) live in a static device array of RTX_STATIC_MAX_TILES * 12 bytes
 * (3 MiB) that every GPU the library runs on holds until the process ends.  RTX_ERR_UNSUPPORTED for an image of more
 * than RTX_STATIC_MAX_TILES tiles of 8 x 8 pixels (e.g. beyond 4096 x 4096) and for the wavefront
 * variant on a scene whose BVH is deep (its path state is then a device block), and on a context
 * with collect_stats = 1 (reading counters back waits for the stream).  The call records no timing
 * events, so rtx_last_kernel_ms does not cover it; nothing it does is illegal on a capturing stream.
 * All contexts on one GPU share that GPU's array: calls that overlap in time on one GPU race,
 * whatever their streams -- replays of graphs that hold such a call included. */
#define RTX_STATIC_MAX_TILES 262144u
int rtx_render_samples_static(rtx_ctx *ctx, uint32_t sample_first, uint32_t sample_count,
                              float *d_sum, void *hip_stream);

/* As rtx_render_samples, but the per-pixel sums CONTINUE from the values already in
 * d_sum: calls over [0, a) and then [a, b) leave the bits one call over [0, b)
 * writes.  src/main.rs:23-27 keeps the image a valid mean after every sample step so
 * the loop can stop anywhere; this is that property for a sum kept on the device. */
int rtx_accumulate_samples(rtx_ctx *ctx, uint32_t sample_first, uint32_t sample_count,
                           float *d_sum, void *hip_stream);

/* d_rgb[i] = d_rgb[i] / divisor on the device: turns summed radiance into the
 * mean that src/main.rs:26 maintains. */
int rtx_divide(rtx_ctx *ctx, float *d_rgb, uint64_t n_floats, float divisor, void *hip_stream);

/* The output stage of src/main.rs:40-41 on the device: for each of n_floats values of
 * d_sum, x = d_sum[i] / divisor, the ACES curve and saturate (src/image.rs:65-79), gamma
 * 1/2.2 (src/image.rs:60-63) and the byte conversion of src/image.rs:44 into d_bytes
 * (DEVICE buffers; d_sum is not modified).  Float operations and their order are the
 * reference's; powf comes from the device library, so a byte may differ by one from
 * rtx_host_color_correction + rtx_host_to_bytes where the value sits on a rounding
 * boundary. */
int rtx_tonemap_bytes(rtx_ctx *ctx, const float *d_sum, uint64_t n_floats, float divisor,
                      uint8_t *d_bytes, void *hip_stream);

/* rtx_render (src/main.rs:13-31) with the image left on the device: renders n_samples samples per
 * pixel of the scene given to rtx_set_scene and writes the mean linear radiance to d_rgb, a DEVICE
 * buffer of width*height*3 floats (row 0 = top) on the context's first GPU, which the caller has
 * finished writing to.  A multi-GPU context deals the frame out, and GPU g adds slice g of all sums,
 * divides and writes that slice straight into d_rgb (peer access), or -- RTX_REDUCE_RCCL -- the
 * first GPU copies the all-reduced image there.  Same bits as rtx_render.  Returns when the image
 * is complete. */
int rtx_render_device(rtx_ctx *ctx, uint32_t n_samples, float *d_rgb);

/* render + color_correction + the byte conversion of write (src/main.rs:38-41) without
 * the float image leaving the device: bytes = width*height*3 values of the PPM body
 * (host memory, row 0 = top). */
int rtx_render_bytes(rtx_ctx *ctx, const rtx_scene *scene, uint8_t *bytes);

/* Progressive rendering on a sum buffer the context owns (stop / checkpoint / resume;
 * the reference's loop is interruptible after any sample step, src/main.rs:14-28).
 *   begin: set the scene; start from zero samples (sum = NULL, samples_done = 0) or
 *          from a saved state (sum = width*height*3 host floats, the sum of samples
 *          [0, samples_done)).
 *   step:  render the next sample_count samples onto the sum; returns when done.
 *   read:  any of rgb_mean (sum / samples, what rtx_render returns), sum (the state to
 *          save), bytes (tone-mapped, as rtx_render_bytes) and samples_done may be NULL.
 * After begin + steps totalling n samples, read returns the bits of rtx_render with n_samples = n,
 * however the steps were cut, wherever a save / resume happened and however many GPUs the context
 * drives: a resumed sum continues, pixel by pixel, on the GPU that owns the pixel.  (One exception:
 * a saved value of -0 comes back from a multi-GPU context as +0.)  read adds the GPUs' buffers
 * once, whatever combination of outputs is asked for.
 * rtx_render, rtx_render_device and rtx_render_bytes on the same context end the progressive render. */
int rtx_progress_begin(rtx_ctx *ctx, const rtx_scene *scene, const float *sum, uint32_t samples_done);
int rtx_progress_step(rtx_ctx *ctx, uint32_t sample_count);
int rtx_progress_read(rtx_ctx *ctx, float *rgb_mean, float *sum, uint8_t *bytes, uint32_t *samples_done);

/* Deterministic primary-visibility pass: one ray through every pixel centre
 * (src/main.rs:19-21 with du = dv = 0.5) and its closest hit (src/trace.rs:102-121), on the
 * context's first GPU.
 * Host output arrays of width*height entries (normal: *3), row 0 = top; any may
 * be NULL.  hit_id = object index or -1. */
int rtx_primary(rtx_ctx *ctx, const rtx_scene *scene, int32_t *hit_id, float *t,
                float *normal, uint8_t *inside);

/* Event counters of the last render on this context (needs collect_stats = 1). */
int rtx_get_stats(rtx_ctx *ctx, rtx_stats *out);

/* Milliseconds the device spent in the kernels of the last rtx_render /
 * rtx_render_samples call, from HIP events on that call's stream.  Waits for
 * the call to finish. */
int rtx_last_kernel_ms(rtx_ctx *ctx, float *ms);

/* Number of lights derived from the current scene, and their object indices. */
int rtx_get_lights(rtx_ctx *ctx, int32_t *indices, int32_t capacity);

/* Page-locked host memory for the images rtx_render, rtx_render_bytes and rtx_progress_read write.
 * No counterpart in the reference (its image is a Vec, src/image.rs:7-10).  Any host memory is
 * accepted by those calls; into pageable memory the runtime stages the device-to-host copies one
 * after the other, into memory from here the GPUs of a multi-GPU context deliver their slices of the
 * image side by side.  *out = NULL for 0 bytes. */
int rtx_alloc_pinned(void **out, uint64_t bytes);
void rtx_free_pinned(void *p);

/* Number of GPUs the context drives (1 unless rtx_opts.n_devices asked for more). */
int rtx_device_count(rtx_ctx *ctx);

/* Message for the calling thread's most recent error ("" if none). */
const char *rtx_last_error(void);

/* ABI version of this header: bumped on any layout or signature change. */
uint32_t rtx_abi_version(void);
#define RTX_ABI_VERSION 5u

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
