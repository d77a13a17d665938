/*
 * rt_oracle.h -- CPU oracle for the per-pixel trace loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker that the HIP path is
 * compared with; nothing shipped may link, import or call it.  Only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py use it.
 *
 * PARITY UNPINNED: the reference (a Rust crate) ships no tests, golden vectors
 * or expected outputs and cannot be compiled where this oracle is built, and its
 * random stream (an OS-seeded ThreadRng shared by all pixels) is not
 * reproducible.  The oracle is therefore a scalar, single-thread restatement of
 * the reference's source, function by function, pinned only by analytic
 * known-answer vectors derived from the reference's formulas.
 *
 * Every function in rt_oracle.c cites the reference file:line it follows.
 * All arithmetic is IEEE binary32, evaluated in the order the reference's
 * expressions (and nalgebra's small-vector kernels) evaluate it; the library is
 * built with -ffp-contract=off so that no multiply-add is fused.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* primitive kinds: parser.rs:43-47 */
enum { ORC_PLANE = 0, ORC_ELLIPSOID = 1, ORC_BOX = 2 };
/* materials: objects/object.rs:5-9 */
enum { ORC_DIFFUSE = 0, ORC_METALLIC = 1, ORC_DIELECTRIC = 2 };

/* One scene object: objects/object.rs:11-17 + objects/figures.rs:4-23.
 * param = plane normal | ellipsoid radii | box half-sizes.
 * rotation = unit quaternion (x, y, z, w). */
typedef struct {
    uint32_t kind;
    float param[3];
    float position[3];
    float rotation[4];
    float color[3];
    float emission[3];
    uint32_t material;
    float ior;
} orc_prim;

/* The render input: parser.rs:12-24 with the camera of camera.rs:6-12.
 * cam_axis holds the columns right, up, forward (parser.rs:64). */
typedef struct {
    uint32_t width, height, ray_depth, n_samples;
    float background[3];
    float cam_pos[3];
    float cam_axis[9];
    float tg_fov_x, tg_fov_y;
    uint32_t n_prims;
    const orc_prim *prims;
} orc_scene;

enum { ORC_RNG_PHILOX = 0, ORC_RNG_CONST = 1 };
/* ORC_ORDER_REFERENCE: recursion and f32 running mean exactly as main.rs:13-31
 * and trace.rs:11-71 evaluate them.
 * ORC_ORDER_FORWARD: the same paths accumulated front to back
 * (radiance += throughput * emission), per-pixel sum divided once. */
enum { ORC_ORDER_REFERENCE = 0, ORC_ORDER_FORWARD = 1 };

typedef struct {
    uint64_t seed;
    uint32_t rng_mode;       /* ORC_RNG_* */
    uint32_t const_words[4]; /* the 4 words every draw block returns in ORC_RNG_CONST */
    uint32_t order;          /* ORC_ORDER_* */
    uint32_t sample_first;   /* FORWARD only: first global sample index */
    uint32_t sample_count;   /* FORWARD only: 0 = all n_samples */
    uint32_t output_sum;     /* FORWARD only: 1 = leave the per-pixel sum undivided */
    uint32_t full_paths;     /* FORWARD only: 1 = never end a path early on a zero throughput (see
                              * orc_scene_is_tame); the image is the same bit for bit either way */
    uint32_t col_first;      /* render only image columns [col_first, col_first + col_count) and leave */
    uint32_t col_count;      /* the rest of rgb untouched; 0 = the whole image (rgb is zeroed first).
                              * Pixels are independent, so disjoint bands may be rendered into one
                              * buffer concurrently by several threads (a test convenience). */
} orc_opts;

/* event counters of one render (FORWARD order), for FLOP accounting */
typedef struct {
    uint64_t samples;         /* (pixel, step) iterations: main.rs:14-28 */
    uint64_t rays;            /* closest-hit queries: trace.rs:16 */
    uint64_t hits[3];         /* closest hits by primitive kind */
    uint64_t cosine_samples;  /* random.rs:33-53 */
    uint64_t light_samples[3];/* random.rs:73-81 by light kind (index 0 unused) */
    uint64_t pdf_evals;       /* random.rs:143-152 */
    uint64_t light_pdf[3];    /* per (pdf eval, light): 0 / 1 / 2 intersections found */
    uint64_t diffuse_bounces; /* trace.rs:43-48 reached */
    uint64_t metallic_bounces;
    uint64_t dielectric_bounces;
    uint64_t prim_tests[3];   /* geometry.rs intersect calls inside closest-hit scans, by kind */
    uint64_t light_tests[3];  /* geometry.rs intersect calls inside ToLight::pdf, by kind */
} orc_stats;

/* full render: rgb = width*height*3 floats, row 0 = TOP row (image.rs:21-29) */
int orc_render(const orc_scene *scene, const orc_opts *opts, float *rgb, orc_stats *stats_or_null);

/* 1 if every number of the scene is finite, no object is degenerate (zero plane normal, zero
 * radius or half-size) and magnitudes are moderate.  In such a scene a path whose throughput
 * has become exactly (0, 0, 0) -- it bounced off a black object, like the light of
 * assets/scene.txt:38-41, which has no COLOR -- can only add T * emission = +-0 from then on,
 * and x + (+-0) = x: the FORWARD order ends the path there (the reference goes on tracing it,
 * trace.rs:43-48, to no effect) -- and already at the black surface itself when it is diffuse
 * or metallic, before the scatter whose only possible outcomes are a rejected sample and a zero
 * throughput.  The device applies the same rule with the same predicate. */
int orc_scene_is_tame(const orc_scene *scene);

/* deterministic primary visibility through pixel centres (du = dv = 0.5):
 * hit_id = -1 on a miss; t, normal (w*h*3) and inside undefined there (written 0). */
int orc_primary(const orc_scene *scene, int32_t *hit_id, float *t, float *normal, uint8_t *inside);

/* one closest-hit query (trace.rs:102-121); dir is used as given (not normalised).
 * returns the object index or -1. */
int orc_closest_hit(const orc_scene *scene, const float origin[3], const float dir[3],
                    float *t, float normal[3], int *inside);

/* ---- unit-level entry points for known-answer tests ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_sincos(float x, float *s, float *c);
void orc_camera_ray(const orc_scene *scene, float u, float v, float origin[3], float dir[3]);
/* figure-local tests (geometry.rs:56-147): returns 1 on hit */
int orc_figure_intersect(uint32_t kind, const float param[3], const float origin[3], const float dir[3],
                         float *t, float normal[3], int *inside);
void orc_reflect(const float dir[3], const float point[3], const float normal[3],
                 float out_origin[3], float out_dir[3]);
int orc_refract(const float dir[3], const float point[3], const float normal[3], float eta,
                float out_origin[3], float out_dir[3]);
float orc_schlick(float eta, float cosv);
void orc_cosine_sample(const float n[3], uint32_t w_theta, uint32_t w_r, float out[3]);
/* light surface sampling (sample.rs:13-17, 25-51, 60-64) of object prim_index with draw block w */
void orc_light_sample(const orc_scene *scene, uint32_t prim_index, const uint32_t w[4], float out_point[3]);
/* MIS pdf at p with normal n for direction d (random.rs:143-152) */
float orc_mis_pdf(const orc_scene *scene, const float p[3], const float n[3], const float d[3]);
float orc_light_pdf(const orc_scene *scene, const float p[3], const float d[3]);
/* number of lights and their object indices (parser.rs:69-89) */
int orc_lights(const orc_scene *scene, int32_t *indices, int capacity);

/* host-side colour pipeline: image.rs:51-83 (ACES + gamma) and :38-46 (byte conversion) */
void orc_color_correction(float *rgb, uint64_t n_floats);
void orc_to_bytes(const float *rgb, uint64_t n_floats, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
