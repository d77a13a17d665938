/*
 * rtx_host.h -- C ABI of librtx_host: the host side that stays outside the GPU path,
 * restated in C++ because the reference's own toolchain (Rust) is not available here.
 * It mirrors three pieces of the reference one to one:
 *   rtx_host_parse_file / _text   = parse_scene            (src/parser.rs:104-199, :50-101)
 *   rtx_host_color_correction     = Image::color_correction (src/image.rs:51-83)
 *   rtx_host_write_ppm            = Image::write            (src/image.rs:31-49)
 * No function here touches the GPU or computes radiance.
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H

#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A parsed scene: `scene.prims` points at `prims` (owned by this struct).
 * fov_x is kept for reference; the tangents in `scene` are what the renderer uses. */
typedef struct rtx_host_scene {
    rtx_scene scene;
    rtx_prim *prims;
    float fov_x;
} rtx_host_scene;

/* parse_scene(path) (src/parser.rs:104).  Where the reference panics -- unreadable file,
 * unparsable number, missing operand, a modifier line before any primitive, a missing
 * mandatory line -- these return RTX_ERR_INVALID_SCENE (RTX_ERR_INVALID_ARG for a null
 * argument) and rtx_host_last_error() says which line.  Unknown keywords and blank lines
 * are ignored, as in src/parser.rs:194.  Free the result with rtx_host_free_scene. */
int rtx_host_parse_file(const char *path, rtx_host_scene *out);
/* The same on an in-memory copy of the file's text. */
int rtx_host_parse_text(const char *text, rtx_host_scene *out);
void rtx_host_free_scene(rtx_host_scene *scene);

/* Image::color_correction (src/image.rs:51-83): ACES tone map, clamp to [0,1], gamma 1/2.2,
 * in place on n_floats linear radiance values. */
void rtx_host_color_correction(float *rgb, uint64_t n_floats);
/* The byte conversion of Image::write (src/image.rs:38-46): (255 x).round() as u8. */
void rtx_host_to_bytes(const float *rgb, uint64_t n_floats, uint8_t *out);
/* Image::write (src/image.rs:31-49): binary P6 with the header "P6\n<w> <h>\n255\n";
 * rgb = width*height*3 floats, row 0 = top. */
int rtx_host_write_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height);

/* Image::write for values that are bytes already (the body rtx_render_bytes returns). */
int rtx_host_write_ppm_bytes(const char *path, const uint8_t *bytes, uint32_t width, uint32_t height);

/* Checkpoints of a progressive render (no counterpart in the reference, whose only state is
 * the image in memory).  File = 64-byte header {magic "RTXCKPT1", width, height,
 * samples_done, reserved, seed, scene fingerprint, ...zero} + width*height*3 little-endian
 * floats: the per-pixel sum of samples [0, samples_done).  The save is atomic (temporary
 * file + rename), so an interrupted run leaves the previous checkpoint intact. */
typedef struct rtx_host_checkpoint {
    uint32_t width, height;
    uint32_t samples_done;
    uint64_t seed;
    uint64_t fingerprint; /* rtx_host_scene_fingerprint of the scene being rendered */
} rtx_host_checkpoint;
/* FNV-1a over everything that determines a sample's value: image size, depth, background,
 * camera and every object (n_samples excluded: a resumed run may ask for more). */
uint64_t rtx_host_scene_fingerprint(const rtx_scene *scene);
int rtx_host_save_checkpoint(const char *path, const rtx_host_checkpoint *info, const float *sum);
/* Reads the header into info; when sum is not NULL also the body (capacity_floats must be
 * at least width*height*3).  RTX_ERR_INVALID_ARG: unreadable, truncated or not a checkpoint. */
int rtx_host_load_checkpoint(const char *path, rtx_host_checkpoint *info, float *sum, uint64_t capacity_floats);

const char *rtx_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_HOST_H */
