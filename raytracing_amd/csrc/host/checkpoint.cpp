// checkpoint.cpp -- save / load of a progressive render's state (sum buffer + sample counter).
// The reference has nothing like it: its render loop (src/main.rs:13-31) keeps the image in
// memory only.  The state is what rtx_progress_read / rtx_progress_begin exchange.
#include "../../../include/rtx_host.h"

#include <cstdio>
#include <cstring>
#include <string>

namespace {

const char kMagic[8] = {'R', 'T', 'X', 'C', 'K', 'P', 'T', '1'};

struct Header {
    char magic[8];
    uint32_t width, height;
    uint32_t samples_done, reserved;
    uint64_t seed;
    uint64_t fingerprint;
    uint8_t zero[24];
};
static_assert(sizeof(Header) == 64, "checkpoint header is 64 bytes");

void fnv(uint64_t &h, const void *data, size_t n)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
}

}  // namespace

extern "C" {

uint64_t rtx_host_scene_fingerprint(const rtx_scene *s)
{
    uint64_t h = 0xcbf29ce484222325ull;
    if (!s) return h;
    fnv(h, &s->width, sizeof(s->width));
    fnv(h, &s->height, sizeof(s->height));
    fnv(h, &s->ray_depth, sizeof(s->ray_depth));
    fnv(h, s->background, sizeof(s->background));
    fnv(h, s->cam_pos, sizeof(s->cam_pos));
    fnv(h, s->cam_axis, sizeof(s->cam_axis));
    fnv(h, &s->tg_fov_x, sizeof(float));
    fnv(h, &s->tg_fov_y, sizeof(float));
    fnv(h, &s->n_prims, sizeof(s->n_prims));
    for (uint32_t i = 0; i < s->n_prims && s->prims; ++i) {
        // field by field: the struct has no padding today, but the hash must not depend on that
        const rtx_prim &p = s->prims[i];
        fnv(h, &p.kind, 4);
        fnv(h, p.param, 12);
        fnv(h, p.position, 12);
        fnv(h, p.rotation, 16);
        fnv(h, p.color, 12);
        fnv(h, p.emission, 12);
        fnv(h, &p.material, 4);
        fnv(h, &p.ior, 4);
    }
    return h;
}

int rtx_host_save_checkpoint(const char *path, const rtx_host_checkpoint *info, const float *sum)
{
    if (!path || !info || !sum) return RTX_ERR_INVALID_ARG;
    Header hd;
    std::memset(&hd, 0, sizeof(hd));
    std::memcpy(hd.magic, kMagic, 8);
    hd.width = info->width;
    hd.height = info->height;
    hd.samples_done = info->samples_done;
    hd.seed = info->seed;
    hd.fingerprint = info->fingerprint;
    const std::string tmp = std::string(path) + ".tmp";
    std::FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return RTX_ERR_INVALID_ARG;
    const size_t n = (size_t)info->width * info->height * 3u;
    bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1;
    ok = ok && (n == 0 || std::fwrite(sum, sizeof(float), n, f) == n);
    ok = (std::fclose(f) == 0) && ok;
    if (ok) ok = std::rename(tmp.c_str(), path) == 0;
    if (!ok) std::remove(tmp.c_str());
    return ok ? RTX_OK : RTX_ERR_INVALID_ARG;
}

int rtx_host_load_checkpoint(const char *path, rtx_host_checkpoint *info, float *sum, uint64_t capacity_floats)
{
    if (!path || !info) return RTX_ERR_INVALID_ARG;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return RTX_ERR_INVALID_ARG;
    Header hd;
    bool ok = std::fread(&hd, sizeof(hd), 1, f) == 1 && std::memcmp(hd.magic, kMagic, 8) == 0;
    if (ok) {
        info->width = hd.width;
        info->height = hd.height;
        info->samples_done = hd.samples_done;
        info->seed = hd.seed;
        info->fingerprint = hd.fingerprint;
        const uint64_t n = (uint64_t)hd.width * hd.height * 3u;
        if (sum) {
            ok = n <= capacity_floats && (n == 0 || std::fread(sum, sizeof(float), (size_t)n, f) == (size_t)n);
            // a longer file is not this format either
            ok = ok && std::fgetc(f) == EOF;
        }
    }
    std::fclose(f);
    return ok ? RTX_OK : RTX_ERR_INVALID_ARG;
}

}  // extern "C"
