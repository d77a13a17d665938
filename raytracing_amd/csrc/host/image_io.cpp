// image_io.cpp -- the reference's output stage (src/image.rs:31-83): ACES tone map, gamma,
// byte conversion and the binary PPM writer.
#include "../../../include/rtx_host.h"

#include <cmath>
#include <cstdio>
#include <vector>

extern "C" {

// src/image.rs:65-79 then :60-63.  saturate = clamp to [0, 1].
void rtx_host_color_correction(float *rgb, uint64_t n)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const float inv_gamma = 1.0f / 2.2f;
    for (uint64_t i = 0; i < n; ++i) {
        const float x = rgb[i];
        const float up = x * (a * x + b);
        const float down = x * (c * x + d) + e;
        float y = up / down;
        y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
        rgb[i] = std::pow(y, inv_gamma);
    }
}

// src/image.rs:44: round half away from zero, then a saturating cast (NaN -> 0)
void rtx_host_to_bytes(const float *rgb, uint64_t n, uint8_t *out)
{
    for (uint64_t i = 0; i < n; ++i) {
        const float v = std::round(255.0f * rgb[i]);
        out[i] = (v != v) ? 0 : (v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (uint8_t)v));
    }
}

int rtx_host_write_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height)
{
    if (!path || !rgb) return RTX_ERR_INVALID_ARG;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return RTX_ERR_INVALID_ARG;
    const uint64_t n = (uint64_t)width * height * 3u;
    std::vector<uint8_t> bytes(n);
    rtx_host_to_bytes(rgb, n, bytes.data());
    bool ok = std::fprintf(f, "P6\n%u %u\n255\n", width, height) > 0;
    ok = ok && (n == 0 || std::fwrite(bytes.data(), 1, n, f) == n);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RTX_OK : RTX_ERR_INVALID_ARG;
}

int rtx_host_write_ppm_bytes(const char *path, const uint8_t *bytes, uint32_t width, uint32_t height)
{
    if (!path || !bytes) return RTX_ERR_INVALID_ARG;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return RTX_ERR_INVALID_ARG;
    const uint64_t n = (uint64_t)width * height * 3u;
    bool ok = std::fprintf(f, "P6\n%u %u\n255\n", width, height) > 0;
    ok = ok && (n == 0 || std::fwrite(bytes, 1, n, f) == n);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RTX_OK : RTX_ERR_INVALID_ARG;
}

}  // extern "C"
