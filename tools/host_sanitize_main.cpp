// Exercises librtx_host's sources under AddressSanitizer + UBSan (host code only):
// parser on valid, invalid and randomly mangled inputs, tone map, byte conversion, PPM and
// checkpoint files.  Built and run by tools/host_sanitize.sh; prints "clean" at the end.
#include "../include/rtx_host.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

int main()
{
    const std::string head =
        "DIMENSIONS 4 2\nRAY_DEPTH 3\nSAMPLES 1\nBG_COLOR 0 0 0\nCAMERA_POSITION 0 0 0\nCAMERA_RIGHT 1 0 0\n"
        "CAMERA_UP 0 1 0\nCAMERA_FORWARD 0 0 1\nCAMERA_FOV_X 1.0\n";
    const char *bodies[] = {"PLANE 0 1\n", "PLANE 0  1 0\n", "POSITION 0 0 0\nPLANE 0 1 0\n", "PLANE 0 1 zero\n",
                            "DIMENSIONS -4 4\n", "SAMPLES 1.5\n", "PLANE 0 1 0\nROTATION 1 2 3 4\nEMISSION 1 inf -0\n",
                            "BOX 1 1 1\nDIELECTRIC\nIOR 1.5\nELLIPSOID 1 2 3\nMETALLIC\n", ""};
    int parsed = 0, refused = 0;
    for (const char *b : bodies) {
        rtx_host_scene hs;
        const std::string t = head + b;
        if (rtx_host_parse_text(t.c_str(), &hs) == RTX_OK) { parsed++; rtx_host_free_scene(&hs); }
        else refused++;
    }
    // mangled copies of a valid file: deletions, duplications and byte flips
    const std::string good = head + bodies[7] + bodies[6];
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int it = 0; it < 20000; ++it) {
        std::string t = good;
        for (int k = 0, n = 1 + (int)(rnd() % 4); k < n && !t.empty(); ++k) {
            size_t p = rnd() % t.size();
            switch (rnd() % 4) {
            case 0: t.erase(p, 1 + rnd() % 6); break;
            case 1: t.insert(p, t.substr(rnd() % t.size(), rnd() % 9)); break;
            case 2: t[p] = (char)(rnd() % 256 ? rnd() % 128 : 0); break;
            default: t.insert(p, 1, " \n\t-.e0"[rnd() % 8]); break;
            }
        }
        rtx_host_scene hs;
        if (rtx_host_parse_text(t.c_str(), &hs) == RTX_OK) { parsed++; rtx_host_free_scene(&hs); }
        else refused++;
    }
    rtx_host_scene none;
    if (rtx_host_parse_file("/nonexistent/scene.txt", &none) == RTX_OK) return 1;
    if (rtx_host_parse_text(nullptr, &none) == RTX_OK || rtx_host_parse_text("", nullptr) == RTX_OK) return 1;

    std::vector<float> img = {0.0f, 0.5f, 2.0f, -1.0f, 1e9f, 1e-30f, -0.0f, 3.0f, 0.25f, 1.0f, 7.0f, 0.1f};
    std::vector<float> sum = img;
    rtx_host_color_correction(img.data(), img.size());
    std::vector<uint8_t> bytes(img.size());
    rtx_host_to_bytes(img.data(), img.size(), bytes.data());
    if (rtx_host_write_ppm("/tmp/rtx_sanitize.ppm", img.data(), 2, 2) != RTX_OK) return 1;
    if (rtx_host_write_ppm_bytes("/tmp/rtx_sanitize.ppm", bytes.data(), 2, 2) != RTX_OK) return 1;

    rtx_host_scene hs;
    if (rtx_host_parse_text(good.c_str(), &hs) != RTX_OK) return 1;
    rtx_host_checkpoint info = {2, 2, 5, 77, rtx_host_scene_fingerprint(&hs.scene)};
    rtx_host_free_scene(&hs);
    if (rtx_host_save_checkpoint("/tmp/rtx_sanitize.ckpt", &info, sum.data()) != RTX_OK) return 1;
    rtx_host_checkpoint back;
    std::vector<float> got(12);
    if (rtx_host_load_checkpoint("/tmp/rtx_sanitize.ckpt", &back, got.data(), got.size()) != RTX_OK) return 1;
    if (rtx_host_load_checkpoint("/tmp/rtx_sanitize.ckpt", &back, got.data(), 11) == RTX_OK) return 1;   // too small
    if (rtx_host_load_checkpoint("/tmp/rtx_sanitize.ppm", &back, nullptr, 0) == RTX_OK) return 1;        // not a checkpoint
    std::remove("/tmp/rtx_sanitize.ppm");
    std::remove("/tmp/rtx_sanitize.ckpt");
    std::printf("clean: %d inputs parsed, %d refused\n", parsed, refused);
    return 0;
}
