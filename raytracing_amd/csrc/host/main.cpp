// main.cpp -- the reference's driver (src/main.rs:33-42) with `render` replaced by librtx:
//   raytracing [scene.txt] [out.ppm]      defaults: assets/scene.txt, /tmp/out.ppm
// Options after the two positional arguments override what the scene file says, so the
// benchmark configurations can reuse one scene file:
//   --size W H   --spp N   --depth N   --seed N   --mode megakernel|wavefront
#include "../../../include/rtx.h"
#include "../../../include/rtx_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    std::vector<std::string> pos;
    rtx_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    opts.seed = 1;
    opts.device = -1;
    long width = -1, height = -1, spp = -1, depth = -1;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](int k) {
            if (i + k >= argc) { std::fprintf(stderr, "%s needs %d value(s)\n", a.c_str(), k); std::exit(2); }
        };
        if (a == "--size") { need(2); width = std::atol(argv[++i]); height = std::atol(argv[++i]); }
        else if (a == "--spp") { need(1); spp = std::atol(argv[++i]); }
        else if (a == "--depth") { need(1); depth = std::atol(argv[++i]); }
        else if (a == "--seed") { need(1); opts.seed = std::strtoull(argv[++i], nullptr, 10); }
        else if (a == "--mode") { need(1); opts.mode = std::string(argv[++i]) == "wavefront" ? RTX_MODE_WAVEFRONT : RTX_MODE_MEGAKERNEL; }
        else pos.push_back(a);
    }
    const std::string input = pos.size() > 0 ? pos[0] : "assets/scene.txt";
    const std::string output = pos.size() > 1 ? pos[1] : "/tmp/out.ppm";

    rtx_host_scene hs;
    if (rtx_host_parse_file(input.c_str(), &hs) != RTX_OK) {
        std::fprintf(stderr, "parse_scene: %s\n", rtx_host_last_error());
        return 1;
    }
    if (width > 0 && height > 0) {
        // the tangents depend on the aspect ratio (src/parser.rs:54-55)
        hs.scene.width = (uint32_t)width;
        hs.scene.height = (uint32_t)height;
        hs.scene.tg_fov_y = ((float)hs.scene.height / (float)hs.scene.width) * hs.scene.tg_fov_x;
    }
    if (spp >= 0) hs.scene.n_samples = (uint32_t)spp;
    if (depth >= 0) hs.scene.ray_depth = (uint32_t)depth;

    rtx_ctx *ctx = nullptr;
    if (rtx_create(&ctx, &opts) != RTX_OK) {
        std::fprintf(stderr, "rtx_create: %s\n", rtx_last_error());
        return 1;
    }
    std::vector<float> image((size_t)hs.scene.width * hs.scene.height * 3u);
    int rc = rtx_render(ctx, &hs.scene, image.data());  // render(&mut scene)
    if (rc != RTX_OK) std::fprintf(stderr, "rtx_render: %s\n", rtx_last_error());
    if (rc == RTX_OK) {
        rtx_host_color_correction(image.data(), image.size());  // scene.image.color_correction()
        rc = rtx_host_write_ppm(output.c_str(), image.data(), hs.scene.width, hs.scene.height);  // .write(&output)
        if (rc != RTX_OK) std::fprintf(stderr, "cannot write %s\n", output.c_str());
    }
    rtx_destroy(ctx);
    rtx_host_free_scene(&hs);
    return rc == RTX_OK ? 0 : 1;
}
