// main.cpp -- the reference's driver (src/main.rs:33-42) with `render` replaced by librtx:
//   raytracing [scene.txt] [out.ppm]      defaults: assets/scene.txt, /tmp/out.ppm
// Options after the two positional arguments override what the scene file says, so the
// benchmark configurations can reuse one scene file:
//   --size W H   --spp N   --depth N   --seed N   --mode megakernel|wavefront
//   --gpus N              deal the frame out to GPUs 0..N-1 of the node (buffers added on the GPUs; same bits as one GPU)
//   --devices a,b,...     the same over the listed HIP ordinals (an ordinal may repeat)
//   --reduce p2p|rccl     how: peer-access slice reduce (default) or RCCL all-reduce
// Output stage:
//   --device-tonemap      tone map and quantise on the GPU; only bytes come back
// Progressive rendering (the image equals the one-shot render bit for bit):
//   --chunk N             render N samples per step
//   --checkpoint FILE     save the state after every step; resume from FILE if it matches
//   --stop-after N        leave after N samples in total (exit status 3, no image written)
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
    long width = -1, height = -1, spp = -1, depth = -1, chunk = -1, stop_after = -1;
    bool device_tonemap = false;
    std::string checkpoint;
    std::vector<int32_t> devices;
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
        else if (a == "--gpus") { need(1); opts.n_devices = (int32_t)std::atol(argv[++i]); }
        else if (a == "--devices") {
            need(1);
            devices.clear();
            for (const char *q = argv[++i]; *q;) {
                char *end = nullptr;
                devices.push_back((int32_t)std::strtol(q, &end, 10));
                if (end == q) { std::fprintf(stderr, "--devices wants a comma-separated list of ordinals\n"); return 2; }
                q = *end == ',' ? end + 1 : end;
            }
            opts.n_devices = (int32_t)devices.size();
        }
        else if (a == "--reduce") { need(1); opts.reduce = std::string(argv[++i]) == "rccl" ? RTX_REDUCE_RCCL : RTX_REDUCE_P2P; }
        else if (a == "--chunk") { need(1); chunk = std::atol(argv[++i]); }
        else if (a == "--stop-after") { need(1); stop_after = std::atol(argv[++i]); }
        else if (a == "--checkpoint") { need(1); checkpoint = argv[++i]; }
        else if (a == "--device-tonemap") device_tonemap = true;
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

    if (!devices.empty()) opts.devices = devices.data();
    rtx_ctx *ctx = nullptr;
    if (rtx_create(&ctx, &opts) != RTX_OK) {
        std::fprintf(stderr, "rtx_create: %s\n", rtx_last_error());
        return 1;
    }
    const uint32_t w = hs.scene.width, h = hs.scene.height, total = hs.scene.n_samples;
    const size_t floats = (size_t)w * h * 3u;
    std::vector<float> image;
    std::vector<uint8_t> bytes;
    const bool progressive = chunk > 0 || stop_after >= 0 || !checkpoint.empty();
    int rc = RTX_OK;
    bool stopped = false;
    if (!progressive) {
        if (device_tonemap) {
            bytes.resize(floats);
            rc = rtx_render_bytes(ctx, &hs.scene, bytes.data());
        } else {
            image.resize(floats);
            rc = rtx_render(ctx, &hs.scene, image.data());  // render(&mut scene)
        }
        if (rc != RTX_OK) std::fprintf(stderr, "rtx_render: %s\n", rtx_last_error());
    } else {
        const uint64_t fingerprint = rtx_host_scene_fingerprint(&hs.scene);
        std::vector<float> sum;
        uint32_t done = 0;
        if (!checkpoint.empty()) {
            rtx_host_checkpoint info;
            if (rtx_host_load_checkpoint(checkpoint.c_str(), &info, nullptr, 0) == RTX_OK) {
                const bool same = info.width == w && info.height == h && info.seed == opts.seed &&
                                  info.fingerprint == fingerprint && info.samples_done <= total;
                if (!same) {
                    std::fprintf(stderr, "%s belongs to another scene, size, seed or a longer run\n", checkpoint.c_str());
                    rtx_destroy(ctx);
                    rtx_host_free_scene(&hs);
                    return 1;
                }
                sum.resize(floats);
                if (rtx_host_load_checkpoint(checkpoint.c_str(), &info, sum.data(), floats) != RTX_OK) {
                    std::fprintf(stderr, "%s is truncated\n", checkpoint.c_str());
                    rtx_destroy(ctx);
                    rtx_host_free_scene(&hs);
                    return 1;
                }
                done = info.samples_done;
                std::fprintf(stderr, "resuming at sample %u of %u\n", done, total);
            }
        }
        rc = rtx_progress_begin(ctx, &hs.scene, done ? sum.data() : nullptr, done);
        const uint32_t goal = stop_after >= 0 && (uint32_t)stop_after < total ? (uint32_t)stop_after : total;
        const uint32_t step = chunk > 0 ? (uint32_t)chunk : total;
        while (rc == RTX_OK && done < goal) {
            const uint32_t n = goal - done < step ? goal - done : step;
            rc = rtx_progress_step(ctx, n);
            if (rc != RTX_OK) break;
            done += n;
            if (!checkpoint.empty()) {
                sum.resize(floats);
                rc = rtx_progress_read(ctx, nullptr, sum.data(), nullptr, nullptr);
                rtx_host_checkpoint info = {w, h, done, opts.seed, fingerprint};
                if (rc == RTX_OK && rtx_host_save_checkpoint(checkpoint.c_str(), &info, sum.data()) != RTX_OK) {
                    std::fprintf(stderr, "cannot write %s\n", checkpoint.c_str());
                    rc = RTX_ERR_INVALID_ARG;
                }
            }
        }
        stopped = rc == RTX_OK && done < total;
        if (rc == RTX_OK && !stopped) {
            if (device_tonemap) {
                bytes.resize(floats);
                rc = rtx_progress_read(ctx, nullptr, nullptr, bytes.data(), nullptr);
            } else {
                image.resize(floats);
                rc = rtx_progress_read(ctx, image.data(), nullptr, nullptr, nullptr);
            }
        }
        if (rc != RTX_OK) std::fprintf(stderr, "progressive render: %s\n", rtx_last_error());
    }
    if (rc == RTX_OK && !stopped) {
        if (device_tonemap) {
            rc = rtx_host_write_ppm_bytes(output.c_str(), bytes.data(), w, h);
        } else {
            rtx_host_color_correction(image.data(), image.size());  // scene.image.color_correction()
            rc = rtx_host_write_ppm(output.c_str(), image.data(), w, h);  // .write(&output)
        }
        if (rc != RTX_OK) std::fprintf(stderr, "cannot write %s\n", output.c_str());
    }
    rtx_destroy(ctx);
    rtx_host_free_scene(&hs);
    if (stopped) return 3;
    return rc == RTX_OK ? 0 : 1;
}
