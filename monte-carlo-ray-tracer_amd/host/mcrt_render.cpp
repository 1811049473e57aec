// Stand-alone C++ host on top of the C ABI (include/mcrt.h): the same sequence a patched
// Camera::sampleImage() runs, but the flattened scene comes from a scene image
// (*.mcrt) written by the flattener inside the reference host.
//
//   mcrt_render scene.mcrt out.f64 [--width W --height H --sqrtspp S] [--seed N] [--photon] [--device D]
//
// Writes the frame as raw FP64 RGB, row-major (what Image::operator() holds, camera/image.cpp:53-56),
// and prints the statistics. Tonemapping and the TGA writer stay with the reference (Image::save).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcrt.h"

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s scene.mcrt out.f64 [--width W --height H --sqrtspp S] [--seed N] [--photon] [--device D]\n", argv[0]);
        return 2;
    }
    mcrt_image* img = nullptr;
    if (mcrt_image_load(argv[1], &img) != MCRT_OK) {
        std::fprintf(stderr, "cannot load scene image %s\n", argv[1]);
        return 1;
    }
    mcrt_camera_desc cam = *mcrt_image_camera(img);
    uint32_t seed = (uint32_t)mcrt_image_param(img, "global_seed");
    int photon = (int)mcrt_image_param(img, "photon_mapping"), device = 0;
    for (int i = 3; i < argc; i++) {
        std::string k = argv[i];
        auto val = [&]() { return i + 1 < argc ? std::strtoul(argv[++i], nullptr, 0) : 0ul; };
        if (k == "--width") cam.width = (uint32_t)val();
        else if (k == "--height") cam.height = (uint32_t)val();
        else if (k == "--sqrtspp") cam.sqrtspp = (uint32_t)val();
        else if (k == "--seed") seed = (uint32_t)val();
        else if (k == "--device") device = (int)val();
        else if (k == "--photon") photon = 1;
    }
    cam.shard_index = 0;
    cam.shard_count = 1;
    mcrt_ctx* ctx = nullptr;
    if (mcrt_create(&ctx, device) != MCRT_OK) {
        std::fprintf(stderr, "mcrt_create: %s\n", mcrt_last_error(nullptr));
        return 1;
    }
    int rc = mcrt_upload_scene(ctx, mcrt_image_scene(img));
    if (rc == MCRT_OK && photon)
        rc = mcrt_upload_photons(ctx, mcrt_image_photons(img, 0), mcrt_image_photons(img, 1),
                                 (uint32_t)mcrt_image_param(img, "k_nearest_photons"), (int)mcrt_image_param(img, "direct_visualization"));
    std::vector<double> rgb((size_t)cam.width * cam.height * 3);
    mcrt_stats st;
    if (rc == MCRT_OK) rc = mcrt_render(ctx, &cam, seed, photon ? MCRT_INTEGRATOR_PHOTON_MAPPER : MCRT_INTEGRATOR_PATH_TRACER, rgb.data(), &st);
    if (rc != MCRT_OK) {
        std::fprintf(stderr, "mcrt error %d: %s\n", rc, mcrt_last_error(ctx));
        return 1;
    }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f || std::fwrite(rgb.data(), sizeof(double), rgb.size(), f) != rgb.size()) {
        std::fprintf(stderr, "cannot write %s\n", argv[2]);
        return 1;
    }
    std::fclose(f);
    std::printf("{\"width\":%u,\"height\":%u,\"spp\":%u,\"paths\":%llu,\"rays\":%llu,\"kernel_ms\":%.3f,\"total_ms\":%.3f,\"Mray_s\":%.1f}\n",
                cam.width, cam.height, cam.sqrtspp * cam.sqrtspp, (unsigned long long)st.paths, (unsigned long long)st.rays,
                st.kernel_ms, st.total_ms, st.rays / st.kernel_ms / 1e3);
    mcrt_destroy(ctx);
    mcrt_image_free(img);
    return 0;
}
