// Stand-alone C++ host on top of the C ABI (include/mcrt.h): the same sequence a patched
// Camera::sampleImage() runs, but the flattened scene comes from a scene image
// (*.mcrt) written by the flattener inside the reference host.
//
//   mcrt_render scene.mcrt out.f64 [--width W --height H --sqrtspp S] [--seed N] [--photon] [--device D | --devices D0,D1,...]
//               [--tga out.tga [--tonemapper hable|aces] [--exposure EV] [--gain EV] [--plain]]
//
// Writes the frame as raw FP64 RGB, row-major (what Image::operator() holds, camera/image.cpp:53-56),
// and prints the statistics. --tga also develops it the way Image::save does (auto exposure / gain, tone map, sRGB bytes:
// mcrt_tonemap) and writes the reference's .tga; the "image" options default to the ones stored in the scene image.
// --devices renders the frame on several GPUs from this one process (mcrt_render_multi: one host thread per GPU).
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
    auto from_bits = [](uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; };
    mcrt_image_desc image{};
    image.tonemapper = (uint32_t)mcrt_image_param(img, "image_tonemapper");
    image.plain = (uint32_t)mcrt_image_param(img, "image_plain");
    image.exposure_compensation = from_bits(mcrt_image_param(img, "image_exposure_ev_bits"));
    image.gain_compensation = from_bits(mcrt_image_param(img, "image_gain_ev_bits"));
    std::string tga;
    std::vector<int> devices;
    for (int i = 3; i < argc; i++) {
        std::string k = argv[i];
        auto val = [&]() { return i + 1 < argc ? std::strtoul(argv[++i], nullptr, 0) : 0ul; };
        if (k == "--width") cam.width = (uint32_t)val();
        else if (k == "--height") cam.height = (uint32_t)val();
        else if (k == "--sqrtspp") cam.sqrtspp = (uint32_t)val();
        else if (k == "--seed") seed = (uint32_t)val();
        else if (k == "--device") device = (int)val();
        else if (k == "--devices" && i + 1 < argc) {
            for (const char* p = argv[++i]; *p;) {
                devices.push_back((int)std::strtol(p, const_cast<char**>(&p), 10));
                if (*p == ',') p++;
            }
        }
        else if (k == "--photon") photon = 1;
        else if (k == "--tga" && i + 1 < argc) tga = argv[++i];
        else if (k == "--tonemapper" && i + 1 < argc) image.tonemapper = (argv[++i][0] | 0x20) == 'a' ? MCRT_TONEMAP_ACES : MCRT_TONEMAP_HABLE;
        else if (k == "--exposure" && i + 1 < argc) image.exposure_compensation = std::strtod(argv[++i], nullptr);
        else if (k == "--gain" && i + 1 < argc) image.gain_compensation = std::strtod(argv[++i], nullptr);
        else if (k == "--plain") image.plain = 1;
    }
    cam.shard_index = 0;
    cam.shard_count = 1;
    if (devices.empty()) devices.push_back(device);
    std::vector<mcrt_ctx*> ctxs(devices.size(), nullptr);
    int rc = MCRT_OK;
    for (size_t d = 0; d < devices.size() && rc == MCRT_OK; d++) {
        if (mcrt_create(&ctxs[d], devices[d]) != MCRT_OK) {
            std::fprintf(stderr, "mcrt_create(device %d): %s\n", devices[d], mcrt_last_error(nullptr));
            return 1;
        }
        rc = mcrt_upload_scene(ctxs[d], mcrt_image_scene(img));
        if (rc == MCRT_OK && photon)
            rc = mcrt_upload_photons(ctxs[d], mcrt_image_photons(img, 0), mcrt_image_photons(img, 1),
                                     (uint32_t)mcrt_image_param(img, "k_nearest_photons"), (int)mcrt_image_param(img, "direct_visualization"));
        if (rc != MCRT_OK) std::fprintf(stderr, "device %d: %s\n", devices[d], mcrt_last_error(ctxs[d]));
    }
    mcrt_ctx* ctx = ctxs[0];
    std::vector<double> rgb((size_t)cam.width * cam.height * 3);
    mcrt_stats st;
    const int mode = photon ? MCRT_INTEGRATOR_PHOTON_MAPPER : MCRT_INTEGRATOR_PATH_TRACER;
    if (rc == MCRT_OK)
        rc = ctxs.size() > 1 ? mcrt_render_multi(ctxs.data(), (uint32_t)ctxs.size(), &cam, seed, mode, rgb.data(), &st)
                             : mcrt_render(ctx, &cam, seed, mode, rgb.data(), &st);
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
    if (!tga.empty()) {
        image.width = cam.width;
        image.height = cam.height;
        std::vector<uint8_t> bgr((size_t)cam.width * cam.height * 3);
        double factors[2];
        rc = mcrt_tonemap(ctx, rgb.data(), &image, bgr.data(), factors);
        if (rc == MCRT_OK) rc = mcrt_tga_save(tga.c_str(), cam.width, cam.height, bgr.data());
        if (rc != MCRT_OK) {
            std::fprintf(stderr, "mcrt error %d writing %s: %s\n", rc, tga.c_str(), mcrt_last_error(ctx));
            return 1;
        }
        std::printf("{\"tga\":\"%s\",\"exposure_factor\":%.17g,\"gain_factor\":%.17g}\n", tga.c_str(), factors[0], factors[1]);
    }
    std::printf("{\"width\":%u,\"height\":%u,\"spp\":%u,\"paths\":%llu,\"rays\":%llu,\"kernel_ms\":%.3f,\"total_ms\":%.3f,\"Mray_s\":%.1f}\n",
                cam.width, cam.height, cam.sqrtspp * cam.sqrtspp, (unsigned long long)st.paths, (unsigned long long)st.rays,
                st.kernel_ms, st.total_ms, st.rays / st.kernel_ms / 1e3);
    for (mcrt_ctx* c : ctxs) mcrt_destroy(c);
    mcrt_image_free(img);
    return 0;
}
