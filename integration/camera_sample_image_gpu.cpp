// integration/camera_sample_image_gpu.cpp — the drop-in, built for real: the replacement body of
//
//     void Camera::sampleImage();          // reference: source/camera/camera.cpp:101-145
//
// that INTEGRATION.md §2 describes, linked with the reference's OWN objects (every translation unit of
// /root/reference/source incl. main.cpp, compiled unmodified by oracle/Makefile) and libmcrt_hip.so into
// oracle/_ref/mcrt_ref_gpu. The reference's definition of Camera::sampleImage() in camera.o is made a weak symbol
// (objcopy --weaken-symbol, oracle/Makefile) so that this strong definition is the one Camera::capture() (camera.cpp:170-181)
// calls; everything else — main(), the stdin menu, JSON/OBJ loading, BVH::BVH, Scene::generateEmissives, the photon pass of
// PhotonMapper::PhotonMapper, Camera::saveImage / Image::save — is the reference's code, untouched.
//
// Contract kept (camera.cpp:138-144): on return image(x, y) holds the filtered mean radiance of every pixel.
// Test infrastructure (reference code + our library), not the product: tests/test_gpu_dropin.py runs the binary on the GPU box
// and compares the .tga files the reference's Image::save writes with the committed reference-written ones.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "ref_flatten.hpp"

void Camera::sampleImage()
{
    static mcrt_ctx* ctx = nullptr;  // one context (device 0) for the life of the process
    if (!ctx && mcrt_create(&ctx, 0) != MCRT_OK)
        throw std::runtime_error(mcrt_last_error(nullptr));  // the reference's convention: exceptions reach main (main.cpp:48-56)

    Flat flat;
    flattenScene(integrator->scene, flat);  // arrays stay owned by the host, copied during the call
    if (mcrt_upload_scene(ctx, &flat.desc) != MCRT_OK) throw std::runtime_error(mcrt_last_error(ctx));

    int mode = MCRT_INTEGRATOR_PATH_TRACER;
    if (auto pm = dynamic_cast<PhotonMapper*>(integrator.get())) {
        FlatMap g, c;
        flattenMap(pm->global_map, g);
        flattenMap(pm->caustic_map, c);
        if (mcrt_upload_photons(ctx, g.desc.num_octants ? &g.desc : nullptr, c.desc.num_octants ? &c.desc : nullptr,
                                (uint32_t)pm->k_nearest_photons, pm->direct_visualization ? 1 : 0) != MCRT_OK)
            throw std::runtime_error(mcrt_last_error(ctx));
        mode = MCRT_INTEGRATOR_PHOTON_MAPPER;
    }

    const mcrt_camera_desc cam = flattenCamera(*this);  // shard_count = 1: the whole frame on this GPU
    std::vector<double> rgb(image.width * image.height * 3);
    mcrt_stats stats;
    if (mcrt_render(ctx, &cam, Sampler::global_seed, mode, rgb.data(), &stats) != MCRT_OK)
        throw std::runtime_error(mcrt_last_error(ctx));
    num_sampled_pixels = image.width * image.height;

    for (size_t y = 0; y < image.height; y++)  // camera.cpp:138-144
        for (size_t x = 0; x < image.width; x++) {
            const double* p = &rgb[(y * image.width + x) * 3];
            image(x, y) = glm::dvec3(p[0], p[1], p[2]);
        }

    std::printf("\n[mcrt_hip] Camera::sampleImage on the GPU: %llu paths, %llu rays, kernel %u (%u launches), %.3f ms, %.1f Mray/s\n",
                (unsigned long long)stats.paths, (unsigned long long)stats.rays, stats.kernel_id, stats.kernel_launches, stats.kernel_ms,
                stats.kernel_ms > 0 ? stats.rays / stats.kernel_ms / 1e3 : 0.0);
    if (const char* dump = std::getenv("MCRT_DROPIN_DUMP")) {  // FP64 frame for the parity test (the .tga is 8 bit)
        if (FILE* f = std::fopen(dump, "wb")) {
            std::fwrite(rgb.data(), sizeof(double), rgb.size(), f);
            std::fclose(f);
        }
    }
}
