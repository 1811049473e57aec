// MemorySanitizer run of the device code built for the host (tests/emu/mcrt_emu.cpp) on a scene image: the wavefront integrator, and —
// argument 4 — the megakernels' code (mega / flat / sm) or the photon mapper (pm, wfpm: images that carry photon maps).
// Built and run by tools/msan_emu.sh. Reports inside mcrt_image.cpp come from the uninstrumented libstdc++ (std::map of
// std::string) and are filtered out by the script; anything else is a use of an uninitialised value in the integrator.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../include/mcrt.h"

extern "C" int emu_render_wf(const mcrt_scene_desc* scene, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t slots, uint32_t owned_rows,
                             double* out_rgb, uint64_t* counters);
extern "C" int emu_render_wf_pm(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k_nearest,
                                int direct_visualization, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t slots, uint32_t owned_rows,
                                double* out_rgb, uint64_t* counters);
extern "C" int emu_render(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k_nearest,
                          int direct_visualization, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, uint32_t row0, uint32_t row1,
                          int stage_lds, double* out_rgb, uint64_t* counters);
extern "C" int emu_render_sm(const mcrt_scene_desc* scene, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t row0, uint32_t row1,
                             int stage_all, double* out_rgb, uint64_t* counters);

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    mcrt_image* img = nullptr;
    if (mcrt_image_load(argv[1], &img)) {
        fprintf(stderr, "load failed\n");
        return 2;
    }
    fprintf(stderr, "MSAN-DRIVER: image loaded\n");
    mcrt_camera_desc cam = *mcrt_image_camera(img);
    cam.width = argc > 3 ? (uint32_t)atoi(argv[3]) : 48u;
    cam.height = cam.width * 9u / 16u;
    cam.sqrtspp = 2;
    std::vector<double> out((size_t)cam.width * cam.height * 3);
    uint64_t counters[8] = {0};
    const uint32_t slots = argc > 2 ? (uint32_t)atoi(argv[2]) : 333u;
    const std::string mode = argc > 4 ? argv[4] : "wf";
    const mcrt_scene_desc* sc = mcrt_image_scene(img);
    const mcrt_photon_map_desc *g = mcrt_image_photons(img, 0), *c = mcrt_image_photons(img, 1);
    int rc;
    if (mode == "wf") rc = emu_render_wf(sc, &cam, 12345u, slots, cam.height, out.data(), counters);
    else if (mode == "wfpm") rc = emu_render_wf_pm(sc, g, c, 50, 0, &cam, 12345u, slots, cam.height, out.data(), counters);
    else if (mode == "pm") rc = emu_render(sc, g, c, 50, 0, &cam, 12345u, MCRT_INTEGRATOR_PHOTON_MAPPER, 0, cam.height, 1, out.data(), counters);
    else if (mode == "sm") rc = emu_render_sm(sc, &cam, 12345u, 0, cam.height, 0, out.data(), counters);
    else rc = emu_render(sc, nullptr, nullptr, 0, 0, &cam, 12345u, MCRT_INTEGRATOR_PATH_TRACER, 0, cam.height, mode == "flat" ? 3 : mode == "top" ? 0 : 1, out.data(), counters);
    double s = 0;
    for (double v : out) s += v;
    printf("rc %d frame sum %.6f rays %llu\n", rc, s, (unsigned long long)counters[0]);
    return 0;
}
