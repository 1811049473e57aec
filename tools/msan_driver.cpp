// MemorySanitizer run of the device code built for the host (tests/emu/mcrt_emu.cpp): the wavefront integrator on a scene image.
// Built and run by tools/msan_emu.sh. Reports inside mcrt_image.cpp come from the uninstrumented libstdc++ (std::map of
// std::string) and are filtered out by the script; anything else is a use of an uninitialised value in the integrator.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/mcrt.h"

extern "C" int emu_render_wf(const mcrt_scene_desc* scene, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t slots, uint32_t owned_rows,
                             double* out_rgb, uint64_t* counters);

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
    const int rc = emu_render_wf(mcrt_image_scene(img), &cam, 12345u, slots, cam.height, out.data(), counters);
    double s = 0;
    for (double v : out) s += v;
    printf("rc %d frame sum %.6f rays %llu\n", rc, s, (unsigned long long)counters[0]);
    return 0;
}
