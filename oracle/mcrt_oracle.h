/*
 * oracle/mcrt_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the reference hot path (linusmossberg/monte-carlo-ray-tracer):
 * Camera::samplePixel -> PathTracer/PhotonMapper::sampleRay -> Scene/BVH::intersect -> Interaction /
 * Material / GGX / Fresnel / Sampler -> LinearOctree::knnSearch -> Film::deposit, operating on the
 * flattened arrays of include/mcrt.h. Every function cites the reference file:line it follows and
 * keeps the reference's operation order, so that on the same libm it reproduces the reference's
 * FP64 output bit for bit (pinned in tests/test_oracle_vs_reference.py against dumps produced by the
 * reference itself, oracle/_ref/mcrt_ref, and against tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libmcrt_hip.so) never links, imports or calls anything under oracle/.
 */
#ifndef MCRT_ORACLE_H
#define MCRT_ORACLE_H

#include <stdint.h>
#include "../include/mcrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_counters {
    uint64_t paths, rays, node_tests, prim_tests, knn_searches, knn_octants, knn_photons, sphere_tests;
} oracle_counters;

/* Sampler (sampling/sampler.hpp:13-90): initiate(pixel), setIndex(index), `shuffles` x shuffle(),
 * then get<0,7>(). */
void oracle_sampler(uint32_t global_seed, uint32_t pixel, uint32_t index, uint32_t shuffles, double out[7]);

/* Scene::intersect (scene/scene.cpp:151-176) for n rays (start[3], direction[3]). */
void oracle_intersect(const mcrt_scene_desc* scene, uint64_t n, const double* start,
                      const double* direction, double* out_t, uint32_t* out_surface, double* out_uv,
                      oracle_counters* counters);

/* 1: closest hits as the true minimum with ties to the lowest surface index (the HIP walks' rule) instead of the reference's heap
 * order; 0 (default): the reference. See mcrt_oracle.c. Not thread safe against running renders. */
void oracle_set_true_minimum(int on);

/* LinearOctree<Photon>::knnSearch (octree/linear-octree.cpp:25-117); outputs sorted by
 * (distance2, index) ascending, [n][k]. */
void oracle_knn(const mcrt_photon_map_desc* map, uint64_t n, const double* p, uint32_t k,
                uint32_t* out_count, uint32_t* out_index, double* out_distance2);

/* Camera::samplePixel + Film (camera/camera.cpp:66-99, camera/film.cpp:61-113) over rows
 * [row0,row1): out_rgb[(row1-row0)*width*3]; out_samples (optional) [(row1-row0)*width*spp*3].
 * integrator: MCRT_INTEGRATOR_*. threads <= 0 -> all hardware threads. Returns 0, or <0 on error. */
int oracle_render(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* global_map,
                  const mcrt_photon_map_desc* caustic_map, uint32_t k_nearest, int direct_visualization,
                  const mcrt_camera_desc* cam, uint32_t global_seed, int integrator,
                  uint32_t row0, uint32_t row1, int threads, double* out_rgb, double* out_samples,
                  oracle_counters* counters, double* seconds);

/* Photon emission pass (integrator/photon-mapper/photon-mapper.cpp:24-115 work split and per-emission
 * set-up, :225-277 emitPhoton). emissions / caustic_factor are the "photon_map" JSON values. Photons are
 * written in (light, emission index, bounce) order as [n][8] floats (flux rgb, position xyz, phi, theta:
 * photon.hpp:7-12,36-37) with keys = light << 48 | emission index << 16 | bounce. Returns 0, or -1 when
 * a capacity was too small (counts then hold the required sizes). */
int oracle_emit_photons(const mcrt_scene_desc* scene, double emissions, double caustic_factor, uint32_t global_seed,
                        float* global_photons, uint64_t* global_keys, uint64_t global_capacity, uint64_t* global_count,
                        float* caustic_photons, uint64_t* caustic_keys, uint64_t caustic_capacity, uint64_t* caustic_count,
                        uint64_t* emission_paths, uint64_t* rays);

/* Study hooks (round 2's kNN hint study): record the kNN searches of a SINGLE-THREADED oracle_render into buf ([cap][6]:
 * map, x, y, z, pixel, sample); and run searches with a caller-given initial squared bound, counting octants / photons. */
void oracle_knn_recorder(double* buf, uint64_t cap);
uint64_t oracle_knn_recorded(void);
void oracle_knn_hinted(const mcrt_photon_map_desc* map, uint64_t n, const double* p, uint32_t k, const double* bound2,
                       double* out_kth_distance2, uint32_t* out_count, uint64_t* octants, uint64_t* photons);

/* Known-answer helpers for the BSDF building blocks (material/fresnel.cpp, material/ggx.cpp,
 * material/material.cpp). in[11] = wi(3) wo(3) n1 n2 alpha u v; consts[10] = roughness,
 * reflectance(3), complex ior real(3), imag(3); out[18] as written by oracle/ref_main.cpp doKat. */
void oracle_bsdf_kat(uint64_t n, const double* in, const double* consts, double* out);

/* Image::save (camera/image.cpp:37-88) of a width x height FP64 RGB frame: exposure and gain factors from the two
 * 65 536-bin histograms (image.cpp:62-87, common/histogram.cpp), the tone map (camera/pixel-operators.cpp), sRGB gamma
 * (color/srgb.hpp:55-63) and truncation to B,G,R bytes — the payload of the reference's .tga. tonemapper: MCRT_TONEMAP_*.
 * factors (may be NULL) receives {exposure_factor, gain_factor}. */
int oracle_image_save(const double* rgb, uint32_t width, uint32_t height, uint32_t tonemapper, int plain,
                      double exposure_compensation, double gain_compensation, uint8_t* bgr, double* factors);

int oracle_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
