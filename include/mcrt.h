/*
 * mcrt.h — C ABI of libmcrt_hip.so, the MI355X (gfx950) path-tracing / BVH-traversal /
 * photon-kNN engine that drops in behind linusmossberg/monte-carlo-ray-tracer's
 * Camera::sampleImage() (reference: source/camera/camera.cpp:101-145).
 *
 * The reference has no FFI; its seam is the C++ call Camera::sampleImage(), whose contract is
 * "on return camera.image(x,y) holds the filtered mean radiance of every pixel"
 * (camera/camera.cpp:138-144, camera/image.cpp:53-56). The entry points below are what a
 * maintainer binds in its place (binding stub: INTEGRATION.md). Everything is plain pointers and
 * sizes; no C++ types, no torch types, no exceptions cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success or a negative mcrt_status; mcrt_last_error() gives text
 *     (the reference's convention is exceptions caught in main, source/main.cpp:24-32,48-56);
 *   - all *_desc arrays are host memory owned by the caller and are copied during the call;
 *   - all floating point is FP64 unless the field says otherwise (the reference computes in
 *     glm::dvec3 everywhere; only stored photons are FP32, integrator/photon-mapper/photon.hpp:36-37);
 *   - the library never falls back to a CPU path: without a gfx950 device mcrt_create fails.
 */
#ifndef MCRT_H
#define MCRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCRT_ABI_VERSION 2u

typedef enum mcrt_status {
    MCRT_OK = 0,
    MCRT_ERR_INVALID = -1,      /* bad argument / inconsistent descriptor            */
    MCRT_ERR_NO_DEVICE = -2,    /* no gfx950 device / HIP runtime error at create    */
    MCRT_ERR_HIP = -3,          /* HIP runtime error (text in mcrt_last_error)       */
    MCRT_ERR_NO_SCENE = -4,     /* render called before mcrt_upload_scene            */
    MCRT_ERR_NO_PHOTONS = -5,   /* photon-mapping render without mcrt_upload_photons */
    MCRT_ERR_IO = -6,           /* scene-image file could not be read / parsed       */
    MCRT_ERR_UNSUPPORTED = -7   /* e.g. quadric surfaces, non-box film filter        */
} mcrt_status;

/* ------------------------------------------------------------------------------------------
 * Scene. Replaces the data reached through Integrator::scene (integrator/integrator.hpp:26):
 * Scene::{surfaces, emissives, cumulative_emissives_importance, bvh, ior} (scene/scene.hpp:27-40),
 * BVH::{linear_tree, ordered_surfaces} (bvh/bvh.hpp:105-108) and the per-surface / per-material
 * fields the hot path reads.
 * ---------------------------------------------------------------------------------------- */

enum { MCRT_SURF_TRIANGLE = 0, MCRT_SURF_SPHERE = 1, MCRT_SURF_QUADRIC = 2 };

/* Material flag bits = the bools of class Material (material/material.hpp:41-44) as they stand
 * AFTER scene construction (scene/scene.cpp:83-89 copies materials without recomputing them). */
enum {
    MCRT_MAT_ROUGH = 1u << 0,
    MCRT_MAT_ROUGH_SPECULAR = 1u << 1,
    MCRT_MAT_OPAQUE = 1u << 2,
    MCRT_MAT_EMISSIVE = 1u << 3,
    MCRT_MAT_DIRAC_DELTA = 1u << 4,
    MCRT_MAT_PERFECT_MIRROR = 1u << 5,
    MCRT_MAT_COMPLEX_IOR = 1u << 6
};

/* One record per Material object (material/material.hpp:9-55). emittance is the value left by
 * Scene::generateEmissives (scene/scene.cpp:202), i.e. radiosity, not flux. */
typedef struct mcrt_material {
    double reflectance[3];
    double specular_reflectance[3];
    double transmittance[3];
    double emittance[3];
    double roughness, specular_roughness, ior, transparency;
    double A, B;           /* Oren-Nayar terms, material.cpp:106-108       */
    double a[2];           /* GGX alpha, material.cpp:110                  */
    double ior_real[3];    /* ComplexIOR::real      (material/fresnel.hpp:6-11) */
    double ior_imag[3];    /* ComplexIOR::imaginary                         */
    uint32_t flags;        /* MCRT_MAT_*                                    */
    uint32_t reserved;
} mcrt_material;

typedef struct mcrt_scene_desc {
    uint32_t abi_version;              /* MCRT_ABI_VERSION */

    /* BVH::linear_tree in the reference's depth-first order (bvh/bvh.hpp:68-74).
     * num_nodes == 0 selects the brute-force loop of Scene::intersect (scene/scene.cpp:161-173). */
    uint32_t num_nodes;
    const double*   node_bounds;        /* [num_nodes][6]  BB.min xyz, BB.max xyz      */
    const uint32_t* node_start_surface; /* [num_nodes]                                  */
    const uint32_t* node_num_surfaces;  /* [num_nodes]  >0 ⇒ leaf (uint8 in reference) */
    const uint32_t* node_next_sibling;  /* [num_nodes]  0 ⇒ none                       */

    /* Surfaces in BVH::ordered_surfaces order (Scene::surfaces order when num_nodes == 0). */
    uint32_t num_surfaces;
    const uint8_t*  surf_kind;          /* MCRT_SURF_*                                          */
    const uint8_t*  surf_interpolate;   /* 1 ⇔ Triangle::N != nullptr (surface/triangle.cpp:56) */
    const uint32_t* surf_material;      /* index into materials                                 */
    const double*   surf_area;          /* Base::area_                                          */
    const double*   surf_v;             /* [n][9] triangle: v0,v1,v2 · sphere: origin,radius,0… · quadric: record index,0… */
    const double*   surf_e;             /* [n][9] triangle: E1,E2,normal_ · sphere: unused      */
    const double*   surf_vn;            /* [n][9] vertex normals N[0..2], or NULL if none       */

    uint32_t num_materials;
    const mcrt_material* materials;

    /* Scene::emissives / cumulative_emissives_importance in the order the reference produced. */
    uint32_t num_lights;
    const uint32_t* light_surface;      /* index into the surface arrays above */
    const double*   light_cdf;

    double scene_ior;                   /* Scene::ior (scene/scene.cpp:22)   */
    double bb_min[3], bb_max[3];        /* Scene::BB()                        */

    /* Surface::Quadric (surface/surface.hpp:99-116, surface/quadric.cpp). One record of 22 doubles per quadric:
     * Q as glm stores it (column-major: Q[c][r] at 4*c + r; the gradient matrix G is 2 * its upper 3 rows,
     * quadric.cpp:38-45), then BB_.min, BB_.max — the box that slices the quadric (quadric.cpp:36,72-76,92).
     * Surface i of kind MCRT_SURF_QUADRIC names its record in surf_v[9*i] (an integer stored as a double).
     * Quadrics cannot be emissive (scene/scene.cpp:125). */
    uint32_t num_quadrics;
    const double*   quadrics;           /* [num_quadrics][22] */
} mcrt_scene_desc;

/* ------------------------------------------------------------------------------------------
 * Photon maps. Replaces PhotonMapper::{caustic_map, global_map} =
 * LinearOctree<Photon>::{linear_tree, ordered_data} (octree/linear-octree.hpp:19-29).
 * ---------------------------------------------------------------------------------------- */
typedef struct mcrt_photon_map_desc {
    uint32_t num_octants;
    const double*   octant_bounds;          /* [n][6] tight BB min,max (linear-octree.cpp:220,241) */
    const uint64_t* octant_start_data;
    const uint64_t* octant_contained_data;
    const uint32_t* octant_next_sibling;    /* 0xFFFFFFFF ⇒ none (linear-octree.hpp:36) */
    const uint8_t*  octant_leaf;
    uint64_t num_photons;
    const float*    photons;                /* [n][8] flux rgb, position xyz, phi, theta (photon.hpp:36-37) */
} mcrt_photon_map_desc;

/* ------------------------------------------------------------------------------------------
 * Camera. The public fields of class Camera read by samplePixel (camera/camera.hpp:39-51,
 * camera/camera.cpp:66-99) plus the work split.
 * ---------------------------------------------------------------------------------------- */
typedef struct mcrt_camera_desc {
    double eye[3], forward[3], left[3], up[3];
    double focal_length, sensor_width, aperture_radius, focus_distance;
    uint32_t thin_lens;
    uint32_t width, height;
    uint32_t sqrtspp;
    /* Image-space sharding (multi-GPU): rows are dealt to shards in groups of shard_rows rows,
     * group g belongs to shard g % shard_count. shard_count <= 1 renders every row. Per-pixel
     * seeding stays hashCombine(global_seed, hash(y*width+x)) (camera.cpp:73, sampler.hpp:32-35)
     * so the image is independent of the split. */
    uint32_t shard_index, shard_count, shard_rows;
    /* Film reconstruction filter (camera/film.cpp:19-58, camera/filter.hpp): MCRT_FILM_BOX is the reference's default
     * Film(width, height): radius 0.5, every sample lands in its own pixel with weight 1. Any other filter makes every
     * sample a splat over the pixels within film_radius (0 = the filter's default radius, film.cpp:31-44), weights from
     * the filter function or, when film_cache_size > 0, from a table of that many samples of it (film.cpp:49-57,86-97);
     * such frames are rendered by the wavefront pipeline (either integrator; shard_count <= 1, or sharded through
     * mcrt_render_film_device; a scene without a BVH is walked through a tree over index ranges there). */
    uint32_t film_filter;
    double   film_radius;
    uint32_t film_cache_size;
    uint32_t reserved;
} mcrt_camera_desc;

enum { MCRT_FILM_BOX = 0, MCRT_FILM_MITCHELL_NETRAVALI = 1, MCRT_FILM_CATMULL_ROM = 2, MCRT_FILM_B_SPLINE = 3, MCRT_FILM_HERMITE = 4,
       MCRT_FILM_GAUSSIAN = 5, MCRT_FILM_LANCZOS = 6 };

enum { MCRT_INTEGRATOR_PATH_TRACER = 0, MCRT_INTEGRATOR_PHOTON_MAPPER = 1 };

typedef struct mcrt_stats {
    uint64_t paths;         /* pixel samples traced = Integrator::sampleRay calls             */
    uint64_t rays;          /* closest-hit queries  = Scene::intersect calls (bounce + shadow) */
    uint64_t node_tests;    /* BoundingBox::intersect calls performed by the GPU traversal     */
    uint64_t prim_tests;    /* primitive intersect calls performed by the GPU traversal        */
    uint64_t knn_searches;  /* LinearOctree::knnSearch calls                                   */
    double   kernel_ms;     /* HIP-event time of the integrator kernel(s) on their stream      */
    double   total_ms;      /* wall time of the call incl. copies                              */
    uint32_t kernel_launches;
    uint32_t kernel_id;     /* MCRT_KERNEL_*: which kernel form traced the frame (tests pin it, so a silent
                             * change of the selection rule in launchRender cannot pass unnoticed)            */
} mcrt_stats;

/* Kernel forms of the integrator; mcrt_stats.kernel_id of the last mcrt_render*. */
enum {
    MCRT_KERNEL_NONE = 0,          /* nothing launched (a shard that owns no rows)                                   */
    MCRT_KERNEL_FLAT = 1,          /* renderKernel, flat-scene instance: wave-uniform loop over all primitives, scene in LDS */
    MCRT_KERNEL_WAVESYNC = 2,      /* renderKernel<path tracer>, wave-synchronous bounce loop (MCRT_KERNEL=legacy, instrumented runs) */
    MCRT_KERNEL_LANE_SM = 3,       /* renderKernelSM: per-lane state machine megakernel                               */
    MCRT_KERNEL_WAVEFRONT = 4,     /* wfShadeKernel + wfTraceKernel over the slot pool                               */
    MCRT_KERNEL_PM_WAVE = 5,       /* renderKernelPM: photon mapper, wave-cooperative kNN estimates                  */
    MCRT_KERNEL_PM_LANE = 6,       /* renderKernel<photon mapper>: per-lane kNN (k > 768, MCRT_KERNEL=legacy)        */
    MCRT_KERNEL_WAVEFRONT_PM = 7   /* wavefront pipeline with kNN launches (MCRT_KERNEL=wf, photon-mapped frames)    */
};

typedef struct mcrt_ctx mcrt_ctx; /* opaque; owns all device memory and one HIP stream */

/* Lifecycle. device_id = HIP ordinal of the gfx950 device this context drives (one context per
 * device / per process rank). */
int  mcrt_create(mcrt_ctx** out, int device_id);
/* Number of HIP devices this process sees (0: none, or the runtime cannot start) - what a one-process host sizes its set of contexts
 * by before mcrt_render_multi; the reference's counterpart is std::thread::hardware_concurrency() (integrator/integrator.cpp:20-24).
 * mcrt_create still refuses a device that is not gfx950. */
int  mcrt_device_count(void);
void mcrt_destroy(mcrt_ctx* ctx);
const char* mcrt_last_error(const mcrt_ctx* ctx); /* ctx may be NULL: last create error */

/* Run-time options of a context: kernel selection and tuning knobs for A/B runs and parity tests (DESIGN.md "Run-time
 * options" lists them; defaults are the measured best). Keys are MCRT_* names, values decimal numbers or words. The process
 * environment SEEDS the options once, in mcrt_create (so `MCRT_KERNEL=wf ./host` still works); after that the library never
 * reads the environment: a host changes behaviour with mcrt_set_option, between frames (value NULL = back to the default).
 * Options that shape the uploaded scene (MCRT_FLAT_MAX) take effect at the next mcrt_upload_scene.
 * The reference has no counterpart (its knobs are compile-time constants); mcrt_get_option returns the value or NULL. */
int mcrt_set_option(mcrt_ctx* ctx, const char* key, const char* value);
const char* mcrt_get_option(const mcrt_ctx* ctx, const char* key);

int mcrt_upload_scene(mcrt_ctx* ctx, const mcrt_scene_desc* scene);
int mcrt_upload_photons(mcrt_ctx* ctx, const mcrt_photon_map_desc* global_map,
                        const mcrt_photon_map_desc* caustic_map,
                        uint32_t k_nearest_photons, int direct_visualization);

/* Replaces the thread fan-out of Camera::sampleImage (camera.cpp:120-144): renders every owned
 * pixel with spp = sqrtspp^2 samples and writes image(x,y) as FP64 RGB, row-major, width*height*3
 * doubles, into caller-allocated HOST memory (rows not owned by this shard are left untouched).
 * global_seed replaces Sampler::global_seed (sampling/sampler.hpp:58).
 * PARITY: a path-traced frame (MCRT_INTEGRATOR_PATH_TRACER) of the default (exact) library is the reference's bits. A photon-mapped
 * frame of the default kernels is held to 1e-10 relative of the reference, not to its bits: the k photons of a radiance estimate are
 * added by a wave reduction instead of in the reference's heap order, and Photon::dir takes the platform's sinf / cosf (measured
 * <= 1e-12). Option MCRT_KERNEL=legacy selects the per-lane photon kernel, which keeps the reference's heap discipline and its
 * sincosf - its frames ARE the reference's bits, at a tenth of the speed. The opt-in tolerance library (libmcrt_hip_tol.so) is held
 * to BASELINE.json's 1e-4 for every frame. */
int mcrt_render(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator,
                double* out_rgb, mcrt_stats* stats /* may be NULL */);

/* Same, but the image stays in DEVICE memory owned by the caller (e.g. the buffer RCCL gathers
 * from) and the launch is asynchronous on `stream` (a hipStream_t; NULL = the context's stream).
 * d_out_rgb holds owned rows only, packed in ascending row order: mcrt_shard_rows() rows of
 * width*3 doubles. Call mcrt_render_finish() to wait and collect stats. (Path-traced frames whose tree is walked by the
 * wavefront pipeline run as a host-driven sequence of launches on `stream`; for them the call BLOCKS until the frame is complete and
 * mcrt_render_finish() only collects the statistics. The pipeline is chosen for BVHs of 65 536 nodes or more (MCRT_WF_MIN_NODES) and,
 * for ANY tree that is not staged whole in LDS, for calls of 32 M path samples or more (MCRT_WF_MIN_PATHS) — counted over the rows THIS
 * call owns, the work the pipeline's launches are amortised over: the measured crossover. Shards of one frame that
 * straddle the threshold may therefore run different kernel forms on different ranks; every form returns the same bits, so the frame
 * does not depend on it, only mcrt_stats.kernel_id and whether this call blocks. MCRT_KERNEL=wf / sm pins the form.)
 * ORDERING: queue consumers of d_out_rgb (copies, reductions, the gather) AFTER mcrt_render_finish() has returned MCRT_OK, not right
 * after this call: a megakernel frame in which a path nests deeper than the eight dielectric media a lane keeps in LDS is rendered
 * AGAIN by mcrt_render_finish through the wavefront pipeline (32 media; mcrt_stats.kernel_id then says MCRT_KERNEL_WAVEFRONT[_PM] and
 * kernel_ms / kernel_launches describe that second run), into the same d_out_rgb on the same stream. */
int mcrt_render_device(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed,
                       int integrator, double* d_out_rgb, void* stream);
int mcrt_render_finish(mcrt_ctx* ctx, mcrt_stats* stats /* may be NULL */);

/* Reconstruction-filter frames across GPUs. With a filter other than the box a sample is a splat over
 * the pixels within the filter radius (Film::deposit, camera/film.cpp:61-79), so the samples of one shard's rows also
 * land in its neighbours' rows: every shard accumulates into a FULL-frame copy of Film's blob — width*height records of
 * {rgb_sum[3], weight_sum} (camera/film.hpp:22-37), 4 doubles each, overwritten by this call — the host sums the shards'
 * buffers (one RCCL all-reduce / reduce; with shard_count 1 there is nothing to sum) and mcrt_film_resolve_device applies
 * Splat::get (film.hpp:31-35, Film::scan film.cpp:81-84) to the sum: width*height*3 doubles, full frame. The reference
 * adds the same splats with std::atomic<double> in thread-timing order, so sums agree to rounding (1e-12), not bits.
 * cam->film_filter must not be MCRT_FILM_BOX. Returns when the shard's samples are complete
 * (mcrt_render_finish() then collects the statistics). */
int mcrt_render_film_device(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, double* d_rgbw,
                            void* stream);
int mcrt_film_resolve_device(mcrt_ctx* ctx, uint32_t width, uint32_t height, const double* d_rgbw, double* d_out_rgb, void* stream);

/* One host process driving several GPUs — the shape of the reference's host, whose Camera::sampleImage fans out to worker
 * threads (camera/camera.cpp:120-136). ctxs[i] is a context on device i's GPU with the scene (and photon maps) already
 * uploaded; the frame's rows are dealt over the contexts (cam->shard_rows per group, 0 = 8; cam->shard_index/count are
 * ignored), every context is driven by its own host thread and copies its rows into out_rgb (full frame, host memory). No
 * collective: shards are independent and Image::save wants the frame on the host anyway. Reconstruction-filter frames are
 * summed and resolved on the host. stats: counters summed over the contexts, times = the slowest context's. The result does
 * not depend on the number of contexts (box filter: bit for bit). */
int mcrt_render_multi(mcrt_ctx* const* ctxs, uint32_t count, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator,
                      double* out_rgb, mcrt_stats* stats /* may be NULL */);

/* Number of rows owned by (shard_index, shard_count, shard_rows) of `cam`, and their indices. */
uint32_t mcrt_shard_rows(const mcrt_camera_desc* cam, uint32_t* rows /* may be NULL */);

/* Photon emission pass on the GPU. Replaces the thread fan-out of
 * PhotonMapper::PhotonMapper (integrator/photon-mapper/photon-mapper.cpp:80-115: per emission
 * Sampler::initiate(light), setIndex(offset+i), light point + cosine direction, emitPhoton :225-277) for
 * the uploaded scene. `emissions` and `caustic_factor` are the "photon_map" JSON values (:31-38); the
 * split of emissions over lights follows :43-78. The photons come back as two unordered lists in the
 * reference's 32-byte Photon layout (photon.hpp:36-37) — the host then builds its octrees from them
 * exactly as it does from the per-thread vectors (:190-207) and calls mcrt_upload_photons.
 * keys (optional diagnostics) identify a photon: light << 48 | emission index << 16 | bounce.
 * The arrays belong to the context and stay valid until the next mcrt_emit_photons / mcrt_destroy. */
typedef struct mcrt_photon_emission {
    uint64_t global_count, caustic_count;
    const float* global_photons;    /* [global_count][8]  flux rgb, position xyz, phi, theta */
    const float* caustic_photons;   /* [caustic_count][8]                                   */
    const uint64_t* global_keys;    /* [global_count]  */
    const uint64_t* caustic_keys;   /* [caustic_count] */
    uint64_t emission_paths;        /* photon paths traced = emissions * caustic_factor, split per light */
    uint64_t rays;                  /* Scene::intersect calls */
    double kernel_ms;
} mcrt_photon_emission;
int mcrt_emit_photons(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed,
                      mcrt_photon_emission* out);
/* Same, restricted to one shard of the photon paths (multi-GPU emission): the paths
 * are numbered 0..emission_paths-1 in (light, emission index) order and shard k of n takes the k-th
 * contiguous block, so the union of the n lists is exactly the unsharded result; the ranks then
 * all-gather their lists (RCCL) and every rank builds/uploads the full maps. */
int mcrt_emit_photons_shard(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed,
                            uint32_t shard_index, uint32_t shard_count, mcrt_photon_emission* out);

/* ---- The photon pass without host round trips (SURVEY.md §8(f) ranks 1 + 2; replaces PhotonMapper::PhotonMapper,
 * integrator/photon-mapper/photon-mapper.cpp:31-203, as a whole): the emission pass, then both maps built where the
 * photons are — cell codes, radix sort, gather, the octants level by level from the sorted codes, depth-first numbering,
 * leaf boxes merged upwards, the search's record lists — and installed as mcrt_upload_photons installs them. Same photons
 * as mcrt_emit_photons, same octants / boxes / photons per leaf as mcrt_photon_map_build (Octree<Photon>::insert,
 * octree/octree.cpp:35-80 + LinearOctree, octree/linear-octree.cpp:202-244) with root cell [bb_min, bb_max] = Scene::BB().
 * Only counters cross PCIe. MCRT_ERR_UNSUPPORTED when more than max_photons_per_leaf photons share one 2^-21 cell (the
 * recursive host builder's case). */
typedef struct mcrt_photon_pass_stats {
    uint64_t global_count, caustic_count;   /* photons stored */
    uint64_t global_octants, caustic_octants;
    uint64_t emission_paths, rays;
    double emission_ms;                     /* emission kernel (HIP events) */
    double sort_ms, octant_ms, finish_ms;   /* both maps: codes + sort + gather / octants + numbering / boxes + record lists (host clock) */
    double total_ms;                        /* whole call (host clock) */
} mcrt_photon_pass_stats;
int mcrt_photon_pass_device(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed, const double bb_min[3],
                            const double bb_max[3], uint32_t max_photons_per_leaf, uint32_t k_nearest_photons,
                            int direct_visualization, mcrt_photon_pass_stats* stats);
/* The same pass for the contexts of ONE host process (the reference's shape: one executable, PhotonMapper::PhotonMapper
 * fanning its emission work out to worker threads, integrator/photon-mapper/photon-mapper.cpp:40-115), sharded: context i traces
 * shard i of `count` of the emission paths, the lists cross between the GPUs on device pointers (hipMemcpyPeer: xGMI between
 * two devices), every context builds both maps from the same concatenation in shard order - the same maps everywhere, the
 * emission's time divided by `count`. ctxs[i] holds the scene already; count == 1 is mcrt_photon_pass_device.
 * stats: one record PER CONTEXT ([count]; may be NULL): emission_paths / rays / emission_ms of its own shard, the maps' counts. */
int mcrt_photon_pass_multi(mcrt_ctx* const* ctxs, uint32_t count, double emissions, double caustic_factor, uint32_t global_seed,
                           const double bb_min[3], const double bb_max[3], uint32_t max_photons_per_leaf, uint32_t k_nearest_photons,
                           int direct_visualization, mcrt_photon_pass_stats* stats);
/* The emission pass alone with the lists left in device memory (owned by the context, valid until the next emission): for
 * hosts that exchange the lists between GPUs (RCCL all-gather on the device pointers) before building the maps. */
typedef struct mcrt_photon_emission_device {
    uint64_t global_count, caustic_count;
    const float* d_global_photons;    /* DEVICE [global_count][8]  */
    const float* d_caustic_photons;   /* DEVICE [caustic_count][8] */
    uint64_t emission_paths, rays;
    double kernel_ms;
} mcrt_photon_emission_device;
int mcrt_emit_photons_device(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed, uint32_t shard_index,
                             uint32_t shard_count, mcrt_photon_emission_device* out);
/* Both maps from photon lists in device memory (not modified), installed like mcrt_upload_photons.
 * STREAM CONTRACT for every device-pointer INPUT of this header (these lists; mcrt_film_resolve_device's blob): the library reads them
 * on the context's own non-blocking stream, which is not ordered after any stream of the caller - the data must be COMPLETE
 * (producing stream synchronised, e.g. hipStreamSynchronize / torch.cuda.synchronize) when the call is made. Device OUTPUTS
 * (mcrt_render_device's d_out) are written on the stream the caller passes and are ordered like any work on that stream. */
int mcrt_upload_photons_device(mcrt_ctx* ctx, const float* d_global_photons, uint64_t global_count, const float* d_caustic_photons,
                               uint64_t caustic_count, const double bb_min[3], const double bb_max[3], uint32_t max_photons_per_leaf,
                               uint32_t k_nearest_photons, int direct_visualization, mcrt_photon_pass_stats* stats);
/* Copy of an installed map (0 global, 1 caustic) as a host object — octants, boxes, photons — for inspection and tests. */
int mcrt_photon_map_download(mcrt_ctx* ctx, int which, struct mcrt_photon_map** out);

/* ---- operator-level entry points (each mirrors one reference function; used by parity tests
 * and by hosts that only want the traversal / kNN engine) -------------------------------- */

/* Scene::intersect (scene/scene.cpp:151-176) for n rays given as start[3], direction[3] (FP64,
 * host). Outputs: t (DBL_MAX when no hit, ray/intersection.hpp:14), surface index (0xFFFFFFFF when
 * no hit), uv[2] (meaningful when the surface interpolates normals). */
int mcrt_intersect(mcrt_ctx* ctx, uint64_t n, const double* start, const double* direction,
                   double* out_t, uint32_t* out_surface, double* out_uv);

/* Sampler (sampling/sampler.hpp:13-90): for each (pixel[i], index[i]) runs initiate(pixel),
 * setIndex(index), then `shuffles` times shuffle(); writes get<0,7>() after the last step
 * (shuffles == 0 gives the un-shuffled camera dimensions). out[n][7]. */
int mcrt_sampler(mcrt_ctx* ctx, uint64_t n, const uint32_t* pixel, const uint32_t* index,
                 uint32_t shuffles, uint32_t global_seed, double* out);

/* The lobe functions Interaction::BSDF is made of (ray/interaction.cpp:84-153), on n local-frame vectors (host arrays):
 * in[n][11] = wi[3], wo[3], n1, n2, alpha, u, v (wi is folded into the upper hemisphere as |z| + 1e-3, renormalised, for the
 * reflection lobes and negated for transmission); consts[10] = roughness, reflectance[3] of an Oren-Nayar material and a
 * complex IOR real[3], imaginary[3]. out[n][18] = Fresnel::dielectric(n1, n2, wo.z) (material/fresnel.cpp:16-27) ·
 * Fresnel::conductor rgb (:30-49) · GGX::reflection f, pdf (material/ggx.cpp:46-52) · GGX::transmission f, pdf (:54-65) ·
 * GGX::visibleMicrofacet(u, v, wo) xyz (:67-88) · GGX::D(m) (:21-24) · GGX::Lambda(wo) (:31-34) ·
 * Material::diffuseReflection rgb, pdf (material/material.cpp:17-27,82-95) · 0. No scene needed. */
int mcrt_bsdf(mcrt_ctx* ctx, uint64_t n, const double* in, const double* consts, double* out);

/* The libm calls of the path as the DEVICE computes them (csrc/mcrt_libm.hpp: glibc 2.35's algorithms restated so that the GPU returns
 * the bits the reference's std::sin / std::cos pairs (= sincos: sampling/sampling.hpp:29-44, material/ggx.cpp:77-79, surface/sphere.cpp:43),
 * std::sin alone (camera/filter.hpp:64), std::asin (scene/scene.cpp:221) and std::atan2 (integrator/photon-mapper/photon.hpp:10-11)
 * return on an x86-64 host with FMA; SINCOSF: sincosf, the sine / cosine pairs of Photon::dir's two float angles, photon.hpp:19-27 - a[n]
 * holds float values, out0 / out1 the float results widened). fn selects the function; a[n] (and b[n] for atan2: a = y, b = x, and for pow: a^b) are the
 * arguments, out0[n] (and out1[n] for sincos / sincosf: out0 = sine, out1 = cosine) the results. Known-answer tests only; no scene needed. */
enum { MCRT_LIBM_SINCOS = 0, MCRT_LIBM_SIN = 1, MCRT_LIBM_COS = 2, MCRT_LIBM_ASIN = 3, MCRT_LIBM_ATAN2 = 4, MCRT_LIBM_SINCOSF = 5,
       MCRT_LIBM_POW = 6 /* std::pow(a, b): sRGB::gammaCompress, color/srgb.hpp:54-62 - the one libm call of Image::save (csrc/mcrt_libm_pow.hpp) */ };
int mcrt_libm(mcrt_ctx* ctx, int fn, uint64_t n, const double* a, const double* b, double* out0, double* out1);

/* LinearOctree<Photon>::knnSearch (octree/linear-octree.cpp:25-117) on the uploaded map
 * (which = 0 global, 1 caustic) for n query points p[n][3]. Outputs per query: count found
 * (≤ k), photon indices and squared distances sorted by ascending distance (ties by index),
 * each [n][k]; unused slots get 0xFFFFFFFF / +inf. Any k: k <= 768 is served by the wave-cooperative search,
 * larger k by the per-lane search. The search's frontier is unbounded like the reference's priority queue
 * (linear-octree.cpp:33): a wave-cooperative search that fills its 128 + 1 024 entries is repeated by the per-lane
 * search, whose frontier grows on demand (the same holds for photon-mapped frames: mcrt_render_finish). */
int mcrt_knn(mcrt_ctx* ctx, int which, uint64_t n, const double* p, uint32_t k,
             uint32_t* out_count, uint32_t* out_index, double* out_distance2);

/* ------------------------------------------------------------------------------------------
 * Scene-image files (*.mcrt): a flat dump of the three descriptors above, written by the
 * flattener that runs inside the reference host and read by stand-alone hosts.
 * Host-only helpers; they never touch the GPU.
 * ---------------------------------------------------------------------------------------- */
typedef struct mcrt_image mcrt_image; /* opaque; owns the host arrays the descs point into */

int  mcrt_image_load(const char* path, mcrt_image** out);
void mcrt_image_free(mcrt_image* img);
const mcrt_scene_desc*      mcrt_image_scene(const mcrt_image* img);
const mcrt_camera_desc*     mcrt_image_camera(const mcrt_image* img);
const mcrt_photon_map_desc* mcrt_image_photons(const mcrt_image* img, int which /*0 global,1 caustic*/);
/* key = "k_nearest_photons" | "direct_visualization" | "global_seed" | "photon_mapping"; 0 if absent */
uint64_t mcrt_image_param(const mcrt_image* img, const char* key);
int mcrt_image_save(const char* path, const mcrt_scene_desc* scene, const mcrt_camera_desc* cam,
                    const mcrt_photon_map_desc* global_map, const mcrt_photon_map_desc* caustic_map,
                    const char* const* param_keys, const uint64_t* param_values, uint32_t num_params);

/* Host-side photon-map builder: photon list (e.g. from mcrt_emit_photons) -> linear octree with the
 * semantics of Octree<Photon>::insert + LinearOctree<Photon> (octree/octree.cpp:35-80,
 * octree/linear-octree.cpp:202-244): root cell = Scene::BB(), a cell with more than
 * max_photons_per_leaf photons is split into its 8 octants, empty octants are dropped, node boxes are
 * tight. For hosts that do not carry the reference's builder; never touches the GPU. */
typedef struct mcrt_photon_map mcrt_photon_map; /* opaque; owns the arrays its descriptor points into */
int  mcrt_photon_map_build(const float* photons, uint64_t num_photons, const double bb_min[3], const double bb_max[3],
                           uint32_t max_photons_per_leaf, mcrt_photon_map** out);
/* The same tree (same octants, boxes and photons per leaf; photons of a leaf in cell-code order instead of input
 * order) built with the GPU of `ctx`: per-photon octant path codes, radix sort, gather and leaf boxes on the device,
 * octant assembly from the sorted codes on the host (replaces the serial insert loop of
 * PhotonMapper::PhotonMapper, photon-mapper.cpp:169-203, and LinearOctree's compaction, linear-octree.cpp:202-244).
 * Falls back to mcrt_photon_map_build when more than max_photons_per_leaf photons share one 2^-21 cell. */
int  mcrt_photon_map_build_gpu(mcrt_ctx* ctx, const float* photons, uint64_t num_photons, const double bb_min[3],
                               const double bb_max[3], uint32_t max_photons_per_leaf, mcrt_photon_map** out);
const mcrt_photon_map_desc* mcrt_photon_map_get(const mcrt_photon_map* map);
void mcrt_photon_map_free(mcrt_photon_map* map);

/* ------------------------------------------------------------------------------------------
 * BVH builder for hosts that do not carry the reference's: the reference's DEFAULT hierarchy ("bvh": {"type":
 * "octree"}, bvh/bvh.cpp:41-56,130-163,428-449) — an octree over the surfaces' box centroids, leaves of at most 8
 * surfaces, one node per non-empty octant with the union of its surfaces' boxes — built by sorting per-surface octant
 * path codes instead of inserting one surface at a time. The result is the reference's tree
 * bit for bit: same LinearNode arrays, same surface order (tests/test_bvh_build.py rebuilds every golden octree BVH,
 * up to the 6.9 M-triangle C5 scene). `scene` needs its surface arrays, quadrics, bb_min/bb_max; its node arrays are
 * ignored. ctx != NULL: boxes, codes and the radix sort run on the GPU of ctx; ctx == NULL: host only.
 * MCRT_ERR_UNSUPPORTED when more than 8 centroids share one 2^-21 cell of the root cube. */
typedef struct mcrt_bvh_desc {
    uint32_t num_nodes;
    const double*   node_bounds;        /* as mcrt_scene_desc */
    const uint32_t* node_start_surface;
    const uint32_t* node_num_surfaces;
    const uint32_t* node_next_sibling;
    uint32_t num_surfaces;
    const uint32_t* order;              /* [num_surfaces] BVH::ordered_surfaces: new position -> index in `scene` */
} mcrt_bvh_desc;
typedef struct mcrt_bvh mcrt_bvh;       /* opaque; owns the arrays its descriptor points into */
int  mcrt_bvh_build_octree(mcrt_ctx* ctx /* may be NULL */, const mcrt_scene_desc* scene, mcrt_bvh** out);
/* The reference's other two hierarchies, "binary_sah" (arity 2) and "quaternary_sah" (arity 4): its binned surface-area
 * builders (bvh/bvh.cpp:165-426) restated with the same arithmetic and tie rules — same tree — and run on `threads` host
 * threads (0 = all), one subtree per thread. bins_per_axis 0 = the reference's default (16 / 8). Host only. */
int  mcrt_bvh_build_sah(const mcrt_scene_desc* scene, int arity, uint32_t bins_per_axis, uint32_t threads, mcrt_bvh** out);
/* The same two hierarchies built LEVEL BY LEVEL — all open nodes of a depth at once: centroid bounds, binning and the
 * order-preserving partition are passes over the surfaces on the GPU of ctx (atomics on exact minima / maxima / counts, one
 * prefix sum per level), the split of every open node is decided by one thread with the reference's cost loop, the host
 * only strings the nodes together (csrc/mcrt_sah_shared.hpp, mcrt_sah_gpu.hip). Same tree as mcrt_bvh_build_sah and the
 * reference, bit for bit. ctx == NULL runs the same level loop on the host (one thread). bins_per_axis <= 16. */
int  mcrt_bvh_build_sah_gpu(mcrt_ctx* ctx /* may be NULL */, const mcrt_scene_desc* scene, int arity, uint32_t bins_per_axis, mcrt_bvh** out);
const mcrt_bvh_desc* mcrt_bvh_get(const mcrt_bvh* bvh);
void mcrt_bvh_free(mcrt_bvh* bvh);
/* `scene` with its surfaces put in bvh->order, its lights re-indexed and the node arrays of `bvh`: an owning copy whose
 * descriptor can go to mcrt_upload_scene / mcrt_image_save. */
typedef struct mcrt_scene mcrt_scene;
int  mcrt_scene_with_bvh(const mcrt_scene_desc* scene, const mcrt_bvh_desc* bvh, mcrt_scene** out);
const mcrt_scene_desc* mcrt_scene_get(const mcrt_scene* scene);
void mcrt_scene_free(mcrt_scene* scene);

/* ------------------------------------------------------------------------------------------
 * Image::save on the GPU (camera/image.cpp:37-88; SURVEY.md §8(f) rank 4, "tonemap/exposure"): what the reference does with
 * the frame after Camera::sampleImage — auto exposure from the median of a 65 536-bin brightness histogram (getExposure
 * :62-72, common/histogram.cpp:6-41), auto gain from the 99th percentile of the tone-mapped brightness (getGain :77-87),
 * the tone map (camera/pixel-operators.cpp:7-44), sRGB gamma (color/srgb.hpp:55-63) and truncation to bytes in B,G,R
 * order (pixel-operators.cpp:51-55) — as five kernels on the frame where mcrt_render_device left it, so that 3 bytes per
 * pixel cross PCIe instead of 24. Every step but pow() is IEEE-exact and in the reference's order; a byte can differ from
 * the reference's only where pow's last bit moves a value across an integer (tests allow 1 step on < 1e-4 of the bytes).
 * The fields are the scene file's camera "image" object (image.cpp:10-35). */
enum { MCRT_TONEMAP_HABLE = 0, MCRT_TONEMAP_ACES = 1 };
typedef struct mcrt_image_desc {
    uint32_t width, height;          /* of the buffer handed in (the full frame: exposure is a whole-image statistic) */
    uint32_t tonemapper;             /* MCRT_TONEMAP_*; "tonemapper" (default Hable)                                  */
    uint32_t plain;                  /* "plain": no tone map, exposure and gain 1                                      */
    double exposure_compensation;    /* EV: exposure = 0.5 / median * 2^EV                                             */
    double gain_compensation;        /* EV                                                                             */
} mcrt_image_desc;
/* d_rgb: width*height*3 doubles in device memory; d_bgr: width*height*3 bytes in device memory (rows top to bottom, the
 * TGA payload). factors (host, may be NULL) receives {exposure_factor, gain_factor}. Synchronous on `stream`. */
int mcrt_tonemap_device(mcrt_ctx* ctx, const double* d_rgb, const mcrt_image_desc* image, uint8_t* d_bgr, double* factors,
                        void* stream);
/* Same from/to host memory (copies in, runs the kernels, copies the bytes out). */
int mcrt_tonemap(mcrt_ctx* ctx, const double* rgb, const mcrt_image_desc* image, uint8_t* bgr, double* factors);
/* HeaderTGA + payload (camera/image.hpp:39-50, image.cpp:42-51): uncompressed 24 bpp, top-left origin. Host only. */
int mcrt_tga_save(const char* path, uint32_t width, uint32_t height, const uint8_t* bgr);

uint32_t mcrt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MCRT_H */
