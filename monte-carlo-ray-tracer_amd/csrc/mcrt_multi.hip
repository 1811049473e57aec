// mcrt_render_multi: one host process driving several GPUs — the shape of the reference's host (a single executable,
// Camera::sampleImage fanning out to worker threads, camera/camera.cpp:120-136). One host thread per context renders that
// context's shard of the frame (rows dealt in groups of 8, round-robin) and copies its rows into the caller's frame; no
// collective, no inter-GPU traffic: the shards are independent and the frame is assembled in host memory, which is where
// Image::save wants it. Frames with a reconstruction filter accumulate per-context {rgb_sum, weight_sum} buffers that are
// summed and resolved on the host (Splat::get, film.cpp:107-113).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "mcrt_film.hpp"
#include "mcrt_internal.hpp"

using namespace mcrt;

extern "C" int mcrt_render_multi(mcrt_ctx* const* ctxs, uint32_t count, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator,
                                 double* out_rgb, mcrt_stats* stats) {
    if (!ctxs || count == 0 || !ctxs[0]) return MCRT_ERR_INVALID;
    mcrt_ctx* first = ctxs[0];
    if (!cam || !out_rgb) return ctxFail(first, MCRT_ERR_INVALID, "mcrt_render_multi: cam or out_rgb is NULL");
    for (uint32_t i = 0; i < count; i++)
        if (!ctxs[i]) return ctxFail(first, MCRT_ERR_INVALID, "mcrt_render_multi: NULL context");
    const bool splats = filmSplats(cam->film_filter, cam->film_radius);
    const size_t pixels = (size_t)cam->width * cam->height;

    std::vector<int> rc(count, MCRT_OK);
    std::vector<mcrt_stats> st(count);
    std::vector<std::vector<double>> rgbw(splats ? count : 0);
    auto workBody = [&](uint32_t i) {
        mcrt_camera_desc shard = *cam;
        shard.shard_index = i;
        shard.shard_count = count;
        if (shard.shard_rows == 0) shard.shard_rows = 8;
        memset(&st[i], 0, sizeof(mcrt_stats));
        if (!splats) {  // owned rows straight into the caller's frame (disjoint rows per context)
            rc[i] = mcrt_render(ctxs[i], &shard, global_seed, integrator, out_rgb, &st[i]);
            return;
        }
        void* d = nullptr;
        if (hipSetDevice(ctxDevice(ctxs[i])) != hipSuccess || hipMalloc(&d, pixels * 4 * sizeof(double)) != hipSuccess) {
            rc[i] = ctxFail(ctxs[i], MCRT_ERR_HIP, "mcrt_render_multi: cannot allocate the splat buffer");
            return;
        }
        rc[i] = mcrt_render_film_device(ctxs[i], &shard, global_seed, integrator, (double*)d, nullptr);
        if (rc[i] == MCRT_OK) rc[i] = mcrt_render_finish(ctxs[i], &st[i]);
        if (rc[i] == MCRT_OK) {
            rgbw[i].resize(pixels * 4);
            if (hipMemcpy(rgbw[i].data(), d, pixels * 4 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                rc[i] = ctxFail(ctxs[i], MCRT_ERR_HIP, "mcrt_render_multi: copying the splat buffer failed");
        }
        (void)hipFree(d);
    };
    auto work = [&](uint32_t i) {  // a worker thread must not let an exception (bad_alloc from resize) reach std::terminate
        try {
            workBody(i);
        } catch (const std::exception& e) {
            rc[i] = ctxFail(ctxs[i], MCRT_ERR_HIP, std::string("mcrt_render_multi: ") + e.what());
        } catch (...) {
            rc[i] = ctxFail(ctxs[i], MCRT_ERR_HIP, "mcrt_render_multi: unknown exception in a worker");
        }
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve(count);
        for (uint32_t i = 1; i < count; i++) pool.emplace_back(work, i);
    } catch (const std::exception& e) {  // std::system_error: no more threads
        for (auto& t : pool) t.join();
        return ctxFail(first, MCRT_ERR_HIP, std::string("mcrt_render_multi: cannot start a worker thread: ") + e.what());
    }
    work(0);
    for (auto& t : pool) t.join();
    for (uint32_t i = 0; i < count; i++)
        if (rc[i] != MCRT_OK) {
            if (i != 0) ctxFail(first, rc[i], std::string("context ") + std::to_string(i) + ": " + mcrt_last_error(ctxs[i]));
            return rc[i];
        }
    if (splats) {  // sum in context order, then Splat::get
        std::vector<double>& sum = rgbw[0];
        for (uint32_t i = 1; i < count; i++)
            for (size_t k = 0; k < pixels * 4; k++) sum[k] += rgbw[i][k];
        for (size_t p = 0; p < pixels; p++) filmResolve(&sum[p * 4], out_rgb + p * 3);
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (uint32_t i = 0; i < count; i++) {
            stats->paths += st[i].paths;
            stats->rays += st[i].rays;
            stats->node_tests += st[i].node_tests;
            stats->prim_tests += st[i].prim_tests;
            stats->knn_searches += st[i].knn_searches;
            stats->kernel_launches += st[i].kernel_launches;
            if (st[i].kernel_id != MCRT_KERNEL_NONE) stats->kernel_id = st[i].kernel_id;  // same scene, same rule: one form
            stats->kernel_ms = std::max(stats->kernel_ms, st[i].kernel_ms);  // the contexts run side by side
            stats->total_ms = std::max(stats->total_ms, st[i].total_ms);
        }
    }
    return MCRT_OK;
}

// mcrt_photon_pass_multi: PhotonMapper::PhotonMapper's pass (photon-mapper.cpp:40-207) for the contexts of ONE host process, sharded:
// context i traces shard i of `count` of the emission paths (mcrt_emit_photons_device; a path's photons are a function of the seed and
// its index, so the shards' lists are disjoint pieces of the one-context list), the lists cross between the GPUs on device pointers
// (hipMemcpyPeer: xGMI between two devices, an ordinary copy when two contexts share one), every context concatenates them IN SHARD
// ORDER and builds both maps from the same concatenation (mcrt_upload_photons_device) - the same maps in every context. What
// bench.py's N > 1 branch does through torch's RCCL all-gather, for hosts that are one process: the reference's own main(). Against
// every context tracing all paths (round 5's drop-in): the emission, 0.37 of the pass's 0.40 s at C5's 1e8 paths, is divided by `count`.
// stats: one record PER CONTEXT ([count], may be NULL); emission_paths / rays / emission_ms are the context's own shard, the counts the maps'.
extern "C" int mcrt_photon_pass_multi(mcrt_ctx* const* ctxs, uint32_t count, double emissions, double caustic_factor, uint32_t global_seed,
                                      const double bb_min[3], const double bb_max[3], uint32_t max_photons_per_leaf, uint32_t k_nearest_photons,
                                      int direct_visualization, mcrt_photon_pass_stats* stats) {
    if (!ctxs || count == 0 || !ctxs[0]) return MCRT_ERR_INVALID;
    mcrt_ctx* first = ctxs[0];
    for (uint32_t i = 0; i < count; i++)
        if (!ctxs[i]) return ctxFail(first, MCRT_ERR_INVALID, "mcrt_photon_pass_multi: NULL context");
    if (!bb_min || !bb_max) return ctxFail(first, MCRT_ERR_INVALID, "mcrt_photon_pass_multi: bb_min or bb_max is NULL");
    if (count == 1) return mcrt_photon_pass_device(first, emissions, caustic_factor, global_seed, bb_min, bb_max, max_photons_per_leaf, k_nearest_photons,
                                                   direct_visualization, stats);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int> rc(count, MCRT_OK);
    std::vector<mcrt_photon_emission_device> em(count);
    std::vector<mcrt_photon_pass_stats> st(count);
    auto everyContext = [&](auto&& body) -> int {  // one host thread per context; the first failure, recorded on the first context too
        std::vector<std::thread> pool;
        auto guarded = [&](uint32_t i) {
            try {
                rc[i] = body(i);
            } catch (const std::exception& e) {
                rc[i] = ctxFail(ctxs[i], MCRT_ERR_HIP, std::string("mcrt_photon_pass_multi: ") + e.what());
            }
        };
        try {
            pool.reserve(count);
            for (uint32_t i = 1; i < count; i++) pool.emplace_back(guarded, i);
        } catch (const std::exception& e) {
            for (auto& t : pool) t.join();
            return ctxFail(first, MCRT_ERR_HIP, std::string("mcrt_photon_pass_multi: cannot start a worker thread: ") + e.what());
        }
        guarded(0);
        for (auto& t : pool) t.join();
        for (uint32_t i = 0; i < count; i++)
            if (rc[i] != MCRT_OK) {
                if (i != 0) ctxFail(first, rc[i], std::string("context ") + std::to_string(i) + ": " + mcrt_last_error(ctxs[i]));
                return rc[i];
            }
        return MCRT_OK;
    };
    // 1. every context its shard of the emission paths; the lists stay in that context's memory until its next emission
    if (int e = everyContext([&](uint32_t i) { return mcrt_emit_photons_device(ctxs[i], emissions, caustic_factor, global_seed, i, count, &em[i]); })) return e;
    uint64_t total_g = 0, total_c = 0;
    for (uint32_t j = 0; j < count; j++) {
        total_g += em[j].global_count;
        total_c += em[j].caustic_count;
    }
    // 2. every context pulls all shards into one list per map, in shard order (mcrt_emit_photons_device returned with the lists
    //    complete: the copies need no ordering against the emitting streams), and builds the maps from it
    std::vector<double> exchange_ms(count, 0.0);
    if (int e = everyContext([&](uint32_t i) -> int {
            const int dev = ctxDevice(ctxs[i]);
            if (hipSetDevice(dev) != hipSuccess) return ctxFail(ctxs[i], MCRT_ERR_HIP, "mcrt_photon_pass_multi: hipSetDevice failed");
            const auto t1 = std::chrono::steady_clock::now();
            float *dg = nullptr, *dc = nullptr;
            auto release = [&]() {
                if (dg) (void)hipFree(dg);
                if (dc) (void)hipFree(dc);
            };
            if ((total_g && hipMalloc(reinterpret_cast<void**>(&dg), total_g * 8 * sizeof(float)) != hipSuccess) ||
                (total_c && hipMalloc(reinterpret_cast<void**>(&dc), total_c * 8 * sizeof(float)) != hipSuccess)) {
                release();
                return ctxFail(ctxs[i], MCRT_ERR_HIP, "mcrt_photon_pass_multi: cannot allocate the gathered photon lists");
            }
            uint64_t og = 0, oc = 0;
            for (uint32_t j = 0; j < count; j++) {
                const int src = ctxDevice(ctxs[j]);
                if (src != dev) {  // direct xGMI copies where the devices can reach each other (without it hipMemcpyPeer stages through the host: correct, slower)
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, dev, src) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(src, 0);  // (hipErrorPeerAccessAlreadyEnabled is fine)
                    (void)hipGetLastError();
                }
                if (em[j].global_count && hipMemcpyPeer(dg + og * 8, dev, em[j].d_global_photons, src, em[j].global_count * 8 * sizeof(float)) != hipSuccess) {
                    release();
                    return ctxFail(ctxs[i], MCRT_ERR_HIP, "mcrt_photon_pass_multi: copying a global photon list between devices failed");
                }
                if (em[j].caustic_count && hipMemcpyPeer(dc + oc * 8, dev, em[j].d_caustic_photons, src, em[j].caustic_count * 8 * sizeof(float)) != hipSuccess) {
                    release();
                    return ctxFail(ctxs[i], MCRT_ERR_HIP, "mcrt_photon_pass_multi: copying a caustic photon list between devices failed");
                }
                og += em[j].global_count;
                oc += em[j].caustic_count;
            }
            if (hipDeviceSynchronize() != hipSuccess) {  // the lists must be complete when the build reads them (include/mcrt.h, stream contract)
                release();
                return ctxFail(ctxs[i], MCRT_ERR_HIP, "mcrt_photon_pass_multi: the list exchange failed");
            }
            exchange_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            memset(&st[i], 0, sizeof(st[i]));
            const int r = mcrt_upload_photons_device(ctxs[i], dg, total_g, dc, total_c, bb_min, bb_max, max_photons_per_leaf, k_nearest_photons,
                                                     direct_visualization, &st[i]);
            release();
            return r;
        }))
        return e;
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats)
        for (uint32_t i = 0; i < count; i++) {
            st[i].emission_paths = em[i].emission_paths;
            st[i].rays = em[i].rays;
            st[i].emission_ms = em[i].kernel_ms;
            st[i].total_ms = total_ms;
            stats[i] = st[i];
        }
    return MCRT_OK;
}
