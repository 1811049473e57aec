// mcrt_render_multi: one host process driving several GPUs — the shape of the reference's host (a single executable,
// Camera::sampleImage fanning out to worker threads, camera/camera.cpp:120-136). One host thread per context renders that
// context's shard of the frame (rows dealt in groups of 8, round-robin) and copies its rows into the caller's frame; no
// collective, no inter-GPU traffic: the shards are independent and the frame is assembled in host memory, which is where
// Image::save wants it. Frames with a reconstruction filter accumulate per-context {rgb_sum, weight_sum} buffers that are
// summed and resolved on the host (Splat::get, film.cpp:107-113).
#include <hip/hip_runtime.h>

#include <algorithm>
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
