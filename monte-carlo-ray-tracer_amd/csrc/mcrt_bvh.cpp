// Octree BVH builder, host path and the scene re-ordering helper (include/mcrt.h: mcrt_bvh_build_octree,
// mcrt_scene_with_bvh). Algorithm and its equivalence with the reference's builder: mcrt_bvh_shared.hpp. The GPU path
// (boxes, codes, radix sort on the device) is bvhOctreeGpu in mcrt_octree_gpu.hip.
#include <numeric>

#include "mcrt_bvh_shared.hpp"
#include "mcrt_internal.hpp"

using namespace mcrt;

struct mcrt_scene {
    std::vector<uint8_t> kind, interp;
    std::vector<uint32_t> material, light_surface;
    std::vector<double> area, v, e, vn;
    mcrt_scene_desc desc;
};

namespace {

bool sceneUsable(const mcrt_scene_desc* s) {
    if (!s || s->abi_version != MCRT_ABI_VERSION || s->num_surfaces == 0 || !s->surf_kind || !s->surf_v) return false;
    for (uint32_t i = 0; i < s->num_surfaces; i++)
        if (s->surf_kind[i] > MCRT_SURF_QUADRIC || (s->surf_kind[i] == MCRT_SURF_QUADRIC && (!s->quadrics || s->surf_v[9 * (size_t)i] >= s->num_quadrics)))
            return false;
    return true;
}

int buildHost(const mcrt_scene_desc* s, mcrt_bvh* B) {
    const uint64_t n = s->num_surfaces;
    double mn[3], mx[3];
    bvhRootCube(s, mn, mx);
    std::vector<double> bb(n * 6);
    std::vector<unsigned long long> code(n);
    for (uint64_t i = 0; i < n; i++) {
        double* b = &bb[i * 6];
        surfaceBounds(s->surf_kind[i], s->surf_v + 9 * i, s->quadrics, b);
        // SurfaceCentroid = BB().centroid() = (max + min) / 2 (bvh.cpp:475-476, bounding-box.cpp:30-33)
        code[i] = cellCode((b[3] + b[0]) / 2.0, (b[4] + b[1]) / 2.0, (b[5] + b[2]) / 2.0, mn, mx);
    }
    std::vector<uint32_t> index(n);
    std::iota(index.begin(), index.end(), 0u);
    std::stable_sort(index.begin(), index.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    std::vector<unsigned long long> keys(n);
    std::vector<double> sorted_bb(n * 6);
    for (uint64_t i = 0; i < n; i++) {
        keys[i] = code[index[i]];
        memcpy(&sorted_bb[i * 6], &bb[(size_t)index[i] * 6], 48);
    }
    return assembleOctreeBvh(keys.data(), index.data(), sorted_bb.data(), n, B) ? MCRT_OK : MCRT_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

int mcrt_bvh_build_octree(mcrt_ctx* ctx, const mcrt_scene_desc* scene, mcrt_bvh** out) {
    if (!out || !sceneUsable(scene)) return ctx ? ctxFail(ctx, MCRT_ERR_INVALID, "mcrt_bvh_build_octree: bad scene descriptor") : MCRT_ERR_INVALID;
    mcrt_bvh* B = new mcrt_bvh();
    const int rc = ctx ? bvhOctreeGpu(ctx, scene, B) : buildHost(scene, B);
    if (rc != MCRT_OK) {
        delete B;
        if (ctx && rc == MCRT_ERR_UNSUPPORTED) return ctxFail(ctx, rc, "more than 8 surface centroids inside one 2^-21 cell of the scene cube");
        return rc;
    }
    *out = B;
    return MCRT_OK;
}

const mcrt_bvh_desc* mcrt_bvh_get(const mcrt_bvh* bvh) { return bvh ? &bvh->desc : nullptr; }
void mcrt_bvh_free(mcrt_bvh* bvh) { delete bvh; }

int mcrt_scene_with_bvh(const mcrt_scene_desc* s, const mcrt_bvh_desc* bvh, mcrt_scene** out) {
    if (!out || !s || !bvh || bvh->num_surfaces != s->num_surfaces || !bvh->order || !s->surf_interpolate || !s->surf_material || !s->surf_area ||
        !s->surf_e)
        return MCRT_ERR_INVALID;
    const size_t n = s->num_surfaces;
    mcrt_scene* S = new mcrt_scene();
    S->kind.resize(n);
    S->interp.resize(n);
    S->material.resize(n);
    S->area.resize(n);
    S->v.resize(n * 9);
    S->e.resize(n * 9);
    if (s->surf_vn) S->vn.resize(n * 9);
    std::vector<uint32_t> where(n);  // input index -> new position
    for (size_t i = 0; i < n; i++) {
        const size_t src = bvh->order[i];
        if (src >= n) {
            delete S;
            return MCRT_ERR_INVALID;
        }
        where[src] = (uint32_t)i;
        S->kind[i] = s->surf_kind[src];
        S->interp[i] = s->surf_interpolate[src];
        S->material[i] = s->surf_material[src];
        S->area[i] = s->surf_area[src];
        memcpy(&S->v[i * 9], s->surf_v + src * 9, 72);
        memcpy(&S->e[i * 9], s->surf_e + src * 9, 72);
        if (s->surf_vn) memcpy(&S->vn[i * 9], s->surf_vn + src * 9, 72);
    }
    S->light_surface.resize(s->num_lights);
    for (uint32_t i = 0; i < s->num_lights; i++) S->light_surface[i] = where[s->light_surface[i]];
    S->desc = *s;  // materials, light_cdf, quadrics, scalars stay the caller's arrays
    S->desc.num_nodes = bvh->num_nodes;
    S->desc.node_bounds = bvh->node_bounds;
    S->desc.node_start_surface = bvh->node_start_surface;
    S->desc.node_num_surfaces = bvh->node_num_surfaces;
    S->desc.node_next_sibling = bvh->node_next_sibling;
    S->desc.surf_kind = S->kind.data();
    S->desc.surf_interpolate = S->interp.data();
    S->desc.surf_material = S->material.data();
    S->desc.surf_area = S->area.data();
    S->desc.surf_v = S->v.data();
    S->desc.surf_e = S->e.data();
    S->desc.surf_vn = s->surf_vn ? S->vn.data() : nullptr;
    S->desc.light_surface = S->light_surface.data();
    *out = S;
    return MCRT_OK;
}

const mcrt_scene_desc* mcrt_scene_get(const mcrt_scene* scene) { return scene ? &scene->desc : nullptr; }
void mcrt_scene_free(mcrt_scene* scene) { delete scene; }

}  // extern "C"
