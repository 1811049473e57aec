// integration/ref_flatten.hpp — the FLATTENER a maintainer of the reference adds next to its sources:
// walks the reference's already-built Scene / BVH / LinearOctree<Photon> / Camera objects and fills the descriptors of
// include/mcrt.h. Reference-side integration code, compiled only into hosts that link the reference's own translation units
// (oracle/_ref/mcrt_ref: the checker; oracle/_ref/mcrt_ref_gpu: the reference's main() with Camera::sampleImage() replaced by
// libmcrt_hip.so, integration/camera_sample_image_gpu.cpp). Never part of the product library.
//
// Private members (BVH::linear_tree, Triangle::v0 …, PhotonMapper::global_map, Film::filter_function) are reached by compiling
// the including TU with -fno-access-control (layout is unaffected); a maintainer would add friend declarations instead.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <vector>

#include <glm/glm.hpp>

#include "bvh/bvh.hpp"
#include "camera/camera.hpp"
#include "camera/filter.hpp"
#include "integrator/integrator.hpp"
#include "integrator/photon-mapper/photon-mapper.hpp"
#include "material/fresnel.hpp"
#include "material/material.hpp"
#include "octree/linear-octree.hpp"
#include "sampling/sampler.hpp"
#include "scene/scene.hpp"
#include "surface/surface.hpp"

#include "../include/mcrt.h"

namespace {

struct Flat {
    std::vector<double> node_bounds;
    std::vector<uint32_t> node_start, node_count, node_next;
    std::vector<uint8_t> kind, interp;
    std::vector<uint32_t> surf_material;
    std::vector<double> area, v, e, vn, quadrics;
    std::vector<mcrt_material> materials;
    std::vector<uint32_t> light_surface;
    std::vector<double> light_cdf;
    std::unordered_map<const Surface::Base*, uint32_t> index;
    mcrt_scene_desc desc;
};

void put3(std::vector<double>& dst, const glm::dvec3& x) { dst.push_back(x.x); dst.push_back(x.y); dst.push_back(x.z); }

void flattenScene(const Scene& scene, Flat& F) {
    const std::vector<std::shared_ptr<Surface::Base>>* ordered = &scene.surfaces;
    if (scene.bvh) {
        const BVH& bvh = *scene.bvh;
        ordered = &bvh.ordered_surfaces;
        for (const auto& n : bvh.linear_tree) {
            put3(F.node_bounds, n.BB.min);
            put3(F.node_bounds, n.BB.max);
            F.node_start.push_back(n.start_surface);
            F.node_count.push_back(n.num_surfaces);
            F.node_next.push_back(n.next_sibling);
        }
    }
    std::unordered_map<const Material*, uint32_t> mat_index;
    bool any_vn = false;
    for (const auto& sp : *ordered) {
        const Surface::Base* s = sp.get();
        F.index[s] = (uint32_t)F.kind.size();
        const Material* m = s->material.get();
        auto it = mat_index.find(m);
        if (it == mat_index.end()) {
            mcrt_material mm;
            std::memset(&mm, 0, sizeof(mm));
            for (int c = 0; c < 3; c++) {
                mm.reflectance[c] = m->reflectance[c];
                mm.specular_reflectance[c] = m->specular_reflectance[c];
                mm.transmittance[c] = m->transmittance[c];
                mm.emittance[c] = m->emittance[c];
                if (m->complex_ior) {
                    mm.ior_real[c] = m->complex_ior->real[c];
                    mm.ior_imag[c] = m->complex_ior->imaginary[c];
                }
            }
            mm.roughness = m->roughness;
            mm.specular_roughness = m->specular_roughness;
            mm.ior = m->ior;
            mm.transparency = m->transparency;
            mm.A = m->A;
            mm.B = m->B;
            mm.a[0] = m->a.x;
            mm.a[1] = m->a.y;
            mm.flags = (m->rough ? MCRT_MAT_ROUGH : 0) | (m->rough_specular ? MCRT_MAT_ROUGH_SPECULAR : 0) |
                       (m->opaque ? MCRT_MAT_OPAQUE : 0) | (m->emissive ? MCRT_MAT_EMISSIVE : 0) |
                       (m->dirac_delta ? MCRT_MAT_DIRAC_DELTA : 0) |
                       (m->perfect_mirror ? MCRT_MAT_PERFECT_MIRROR : 0) |
                       (m->complex_ior ? MCRT_MAT_COMPLEX_IOR : 0);
            it = mat_index.emplace(m, (uint32_t)F.materials.size()).first;
            F.materials.push_back(mm);
        }
        F.surf_material.push_back(it->second);
        F.area.push_back(s->area_);
        if (auto t = dynamic_cast<const Surface::Triangle*>(s)) {
            F.kind.push_back(MCRT_SURF_TRIANGLE);
            put3(F.v, t->v0); put3(F.v, t->v1); put3(F.v, t->v2);
            put3(F.e, t->E1); put3(F.e, t->E2); put3(F.e, t->normal_);
            if (t->N) {
                any_vn = true;
                F.interp.push_back(1);
                put3(F.vn, (*t->N)[0]); put3(F.vn, (*t->N)[1]); put3(F.vn, (*t->N)[2]);
            } else {
                F.interp.push_back(0);
                for (int i = 0; i < 9; i++) F.vn.push_back(0.0);
            }
        } else if (auto sph = dynamic_cast<const Surface::Sphere*>(s)) {
            F.kind.push_back(MCRT_SURF_SPHERE);
            F.interp.push_back(0);
            put3(F.v, sph->origin);
            F.v.push_back(sph->radius);
            for (int i = 0; i < 5; i++) F.v.push_back(0.0);
            for (int i = 0; i < 9; i++) { F.e.push_back(0.0); F.vn.push_back(0.0); }
        } else if (auto q = dynamic_cast<const Surface::Quadric*>(s)) {
            F.kind.push_back(MCRT_SURF_QUADRIC);
            F.interp.push_back(0);
            F.v.push_back((double)(F.quadrics.size() / 22));
            for (int i = 0; i < 8; i++) F.v.push_back(0.0);
            for (int i = 0; i < 9; i++) { F.e.push_back(0.0); F.vn.push_back(0.0); }
            for (int c = 0; c < 4; c++)
                for (int rr = 0; rr < 4; rr++) F.quadrics.push_back(q->Q[c][rr]);
            put3(F.quadrics, q->BB_.min);
            put3(F.quadrics, q->BB_.max);
        } else {
            std::fprintf(stderr, "flatten: unsupported surface type\n");
            std::exit(3);
        }
    }
    for (size_t i = 0; i < scene.emissives.size(); i++) {
        F.light_surface.push_back(F.index.at(scene.emissives[i].get()));
        F.light_cdf.push_back(scene.cumulative_emissives_importance[i]);
    }
    mcrt_scene_desc& d = F.desc;
    std::memset(&d, 0, sizeof(d));
    d.abi_version = MCRT_ABI_VERSION;
    d.num_nodes = (uint32_t)F.node_start.size();
    d.node_bounds = F.node_bounds.data();
    d.node_start_surface = F.node_start.data();
    d.node_num_surfaces = F.node_count.data();
    d.node_next_sibling = F.node_next.data();
    d.num_surfaces = (uint32_t)F.kind.size();
    d.surf_kind = F.kind.data();
    d.surf_interpolate = F.interp.data();
    d.surf_material = F.surf_material.data();
    d.surf_area = F.area.data();
    d.surf_v = F.v.data();
    d.surf_e = F.e.data();
    d.surf_vn = any_vn ? F.vn.data() : nullptr;
    d.num_materials = (uint32_t)F.materials.size();
    d.materials = F.materials.data();
    d.num_lights = (uint32_t)F.light_surface.size();
    d.light_surface = F.light_surface.data();
    d.light_cdf = F.light_cdf.data();
    d.scene_ior = scene.ior;
    BoundingBox bb = scene.BB();
    for (int c = 0; c < 3; c++) { d.bb_min[c] = bb.min[c]; d.bb_max[c] = bb.max[c]; }
    d.num_quadrics = (uint32_t)(F.quadrics.size() / 22);
    d.quadrics = F.quadrics.data();
}

struct FlatMap {
    std::vector<double> bounds;
    std::vector<uint64_t> start, contained;
    std::vector<uint32_t> next;
    std::vector<uint8_t> leaf;
    std::vector<float> photons;
    mcrt_photon_map_desc desc;
    FlatMap() { std::memset(&desc, 0, sizeof(desc)); }
};

void flattenMap(const LinearOctree<Photon>& map, FlatMap& M) {
    for (const auto& o : map.linear_tree) {
        put3(M.bounds, o.BB.min);
        put3(M.bounds, o.BB.max);
        M.start.push_back(o.start_data);
        M.contained.push_back(o.contained_data);
        M.next.push_back(o.next_sibling);
        M.leaf.push_back(o.leaf);
    }
    M.photons.resize(map.ordered_data.size() * 8);
    for (size_t i = 0; i < map.ordered_data.size(); i++) {
        const Photon& p = map.ordered_data[i];
        float* o = &M.photons[i * 8];
        o[0] = p.flux_.x; o[1] = p.flux_.y; o[2] = p.flux_.z;
        o[3] = p.position_.x; o[4] = p.position_.y; o[5] = p.position_.z;
        o[6] = p.phi; o[7] = p.theta;
    }
    std::memset(&M.desc, 0, sizeof(M.desc));
    M.desc.num_octants = (uint32_t)M.start.size();
    M.desc.octant_bounds = M.bounds.data();
    M.desc.octant_start_data = M.start.data();
    M.desc.octant_contained_data = M.contained.data();
    M.desc.octant_next_sibling = M.next.data();
    M.desc.octant_leaf = M.leaf.data();
    M.desc.num_photons = map.ordered_data.size();
    M.desc.photons = M.photons.data();
}

// Film keeps its filter as a std::function<double(double)> holding a plain function pointer (film.cpp:26-44): the kind is
// recovered by comparing that pointer with the functions of camera/filter.hpp.
inline uint32_t filmFilterKind(const Film& film) {
    using Fn = double (*)(double);
    const Fn* target = film.filter_function.target<Fn>();
    if (!target) return MCRT_FILM_BOX;
    const Fn kinds[7] = {&Filter::box, &Filter::MitchellNetravali<>, &Filter::CatmullRom, &Filter::BSpline, &Filter::Hermite, &Filter::Gaussian,
                         &Filter::Lanczos};
    for (uint32_t i = 0; i < 7; i++)
        if (*target == kinds[i]) return i;
    return MCRT_FILM_BOX;
}

mcrt_camera_desc flattenCamera(const Camera& c) {
    mcrt_camera_desc d;
    std::memset(&d, 0, sizeof(d));
    for (int i = 0; i < 3; i++) {
        d.eye[i] = c.eye[i]; d.forward[i] = c.forward[i]; d.left[i] = c.left[i]; d.up[i] = c.up[i];
    }
    d.focal_length = c.focal_length;
    d.sensor_width = c.sensor_width;
    d.aperture_radius = c.aperture_radius;
    d.focus_distance = c.focus_distance;
    d.thin_lens = c.thin_lens ? 1 : 0;
    d.width = (uint32_t)c.image.width;
    d.height = (uint32_t)c.image.height;
    d.sqrtspp = (uint32_t)c.sqrtspp;
    d.shard_index = 0; d.shard_count = 1; d.shard_rows = 1;
    d.film_filter = filmFilterKind(c.film);
    // (the box filter's radius too: with another radius than its default 0.5 the film splats, film.cpp:44-46)
    d.film_radius = c.film.radius;
    d.film_cache_size = (uint32_t)c.film.filter_cache.size();
    return d;
}

}  // namespace
