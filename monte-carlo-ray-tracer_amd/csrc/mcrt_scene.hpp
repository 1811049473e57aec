// Device-side scene view and closest-hit traversal (the §8 rows a4-a8):
//   Scene::intersect        scene/scene.cpp:151-176
//   BVH::intersect          bvh/bvh.cpp:80-129
//   BoundingBox::intersect  common/bounding-box.cpp:9-17
//   Triangle::intersect     surface/triangle.cpp:23-63
//   Sphere::intersect       surface/sphere.cpp:13-26, solveQuadratic common/util.hpp:60-83
//
// Data layout (built once by mcrt_upload_scene from the reference's sibling-linked LinearNode array):
//   * nodes are re-ordered breadth-first so that the children of a node are CONTIGUOUS and the top
//     of the tree is a prefix of the array (that prefix is what gets staged in LDS);
//     node_bounds[n][6] FP64 (48 B), node_meta[n] = {a,b}: leaf {start_surface, count},
//     inner {first_child, 0x80000000 | child_count};
//   * primitives keep the reference's ordered_surfaces order (leaf ranges stay contiguous);
//     prim[n][10] FP64 = 80 B intersection record: triangle v0,E1,E2,tag / sphere origin,radius,..,tag
//     with tag = kind + 2*interpolate; everything only needed after the closest hit is in surf_*.
//
// Traversal: the reference pops nodes best-first from a binary heap. The closest hit does not depend
// on visiting order (only exact-t ties between different surfaces do), so each lane runs a
// near-child-first DEPTH-first walk with an 8-byte-per-entry stack {float entry_t rounded down, node}
// kept in LDS ([depth][lane] so a wave's accesses are conflict-free), spilling to a per-lane global
// slab past kLdsStackDepth. Entry distances are only used to cull (conservatively); every box and
// primitive test is the reference's FP64 arithmetic.
#pragma once

#include "mcrt_math.hpp"

namespace mcrt {

constexpr uint32_t kInnerFlag = 0x80000000u;
constexpr int kPrimStride = 10;      // doubles per intersection record
constexpr int kLdsStackDepth = 16;   // entries per lane kept in LDS
constexpr int kMaxStackDepth = 128;  // total entries per lane (LDS + global spill)

struct NodeMeta {
    uint32_t a, b;
};

struct Ray {  // ray/ray.hpp:10-26
    d3 start, direction, inv_direction;
    double medium_ior, refraction_scale;
    int refraction_level;
    uint16_t depth, diffuse_depth;
    bool dirac_delta, refraction;
};

MCRT_HD Ray makeRay(d3 start, d3 direction, double medium_ior) {  // ray.cpp:13-14
    Ray r;
    r.start = start;
    r.direction = direction;
    r.inv_direction = rcp3(direction);
    r.medium_ior = medium_ior;
    r.refraction_scale = 1.0;
    r.refraction_level = 0;
    r.depth = 0;
    r.diffuse_depth = 0;
    r.dirac_delta = false;
    r.refraction = false;
    return r;
}
MCRT_HD Ray makeRayTo(d3 start, d3 end) { return makeRay(start, normalize(end - start), 1.0); }  // ray.cpp:10-11

struct Hit {  // ray/intersection.hpp:9-23
    double t, u, v;
    uint32_t surface;
    bool interpolate;
};

struct StackEntry {
    float t;
    uint32_t node;
};

// Everything a lane needs to traverse. Pointers may address LDS (staged copies) or global memory.
struct SceneView {
    uint32_t num_nodes, num_surfaces;
    const double* node_bounds;   // global, all nodes
    const NodeMeta* node_meta;   // global, all nodes
    const double* prim;          // global, all primitives
    uint32_t lds_nodes;          // nodes [0, lds_nodes) are also in LDS
    const double* lds_node_bounds;
    const NodeMeta* lds_node_meta;
    uint32_t lds_prims;          // primitives [0, lds_prims) are also in LDS
    const double* lds_prim;
};

struct LaneStack {
    StackEntry* lds;       // &lds_stack[lane_in_block]; stride = block size
    uint32_t lds_stride;
    StackEntry* spill;     // &spill[global_lane]; stride = total lanes
    uint32_t spill_stride;
    MCRT_HD void put(int sp, StackEntry e) const {
        if (sp < kLdsStackDepth) lds[(uint32_t)sp * lds_stride] = e;
        else spill[(uint32_t)(sp - kLdsStackDepth) * (size_t)spill_stride] = e;
    }
    MCRT_HD StackEntry get(int sp) const {
        return sp < kLdsStackDepth ? lds[(uint32_t)sp * lds_stride] : spill[(uint32_t)(sp - kLdsStackDepth) * (size_t)spill_stride];
    }
};

struct TraceCounters {
    uint32_t rays, node_tests, prim_tests, overflow;
};

MCRT_HD float floatBelow(double t) {  // largest float <= t (conservative cull key)
    float f = (float)t;
    if ((double)f > t) f = nextafterf(f, -INFINITY);
    return f;
}

// BoundingBox::intersect (bounding-box.cpp:9-17)
MCRT_HD bool boxIntersect(const double* b, const Ray& ray, double& t) {
    d3 t0 = (ld3(b) - ray.start) * ray.inv_direction;
    d3 t1 = (ld3(b + 3) - ray.start) * ray.inv_direction;
    d3 lo = d3{gmin(t0.x, t1.x), gmin(t0.y, t1.y), gmin(t0.z, t1.z)};
    d3 hi = d3{gmax(t0.x, t1.x), gmax(t0.y, t1.y), gmax(t0.z, t1.z)};
    t = gmax(compMax(lo), 0.0);
    return compMin(hi) >= t;
}

// Triangle::intersect (triangle.cpp:23-63) / Sphere::intersect (sphere.cpp:13-26) on one record.
MCRT_HD bool primIntersect(const double* rec, const Ray& ray, Hit& out) {
    const double tag = rec[9];
    if (tag == 1.0) {  // sphere
        d3 so = ray.start - ld3(rec);
        double b = 2.0 * dot(ray.direction, so);
        double c = dot(so, so) - sq(rec[3]);
        // solveQuadratic(1.0, b, c) (util.hpp:60-83) with a == 1
        double d = b * b - 4.0 * 1.0 * c;
        if (d < 0.0) return false;
        double sd = sqrt(d);
        double t = -0.5 * (b + (b < 0.0 ? -sd : sd));
        double t_min = t / 1.0;
        double t_max = c / t;
        if (t_min > t_max) {
            double tmp = t_min;
            t_min = t_max;
            t_max = tmp;
        }
        if (!(t_max >= 0.0)) return false;
        out.t = t_min < 0.0 ? t_max : t_min;
        out.u = 0.0;
        out.v = 0.0;
        out.interpolate = false;
        return true;
    }
    d3 v0 = ld3(rec), E1 = ld3(rec + 3), E2 = ld3(rec + 6);
    d3 P = cross(ray.direction, E2);
    double determinant = dot(P, E1);
    if (determinant < kEpsilon && determinant > -kEpsilon) return false;
    double inv_determinant = 1.0 / determinant;
    d3 T = ray.start - v0;
    double u = dot(P, T) * inv_determinant;
    if (u > 1.0 || u < 0.0) return false;
    d3 Q = cross(T, E1);
    double v = dot(Q, ray.direction) * inv_determinant;
    if (v > 1.0 || v < 0.0 || u + v > 1.0) return false;
    double t = dot(Q, E2) * inv_determinant;
    if (t <= 0.0) return false;
    out.t = t;
    out.interpolate = tag >= 2.0;
    out.u = out.interpolate ? u : 0.0;
    out.v = out.interpolate ? v : 0.0;
    return true;
}

template <bool kCount>
MCRT_HD Hit sceneIntersect(const SceneView& sv, const Ray& ray, const LaneStack& stk, TraceCounters& cnt) {
    Hit best;
    best.t = kDblMax;
    best.u = 0.0;
    best.v = 0.0;
    best.surface = kNoSurface;
    best.interpolate = false;
    cnt.rays++;

    if (sv.num_nodes == 0) {  // brute force, scene.cpp:161-173
        for (uint32_t i = 0; i < sv.num_surfaces; i++) {
            const double* rec = i < sv.lds_prims ? sv.lds_prim + (size_t)i * kPrimStride : sv.prim + (size_t)i * kPrimStride;
            Hit h;
            if (kCount) cnt.prim_tests++;
            if (primIntersect(rec, ray, h) && h.t < best.t) {
                best = h;
                best.surface = i;
            }
        }
        return best;
    }

    double t;
    if (kCount) cnt.node_tests++;
    const double* rb = sv.lds_nodes > 0 ? sv.lds_node_bounds : sv.node_bounds;
    if (!boxIntersect(rb, ray, t)) return best;

    int sp = 0;
    uint32_t node = 0;
    for (;;) {
        NodeMeta m = node < sv.lds_nodes ? sv.lds_node_meta[node] : sv.node_meta[node];
        if (!(m.b & kInnerFlag)) {
            const uint32_t end = m.a + m.b;
            for (uint32_t i = m.a; i < end; i++) {
                const double* rec = i < sv.lds_prims ? sv.lds_prim + (size_t)i * kPrimStride : sv.prim + (size_t)i * kPrimStride;
                Hit h;
                if (kCount) cnt.prim_tests++;
                if (primIntersect(rec, ray, h) && h.t < best.t) {
                    best = h;
                    best.surface = i;
                }
            }
        } else {
            const uint32_t first = m.a, count = m.b & ~kInnerFlag;
            const int sp0 = sp;
            float nearest = INFINITY;
            int nearest_sp = -1;
            for (uint32_t c = first; c < first + count; c++) {
                const double* cb = c < sv.lds_nodes ? sv.lds_node_bounds + (size_t)c * 6 : sv.node_bounds + (size_t)c * 6;
                if (kCount) cnt.node_tests++;
                if (boxIntersect(cb, ray, t) && t < best.t) {
                    if (sp < kMaxStackDepth) {
                        StackEntry e;
                        e.t = floatBelow(t);
                        e.node = c;
                        if (e.t < nearest) {
                            nearest = e.t;
                            nearest_sp = sp;
                        }
                        stk.put(sp++, e);
                    } else {
                        cnt.overflow = 1;
                    }
                }
            }
            // visit the nearest of the children just pushed first: move it to the top of the stack
            if (sp - sp0 > 1 && nearest_sp != sp - 1) {
                StackEntry a = stk.get(nearest_sp), b = stk.get(sp - 1);
                stk.put(nearest_sp, b);
                stk.put(sp - 1, a);
            }
        }
        // pop, culling entries that can no longer beat the current closest hit
        bool found = false;
        while (sp > 0) {
            StackEntry e = stk.get(--sp);
            if ((double)e.t < best.t) {
                node = e.node;
                found = true;
                break;
            }
        }
        if (!found) break;
    }
    return best;
}

}  // namespace mcrt
