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
// depth-first "while-while" walk: an inner loop that only descends through inner nodes (all lanes of
// the wave execute box tests together) and a leaf step (all lanes execute primitive tests together),
// nearest hit child kept in registers, the others on an 8-byte-per-entry stack
// {float entry_t rounded down, node} in LDS ([depth][lane], conflict-free), spilling to a per-lane
// global slab past kLdsStackDepth. Entry distances only cull (conservatively); every box and
// primitive test is the reference's FP64 arithmetic.
//
// Shadow rays (Integrator::sampleDirect, integrator.cpp:45-73) only need to know whether the closest
// hit is the sampled light and, if so, its t. With d = |light point - start| the query is therefore
// bounded to t < d(1+1e-9) (anything farther cannot be closer than the light) and ends at the first
// hit of ANOTHER surface with t < d(1-1e-9) (the light cannot be the closest hit any more); the
// estimate is unchanged.
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

// 64-byte node record of the lane-state-machine traversal (mcrt_lanesm.hpp): a child's meta travels
// with its box. leaf: a = start_surface, m = count (1..255); inner: a = first_child, m = 0x100 | child_count
struct Node64 {
    double b[6];
    uint32_t a, m;
    uint32_t pad0, pad1;
};
constexpr uint32_t kSmInner = 0x100u;

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

// What a lane traverses. kAll: the whole BVH and all primitives are staged in LDS (small scenes);
// otherwise the first `lds_nodes` nodes are in LDS and the rest, and all primitives, in global memory.
template <bool kAll>
struct SceneViewT {
    uint32_t num_nodes, num_surfaces;
    cptr<double, kAll> node_bounds;
    cptr<NodeMeta, kAll> node_meta;
    cptr<double, kAll> prim;
    uint32_t lds_nodes;  // only meaningful when !kAll
    MCRT_LDS_AS const double* lds_node_bounds;
    MCRT_LDS_AS const NodeMeta* lds_node_meta;
    // flat mode (tiny scenes, kAll only): num_nodes == 0 and a second copy of the intersection records
    // sorted by kind (triangles [0, flat_tris), spheres [flat_tris, num_surfaces)); flat_index maps a
    // sorted slot back to the surface index.
    uint32_t flat_tris;
    cptr<double, kAll> flat_prim;
    cptr<uint32_t, kAll> flat_index;
};

struct LaneStack {
    MCRT_LDS_AS StackEntry* lds;  // &lds_stack[lane_in_block]; stride = block size
    uint32_t lds_stride;
    StackEntry* spill;            // &spill[global_lane]; stride = total lanes
    uint32_t spill_stride;
    MCRT_HD void put(int sp, StackEntry e) const {
        if (sp < kLdsStackDepth) lds[(uint32_t)sp * lds_stride] = e;
        else spill[(uint32_t)(sp - kLdsStackDepth) * (size_t)spill_stride] = e;
    }
    MCRT_HD StackEntry get(int sp) const {
        if (sp < kLdsStackDepth) return lds[(uint32_t)sp * lds_stride];
        return spill[(uint32_t)(sp - kLdsStackDepth) * (size_t)spill_stride];
    }
};

struct TraceCounters {
    uint32_t rays, node_tests, prim_tests, overflow;
};

MCRT_HD float floatBelow(double t) {  // largest float <= t (conservative cull key)
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2float_rd(t);
#else
    float f = (float)t;
    if ((double)f > t) f = nextafterf(f, -INFINITY);
    return f;
#endif
}

struct Box {
    double v[6];
};

template <bool kAll>
MCRT_HD Box loadBox(const SceneViewT<kAll>& sv, uint32_t i) {
    Box b;
    if (kAll) {
        cptr<double, kAll> p = sv.node_bounds + (size_t)i * 6;
        for (int k = 0; k < 6; k++) b.v[k] = p[k];
    } else if (i < sv.lds_nodes) {
        MCRT_LDS_AS const double* p = sv.lds_node_bounds + i * 6u;
        for (int k = 0; k < 6; k++) b.v[k] = p[k];
    } else {
        cptr<double, kAll> p = sv.node_bounds + (size_t)i * 6;
        for (int k = 0; k < 6; k++) b.v[k] = p[k];
    }
    return b;
}
template <bool kAll>
MCRT_HD NodeMeta loadMeta(const SceneViewT<kAll>& sv, uint32_t i) {
    if (!kAll && i < sv.lds_nodes) return sv.lds_node_meta[i];
    return sv.node_meta[i];
}

// BoundingBox::intersect (bounding-box.cpp:9-17). kFast uses v_min/v_max_f64; the caller selects it
// only when no slab product can be NaN (all components of ray.inv_direction finite), where it gives
// the same decision and the same t.
template <bool kFast>
MCRT_HD bool boxIntersect(const Box& b, const Ray& ray, double& t) {
    d3 t0 = (d3{b.v[0], b.v[1], b.v[2]} - ray.start) * ray.inv_direction;
    d3 t1 = (d3{b.v[3], b.v[4], b.v[5]} - ray.start) * ray.inv_direction;
    if (kFast) {
        double lo = fastMax(fastMax(fastMin(t0.x, t1.x), fastMin(t0.y, t1.y)), fastMin(t0.z, t1.z));
        double hi = fastMin(fastMin(fastMax(t0.x, t1.x), fastMax(t0.y, t1.y)), fastMax(t0.z, t1.z));
        t = fastMax(lo, 0.0);
        return hi >= t;
    }
    d3 lo = d3{gmin(t0.x, t1.x), gmin(t0.y, t1.y), gmin(t0.z, t1.z)};
    d3 hi = d3{gmax(t0.x, t1.x), gmax(t0.y, t1.y), gmax(t0.z, t1.z)};
    t = gmax(compMax(lo), 0.0);
    return compMin(hi) >= t;
}

// ---- Surface::Quadric (surface/quadric.cpp). Record = 22 doubles: Q as glm stores it (Q[c][r] at 4c + r), BB_.min,
// BB_.max. Primitive records and the device copy of surf_v carry the ADDRESS of a quadric's record (64 bits stored
// in the double's slot; patched in by the host when the records' home is known), so that no view needs another
// pointer for a primitive kind no BASELINE scene uses.
// Quadric code is compiled only into the device kernels of scenes that are NOT staged whole in LDS (kAll == false):
// no BASELINE scene has quadrics and the LDS-resident kernels (configs C1/C2) sit at their register limit — the extra
// branch cost the headline kernel 3 % through spills. mcrt_upload_scene never stages a scene with quadrics whole. The
// host build (tests/emu) keeps the code everywhere.
template <bool kAll>
struct QuadricsIn {
#if defined(__HIP_DEVICE_COMPILE__)
    static constexpr bool value = !kAll;
#else
    static constexpr bool value = true;
#endif
};

MCRT_HD const double* quadricPtr(double slot) { return reinterpret_cast<const double*>((uintptr_t)dBits(slot)); }
MCRT_HD double quadricSlot(const double* record) { return bitsD((unsigned long long)(uintptr_t)record); }

// solveQuadratic, common/util.hpp:60-83
MCRT_HD bool solveQuadratic(double a, double b, double c, double& t_min, double& t_max) {
    if (a != 0.0) {
        double d = b * b - 4.0 * a * c;
        if (d < 0.0) return false;
        double sd = sqrt(d);
        double t = -0.5 * (b + (b < 0.0 ? -sd : sd));
        t_min = t / a;
        t_max = c / t;
        if (t_min > t_max) {
            double tmp = t_min;
            t_min = t_max;
            t_max = tmp;
        }
        return true;
    }
    if (b != 0.0) {
        t_min = t_max = -c / b;
        return true;
    }
    return false;
}

// Quadric::intersect, quadric.cpp:69-100: clip to the slicing box first, solve transpose(r) Q r = 0 from the box
// entry point. glm::dmat4 * dvec4 adds its four column products pairwise, (c0 + c1) + (c2 + c3), and so does the
// dvec4 dot (lib/glm/glm/detail/type_mat4x4.inl:561-573, func_geometric.inl:58-65).
MCRT_HD bool quadricIntersect(const double* q, const Ray& ray, Hit& out) {
    Box bb;
    for (int k = 0; k < 6; k++) bb.v[k] = q[16 + k];
    double t_bb = 0.0;
    if (!boxIntersect<false>(bb, ray, t_bb)) return false;
    const d3 o = ray.start + ray.direction * t_bb;  // Ray::operator()(t_bb), w = 1
    const d3 d = ray.direction;                      // w = 0
    double Qo[4], Qd[4];
    for (int r = 0; r < 4; r++) {
        Qo[r] = (q[0 + r] * o.x + q[4 + r] * o.y) + (q[8 + r] * o.z + q[12 + r] * 1.0);
        Qd[r] = (q[0 + r] * d.x + q[4 + r] * d.y) + (q[8 + r] * d.z + q[12 + r] * 0.0);
    }
    const double a = (d.x * Qd[0] + d.y * Qd[1]) + (d.z * Qd[2] + 0.0 * Qd[3]);
    const double b = ((d.x * Qo[0] + d.y * Qo[1]) + (d.z * Qo[2] + 0.0 * Qo[3])) * 2.0;
    const double c = (o.x * Qo[0] + o.y * Qo[1]) + (o.z * Qo[2] + 1.0 * Qo[3]);
    double t_min, t_max;
    if (solveQuadratic(a, b, c, t_min, t_max) && t_max >= 0.0) {
        const double t = t_bb + (t_min < 0.0 ? t_max : t_min);
        const d3 p = ray.start + ray.direction * t;  // BB_.contains(ray(t)), bounding-box.cpp:19-23
        if (!(p.x >= bb.v[0] && p.y >= bb.v[1] && p.z >= bb.v[2] && p.x <= bb.v[3] && p.y <= bb.v[4] && p.z <= bb.v[5])) return false;
        out.t = t;
        out.u = 0.0;
        out.v = 0.0;
        out.interpolate = false;
        return true;
    }
    return false;
}

// Quadric::normal, quadric.cpp:127-130: normalize(G * (pos, 1)), G = 2 * the upper three rows of Q (quadric.cpp:38-45);
// dmat4x3 * dvec4 adds left to right (type_mat4x3.inl:469-478).
MCRT_HD d3 quadricNormal(const double* q, d3 pos) {
    d3 g;
    g.x = (2.0 * q[0]) * pos.x + (2.0 * q[4]) * pos.y + (2.0 * q[8]) * pos.z + (2.0 * q[12]) * 1.0;
    g.y = (2.0 * q[1]) * pos.x + (2.0 * q[5]) * pos.y + (2.0 * q[9]) * pos.z + (2.0 * q[13]) * 1.0;
    g.z = (2.0 * q[2]) * pos.x + (2.0 * q[6]) * pos.y + (2.0 * q[10]) * pos.z + (2.0 * q[14]) * 1.0;
    return normalize(g);
}

// Triangle::intersect (triangle.cpp:23-63) / Sphere::intersect (sphere.cpp:13-26) / Quadric::intersect on one record
// (tag = rec[9]: 0 triangle, 2 triangle with vertex normals, 1 sphere, 3 quadric).
template <bool kQuadrics, class P>
MCRT_HD bool primIntersect(P rec, const Ray& ray, Hit& out) {
    const double tag = rec[9];
    if constexpr (kQuadrics) {
        if (tag == 3.0) return quadricIntersect(quadricPtr(rec[0]), ray, out);
    }
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
    d3 P_ = cross(ray.direction, E2);
    double determinant = dot(P_, E1);
    if (determinant < kEpsilon && determinant > -kEpsilon) return false;
    double inv_determinant = 1.0 / determinant;
    d3 T = ray.start - v0;
    double u = dot(P_, T) * inv_determinant;
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

// Closest-hit update. The reference keeps the FIRST tested primitive among exact-t ties
// (bvh.cpp:100, scene.cpp:166) and its test order is its heap order; here ties go to the LOWEST surface
// index, which makes the result independent of the visiting order (flat loop, depth-first walk and any
// sharding agree bit for bit). Both rules pick the same primitive unless two different surfaces are
// hit at exactly the same t.
MCRT_HD bool closer(double t, uint32_t surface, const Hit& best) {
    return t < best.t || (t == best.t && surface < best.surface);
}

// Branch-free forms of the two primitive tests for the wave-uniform flat loop: the same arithmetic in
// the same order, all comparisons folded into one accept flag (a primitive is accepted by exactly the
// rays the early-return form accepts), so that independent tests can be interleaved by the compiler.
template <class P>
MCRT_HD bool triangleTestFlat(P rec, d3 start, d3 direction, double& t, double& u, double& v) {
    d3 v0 = ld3(rec), E1 = ld3(rec + 3), E2 = ld3(rec + 6);
    d3 P_ = cross(direction, E2);
    double determinant = dot(P_, E1);
    double inv_determinant = 1.0 / determinant;
    d3 T = start - v0;
    u = dot(P_, T) * inv_determinant;
    d3 Q = cross(T, E1);
    v = dot(Q, direction) * inv_determinant;
    t = dot(Q, E2) * inv_determinant;
    const bool parallel = determinant < kEpsilon && determinant > -kEpsilon;
    const bool u_out = u > 1.0 || u < 0.0;
    const bool v_out = v > 1.0 || v < 0.0 || u + v > 1.0;
    return !parallel & !u_out & !v_out & !(t <= 0.0);
}
template <class P>
MCRT_HD bool sphereTestFlat(P rec, d3 start, d3 direction, double& t_hit) {
    d3 so = start - ld3(rec);
    double b = 2.0 * dot(direction, so);
    double c = dot(so, so) - sq(rec[3]);
    double d = b * b - 4.0 * 1.0 * c;
    double sd = sqrt(d);
    double t = -0.5 * (b + (b < 0.0 ? -sd : sd));
    double t_min = t / 1.0;
    double t_max = c / t;
    const bool swap = t_min > t_max;
    double lo = swap ? t_max : t_min, hi = swap ? t_min : t_max;
    t_hit = lo < 0.0 ? hi : lo;
    return !(d < 0.0) & (hi >= 0.0);
}

MCRT_HD void hitInit(Hit& h, double t_limit) {
    h.t = t_limit;
    h.u = 0.0;
    h.v = 0.0;
    h.surface = kNoSurface;
    h.interpolate = false;
}

struct ShadowQuery {
    uint32_t light;  // surface the shadow ray was aimed at
    double t_near;   // d (1 - 1e-9): a closer hit of another surface decides the query
    double t_far;    // d (1 + 1e-9): nothing farther can matter
};

// kFlat: the caller knows the scene is in flat mode (a kernel instance for such scenes only): the BVH walk is not compiled in.
template <bool kAll, bool kCount, bool kShadow, bool kFlat = false>
MCRT_HD Hit sceneIntersect(const SceneViewT<kAll>& sv, const Ray& ray, const LaneStack& stk, TraceCounters& cnt,
                           const ShadowQuery* sq = nullptr) {
    Hit best;
    best.t = kShadow ? sq->t_far : kDblMax;
    best.u = 0.0;
    best.v = 0.0;
    best.surface = kNoSurface;
    best.interpolate = false;
    cnt.rays++;

    if (kFlat || sv.num_nodes == 0) {  // every primitive, as scene.cpp:161-173 does without a BVH
        if (kFlat || (kAll && sv.flat_prim)) {
            // wave-uniform loops over the kind-sorted copy: no per-lane control flow at all
            const uint32_t nt = sv.flat_tris, ns = sv.num_surfaces;
#pragma unroll 2
            for (uint32_t i = 0; i < nt; i++) {
                double t, u, v;
                if (kCount) cnt.prim_tests++;
                cptr<double, kAll> rec = sv.flat_prim + (size_t)i * kPrimStride;
                const bool ok = triangleTestFlat(rec, ray.start, ray.direction, t, u, v);
                const uint32_t idx = sv.flat_index[i];
                if (ok && closer(t, idx, best)) {
                    const bool interp = rec[9] >= 2.0;
                    best.t = t;
                    best.u = interp ? u : 0.0;
                    best.v = interp ? v : 0.0;
                    best.interpolate = interp;
                    best.surface = idx;
                }
            }
#pragma unroll 2
            for (uint32_t i = nt; i < ns; i++) {
                double t;
                if (kCount) cnt.prim_tests++;
                const bool ok = sphereTestFlat(sv.flat_prim + (size_t)i * kPrimStride, ray.start, ray.direction, t);
                const uint32_t idx = sv.flat_index[i];
                if (ok && closer(t, idx, best)) {
                    best.t = t;
                    best.u = 0.0;
                    best.v = 0.0;
                    best.interpolate = false;
                    best.surface = idx;
                }
            }
            return best;
        }
        for (uint32_t i = 0; i < sv.num_surfaces; i++) {
            Hit h;
            if (kCount) cnt.prim_tests++;
            if (primIntersect<QuadricsIn<kAll>::value>(sv.prim + (size_t)i * kPrimStride, ray, h) && closer(h.t, i, best)) {
                best = h;
                best.surface = i;
                if (kShadow && i != sq->light && h.t < sq->t_near) return best;
            }
        }
        return best;
    }

    // v_min/v_max box test is exact unless a slab product can be 0*inf = NaN
    const bool fast = finite64(ray.inv_direction.x) && finite64(ray.inv_direction.y) && finite64(ray.inv_direction.z);

    double t;
    if (kCount) cnt.node_tests++;
    {
        Box rb = loadBox(sv, 0u);
        if (!(fast ? boxIntersect<true>(rb, ray, t) : boxIntersect<false>(rb, ray, t))) return best;
    }

    int sp = 0;
    uint32_t node = 0;
    bool have = true;
    while (have) {
        NodeMeta m = loadMeta(sv, node);
        // ---- descend: only inner nodes in this loop, so the wave's lanes run box tests together
        while (m.b & kInnerFlag) {
            const uint32_t first = m.a, count = m.b & ~kInnerFlag;
            double near_t = 0.0;
            uint32_t near_node = kNoSurface;
            for (uint32_t c = first; c < first + count; c++) {
                Box cb = loadBox(sv, c);
                if (kCount) cnt.node_tests++;
                const bool hit = fast ? boxIntersect<true>(cb, ray, t) : boxIntersect<false>(cb, ray, t);
                if (hit && t <= best.t) {
                    // keep the nearest child in registers, push the others
                    uint32_t push_node = c;
                    double push_t = t;
                    if (near_node == kNoSurface || t < near_t) {
                        push_node = near_node;
                        push_t = near_t;
                        near_node = c;
                        near_t = t;
                    }
                    if (push_node != kNoSurface) {
                        if (sp < kMaxStackDepth) {
                            StackEntry e;
                            e.t = floatBelow(push_t);
                            e.node = push_node;
                            stk.put(sp++, e);
                        } else {
                            cnt.overflow = 1;
                        }
                    }
                }
            }
            if (near_node != kNoSurface) {
                node = near_node;
            } else {
                have = false;
                while (sp > 0) {
                    StackEntry e = stk.get(--sp);
                    if ((double)e.t <= best.t) {
                        node = e.node;
                        have = true;
                        break;
                    }
                }
                if (!have) break;
            }
            m = loadMeta(sv, node);
        }
        if (!have) break;
        // ---- leaf
        {
            const uint32_t end = m.a + m.b;
            for (uint32_t i = m.a; i < end; i++) {
                Hit h;
                if (kCount) cnt.prim_tests++;
                if (primIntersect<QuadricsIn<kAll>::value>(sv.prim + (size_t)i * kPrimStride, ray, h) && closer(h.t, i, best)) {
                    best = h;
                    best.surface = i;
                    if (kShadow && i != sq->light && h.t < sq->t_near) return best;
                }
            }
        }
        // ---- pop, culling entries that can no longer beat (or tie) the current closest hit
        have = false;
        while (sp > 0) {
            StackEntry e = stk.get(--sp);
            if ((double)e.t <= best.t) {
                node = e.node;
                have = true;
                break;
            }
        }
    }
    return best;
}

}  // namespace mcrt
