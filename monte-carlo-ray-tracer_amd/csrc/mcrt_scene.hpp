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
constexpr int kMaxStackDepth = 128;  // total entries per lane (LDS + global spill) - the MINIMUM: a scene whose tree can hold more on a stack gets its own bound (HostLayout::stack_bound)

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
    // FP32 cull records of the flat loop ("FP32 cull" below); no pairs = test every primitive in FP64
    cptr<float, kAll> flat_pre;
    uint32_t pre_tri_pairs, pre_sph_pairs;
    double pre_cx, pre_cy, pre_cz, pre_bound;
    // the same records as a KERNEL ARGUMENT (renderKernelFlatK, kPreK below): read with scalar loads into SGPRs, not from LDS
    const float* flat_pre_k = nullptr;
};

struct LaneStack {
    MCRT_LDS_AS StackEntry* lds;  // &lds_stack[lane_in_block]; stride = block size
    uint32_t lds_stride;
    StackEntry* spill;            // &spill[global_lane]; stride = total lanes
    uint32_t spill_stride;
    int max_depth = kMaxStackDepth;  // entries per lane in all (LDS + spill slab): the scene's stack bound, at least kMaxStackDepth (DeviceScene::stack_depth)
    MCRT_HD void put(int sp, StackEntry e) const {
        if (sp < kLdsStackDepth) lds[(uint32_t)sp * lds_stride] = e;
        else spill[(uint32_t)(sp - kLdsStackDepth) * (size_t)spill_stride] = e;
    }
    MCRT_HD StackEntry get(int sp) const {
        // two typed reads instead of one read through a selected pointer: the select would be a generic (flat) pointer
        // built from an address-space-3 one, which hipcc 7.2 cannot always select code for (intersectKernel<true>)
        StackEntry e = lds[(uint32_t)(sp < kLdsStackDepth ? sp : kLdsStackDepth - 1) * lds_stride];
        if (sp >= kLdsStackDepth) e = spill[(uint32_t)(sp - kLdsStackDepth) * (size_t)spill_stride];
        return e;
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

// ---- FP32 cull in front of the flat loop ------------------------------------------------------------------------
// The flat loop runs the reference's FP64 Moller-Trumbore / quadratic on EVERY primitive for every ray (~75 / ~60
// instructions each); all but one or two of those tests end in a reject. The cull evaluates the same determinants in
// packed FP32 (two primitives per v_pk_* instruction) and rejects a primitive only when the FP64 test is CERTAIN to
// reject it: every comparison carries an absolute error bound of the FP32 evaluation (inputs rounded to float around
// the scene centre, rays starting within `pre_bound` of it; derivation in DESIGN.md "FP32 cull"), so the survivors are
// a superset of the primitives the reference accepts. The survivors (a bit mask per lane) then go through the
// unchanged FP64 tests, per lane; closest hit and tie rule are untouched, so the result is bit-identical by
// construction. A ray outside the domain of the bounds (start farther than pre_bound from the centre, non-unit or
// non-finite direction) keeps every primitive.
//
// Records (floats, host-built by buildFlatCull, mcrt_layout.hpp), two primitives interleaved per component:
//   triangle pair [16][2]: a = v0 - centre (3), E1 (3), E2 (3), eu, ev, et, ew, pad (3)
//   sphere pair   [4][2]:  c - centre (3), r^2 + gamma ; then [4][2]: 1 - beta, pad (3)       (kSphPairFloats = 16)
// Rejected when (A = uN det, B = vN det, C = tN det, W = (uN + vN) det - det^2):
//   A < -eu (u < 0) | B < -ev (v < 0) | C < -et (t < 0) | W > ew (u + v > 1)
//   sphere: nb^2 (1 + 8u) - so2 (1 - beta) + (r^2 + gamma) < 0 (no real root) | (b > 0 and c > 0: both roots negative)
typedef float f2 __attribute__((vector_size(8)));
constexpr int kTriPairFloats = 32, kSphPairFloats = 16;
constexpr float kCullB2 = 1.0f + 8.0f * 5.9604645e-8f * 1.0001f;

template <class P>
MCRT_HD f2 ldf2(P p) {
    return f2{p[0], p[1]};
}

struct CullRay {
    f2 sx, sy, sz, dx, dy, dz;  // start - centre and direction, each splat over the pair
    bool in_domain;
};
// (the thresholds of the records assume a UNIT direction: a longer one - mcrt_intersect takes what the caller passes - scales the
// error terms by |d|^2 and is therefore out of the domain, like a start beyond `bound`; out of domain = every primitive survives)
MCRT_HD CullRay cullRayAt(double cx, double cy, double cz, double bound, d3 start, d3 direction) {
    CullRay r;
    const double x = start.x - cx, y = start.y - cy, z = start.z - cz;
    r.in_domain = fabs(x) <= bound && fabs(y) <= bound && fabs(z) <= bound &&
                  (direction.x * direction.x + direction.y * direction.y + direction.z * direction.z) <= 1.000001;
    const float fx = (float)x, fy = (float)y, fz = (float)z;
    const float gx = (float)direction.x, gy = (float)direction.y, gz = (float)direction.z;
    r.sx = f2{fx, fx}; r.sy = f2{fy, fy}; r.sz = f2{fz, fz};
    r.dx = f2{gx, gx}; r.dy = f2{gy, gy}; r.dz = f2{gz, gz};
    return r;
}
template <bool kAll>
MCRT_HD CullRay cullRay(const SceneViewT<kAll>& sv, d3 start, d3 direction) {
    return cullRayAt(sv.pre_cx, sv.pre_cy, sv.pre_cz, sv.pre_bound, start, direction);
}

// Bit i of the result is set when triangle i of the pairs given may be hit; pairs <= 16 (one mask word).
template <class P>
MCRT_HD uint32_t cullTriangles(P pre, uint32_t pairs, uint32_t count, const CullRay& r) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    uint32_t rej = 0;
    for (uint32_t p = pairs; p-- > 0;) {
        P q = pre + (size_t)p * kTriPairFloats;
        const f2 ax = ldf2(q + 0), ay = ldf2(q + 2), az = ldf2(q + 4);
        const f2 e1x = ldf2(q + 6), e1y = ldf2(q + 8), e1z = ldf2(q + 10);
        const f2 e2x = ldf2(q + 12), e2y = ldf2(q + 14), e2z = ldf2(q + 16);
        const f2 eu = ldf2(q + 18), ev = ldf2(q + 20), et = ldf2(q + 22), ew = ldf2(q + 24);
        const f2 Px = r.dy * e2z - e2y * r.dz, Py = r.dz * e2x - e2z * r.dx, Pz = r.dx * e2y - e2x * r.dy;  // cross(direction, E2)
        const f2 det = Px * e1x + Py * e1y + Pz * e1z;
        const f2 Tx = r.sx - ax, Ty = r.sy - ay, Tz = r.sz - az;
        const f2 uN = Px * Tx + Py * Ty + Pz * Tz;
        const f2 Qx = Ty * e1z - e1y * Tz, Qy = Tz * e1x - e1z * Tx, Qz = Tx * e1y - e1x * Ty;              // cross(T, E1)
        const f2 vN = Qx * r.dx + Qy * r.dy + Qz * r.dz;
        const f2 tN = Qx * e2x + Qy * e2y + Qz * e2z;
        const f2 A = uN * det + eu, B = vN * det + ev, C = tN * det + et;
        const f2 W = (det * det + ew) - (uN + vN) * det;
        // a set sign bit in any of the four = certain reject
        const uint32_t x1 = floatBits(A[1]) | floatBits(B[1]) | floatBits(C[1]) | floatBits(W[1]);
        const uint32_t x0 = floatBits(A[0]) | floatBits(B[0]) | floatBits(C[0]) | floatBits(W[0]);
        rej = (rej << 1) | (x1 >> 31);
        rej = (rej << 1) | (x0 >> 31);
    }
    const uint32_t all = count >= 32u ? 0xFFFFFFFFu : ((1u << count) - 1u);
    return ((r.in_domain && pairs != 0u) ? ~rej : 0xFFFFFFFFu) & all;
}

// Bit i set: sphere i of the pairs given may be hit; pairs <= 16.
template <class P>
MCRT_HD uint32_t cullSpheres(P pre, uint32_t pairs, uint32_t count, const CullRay& r) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    uint32_t rej = 0;
    for (uint32_t p = pairs; p-- > 0;) {
        P q = pre + (size_t)p * kSphPairFloats;
        const f2 cx = ldf2(q + 0), cy = ldf2(q + 2), cz = ldf2(q + 4), cg = ldf2(q + 6), cb = ldf2(q + 8);
        const f2 ox = cx - r.sx, oy = cy - r.sy, oz = cz - r.sz;     // centre - start = -so
        const f2 nb = ox * r.dx + oy * r.dy + oz * r.dz;             // -dot(direction, so) = -b / 2
        const f2 so2 = ox * ox + oy * oy + oz * oz;
        const f2 inner = cg - so2 * cb;                              // < 0: the start is certainly outside the sphere (c > 0)
        const f2 D = (nb * f2{kCullB2, kCullB2}) * nb + inner;       // < 0: the discriminant is certainly negative
        // outside and heading away (b > 0, i.e. nb < 0): both roots negative
        const uint32_t x1 = (floatBits(nb[1]) & floatBits(inner[1])) | floatBits(D[1]);
        const uint32_t x0 = (floatBits(nb[0]) & floatBits(inner[0])) | floatBits(D[0]);
        rej = (rej << 1) | (x1 >> 31);
        rej = (rej << 1) | (x0 >> 31);
    }
    const uint32_t all = count >= 32u ? 0xFFFFFFFFu : ((1u << count) - 1u);
    return ((r.in_domain && pairs != 0u) ? ~rej : 0xFFFFFFFFu) & all;
}

MCRT_HD uint32_t lowestBit(uint32_t m) { return (uint32_t)__builtin_ctz(m); }  // m != 0

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
    double dist;     // d, the distance to the light point (what the wavefront pipeline queues: WfRayQueue)
    MCRT_HD void setRange(double d) {
        dist = d;
        t_near = d * (1.0 - 1e-9);
        t_far = d * (1.0 + 1e-9);
    }
};

// kFlat: the caller knows the scene is in flat mode (a kernel instance for such scenes only): the BVH walk is not compiled in.
// kPreK (with kFlat; round 5): the cull records are read through sv.flat_pre_k - the kernel's own argument block, i.e. constant
// memory at an address every lane shares - so the compiler loads a pair's record with s_load_dwordx16 / x8 / x2 into SGPRs and the
// packed instructions take them as scalar operands, instead of seven wave-wide ds_read (1 KB returned each for 16 bytes of
// information) into 26 VGPRs. C2 -2.0 %, C2-GGX -1.1 % (profiles/r05_ab_c2_flat_karg.log). Asking for the NEXT pair's record ahead
// of this pair's arithmetic (double-buffered SGPRs) was 10 % SLOWER: 26 s_mov per pair on a wave's single issue stream.
template <bool kAll, bool kCount, bool kShadow, bool kFlat = false, bool kPreK = false>
MCRT_HD Hit sceneIntersect(const SceneViewT<kAll>& sv, const Ray& ray, const LaneStack& stk, TraceCounters& cnt,
                           const ShadowQuery* sq = nullptr) {
    Hit best;
    best.t = kDblMax;
    best.u = 0.0;
    best.v = 0.0;
    best.surface = kNoSurface;
    best.interpolate = false;
    cnt.rays++;
    // Shadow queries, exactly (integrator.cpp:68-73: the closest hit must BE the light). The flat loop below tests every primitive the
    // cull leaves anyway, so it simply returns the true closest hit, unbounded. The walks bound themselves by the light's OWN
    // intersection: the reference's FP64 test on the light's record first, then "is any other surface closer than that" (ties to the
    // lowest index as everywhere); a light its own test does not hit cannot be the closest hit - no walk. (Until round 4 the bound was
    // the distance d to the sampled point, d (1 +- 1e-9): off where the ray grazes the light and the computed t strays farther from d.)
    uint32_t sh_light = kNoSurface;
    double sh_near = 0.0;

    if (kFlat || sv.num_nodes == 0) {  // every primitive, as scene.cpp:161-173 does without a BVH
        if (kFlat || (kAll && sv.flat_prim)) {
            const uint32_t nt = sv.flat_tris, ns = sv.num_surfaces;
            // kFlat instances only carry the culled form; without cull
            // records (MCRT_FLAT_CULL=0) every primitive is a survivor
            if (kFlat || sv.pre_tri_pairs + sv.pre_sph_pairs != 0u) {
                // FP32 cull over all primitives (wave-uniform, packed), then the FP64 tests of each lane's survivors
                const CullRay cr = cullRay(sv, ray.start, ray.direction);
                // 32 primitives (16 pairs) per mask word
                for (uint32_t base = 0; base < nt; base += 32u) {
                    const uint32_t left = nt - base, pairs_left = sv.pre_tri_pairs - base / 2u;
                    uint32_t mt;
                    if constexpr (kPreK)
                        mt = cullTriangles(sv.flat_pre_k + (size_t)(base / 2u) * kTriPairFloats, sv.pre_tri_pairs ? (pairs_left < 16u ? pairs_left : 16u) : 0u,
                                           left < 32u ? left : 32u, cr);
                    else
                        mt = cullTriangles(sv.flat_pre + (size_t)(base / 2u) * kTriPairFloats, sv.pre_tri_pairs ? (pairs_left < 16u ? pairs_left : 16u) : 0u,
                                           left < 32u ? left : 32u, cr);
                    if (kCount) cnt.prim_tests += (uint32_t)__builtin_popcount(mt);
                    while (mt) {
                        const uint32_t i = base + lowestBit(mt);
                        mt &= mt - 1u;
                        double t, u, v;
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
                }
                cptr<float, kAll> spre = sv.flat_pre + (size_t)sv.pre_tri_pairs * kTriPairFloats;
                const float* spre_k = kPreK ? sv.flat_pre_k + (size_t)sv.pre_tri_pairs * kTriPairFloats : nullptr;
                for (uint32_t base = 0; base < ns - nt; base += 32u) {
                    const uint32_t left = ns - nt - base, pairs_left = sv.pre_sph_pairs - base / 2u;
                    uint32_t ms;
                    if constexpr (kPreK)
                        ms = cullSpheres(spre_k + (size_t)(base / 2u) * kSphPairFloats, sv.pre_sph_pairs ? (pairs_left < 16u ? pairs_left : 16u) : 0u,
                                         left < 32u ? left : 32u, cr);
                    else
                        ms = cullSpheres(spre + (size_t)(base / 2u) * kSphPairFloats, sv.pre_sph_pairs ? (pairs_left < 16u ? pairs_left : 16u) : 0u,
                                         left < 32u ? left : 32u, cr);
                    if (kCount) cnt.prim_tests += (uint32_t)__builtin_popcount(ms);
                    while (ms) {
                        const uint32_t i = nt + base + lowestBit(ms);
                        ms &= ms - 1u;
                        double t;
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
                }
                return best;
            }
            if constexpr (kFlat) return best;  // (not reached)
            // wave-uniform loops over the kind-sorted copy: no per-lane control flow at all
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
                if (kShadow && i != sh_light && h.t < sh_near) return best;
            }
        }
        return best;
    }

    if (kShadow) {  // the light's own intersection bounds the walk (see above)
        sh_light = sq->light;
        Hit h;
        if (kCount) cnt.prim_tests++;
        if (!primIntersect<QuadricsIn<kAll>::value>(sv.prim + (size_t)sh_light * kPrimStride, ray, h)) return best;
        // the light IS the hit so far (its leaf's box may start an ulp behind this t and be culled: the walk need not find it again);
        // another surface replaces it when it is closer - or exactly as far with a lower index, the tie rule of every query
        best = h;
        best.surface = sh_light;
        sh_near = h.t;
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
                        if (sp < stk.max_depth) {
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
                    if (kShadow && i != sh_light && h.t < sh_near) return best;
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
