/*
 * oracle/mcrt_oracle.c — TEST INFRASTRUCTURE ONLY (see mcrt_oracle.h).
 *
 * Plain-C restatement of the reference's hot path on the flattened arrays of include/mcrt.h.
 * Citations are path:line under /root/reference/source. Operation ORDER is the reference's
 * (glm 0.9.9.8 semantics: dot = (x*x' + y*y') + z*z'; normalize(v) = v * (1/sqrt(dot(v,v)));
 * min(a,b) = (b<a)?b:a; max(a,b) = (a<b)?b:a; vec/scalar divides per component), so that with
 * -ffp-contract=off and the same libm the output equals the reference's to the last bit.
 */
#define _GNU_SOURCE
#include "mcrt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ---------------------------------------------------------------- vec3 (glm::dvec3 semantics) */
typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* func_geometric.inl:56-59 */
static inline v3 vcross(v3 x, v3 y) { /* func_geometric.inl:79-82 */
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline v3 vnormalize(v3 v) { return vscale(v, 1.0 / sqrt(vdot(v, v))); } /* func_geometric.inl:88 */
static inline double gmin(double x, double y) { return (y < x) ? y : x; }  /* glm::min */
static inline double gmax(double x, double y) { return (x < y) ? y : x; }  /* glm::max == std::max */
static inline double smin(double a, double b) { return (b < a) ? b : a; }  /* std::min */
static inline double pow2(double x) { return x * x; }
static inline double compMax(v3 v) { return gmax(gmax(v.x, v.y), v.z); }   /* gtx/component_wise.inl:120-126 */
static inline double compMin(v3 v) { return gmin(gmin(v.x, v.y), v.z); }
static inline v3 vmix(v3 x, v3 y, double a) { return vadd(vscale(x, 1.0 - a), vscale(y, a)); } /* func_common.inl:110 */
static inline double smix(double x, double y, double a) { return x * (1.0 - a) + y * a; }
static inline v3 ld3(const double* p) { return V(p[0], p[1], p[2]); }

#define PI 3.14159265358979323846
#define INV_PI 0.31830988618379067154
#define TWO_PI 6.283185307179586476925
#define EPSILON 1e-9 /* common/constants.hpp:3-9 */

/* ------------------------------------------------------------------ Sampler (sampling/) */
static uint32_t g_dirs[6][32];
static pthread_once_t g_dirs_once = PTHREAD_ONCE_INIT;

static uint32_t reverseBits(uint32_t x) { /* sampling/sobol.hpp:7-14 */
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
}

static void initDirections(void) { /* sampling/sobol.hpp:18-54 (new-joe-kuo-6.21201, dims 2..7) */
    static const uint32_t s[6] = {1, 2, 3, 3, 4, 4};
    static const uint32_t a[6] = {0, 1, 1, 2, 1, 4};
    static const uint32_t m[6][4] = {{1}, {1, 3}, {1, 3, 1}, {1, 1, 1}, {1, 1, 3, 3}, {1, 3, 5, 13}};
    for (uint32_t dim = 0; dim < 6; dim++) {
        uint32_t* Vd = g_dirs[dim];
        for (uint32_t bit = 0; bit < s[dim]; bit++) Vd[bit] = m[dim][bit] << (31 - bit);
        for (uint32_t bit = s[dim]; bit < 32; bit++) {
            Vd[bit] = Vd[bit - s[dim]] ^ (Vd[bit - s[dim]] >> s[dim]);
            for (uint32_t k = 1; k < s[dim]; k++)
                Vd[bit] ^= (((a[dim] >> (s[dim] - 1 - k)) & 1) * Vd[bit - k]);
        }
        for (uint32_t bit = 0; bit < 32; bit++) Vd[bit] = reverseBits(Vd[bit]);
    }
}

static uint32_t bitReversedSample(int dim, uint32_t index) { /* sampling/sobol.hpp:58-71 */
    if (dim == 0) return index;
    uint32_t x = 0u;
    for (int bit = 0; index; index >>= 1u, bit++) x ^= (index & 1u) * g_dirs[dim - 1][bit];
    return x;
}

static uint32_t hash32(uint32_t x) { /* sampler.hpp:76-84 */
    x ^= x >> 15; x *= 0xd168aaadu; x ^= x >> 15; x *= 0xaf723597u; x ^= x >> 15;
    return x;
}
static uint32_t hashCombine(uint32_t seed, uint32_t v) { /* sampler.hpp:87-90 */
    return seed ^ (v + 0x9e3779b9u + (seed << 6) + (seed >> 2));
}
static uint32_t scramble(uint32_t x, uint32_t seed) { /* sampler.hpp:61-72 */
    x ^= x * 0x3d20adeau;
    x += seed;
    x *= (seed >> 16) | 1u;
    x ^= x * 0x05526c56u;
    x ^= x * 0x53a22864u;
    return reverseBits(x);
}

typedef struct { uint32_t global_seed, base_seed, seed, sequence, bit_reversed_index, shuffled_index; } Sampler;

static void samplerInitiate(Sampler* s, uint32_t start_seed) { s->base_seed = hashCombine(s->global_seed, hash32(start_seed)); } /* :32-35 */
static void samplerSetIndex(Sampler* s, uint32_t index) { /* :38-44 */
    s->sequence = 0u; s->seed = s->base_seed; s->bit_reversed_index = reverseBits(index); s->shuffled_index = index;
}
static void samplerShuffle(Sampler* s) { /* :48-52 */
    s->seed = hashCombine(s->base_seed, hash32(++s->sequence));
    s->shuffled_index = scramble(s->bit_reversed_index, s->seed);
}
static double samplerGet(const Sampler* s, int dim) { /* :20-30 */
    return scramble(bitReversedSample(dim, s->shuffled_index), hashCombine(s->seed, hash32((uint32_t)dim))) * 0x1p-32;
}
enum { DIM_PIXEL = 0, DIM_LENS = 2, DIM_LIGHT = 0, DIM_BSDF = 3, DIM_INTERACTION = 5, DIM_ABSORB = 6 }; /* sampling.hpp:59-76 */

void oracle_sampler(uint32_t global_seed, uint32_t pixel, uint32_t index, uint32_t shuffles, double out[7]) {
    pthread_once(&g_dirs_once, initDirections);
    Sampler s; memset(&s, 0, sizeof(s)); s.global_seed = global_seed;
    samplerInitiate(&s, pixel);
    samplerSetIndex(&s, index);
    for (uint32_t i = 0; i < shuffles; i++) samplerShuffle(&s);
    for (int d = 0; d < 7; d++) out[d] = samplerGet(&s, d);
}

/* ------------------------------------------------------------------ Ray (ray/ray.hpp:10-36) */
typedef struct {
    v3 start, direction, inv_direction;
    double medium_ior, refraction_scale;
    int dirac_delta, refraction;
    uint16_t depth, diffuse_depth;
    int refraction_level;
} Ray;

static Ray rayDir(v3 start, v3 direction, double medium_ior) { /* ray.cpp:13-14 */
    Ray r; memset(&r, 0, sizeof(r));
    r.start = start; r.direction = direction;
    r.inv_direction = V(1.0 / direction.x, 1.0 / direction.y, 1.0 / direction.z);
    r.medium_ior = medium_ior; r.refraction_scale = 1.0;
    return r;
}
static Ray rayTo(v3 start, v3 end) { return rayDir(start, vnormalize(vsub(end, start)), 1.0); } /* ray.cpp:10-11 */
static v3 rayAt(const Ray* r, double t) { return vadd(r->start, vscale(r->direction, t)); }      /* ray.cpp:69-72 */

#define MAX_IORS 1024 /* (the reference's std::vector is unbounded; the GPU tests nest 150 media) */
typedef struct { double iors[MAX_IORS]; int size; } RefractionHistory; /* ray.cpp:74-98 */
static void rhInit(RefractionHistory* h, const Ray* ray) { h->iors[0] = ray->medium_ior; h->size = 1; }
static void rhUpdate(RefractionHistory* h, const Ray* ray) {
    if (ray->refraction_level > 0) {
        if (ray->refraction_level == h->size) { if (h->size < MAX_IORS) h->iors[h->size++] = ray->medium_ior; }
        else if (ray->refraction_level < h->size - 1) h->size--;
    }
}
static double rhExternalIOR(const RefractionHistory* h, const Ray* ray) {
    int i = ray->refraction_level - 1;
    if (i < 0) i = 0;
    if (i > h->size - 1) i = h->size - 1;
    return h->iors[i];
}

/* ------------------------------------------------------------------ intersection */
typedef struct { double t; uint32_t surface; double u, v; int interpolate; } Hit; /* ray/intersection.hpp:9-23 */
#define NO_SURFACE 0xFFFFFFFFu

typedef struct {
    const mcrt_scene_desc* s;
    oracle_counters* c;
} SceneRef;

static int bbIntersect(const double* b, const Ray* ray, double* t) { /* common/bounding-box.cpp:9-17 */
    v3 mn = ld3(b), mx = ld3(b + 3);
    v3 t0 = vmul(vsub(mn, ray->start), ray->inv_direction);
    v3 t1 = vmul(vsub(mx, ray->start), ray->inv_direction);
    v3 lo = V(gmin(t0.x, t1.x), gmin(t0.y, t1.y), gmin(t0.z, t1.z));
    v3 hi = V(gmax(t0.x, t1.x), gmax(t0.y, t1.y), gmax(t0.z, t1.z));
    *t = gmax(compMax(lo), 0.0);
    return compMin(hi) >= *t;
}

static int triIntersect(const mcrt_scene_desc* s, uint32_t i, const Ray* ray, Hit* out) { /* surface/triangle.cpp:23-63 */
    v3 v0 = ld3(s->surf_v + 9 * (size_t)i), E1 = ld3(s->surf_e + 9 * (size_t)i), E2 = ld3(s->surf_e + 9 * (size_t)i + 3);
    v3 P = vcross(ray->direction, E2);
    double determinant = vdot(P, E1);
    if (determinant < EPSILON && determinant > -EPSILON) return 0;
    double inv_determinant = 1.0 / determinant;
    v3 T = vsub(ray->start, v0);
    double u = vdot(P, T) * inv_determinant;
    if (u > 1.0 || u < 0.0) return 0;
    v3 Q = vcross(T, E1);
    double v = vdot(Q, ray->direction) * inv_determinant;
    if (v > 1.0 || v < 0.0 || u + v > 1.0) return 0;
    double t = vdot(Q, E2) * inv_determinant;
    if (t <= 0.0) return 0;
    out->t = t; out->u = 0.0; out->v = 0.0; out->interpolate = 0;
    if (s->surf_interpolate[i]) { out->u = u; out->v = v; out->interpolate = 1; }
    return 1;
}

static int solveQuadratic(double a, double b, double c, double* t_min, double* t_max) { /* common/util.hpp:60-83 */
    if (a != 0.0) {
        double d = b * b - 4.0 * a * c;
        if (d < 0.0) return 0;
        double t = -0.5 * (b + (b < 0.0 ? -sqrt(d) : sqrt(d)));
        *t_min = t / a;
        *t_max = c / t;
        if (*t_min > *t_max) { double tmp = *t_min; *t_min = *t_max; *t_max = tmp; }
        return 1;
    }
    if (b != 0.0) { *t_min = *t_max = -c / b; return 1; }
    return 0;
}

static int sphIntersect(const mcrt_scene_desc* s, uint32_t i, const Ray* ray, Hit* out) { /* surface/sphere.cpp:13-26 */
    const double* p = s->surf_v + 9 * (size_t)i;
    v3 so = vsub(ray->start, ld3(p));
    double b = 2.0 * vdot(ray->direction, so);
    double c = vdot(so, so) - pow2(p[3]);
    double t_min, t_max;
    if (solveQuadratic(1.0, b, c, &t_min, &t_max) && t_max >= 0.0) {
        out->t = t_min < 0.0 ? t_max : t_min; out->u = out->v = 0.0; out->interpolate = 0;
        return 1;
    }
    return 0;
}

/* Surface::Quadric, surface/quadric.cpp. Record: Q as glm stores it (Q[c][r] at 4c + r), BB_.min, BB_.max. */
static const double* quadricRecord(const mcrt_scene_desc* s, uint32_t i) { return s->quadrics + 22 * (size_t)s->surf_v[9 * (size_t)i]; }

static int quadricIntersect(const mcrt_scene_desc* s, uint32_t i, const Ray* ray, Hit* out) { /* quadric.cpp:69-100 */
    const double* q = quadricRecord(s, i);
    double t_bb = 0.0;
    if (!bbIntersect(q + 16, ray, &t_bb)) return 0;
    v3 o = rayAt(ray, t_bb), d = ray->direction; /* dvec4 o(ray(t_bb), 1.0), d(ray.direction, 0.0) */
    double Qo[4], Qd[4];
    for (int r = 0; r < 4; r++) { /* dmat4 * dvec4 = (col0*x + col1*y) + (col2*z + col3*w), type_mat4x4.inl:561-573 */
        Qo[r] = (q[0 + r] * o.x + q[4 + r] * o.y) + (q[8 + r] * o.z + q[12 + r] * 1.0);
        Qd[r] = (q[0 + r] * d.x + q[4 + r] * d.y) + (q[8 + r] * d.z + q[12 + r] * 0.0);
    }
    /* dvec4 dot = (x*x' + y*y') + (z*z' + w*w'), func_geometric.inl:58-65 */
    double a = (d.x * Qd[0] + d.y * Qd[1]) + (d.z * Qd[2] + 0.0 * Qd[3]);
    double b = ((d.x * Qo[0] + d.y * Qo[1]) + (d.z * Qo[2] + 0.0 * Qo[3])) * 2.0;
    double c = (o.x * Qo[0] + o.y * Qo[1]) + (o.z * Qo[2] + 1.0 * Qo[3]);
    double t_min, t_max;
    if (solveQuadratic(a, b, c, &t_min, &t_max) && t_max >= 0.0) {
        double t = t_bb + (t_min < 0.0 ? t_max : t_min);
        v3 p = rayAt(ray, t); /* BB_.contains, bounding-box.cpp:19-23 */
        if (!(p.x >= q[16] && p.y >= q[17] && p.z >= q[18] && p.x <= q[19] && p.y <= q[20] && p.z <= q[21])) return 0;
        out->t = t; out->u = out->v = 0.0; out->interpolate = 0;
        return 1;
    }
    return 0;
}

static v3 quadricNormal(const mcrt_scene_desc* s, uint32_t i, v3 pos) { /* quadric.cpp:127-130, G = 2 * Q's upper rows (:38-45) */
    const double* q = quadricRecord(s, i);
    v3 g; /* dmat4x3 * dvec4 adds left to right, type_mat4x3.inl:469-478 */
    g.x = (2.0 * q[0]) * pos.x + (2.0 * q[4]) * pos.y + (2.0 * q[8]) * pos.z + (2.0 * q[12]) * 1.0;
    g.y = (2.0 * q[1]) * pos.x + (2.0 * q[5]) * pos.y + (2.0 * q[9]) * pos.z + (2.0 * q[13]) * 1.0;
    g.z = (2.0 * q[2]) * pos.x + (2.0 * q[6]) * pos.y + (2.0 * q[10]) * pos.z + (2.0 * q[14]) * 1.0;
    return vnormalize(g);
}

static int surfIntersect(const SceneRef* S, uint32_t i, const Ray* ray, Hit* out) {
    if (S->c) { S->c->prim_tests++; if (S->s->surf_kind[i] == MCRT_SURF_SPHERE) S->c->sphere_tests++; }
    if (S->s->surf_kind[i] == MCRT_SURF_QUADRIC) return quadricIntersect(S->s, i, ray, out);
    return S->s->surf_kind[i] == MCRT_SURF_SPHERE ? sphIntersect(S->s, i, ray, out) : triIntersect(S->s, i, ray, out);
}

/* common/priority-queue.hpp:11-126 (binary heap; `less(a,b)` is the element's operator<) */
typedef struct { double t; uint32_t node; } NodeIsect; /* bvh/bvh.hpp:76-81: a < b  <=>  b.t < a.t */
typedef struct { NodeIsect* H; size_t size, cap; } NodeQueue;
static inline int niLess(NodeIsect a, NodeIsect b) { return b.t < a.t; }
static void nqPush(NodeQueue* q, NodeIsect value) { /* :19-31 */
    if (q->size == q->cap) { q->cap = q->cap ? q->cap * 2 : 64; q->H = (NodeIsect*)realloc(q->H, q->cap * sizeof(NodeIsect)); }
    size_t index = q->size++;
    while (index > 0) {
        size_t parent = (index - 1) / 2;
        if (!niLess(q->H[parent], value)) break;
        q->H[index] = q->H[parent];
        index = parent;
    }
    q->H[index] = value;
}
static void nqShiftDown(NodeQueue* q, NodeIsect value, size_t index) { /* :103-123 */
    for (;;) {
        size_t left = 2 * index + 1, right = left + 1, max_child;
        if (right < q->size) max_child = left + (size_t)niLess(q->H[left], q->H[right]);
        else if (left < q->size) max_child = left;
        else break;
        if (!niLess(value, q->H[max_child])) break;
        q->H[index] = q->H[max_child];
        index = max_child;
    }
    q->H[index] = value;
}
static void nqPop(NodeQueue* q) { /* :33-45 */
    if (q->size > 1) { NodeIsect value = q->H[--q->size]; nqShiftDown(q, value, 0); }
    else q->size--;
}

typedef struct { NodeQueue to_visit; void* knn_visit; void* knn_result[2]; } ThreadScratch;

/* oracle_set_true_minimum(1): the closest hit as the TRUE minimum over all primitives with ties to the lowest surface index - what
 * the HIP walks return by construction (DESIGN.md "Ties") - instead of the reference's heap-order result, which differs from it only
 * where two surfaces are hit within an ulp of each other and a child box starts exactly at the first one's t (bvh.cpp:100,120: strict
 * `<` on hits, `top.t >= intersect.t` ends the walk without looking inside). Default 0 = the reference. The tests use it to show that
 * a frame that is not the reference's bits (metal_bunnies: a shelf coplanar with the back wall) differs by this rule and nothing else. */
static int g_true_minimum = 0;
void oracle_set_true_minimum(int on) { g_true_minimum = on; }
static inline int hitCloser(const Hit* h, uint32_t i, const Hit* best) {
    return h->t < best->t || (g_true_minimum && h->t == best->t && i < best->surface);
}

static Hit sceneIntersect(const SceneRef* S, const Ray* ray, ThreadScratch* ts) { /* scene/scene.cpp:151-176, bvh/bvh.cpp:80-129 */
    const mcrt_scene_desc* s = S->s;
    Hit best; best.t = DBL_MAX; best.surface = NO_SURFACE; best.u = best.v = 0.0; best.interpolate = 0;
    if (S->c) S->c->rays++;
    if (s->num_nodes == 0) {
        for (uint32_t i = 0; i < s->num_surfaces; i++) {
            Hit h;
            if (surfIntersect(S, i, ray, &h) && hitCloser(&h, i, &best)) { best = h; best.surface = i; }
        }
        return best;
    }
    NodeQueue* q = &ts->to_visit; q->size = 0;
    double t;
    if (S->c) S->c->node_tests++;
    if (bbIntersect(s->node_bounds, ray, &t)) {
        uint32_t node_idx = 0;
        for (;;) {
            uint32_t ns = s->node_num_surfaces[node_idx];
            if (ns) {
                uint32_t start = s->node_start_surface[node_idx], end = start + ns;
                for (uint32_t i = start; i < end; i++) {
                    Hit h;
                    if (surfIntersect(S, i, ray, &h) && hitCloser(&h, i, &best)) { best = h; best.surface = i; }
                }
            } else {
                uint32_t child = node_idx + 1;
                while (child != 0) {
                    if (S->c) S->c->node_tests++;
                    if (bbIntersect(s->node_bounds + 6 * (size_t)child, ray, &t) && (t < best.t || (g_true_minimum && t == best.t))) {
                        NodeIsect ni = {t, child};
                        nqPush(q, ni);
                    }
                    child = s->node_next_sibling[child];
                }
            }
            if (q->size == 0 || (g_true_minimum ? q->H[0].t > best.t : q->H[0].t >= best.t)) break;
            node_idx = q->H[0].node;
            nqPop(q);
        }
    }
    return best;
}

/* ------------------------------------------------------------------ surfaces */
static v3 surfNormal(const mcrt_scene_desc* s, uint32_t i, v3 pos) { /* triangle.cpp:99-102, sphere.cpp:46-49 */
    if (s->surf_kind[i] == MCRT_SURF_SPHERE) {
        const double* p = s->surf_v + 9 * (size_t)i;
        return vdivs(vsub(pos, ld3(p)), p[3]);
    }
    if (s->surf_kind[i] == MCRT_SURF_QUADRIC) return quadricNormal(s, i, pos);
    return ld3(s->surf_e + 9 * (size_t)i + 6);
}
static v3 surfInterpolatedNormal(const mcrt_scene_desc* s, uint32_t i, double u, double v) { /* triangle.cpp:109-113 */
    const double* n = s->surf_vn + 9 * (size_t)i;
    v3 r = vadd(vadd(vscale(ld3(n), 1.0 - u - v), vscale(ld3(n + 3), u)), vscale(ld3(n + 6), v));
    return vnormalize(r);
}
static v3 surfSample(const mcrt_scene_desc* s, uint32_t i, double u, double v) { /* triangle.cpp:93-97, sphere.cpp:37-44 */
    const double* p = s->surf_v + 9 * (size_t)i;
    if (s->surf_kind[i] == MCRT_SURF_SPHERE) {
        double z = 1.0 - 2.0 * u;
        double r = sqrt(1.0 - pow2(z));
        double phi = TWO_PI * v;
        return vadd(ld3(p), vscale(V(r * cos(phi), r * sin(phi), z), p[3]));
    }
    double su = sqrt(u);
    return vadd(vadd(vscale(ld3(p), 1 - su), vscale(ld3(p + 3), (1 - v) * su)), vscale(ld3(p + 6), v * su));
}

/* ------------------------------------------------------------------ Fresnel / GGX / Material */
static double fresnelDielectric(double n1, double n2, double cos_theta) { /* material/fresnel.cpp:16-27 */
    double g2 = pow2(n2 / n1) + pow2(cos_theta) - 1.0;
    if (g2 < 0.0) return 1.0;
    double g = sqrt(g2);
    double g_p_c = g + cos_theta;
    double g_m_c = g - cos_theta;
    return 0.5 * pow2(g_m_c / g_p_c) * (1.0 + pow2((g_p_c * cos_theta - 1.0) / (g_m_c * cos_theta + 1.0)));
}

static v3 vsqrt(v3 a) { return V(sqrt(a.x), sqrt(a.y), sqrt(a.z)); }
static v3 vpow2(v3 a) { return vmul(a, a); }
static v3 vadds(v3 a, double s) { return V(a.x + s, a.y + s, a.z + s); }
static v3 vsubs(v3 a, double s) { return V(a.x - s, a.y - s, a.z - s); }
static v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }

static v3 fresnelConductor(double n1, v3 real, v3 imag, double cos_theta) { /* material/fresnel.cpp:30-49 */
    double cos_theta2 = pow2(cos_theta);
    double sin_theta2 = 1.0 - cos_theta2;
    v3 eta2 = vpow2(vdivs(real, n1));
    v3 eta_k2 = vpow2(vdivs(imag, n1));
    v3 t0 = vsubs(vsub(eta2, eta_k2), sin_theta2);
    v3 a2_p_b2 = vsqrt(vadd(vpow2(t0), vmul(vscale(eta2, 4.0), eta_k2))); /* 4.0 * eta2 * eta_k2 */
    v3 t1 = vadds(a2_p_b2, cos_theta2);
    v3 t2 = vscale(vsqrt(vscale(vadd(a2_p_b2, t0), 0.5)), 2.0 * cos_theta); /* 2.0*cos * sqrt(0.5*(..)) */
    v3 r_perp = vdiv(vsub(t1, t2), vadd(t1, t2));
    v3 t3 = vadds(vscale(a2_p_b2, cos_theta2), pow2(sin_theta2));
    v3 t4 = vscale(t2, sin_theta2);
    v3 r_par = vdiv(vmul(r_perp, vsub(t3, t4)), vadd(t3, t4));
    return vscale(vadd(r_par, r_perp), 0.5);
}

static double ggxD(v3 m, const double* a) { /* material/ggx.cpp:21-24 */
    return 1.0 / (PI * a[0] * a[1] * pow2(pow2(m.x / a[0]) + pow2(m.y / a[1]) + pow2(m.z)));
}
static double ggxLambda(v3 wo, const double* a) { /* :31-34 */
    return (-1.0 + sqrt(1.0 + (pow2(a[0] * wo.x) + pow2(a[1] * wo.y)) / (pow2(wo.z)))) / 2.0;
}
static double ggxG1(v3 wo, const double* a) { return 1.0 / (1.0 + ggxLambda(wo, a)); }              /* :36-39 */
static double ggxG2(v3 wi, v3 wo, const double* a) { return 1.0 / (1.0 + ggxLambda(wo, a) + ggxLambda(wi, a)); } /* :41-44 */
static double ggxDV(v3 m, v3 wo, const double* a) { return ggxG1(wo, a) * vdot(wo, m) * ggxD(m, a) / wo.z; }     /* :26-29 */
static double ggxReflection(v3 wi, v3 wo, const double* a, double* PDF) { /* :46-52 */
    v3 m = vnormalize(vadd(wo, wi));
    *PDF = ggxDV(m, wo, a) / (4.0 * vdot(m, wo));
    return ggxD(m, a) * ggxG2(wi, wo, a) / (4.0 * wo.z * wi.z);
}
static double ggxTransmission(v3 wi, v3 wo, double n1, double n2, const double* a, double* PDF) { /* :54-65 */
    v3 m = vadd(vscale(wo, n1), vscale(wi, n2));
    double m_length2 = vdot(m, m);
    m = vdivs(m, sqrt(m_length2));
    if (n1 < n2) m = vneg(m);
    double dm_dwi = pow2(n2) * fabs(vdot(wi, m)) / m_length2;
    *PDF = ggxDV(m, wo, a) * dm_dwi;
    return fabs(ggxG2(wi, wo, a) * ggxD(m, a) * vdot(wo, m) * dm_dwi / (wo.z * wi.z));
}
static v3 ggxVisibleMicrofacet(double u, double v, v3 wo, const double* a) { /* :67-88 */
    v3 Vh = vnormalize(V(a[0] * wo.x, a[1] * wo.y, wo.z));
    double len2 = pow2(Vh.x) + pow2(Vh.y);
    v3 T1 = len2 > 0.0 ? vscale(V(-Vh.y, Vh.x, 0.0), 1.0 / sqrt(len2)) : V(1.0, 0.0, 0.0);
    v3 T2 = vcross(Vh, T1);
    double r = sqrt(u);
    double phi = v * TWO_PI;
    double t1 = r * cos(phi);
    double t2 = r * sin(phi);
    double s = 0.5 * (1.0 + Vh.z);
    t2 = (1.0 - s) * sqrt(1.0 - pow2(t1)) + s * t2;
    v3 Nh = vadd(vadd(vscale(T1, t1), vscale(T2, t2)), vscale(Vh, sqrt(gmax(0.0, 1.0 - pow2(t1) - pow2(t2)))));
    return vnormalize(V(a[0] * Nh.x, a[1] * Nh.y, gmax(0.0, Nh.z)));
}

static v3 matLambertian(const mcrt_material* m) { return vscale(ld3(m->reflectance), INV_PI); } /* material.cpp:76-79 */
static v3 matOrenNayar(const mcrt_material* m, v3 wi, v3 wo) { /* material.cpp:82-95 */
    double cos_delta_phi = gmin(gmax((wi.x * wo.x + wi.y * wo.y) /
                                     sqrt((pow2(wi.x) + pow2(wi.y)) * (pow2(wo.x) + pow2(wo.y))), 0.0), 1.0);
    double D = sqrt((1.0 - pow2(wi.z)) * (1.0 - pow2(wo.z))) / gmax(wi.z, wo.z);
    return vscale(matLambertian(m), m->A + m->B * cos_delta_phi * D);
}
static v3 matDiffuseReflection(const mcrt_material* m, v3 wi, v3 wo, double* PDF) { /* material.cpp:17-27 */
    if (wi.z < 0.0) { *PDF = 0.0; return V(0, 0, 0); }
    *PDF = wi.z * INV_PI;
    return (m->flags & MCRT_MAT_ROUGH) ? matOrenNayar(m, wi, wo) : matLambertian(m);
}
static v3 matSpecularReflection(const mcrt_material* m, v3 wi, v3 wo, double* PDF) { /* material.cpp:29-45 */
    if (wi.z < 0.0) { *PDF = 0.0; return V(0, 0, 0); }
    if (m->flags & MCRT_MAT_ROUGH_SPECULAR) return vscale(ld3(m->specular_reflectance), ggxReflection(wi, wo, m->a, PDF));
    *PDF = 1.0;
    return vdivs(ld3(m->specular_reflectance), fabs(wi.z));
}
static v3 matSpecularTransmission(const mcrt_material* m, v3 wi, v3 wo, double n1, double n2, double* PDF, int inside, int flux) { /* material.cpp:47-69 */
    if (wi.z > 0.0) { *PDF = 0.0; return V(0, 0, 0); }
    v3 btdf = !inside ? ld3(m->transmittance) : V(1.0, 1.0, 1.0);
    if (m->flags & MCRT_MAT_ROUGH_SPECULAR) {
        btdf = vscale(btdf, ggxTransmission(wi, wo, n1, n2, m->a, PDF));
        if (flux) btdf = vscale(btdf, pow2(n2 / n1));
    } else {
        *PDF = 1.0;
        btdf = vmul(btdf, vdivs(ld3(m->transmittance), fabs(wi.z)));
        if (!flux) btdf = vscale(btdf, pow2(n1 / n2));
    }
    return btdf;
}

/* ------------------------------------------------------------------ Interaction (ray/interaction.cpp) */
typedef struct { v3 c0, c1, c2; } M3; /* glm::dmat3 columns; common/coordinate-system.cpp:7-18 */
static M3 orthonormalBasis(v3 N) {
    double sign = copysign(1.0, N.z);
    double a = -1.0 / (sign + N.z);
    double b = N.x * N.y * a;
    M3 T = {V(1.0 + sign * N.x * N.x * a, sign * b, -sign * N.x), V(b, sign + N.y * N.y * a, -N.y), N};
    return T;
}
static v3 csFrom(const M3* T, v3 v) { /* T * v, type_mat3x3.inl:468-474 */
    return V(T->c0.x * v.x + T->c1.x * v.y + T->c2.x * v.z,
             T->c0.y * v.x + T->c1.y * v.y + T->c2.y * v.z,
             T->c0.z * v.x + T->c1.z * v.y + T->c2.z * v.z);
}
static v3 csTo(const M3* T, v3 v) { /* transpose(T) * v */
    return V(T->c0.x * v.x + T->c0.y * v.y + T->c0.z * v.z,
             T->c1.x * v.x + T->c1.y * v.y + T->c1.z * v.z,
             T->c2.x * v.x + T->c2.y * v.y + T->c2.z * v.z);
}

enum { REFLECT = 0, REFRACT = 1, DIFFUSE = 2 };
typedef struct {
    int type;
    double t, n1, n2, T, R;
    const mcrt_material* material;
    uint32_t surface;
    v3 position, normal, out;
    M3 shading_cs;
    int inside, dirac_delta;
    Ray ray;
} Interaction;

static void iaSelectType(Interaction* ia, const Sampler* smp) { /* interaction.cpp:156-183 */
    uint32_t f = ia->material->flags;
    if (f & (MCRT_MAT_PERFECT_MIRROR | MCRT_MAT_COMPLEX_IOR)) ia->type = REFLECT;
    else if (ia->n2 < 1.0) ia->type = DIFFUSE;
    else {
        double p = samplerGet(smp, DIM_INTERACTION);
        if (ia->R > p) ia->type = REFLECT;
        else if (ia->R + (1.0 - ia->R) * ia->T > p) ia->type = REFRACT;
        else ia->type = DIFFUSE;
    }
}

static void iaInit(Interaction* ia, const mcrt_scene_desc* s, const Hit* isect, const Ray* ray, double external_ior, const Sampler* smp) { /* interaction.cpp:12-54 */
    ia->t = isect->t; ia->ray = *ray; ia->out = vneg(ray->direction); ia->n1 = ray->medium_ior;
    ia->surface = isect->surface;
    ia->material = &s->materials[s->surf_material[isect->surface]];
    ia->position = rayAt(ray, ia->t);
    ia->normal = surfNormal(s, isect->surface, ia->position);
    double cos_theta = vdot(ray->direction, ia->normal);
    ia->inside = cos_theta > 0.0;
    int opaque = (ia->material->flags & MCRT_MAT_OPAQUE) != 0;
    ia->n2 = (ia->inside && !opaque) ? external_ior : ia->material->ior;
    v3 shading_normal = ia->normal;
    if (isect->interpolate) {
        shading_normal = surfInterpolatedNormal(s, isect->surface, isect->u, isect->v);
        if ((cos_theta < 0.0) != (vdot(ray->direction, shading_normal) < 0.0)) shading_normal = ia->normal;
    }
    if (cos_theta > 0.0) { ia->normal = vneg(ia->normal); shading_normal = vneg(shading_normal); }
    ia->shading_cs = orthonormalBasis(shading_normal);
    ia->R = fresnelDielectric(ia->n1, ia->n2, vdot(shading_normal, ia->out));
    ia->T = ia->material->transparency;
    int rough_specular = (ia->material->flags & MCRT_MAT_ROUGH_SPECULAR) != 0;
    if (rough_specular) ia->R = gmin(gmax(ia->R, 0.1), 0.9);
    iaSelectType(ia, smp);
    ia->dirac_delta = ia->type != DIFFUSE && !rough_specular;
}

static v3 iaBSDFLocal(const Interaction* ia, v3 wo, v3 wi, double* pdf, int flux, int wi_dirac_delta) { /* interaction.cpp:84-153 */
    const mcrt_material* m = ia->material;
    uint32_t f = m->flags;
    double n1 = ia->n1, n2 = ia->n2;
    double cos_theta = wo.z;
    if (f & MCRT_MAT_ROUGH_SPECULAR) {
        if (wi.z > 0.0) cos_theta = vdot(wo, vnormalize(vadd(wo, wi)));
        else {
            v3 mm = vnormalize(vadd(vscale(wo, n1), vscale(wi, n2)));
            cos_theta = vdot(wo, mm);
            if (n1 < n2) cos_theta = -cos_theta;
        }
    }
    if (f & (MCRT_MAT_PERFECT_MIRROR | MCRT_MAT_COMPLEX_IOR)) {
        v3 brdf = matSpecularReflection(m, wi, wo, pdf);
        if (f & MCRT_MAT_COMPLEX_IOR) brdf = vmul(brdf, fresnelConductor(n1, ld3(m->ior_real), ld3(m->ior_imag), cos_theta));
        return brdf;
    }
    if (n2 < 1.0) return matDiffuseReflection(m, wi, wo, pdf);
    double F = fresnelDielectric(n1, n2, cos_theta);
    double pdf_s, pdf_d;
    v3 brdf_s = matSpecularReflection(m, wi, wo, &pdf_s);
    v3 brdf_d = matDiffuseReflection(m, wi, wo, &pdf_d);
    double pdf_t = pdf_s;
    v3 btdf = brdf_s;
    if (F < 1.0) btdf = matSpecularTransmission(m, wi, wo, n1, n2, &pdf_t, ia->inside, flux);
    double R = ia->R, T = ia->T;
    if (wi_dirac_delta) {
        if (ia->type == REFLECT) { *pdf = R; return vscale(brdf_s, F); }
        *pdf = T * (1.0 - R);
        return vscale(vscale(btdf, T), 1.0 - F);
    } else if (!(f & MCRT_MAT_ROUGH_SPECULAR)) {
        *pdf = pdf_d * (1.0 - R) * (1.0 - T);
        return vscale(vscale(brdf_d, 1.0 - F), 1.0 - T);
    }
    *pdf = smix(smix(pdf_d, pdf_t, T), pdf_s, R);
    return vmix(vmix(brdf_d, btdf, T), brdf_s, F);
}

static int iaBSDFWorld(const Interaction* ia, v3* bsdf_absIdotN, v3 world_wi, double* pdf) { /* interaction.cpp:74-82 */
    v3 wi = csTo(&ia->shading_cs, world_wi);
    v3 wo = csTo(&ia->shading_cs, ia->out);
    *bsdf_absIdotN = vscale(iaBSDFLocal(ia, wo, wi, pdf, 0, 0), fabs(wi.z));
    return *pdf > 0.0;
}

static v3 iaSpecularNormal(const Interaction* ia, const Sampler* smp) { /* interaction.cpp:185-193 */
    if (ia->material->flags & MCRT_MAT_ROUGH_SPECULAR) {
        double u0 = samplerGet(smp, DIM_BSDF), u1 = samplerGet(smp, DIM_BSDF + 1);
        return csFrom(&ia->shading_cs, ggxVisibleMicrofacet(u0, u1, csTo(&ia->shading_cs, ia->out), ia->material->a));
    }
    return ia->shading_cs.c2;
}

static v3 cosWeightedHemi(double u, double v) { /* sampling/sampling.hpp:35-44 */
    double r = sqrt(u);
    double azimuth = v * TWO_PI;
    return V(r * cos(azimuth), r * sin(azimuth), sqrt(1 - u));
}

static Ray rayFromInteraction(const Interaction* ia, const Sampler* smp) { /* ray/ray.cpp:16-67 */
    Ray r; memset(&r, 0, sizeof(r));
    r.depth = (uint16_t)(ia->ray.depth + 1); r.diffuse_depth = ia->ray.diffuse_depth;
    r.refraction_scale = ia->ray.refraction_scale; r.start = ia->position;
    r.refraction_level = ia->ray.refraction_level; r.dirac_delta = ia->dirac_delta; r.refraction = 0;
    switch (ia->type) {
    case REFLECT: {
        v3 sn = iaSpecularNormal(ia, smp);
        r.direction = vsub(ia->ray.direction, vscale(vscale(sn, vdot(sn, ia->ray.direction)), 2.0)); /* glm::reflect */
        r.medium_ior = ia->n1;
        r.start = vadd(r.start, vscale(ia->normal, EPSILON));
        break;
    }
    case REFRACT: {
        v3 sn = iaSpecularNormal(ia, smp);
        double inv_eta = ia->n1 / ia->n2;
        double cos_theta = vdot(sn, ia->ray.direction);
        double k = 1.0 - pow2(inv_eta) * (1.0 - pow2(cos_theta));
        if (k >= 0.0) {
            r.direction = vsub(vscale(ia->ray.direction, inv_eta), vscale(sn, inv_eta * cos_theta + sqrt(k)));
            r.medium_ior = ia->n2;
            r.start = vsub(r.start, vscale(ia->normal, EPSILON));
            if (ia->inside) r.refraction_level--; else r.refraction_level++;
            r.refraction_scale *= pow2(1.0 / inv_eta);
            r.refraction = 1;
        } else {
            r.direction = vsub(ia->ray.direction, vscale(vscale(sn, cos_theta), 2.0));
            r.medium_ior = ia->n1;
            r.start = vadd(r.start, vscale(ia->normal, EPSILON));
        }
        break;
    }
    default: {
        r.diffuse_depth++;
        double u0 = samplerGet(smp, DIM_BSDF), u1 = samplerGet(smp, DIM_BSDF + 1);
        r.direction = csFrom(&ia->shading_cs, cosWeightedHemi(u0, u1));
        r.medium_ior = ia->n1;
        r.start = vadd(r.start, vscale(ia->normal, EPSILON));
        break;
    }
    }
    r.inv_direction = V(1.0 / r.direction.x, 1.0 / r.direction.y, 1.0 / r.direction.z);
    return r;
}

static int iaSampleBSDF(const Interaction* ia, v3* bsdf_absIdotN, double* pdf, Ray* new_ray, int flux, const Sampler* smp) { /* interaction.cpp:56-72 */
    *new_ray = rayFromInteraction(ia, smp);
    v3 wi = csTo(&ia->shading_cs, new_ray->direction);
    if ((new_ray->refraction && wi.z >= 0.0) || (!new_ray->refraction && wi.z <= 0.0)) return 0;
    v3 wo = csTo(&ia->shading_cs, ia->out);
    *bsdf_absIdotN = vscale(iaBSDFLocal(ia, wo, wi, pdf, flux, new_ray->dirac_delta), fabs(wi.z));
    return *pdf > 0.0;
}

/* ------------------------------------------------------------------ Integrator (integrator/integrator.cpp) */
typedef struct { double bsdf_pdf, select_probability; uint32_t light; } LightSample; /* integrator.hpp:14-18 */

typedef struct {
    SceneRef S;
    const mcrt_photon_map_desc* maps[2];
    uint32_t k_nearest; int direct_visualization;
    ThreadScratch* ts;
} Ctx;

static double powerHeuristic(double a_pdf, double b_pdf) { double a2 = a_pdf * a_pdf; return a2 / (a2 + b_pdf * b_pdf); } /* util.hpp:85-89 */

static uint32_t selectLight(const mcrt_scene_desc* s, double u, double* select_probability) { /* scene.cpp:225-236, sampling.hpp:13-27 */
    size_t left = 0, right = (size_t)s->num_lights - 1;
    while (left < right) {
        size_t middle = (left + right) / 2;
        if (s->light_cdf[middle] < u) left = middle + 1; else right = middle;
    }
    *select_probability = s->light_cdf[left];
    if (left > 0) *select_probability -= s->light_cdf[left - 1];
    return s->light_surface[left];
}

static v3 sampleDirect(Ctx* C, const Interaction* ia, LightSample* ls, const Sampler* smp) { /* integrator.cpp:31-87 */
    const mcrt_scene_desc* s = C->S.s;
    if (s->num_lights == 0 || (ia->material->flags & MCRT_MAT_DIRAC_DELTA)) { ls->light = NO_SURFACE; return V(0, 0, 0); }
    double u0 = samplerGet(smp, DIM_LIGHT), u1 = samplerGet(smp, DIM_LIGHT + 1), u2 = samplerGet(smp, DIM_LIGHT + 2);
    ls->light = selectLight(s, u2, &ls->select_probability);
    v3 light_pos = surfSample(s, ls->light, u0, u1);
    Ray shadow_ray = rayTo(vadd(ia->position, vscale(ia->normal, EPSILON)), light_pos);
    double cos_light_theta = vdot(vneg(shadow_ray.direction), surfNormal(s, ls->light, light_pos));
    if (cos_light_theta <= 0.0) return V(0, 0, 0);
    double cos_theta = vdot(shadow_ray.direction, ia->normal);
    if (cos_theta <= 0.0) {
        if ((ia->material->flags & MCRT_MAT_OPAQUE) || cos_theta == 0.0) return V(0, 0, 0);
        shadow_ray = rayTo(vsub(ia->position, vscale(ia->normal, EPSILON)), light_pos);
    }
    Hit sh = sceneIntersect(&C->S, &shadow_ray, C->ts);
    if (sh.surface == NO_SURFACE || sh.surface != ls->light) return V(0, 0, 0);
    double light_pdf = pow2(sh.t) / (s->surf_area[ls->light] * cos_light_theta);
    double bsdf_pdf;
    v3 bsdf_absIdotN;
    if (!iaBSDFWorld(ia, &bsdf_absIdotN, shadow_ray.direction, &bsdf_pdf)) return V(0, 0, 0);
    double mis_weight = powerHeuristic(light_pdf, bsdf_pdf);
    const mcrt_material* lm = &s->materials[s->surf_material[ls->light]];
    return vdivs(vmul(vscale(bsdf_absIdotN, mis_weight), ld3(lm->emittance)), light_pdf * ls->select_probability);
}

static v3 sampleEmissive(Ctx* C, const Interaction* ia, const LightSample* ls) { /* integrator.cpp:93-110 */
    if ((ia->material->flags & MCRT_MAT_EMISSIVE) && !ia->inside) {
        if (ia->ray.depth == 0 || ia->ray.dirac_delta) return ld3(ia->material->emittance);
        if (ls->light == ia->surface) {
            double cos_light_theta = vdot(ia->out, ia->normal);
            double light_pdf = pow2(ia->t) / (C->S.s->surf_area[ia->surface] * cos_light_theta);
            double mis_weight = powerHeuristic(ls->bsdf_pdf, light_pdf);
            return vdivs(vscale(ld3(ia->material->emittance), mis_weight), ls->select_probability);
        }
    }
    return V(0, 0, 0);
}

static int absorb(const Ray* ray, v3* throughput, const Sampler* smp) { /* integrator.cpp:112-129, integrator.hpp:28-29 */
    double survive = compMax(*throughput) * ray->refraction_scale;
    if (survive == 0.0) return 1;
    if (ray->diffuse_depth > 3 || ray->depth > 16) {
        survive = smin(0.95, survive);
        if (survive <= samplerGet(smp, DIM_ABSORB)) return 1;
        *throughput = vdivs(*throughput, survive);
    }
    return 0;
}

static v3 skyColor(const Ray* ray) { /* scene.cpp:219-223 */
    double fy = (1.0 + asin(vdot(V(0.0, 1.0, 0.0), ray->direction)) / PI) / 2.0;
    return vmix(V(1.0, 0.5, 0.0), V(0.0, 0.5, 1.0), fy);
}

static v3 pathTracerSampleRay(Ctx* C, Ray ray, Sampler* smp) { /* integrator/path-tracer/path-tracer.cpp:14-51 */
    const mcrt_scene_desc* s = C->S.s;
    v3 radiance = V(0, 0, 0), throughput = V(1, 1, 1);
    RefractionHistory rh; rhInit(&rh, &ray);
    v3 bsdf_absIdotN;
    LightSample ls = {0.0, 0.0, NO_SURFACE};
    for (;;) {
        samplerShuffle(smp);
        Hit isect = sceneIntersect(&C->S, &ray, C->ts);
        if (isect.surface == NO_SURFACE) return vadd(radiance, vmul(skyColor(&ray), throughput));
        Interaction ia;
        iaInit(&ia, s, &isect, &ray, rhExternalIOR(&rh, &ray), smp);
        radiance = vadd(radiance, vmul(sampleEmissive(C, &ia, &ls), throughput));
        radiance = vadd(radiance, vmul(sampleDirect(C, &ia, &ls, smp), throughput));
        if (!iaSampleBSDF(&ia, &bsdf_absIdotN, &ls.bsdf_pdf, &ray, 0, smp)) return radiance;
        throughput = vmul(throughput, vdivs(bsdf_absIdotN, ls.bsdf_pdf));
        if (absorb(&ray, &throughput, smp)) return radiance;
        rhUpdate(&rh, &ray);
    }
}

/* ------------------------------------------------------------------ photon map kNN (octree/linear-octree.cpp) */
typedef struct { double distance2; uint32_t index; } KnnResult;   /* SearchResult<Photon>: a < b <=> a.distance2 < b.distance2 */
typedef struct { KnnResult* H; size_t size, cap; } KnnHeap;       /* PriorityQueue<SearchResult<Photon>> */
typedef struct { double distance2; uint32_t octant; } DNode;      /* linear-octree.cpp:37-42: a < b <=> b.distance2 < a.distance2 */
typedef struct { DNode* H; size_t size, cap; } DQueue;

static inline int krLess(KnnResult a, KnnResult b) { return a.distance2 < b.distance2; }
static void khReserve(KnnHeap* q) { if (q->size == q->cap) { q->cap = q->cap ? q->cap * 2 : 64; q->H = (KnnResult*)realloc(q->H, q->cap * sizeof(KnnResult)); } }
static void khPushUnordered(KnnHeap* q, KnnResult v) { khReserve(q); q->H[q->size++] = v; }
static void khShiftDown(KnnHeap* q, KnnResult value, size_t index) { /* priority-queue.hpp:103-123 */
    for (;;) {
        size_t left = 2 * index + 1, right = left + 1, max_child;
        if (right < q->size) max_child = left + (size_t)krLess(q->H[left], q->H[right]);
        else if (left < q->size) max_child = left;
        else break;
        if (!krLess(value, q->H[max_child])) break;
        q->H[index] = q->H[max_child];
        index = max_child;
    }
    q->H[index] = value;
}
static void khMakeHeap(KnnHeap* q) { /* priority-queue.hpp:57-84 */
    if (q->size <= 1) return;
    const size_t last_index = q->size - 1;
    size_t index = (last_index - 1) / 2;
    if (last_index % 2) {
        size_t left = 2 * index + 1;
        if (krLess(q->H[index], q->H[left])) { KnnResult t = q->H[index]; q->H[index] = q->H[left]; q->H[left] = t; }
        if (index == 0) return;
        index--;
    }
    if (index) {
        size_t lowest_index_with_no_grandchildren = (last_index - 3) / 4 + 1;
        do {
            size_t left = 2 * index + 1;
            size_t max_child = left + (size_t)krLess(q->H[left], q->H[left + 1]);
            if (krLess(q->H[index], q->H[max_child])) { KnnResult t = q->H[index]; q->H[index] = q->H[max_child]; q->H[max_child] = t; }
        } while (index-- != lowest_index_with_no_grandchildren);
    }
    do { khShiftDown(q, q->H[index], index); } while (index--);
}

static inline int dnLess(DNode a, DNode b) { return b.distance2 < a.distance2; }
static void dqPush(DQueue* q, DNode value) {
    if (q->size == q->cap) { q->cap = q->cap ? q->cap * 2 : 64; q->H = (DNode*)realloc(q->H, q->cap * sizeof(DNode)); }
    size_t index = q->size++;
    while (index > 0) {
        size_t parent = (index - 1) / 2;
        if (!dnLess(q->H[parent], value)) break;
        q->H[index] = q->H[parent];
        index = parent;
    }
    q->H[index] = value;
}
static void dqPop(DQueue* q) {
    if (q->size > 1) {
        DNode value = q->H[--q->size];
        size_t index = 0;
        for (;;) {
            size_t left = 2 * index + 1, right = left + 1, max_child;
            if (right < q->size) max_child = left + (size_t)dnLess(q->H[left], q->H[right]);
            else if (left < q->size) max_child = left;
            else break;
            if (!dnLess(value, q->H[max_child])) break;
            q->H[index] = q->H[max_child];
            index = max_child;
        }
        q->H[index] = value;
    } else q->size--;
}

static double bbDistance2(const double* b, v3 p) { /* common/bounding-box.cpp:43-47 */
    v3 mn = ld3(b), mx = ld3(b + 3);
    v3 a = vsub(mn, p), c = vsub(p, mx);
    v3 d = V(gmax(gmax(a.x, c.x), 0.0), gmax(gmax(a.y, c.y), 0.0), gmax(gmax(a.z, c.z), 0.0));
    return vdot(d, d);
}
static double bbMaxDistance2(const double* b, v3 p) { /* common/bounding-box.cpp:50-54 */
    v3 mn = ld3(b), mx = ld3(b + 3);
    v3 a = vsub(mx, p), c = vsub(p, mn);
    v3 d = V(gmax(a.x, c.x), gmax(a.y, c.y), gmax(a.z, c.z));
    return vdot(d, d);
}
static v3 photonPos(const mcrt_photon_map_desc* m, uint64_t i) { const float* p = m->photons + 8 * i; return V((double)p[3], (double)p[4], (double)p[5]); } /* photon.hpp:14-17 */

/* Study hooks (tools/knn_hint_study.py; single-threaded use only): a recorder of the searches of a render, and an initial
 * bound for a search (DBL_MAX = the reference's). Neither is used by any parity test's oracle path. */
static double* g_knn_rec = NULL; static uint64_t g_knn_rec_cap = 0, g_knn_rec_n = 0; static double g_knn_rec_tag[2] = {0, 0};
void oracle_knn_recorder(double* buf, uint64_t cap) { g_knn_rec = buf; g_knn_rec_cap = cap; g_knn_rec_n = 0; }
uint64_t oracle_knn_recorded(void) { return g_knn_rec_n; }
static double g_knn_initial_bound2 = DBL_MAX;

static void knnSearch(const mcrt_photon_map_desc* m, v3 p, size_t k, KnnHeap* result, DQueue* to_visit, oracle_counters* cnt) { /* linear-octree.cpp:25-117 */
    result->size = 0;
    if (!m || m->num_octants == 0) return;
    if (k > m->num_photons) k = (size_t)m->num_photons;
    double max_distance2 = g_knn_initial_bound2;
    to_visit->size = 0;
    DNode current = {bbDistance2(m->octant_bounds, p), 0};
    for (;;) {
        uint32_t oc = current.octant;
        if (cnt) cnt->knn_octants++;
        if (m->octant_leaf[oc] || m->octant_contained_data[oc] <= k) {
            uint64_t start = m->octant_start_data[oc], end_idx = start + m->octant_contained_data[oc];
            for (uint64_t i = start; i < end_idx; i++) {
                v3 d = vsub(p, photonPos(m, i)); /* glm::distance2(data.pos(), p) = length2(p - pos) */
                double distance2 = vdot(d, d);
                if (cnt) cnt->knn_photons++;
                if (distance2 <= max_distance2) {
                    KnnResult r = {distance2, (uint32_t)i};
                    if (result->size < k - 1) khPushUnordered(result, r);
                    else {
                        if (result->size != k) { khPushUnordered(result, r); khMakeHeap(result); }
                        else khShiftDown(result, r, 0); /* pop_push */
                        if (result->H[0].distance2 < max_distance2) max_distance2 = result->H[0].distance2;
                    }
                }
            }
        } else {
            uint32_t child = oc + 1;
            while (child != 0xFFFFFFFFu) {
                double distance2 = bbDistance2(m->octant_bounds + 6 * (size_t)child, p);
                if (distance2 <= max_distance2) {
                    DNode dn = {distance2, child};
                    dqPush(to_visit, dn);
                    if (m->octant_contained_data[child] >= k) {
                        double md = bbMaxDistance2(m->octant_bounds + 6 * (size_t)child, p);
                        if (md < max_distance2) max_distance2 = md;
                    }
                }
                child = m->octant_next_sibling[child];
            }
        }
        if (to_visit->size == 0) break;
        current = to_visit->H[0];
        if (current.distance2 > max_distance2) break;
        dqPop(to_visit);
    }
}

static int cmpKnn(const void* a, const void* b) {
    const KnnResult* x = (const KnnResult*)a; const KnnResult* y = (const KnnResult*)b;
    if (x->distance2 < y->distance2) return -1;
    if (x->distance2 > y->distance2) return 1;
    return x->index < y->index ? -1 : (x->index > y->index ? 1 : 0);
}

/* Study: searches with a caller-given initial bound (squared; DBL_MAX = none) and the octants / photons they touch. */
void oracle_knn_hinted(const mcrt_photon_map_desc* map, uint64_t n, const double* p, uint32_t k, const double* bound2,
                       double* out_kth_distance2, uint32_t* out_count, uint64_t* octants, uint64_t* photons) {
    KnnHeap res = {0, 0, 0}; DQueue dq = {0, 0, 0};
    oracle_counters c; memset(&c, 0, sizeof c);
    for (uint64_t i = 0; i < n; i++) {
        g_knn_initial_bound2 = bound2 ? bound2[i] : DBL_MAX;
        knnSearch(map, ld3(p + 3 * i), k, &res, &dq, &c);
        out_count[i] = (uint32_t)res.size;
        out_kth_distance2[i] = res.size ? res.H[0].distance2 : INFINITY; /* heap top = the farthest kept */
    }
    g_knn_initial_bound2 = DBL_MAX;
    *octants = c.knn_octants; *photons = c.knn_photons;
    free(res.H); free(dq.H);
}

void oracle_knn(const mcrt_photon_map_desc* map, uint64_t n, const double* p, uint32_t k,
                uint32_t* out_count, uint32_t* out_index, double* out_distance2) {
    KnnHeap res = {0, 0, 0}; DQueue dq = {0, 0, 0};
    for (uint64_t i = 0; i < n; i++) {
        knnSearch(map, ld3(p + 3 * i), k, &res, &dq, NULL);
        qsort(res.H, res.size, sizeof(KnnResult), cmpKnn);
        out_count[i] = (uint32_t)res.size;
        for (uint32_t q = 0; q < k; q++) {
            out_index[i * k + q] = q < res.size ? res.H[q].index : 0xFFFFFFFFu;
            out_distance2[i * k + q] = q < res.size ? res.H[q].distance2 : INFINITY;
        }
    }
    free(res.H); free(dq.H);
}

/* ------------------------------------------------------------------ PhotonMapper eye pass (photon-mapper.cpp:279-391) */
static v3 photonDir(const float* ph) { /* photon.hpp:19-27: float overloads of sin/cos */
    float phi = ph[6], theta = ph[7];
    double sin_theta = sinf(theta);
    return V(sin_theta * cosf(phi), sin_theta * sinf(phi), cosf(theta));
}
static v3 photonFlux(const float* ph) { return V((double)ph[0], (double)ph[1], (double)ph[2]); }

static v3 estimateGlobalRadiance(Ctx* C, const Interaction* ia) { /* :343-363 */
    const mcrt_photon_map_desc* m = C->maps[0];
    KnnHeap* photons = (KnnHeap*)C->ts->knn_result[0];
    if (C->S.c) C->S.c->knn_searches++;
    if (g_knn_rec && g_knn_rec_n < g_knn_rec_cap) { double* r = g_knn_rec + 6 * g_knn_rec_n++; r[0] = 0; r[1] = ia->position.x; r[2] = ia->position.y; r[3] = ia->position.z; r[4] = g_knn_rec_tag[0]; r[5] = g_knn_rec_tag[1]; }
    knnSearch(m, ia->position, C->k_nearest, photons, (DQueue*)C->ts->knn_visit, C->S.c);
    if (photons->size == 0) return V(0, 0, 0);
    double bsdf_pdf; v3 bsdf_absIdotN; v3 radiance = V(0, 0, 0);
    for (size_t i = 0; i < photons->size; i++) {
        const float* ph = m->photons + 8 * (size_t)photons->H[i].index;
        if (iaBSDFWorld(ia, &bsdf_absIdotN, photonDir(ph), &bsdf_pdf))
            radiance = vadd(radiance, vdivs(vmul(photonFlux(ph), bsdf_absIdotN), bsdf_pdf));
    }
    return vdivs(radiance, photons->H[0].distance2 * PI);
}

static v3 estimateCausticRadiance(Ctx* C, const Interaction* ia) { /* :368-391 */
    const mcrt_photon_map_desc* m = C->maps[1];
    KnnHeap* photons = (KnnHeap*)C->ts->knn_result[1];
    if (C->S.c) C->S.c->knn_searches++;
    if (g_knn_rec && g_knn_rec_n < g_knn_rec_cap) { double* r = g_knn_rec + 6 * g_knn_rec_n++; r[0] = 1; r[1] = ia->position.x; r[2] = ia->position.y; r[3] = ia->position.z; r[4] = g_knn_rec_tag[0]; r[5] = g_knn_rec_tag[1]; }
    knnSearch(m, ia->position, C->k_nearest, photons, (DQueue*)C->ts->knn_visit, C->S.c);
    if (photons->size == 0) return V(0, 0, 0);
    double inv_max_squared_radius = 1.0 / photons->H[0].distance2;
    double bsdf_pdf; v3 bsdf_absIdotN; v3 radiance = V(0, 0, 0);
    for (size_t i = 0; i < photons->size; i++) {
        const float* ph = m->photons + 8 * (size_t)photons->H[i].index;
        if (iaBSDFWorld(ia, &bsdf_absIdotN, photonDir(ph), &bsdf_pdf)) {
            double wp = gmax(0.0, 1.0 - sqrt(photons->H[i].distance2 * inv_max_squared_radius));
            radiance = vadd(radiance, vdivs(vscale(vmul(photonFlux(ph), bsdf_absIdotN), wp), bsdf_pdf));
        }
    }
    return vscale(vscale(vscale(radiance, 3.0), inv_max_squared_radius), INV_PI);
}

static v3 photonMapperSampleRay(Ctx* C, Ray ray, Sampler* smp) { /* :279-341 */
    const mcrt_scene_desc* s = C->S.s;
    v3 radiance = V(0, 0, 0), throughput = V(1, 1, 1);
    RefractionHistory rh; rhInit(&rh, &ray);
    v3 bsdf_absIdotN;
    LightSample ls = {0.0, 0.0, NO_SURFACE};
    for (;;) {
        samplerShuffle(smp);
        Hit isect = sceneIntersect(&C->S, &ray, C->ts);
        if (isect.surface == NO_SURFACE) return radiance;
        Interaction ia;
        iaInit(&ia, s, &isect, &ray, rhExternalIOR(&rh, &ray), smp);
        radiance = vadd(radiance, vmul(sampleEmissive(C, &ia, &ls), throughput));
        if (ia.dirac_delta) {
            if (!ray.dirac_delta && ray.depth != 0) return radiance;
            if (!iaSampleBSDF(&ia, &bsdf_absIdotN, &ls.bsdf_pdf, &ray, 0, smp)) return radiance;
            throughput = vmul(throughput, vdivs(bsdf_absIdotN, ls.bsdf_pdf));
        } else {
            radiance = vadd(radiance, vmul(estimateCausticRadiance(C, &ia), throughput));
            if (!C->direct_visualization && (ray.dirac_delta || ray.depth == 0)) {
                radiance = vadd(radiance, vmul(sampleDirect(C, &ia, &ls, smp), throughput));
                if (!iaSampleBSDF(&ia, &bsdf_absIdotN, &ls.bsdf_pdf, &ray, 0, smp)) return radiance;
                throughput = vmul(throughput, vdivs(bsdf_absIdotN, ls.bsdf_pdf));
            } else {
                return vadd(radiance, vmul(estimateGlobalRadiance(C, &ia), throughput));
            }
        }
        if (absorb(&ray, &throughput, smp)) return radiance;
        rhUpdate(&rh, &ray);
    }
}

/* ------------------------------------------------------------------ Camera::samplePixel + Film (camera/camera.cpp:66-99) */
/* ---- Film with a reconstruction filter (camera/film.cpp:19-113, camera/filter.hpp). The default box film needs no
 * buffer (a sample lands in its own pixel with weight 1); every other filter splats into job->film. */
static double mitchellNetravali(double B, double C_, double x) { /* filter.hpp:16-40 */
    double k = 6.0 / (6.0 - 2.0 * B);
    if (x < 1.0) {
        double a = k * (12.0 - 9.0 * B - 6.0 * C_) / 6.0, b = k * (-18.0 + 12.0 * B + 6.0 * C_) / 6.0, d = k * (6.0 - 2.0 * B) / 6.0;
        return d + (b + a * x) * x * x;
    }
    double a = k * (-B - 6.0 * C_) / 6.0, b = k * (6.0 * B + 30.0 * C_) / 6.0, c = k * (-12.0 * B - 48.0 * C_) / 6.0, d = k * (8.0 * B + 24.0 * C_) / 6.0;
    return d + (c + (b + a * x) * x) * x;
}
static double filmFilterFunction(uint32_t type, double x) {
    switch (type) {
        case MCRT_FILM_MITCHELL_NETRAVALI: return mitchellNetravali(1.0 / 3.0, 1.0 / 3.0, x);
        case MCRT_FILM_CATMULL_ROM: return mitchellNetravali(0.0, 0.5, x);
        case MCRT_FILM_B_SPLINE: return mitchellNetravali(1.0, 0.0, x);
        case MCRT_FILM_HERMITE: return mitchellNetravali(0.0, 0.0, x * 0.5);
        case MCRT_FILM_GAUSSIAN: return exp(-2.0 * x * x) - exp(-2.0 * 2.0 * 2.0);
        case MCRT_FILM_LANCZOS: return x == 0.0 ? 1.0 : 2.0 * sin(PI * x) * sin(PI * x / 2.0) / (PI * PI * x * x);
        default: return 1.0;
    }
}
typedef struct { uint32_t type, cache_size, width, height; double radius, two_inv_radius, inv_dx; double* cache; double* blob; } Film;
static double filmFilter(const Film* f, double x) { /* film.cpp:86-97 */
    if (f->cache_size == 0) return filmFilterFunction(f->type, f->two_inv_radius * fabs(x));
    return f->cache[(size_t)(f->inv_dx * fabs(x) + 0.5)];
}
static void filmDeposit(Film* f, double px, double py, v3 v) { /* film.cpp:61-79 */
    long long min_x = (long long)(px + 0.5 - f->radius), min_y = (long long)(py + 0.5 - f->radius);
    long long max_x = (long long)(px - 0.5 + f->radius), max_y = (long long)(py - 0.5 + f->radius);
    if (min_x < 0) min_x = 0;
    if (min_y < 0) min_y = 0;
    if (max_x > (long long)f->width - 1) max_x = (long long)f->width - 1;
    if (max_y > (long long)f->height - 1) max_y = (long long)f->height - 1;
    for (long long y = min_y; y <= max_y; y++) {
        double weight_y = filmFilter(f, (double)y + 0.5 - py);
        for (long long x = min_x; x <= max_x; x++) {
            double weight = weight_y * filmFilter(f, (double)x + 0.5 - px);
            double* s = f->blob + ((size_t)y * f->width + (size_t)x) * 4;
            s[0] += v.x * weight; s[1] += v.y * weight; s[2] += v.z * weight; s[3] += weight;
        }
    }
}
static double filmDefaultRadius(uint32_t type) { /* film.cpp:31-44 */
    switch (type) {
        case MCRT_FILM_MITCHELL_NETRAVALI: case MCRT_FILM_CATMULL_ROM: case MCRT_FILM_LANCZOS: return 2.0;
        case MCRT_FILM_B_SPLINE: return 1.39;
        case MCRT_FILM_HERMITE: return 1.0;
        case MCRT_FILM_GAUSSIAN: return 1.71;
        default: return 0.5;
    }
}

typedef struct {
    const mcrt_scene_desc* scene; const mcrt_photon_map_desc* maps[2];
    uint32_t k_nearest; int direct_visualization;
    const mcrt_camera_desc* cam; uint32_t global_seed; int integrator;
    uint32_t row0, row1;
    double* out_rgb; double* out_samples;
    Film* film; /* non-box reconstruction filter: splats (single-threaded) */
    volatile uint32_t next_row;
    oracle_counters total; pthread_mutex_t lock;
} Job;

static void samplePixel(Job* job, Ctx* C, uint32_t x, uint32_t y) {
    const mcrt_camera_desc* cam = job->cam;
    double pixel_size = cam->sensor_width / (double)cam->width;
    size_t spp = (size_t)cam->sqrtspp * cam->sqrtspp;
    double half_x = (double)cam->width * 0.5, half_y = (double)cam->height * 0.5;
    v3 forward = ld3(cam->forward), left = ld3(cam->left), up = ld3(cam->up), eye = ld3(cam->eye);
    Sampler smp; memset(&smp, 0, sizeof(smp)); smp.global_seed = job->global_seed;
    samplerInitiate(&smp, (uint32_t)((size_t)y * cam->width + x));
    double sum[3] = {0.0, 0.0, 0.0}, weight_sum = 0.0; /* Film::Splat, film.cpp:99-113 */
    for (size_t i = 0; i < spp; i++) {
        samplerSetIndex(&smp, (uint32_t)i);
        double u0 = samplerGet(&smp, DIM_PIXEL), u1 = samplerGet(&smp, DIM_PIXEL + 1);
        double px = (double)x + u0, py = (double)y + u1;
        double lx = pixel_size * (half_x - px), ly = pixel_size * (half_y - py);
        v3 direction = vnormalize(vadd(vadd(vscale(forward, cam->focal_length), vscale(left, lx)), vscale(up, ly)));
        Ray ray = rayDir(eye, direction, job->scene->scene_ior);
        if (cam->thin_lens) {
            double l0 = samplerGet(&smp, DIM_LENS), l1 = samplerGet(&smp, DIM_LENS + 1);
            double azimuth = l1 * TWO_PI; /* Sampling::uniformDisk, sampling.hpp:29-33 */
            double su = sqrt(l0);
            double ax = cos(azimuth) * su * cam->aperture_radius, ay = sin(azimuth) * su * cam->aperture_radius;
            v3 focus_point = rayAt(&ray, cam->focus_distance / vdot(ray.direction, forward));
            v3 start = vadd(vadd(eye, vscale(left, ax)), vscale(up, ay));
            ray = rayDir(start, vnormalize(vsub(focus_point, start)), job->scene->scene_ior);
        }
        if (C->S.c) C->S.c->paths++;
        if (g_knn_rec) { g_knn_rec_tag[0] = (double)((size_t)y * cam->width + x); g_knn_rec_tag[1] = (double)i; }
        v3 L = job->integrator == MCRT_INTEGRATOR_PHOTON_MAPPER ? photonMapperSampleRay(C, ray, &smp)
                                                                : pathTracerSampleRay(C, ray, &smp);
        /* Film::deposit with the default box filter, radius 0.5: the sample's own pixel, weight 1 (film.cpp:13-17,61-79) */
        if (job->film) filmDeposit(job->film, px, py, L);
        sum[0] += L.x * 1.0; sum[1] += L.y * 1.0; sum[2] += L.z * 1.0; weight_sum += 1.0;
        if (job->out_samples) {
            double* o = job->out_samples + ((((size_t)(y - job->row0) * cam->width + x) * spp + i) * 3);
            o[0] = L.x; o[1] = L.y; o[2] = L.z;
        }
    }
    if (job->film) return; /* the frame is read from the film when every sample has been deposited */
    double* o = job->out_rgb + ((size_t)(y - job->row0) * cam->width + x) * 3;
    for (int c = 0; c < 3; c++) o[c] = weight_sum == 0.0 ? 0.0 : gmax(sum[c] / weight_sum, 0.0); /* Splat::get, film.cpp:107-113 */
}

static void* worker(void* arg) {
    Job* job = (Job*)arg;
    ThreadScratch ts; memset(&ts, 0, sizeof(ts));
    KnnHeap kh[2] = {{0, 0, 0}, {0, 0, 0}}; DQueue dq = {0, 0, 0};
    ts.knn_result[0] = &kh[0]; ts.knn_result[1] = &kh[1]; ts.knn_visit = &dq;
    oracle_counters cnt; memset(&cnt, 0, sizeof(cnt));
    Ctx C; C.S.s = job->scene; C.S.c = &cnt; C.maps[0] = job->maps[0]; C.maps[1] = job->maps[1];
    C.k_nearest = job->k_nearest; C.direct_visualization = job->direct_visualization; C.ts = &ts;
    for (;;) {
        uint32_t y = __atomic_fetch_add(&job->next_row, 1u, __ATOMIC_RELAXED);
        if (y >= job->row1) break;
        for (uint32_t x = 0; x < job->cam->width; x++) samplePixel(job, &C, x, y);
    }
    pthread_mutex_lock(&job->lock);
    job->total.paths += cnt.paths; job->total.rays += cnt.rays; job->total.node_tests += cnt.node_tests;
    job->total.prim_tests += cnt.prim_tests; job->total.knn_searches += cnt.knn_searches;
    job->total.knn_octants += cnt.knn_octants; job->total.knn_photons += cnt.knn_photons;
    job->total.sphere_tests += cnt.sphere_tests;
    pthread_mutex_unlock(&job->lock);
    free(ts.to_visit.H); free(kh[0].H); free(kh[1].H); free(dq.H);
    return NULL;
}

int oracle_hardware_threads(void) { long n = sysconf(_SC_NPROCESSORS_ONLN); return n > 0 ? (int)n : 1; }

int oracle_render(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* global_map,
                  const mcrt_photon_map_desc* caustic_map, uint32_t k_nearest, int direct_visualization,
                  const mcrt_camera_desc* cam, uint32_t global_seed, int integrator,
                  uint32_t row0, uint32_t row1, int threads, double* out_rgb, double* out_samples,
                  oracle_counters* counters, double* seconds) {
    if (!scene || !cam || !out_rgb || row1 > cam->height || row0 > row1) return -1;
    pthread_once(&g_dirs_once, initDirections);
    Job job; memset(&job, 0, sizeof(job));
    job.scene = scene; job.maps[0] = global_map; job.maps[1] = caustic_map;
    job.k_nearest = k_nearest; job.direct_visualization = direct_visualization;
    job.cam = cam; job.global_seed = global_seed; job.integrator = integrator;
    job.row0 = row0; job.row1 = row1; job.out_rgb = out_rgb; job.out_samples = out_samples;
    job.next_row = row0;
    pthread_mutex_init(&job.lock, NULL);
    if (threads <= 0) threads = oracle_hardware_threads();
    if (threads > 1024) threads = 1024;
    Film film; memset(&film, 0, sizeof(film));
    /* splats for every filter but the default box; a box with another radius than 0.5 splats too (film.cpp:44-46: the radius is
       read after the filter is chosen, Filter::box is 1 everywhere) */
    if (cam->film_filter != MCRT_FILM_BOX || (cam->film_radius != 0.0 && cam->film_radius != 0.5)) { /* Film::Film(width, height, json), film.cpp:19-58 */
        film.type = cam->film_filter; film.width = cam->width; film.height = cam->height;
        film.radius = cam->film_radius > 0.0 ? cam->film_radius : filmDefaultRadius(cam->film_filter);
        film.two_inv_radius = 2.0 / film.radius;
        film.cache_size = cam->film_cache_size;
        if (film.cache_size) {
            film.cache = (double*)malloc(sizeof(double) * film.cache_size);
            for (uint32_t i = 0; i < film.cache_size; i++) film.cache[i] = filmFilterFunction(film.type, (2.0 * (int)i) / (double)(film.cache_size - 1));
            film.inv_dx = (double)(film.cache_size - 1) / film.radius;
        }
        film.blob = (double*)calloc((size_t)cam->width * cam->height * 4, sizeof(double));
        job.film = &film;
        threads = 1; /* plain adds into shared splats; the reference's atomics make its own sums order dependent too */
    }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, &job);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    pthread_mutex_destroy(&job.lock);
    if (job.film) { /* Film::scan of the rendered rows, film.cpp:81-84,107-113 */
        for (uint32_t y = row0; y < row1; y++)
            for (uint32_t x = 0; x < cam->width; x++) {
                const double* s = film.blob + ((size_t)y * cam->width + x) * 4;
                double* o2 = out_rgb + ((size_t)(y - row0) * cam->width + x) * 3;
                for (int c = 0; c < 3; c++) o2[c] = s[3] == 0.0 ? 0.0 : gmax(s[c] / s[3], 0.0);
            }
        free(film.blob); free(film.cache);
    }
    if (counters) *counters = job.total;
    if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return 0;
}

void oracle_intersect(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction,
                      double* out_t, uint32_t* out_surface, double* out_uv, oracle_counters* counters) {
    ThreadScratch ts; memset(&ts, 0, sizeof(ts));
    oracle_counters cnt; memset(&cnt, 0, sizeof(cnt));
    SceneRef S = {scene, &cnt};
    for (uint64_t i = 0; i < n; i++) {
        Ray ray = rayDir(ld3(start + 3 * i), ld3(direction + 3 * i), 1.0);
        Hit h = sceneIntersect(&S, &ray, &ts);
        out_t[i] = h.t; out_surface[i] = h.surface;
        if (out_uv) { out_uv[2 * i] = h.u; out_uv[2 * i + 1] = h.v; }
    }
    if (counters) *counters = cnt;
    free(ts.to_visit.H);
}

/* ------------------------------------------------------------------ photon emission (photon-mapper.cpp) */
typedef struct {
    float* photons; uint64_t* keys; uint64_t cap, count;
} PhotonSink;

static void sinkPhoton(PhotonSink* s, v3 flux, v3 position, v3 direction, uint64_t key) { /* Photon ctor, photon.hpp:7-12 */
    if (s->count < s->cap) {
        float* o = s->photons + 8 * s->count;
        o[0] = (float)flux.x; o[1] = (float)flux.y; o[2] = (float)flux.z;
        o[3] = (float)position.x; o[4] = (float)position.y; o[5] = (float)position.z;
        o[7] = (float)atan2(sqrt(direction.x * direction.x + direction.y * direction.y), direction.z); /* theta */
        o[6] = (float)atan2(direction.y, direction.x);                                                   /* phi   */
        if (s->keys) s->keys[s->count] = key;
    }
    s->count++;
}

static void emitPhoton(Ctx* C, Ray ray, v3 flux, Sampler* smp, double non_caustic_reject, PhotonSink* gs, PhotonSink* cs, uint64_t key) { /* :225-277 */
    const mcrt_scene_desc* s = C->S.s;
    RefractionHistory rh; rhInit(&rh, &ray);
    v3 bsdf_absIdotN; double bsdf_pdf;
    for (uint64_t bounce = 0;; bounce++) {
        samplerShuffle(smp);
        Hit isect = sceneIntersect(&C->S, &ray, C->ts);
        if (isect.surface == NO_SURFACE) return;
        Interaction ia;
        iaInit(&ia, s, &isect, &ray, rhExternalIOR(&rh, &ray), smp);
        if (!(ia.material->flags & MCRT_MAT_DIRAC_DELTA)) {
            if (ray.dirac_delta) sinkPhoton(cs, flux, ia.position, vneg(ray.direction), key | (bounce & 0xFFFFu));
            else if (non_caustic_reject > samplerGet(smp, 2 /* Dim::PM_REJECT */))
                sinkPhoton(gs, vdivs(flux, non_caustic_reject), ia.position, vneg(ray.direction), key | (bounce & 0xFFFFu));
        }
        if (!iaSampleBSDF(&ia, &bsdf_absIdotN, &bsdf_pdf, &ray, 1, smp)) return;
        bsdf_absIdotN = vdivs(bsdf_absIdotN, bsdf_pdf);
        double survive = smin(compMax(bsdf_absIdotN), 0.95);
        if (survive == 0.0 || survive <= samplerGet(smp, DIM_ABSORB)) return;
        flux = vmul(flux, vdivs(bsdf_absIdotN, survive));
        rhUpdate(&rh, &ray);
    }
}

int oracle_emit_photons(const mcrt_scene_desc* scene, double emissions, double caustic_factor, uint32_t global_seed,
                        float* global_photons, uint64_t* global_keys, uint64_t global_capacity, uint64_t* global_count,
                        float* caustic_photons, uint64_t* caustic_keys, uint64_t caustic_capacity, uint64_t* caustic_count,
                        uint64_t* emission_paths, uint64_t* rays) {
    pthread_once(&g_dirs_once, initDirections);
    ThreadScratch ts; memset(&ts, 0, sizeof(ts));
    oracle_counters cnt; memset(&cnt, 0, sizeof(cnt));
    Ctx C; memset(&C, 0, sizeof(C)); C.S.s = scene; C.S.c = &cnt; C.ts = &ts;
    PhotonSink gs = {global_photons, global_keys, global_capacity, 0}, cs = {caustic_photons, caustic_keys, caustic_capacity, 0};
    /* photon-mapper.cpp:28-38 */
    size_t photon_emissions0 = (size_t)emissions;
    double non_caustic_reject = 1.0 / caustic_factor;
    size_t photon_emissions = (size_t)((double)photon_emissions0 * caustic_factor);
    double total_add_flux = 0.0; /* :43-47 */
    for (uint32_t i = 0; i < scene->num_lights; i++) {
        uint32_t ls = scene->light_surface[i];
        v3 lf = vscale(ld3(scene->materials[scene->surf_material[ls]].emittance), scene->surf_area[ls]);
        total_add_flux += 0.0 + lf.x + lf.y + lf.z; /* glm::compAdd: ((0 + x) + y) + z */
    }
    uint64_t paths = 0;
    for (uint32_t i = 0; i < scene->num_lights; i++) { /* :62-78, 86-115 */
        uint32_t ls = scene->light_surface[i];
        v3 light_flux = vscale(ld3(scene->materials[scene->surf_material[ls]].emittance), scene->surf_area[ls]);
        double share = (0.0 + light_flux.x + light_flux.y + light_flux.z) / total_add_flux;
        size_t num_light_emissions = (size_t)((double)photon_emissions * share);
        v3 photon_flux = vdivs(light_flux, (double)num_light_emissions);
        Sampler smp; memset(&smp, 0, sizeof(smp)); smp.global_seed = global_seed;
        samplerInitiate(&smp, i);
        for (size_t j = 0; j < num_light_emissions; j++) {
            samplerSetIndex(&smp, (uint32_t)j);
            double u0 = samplerGet(&smp, 0), u1 = samplerGet(&smp, 1), u2 = samplerGet(&smp, 2), u3 = samplerGet(&smp, 3);
            v3 pos = surfSample(scene, ls, u0, u1);
            v3 normal = surfNormal(scene, ls, pos);
            M3 T = orthonormalBasis(normal);
            v3 dir = csFrom(&T, cosWeightedHemi(u2, u3)); /* CoordinateSystem::from(v, N) */
            pos = vadd(pos, vscale(normal, EPSILON));
            paths++;
            emitPhoton(&C, rayDir(pos, dir, scene->scene_ior), photon_flux, &smp, non_caustic_reject, &gs, &cs,
                       ((uint64_t)i << 48) | ((uint64_t)(uint32_t)j << 16));
        }
    }
    *global_count = gs.count; *caustic_count = cs.count;
    if (emission_paths) *emission_paths = paths;
    if (rays) *rays = cnt.rays;
    free(ts.to_visit.H);
    return (gs.count > gs.cap || cs.count > cs.cap) ? -1 : 0;
}

void oracle_bsdf_kat(uint64_t n, const double* in, const double* consts, double* out) {
    mcrt_material rough; memset(&rough, 0, sizeof(rough));
    rough.roughness = consts[0];
    rough.reflectance[0] = consts[1]; rough.reflectance[1] = consts[2]; rough.reflectance[2] = consts[3];
    double variance = pow2(rough.roughness); /* material.cpp:106-108 */
    rough.A = 1.0 - 0.5 * (variance / (variance + 0.33));
    rough.B = 0.45 * (variance / (variance + 0.09));
    rough.flags = MCRT_MAT_ROUGH;
    v3 real = ld3(consts + 4), imag = ld3(consts + 7);
    for (uint64_t i = 0; i < n; i++) {
        const double* I = in + 11 * i; double* O = out + 18 * i;
        v3 wi = ld3(I), wo = ld3(I + 3);
        double n1 = I[6], n2 = I[7], al[2] = {I[8], I[8]}, u = I[9], v = I[10], pdf;
        O[0] = fresnelDielectric(n1, n2, wo.z);
        v3 fc = fresnelConductor(n1, real, imag, wo.z);
        O[1] = fc.x; O[2] = fc.y; O[3] = fc.z;
        v3 wir = wi; wir.z = fabs(wir.z) + 1e-3; wir = vnormalize(wir);
        O[4] = ggxReflection(wir, wo, al, &pdf); O[5] = pdf;
        v3 wit = vneg(wir);
        O[6] = ggxTransmission(wit, wo, n1, n2, al, &pdf); O[7] = pdf;
        v3 m = ggxVisibleMicrofacet(u, v, wo, al);
        O[8] = m.x; O[9] = m.y; O[10] = m.z;
        O[11] = ggxD(m, al);
        O[12] = ggxLambda(wo, al);
        v3 d = matDiffuseReflection(&rough, wir, wo, &pdf);
        O[13] = d.x; O[14] = d.y; O[15] = d.z; O[16] = pdf; O[17] = 0.0;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Image::save (camera/image.cpp:37-88): exposure and gain from histograms, tone map, gamma, bytes.
 * ------------------------------------------------------------------------------------------- */
/* pixel-operators.cpp:7-18 */
static v3 imgHableF(v3 x) {
    const double A = 0.15, B = 0.50, C = 0.10, D = 0.20, E = 0.02, F = 0.30;
    v3 num = vadds(vmul(x, vadds(vscale(x, A), C * B)), D * E);
    v3 den = vadds(vmul(x, vadds(vscale(x, A), B)), D * F);
    return vadds(vdiv(num, den), -(E / F));
}
static v3 imgFilmicHable(v3 in) { return vdiv(imgHableF(in), imgHableF(V(11.2, 11.2, 11.2))); }

/* pixel-operators.cpp:20-39; glm mat3 * vec3 (type_mat3x3.inl:468-474): column 0 * x + column 1 * y + column 2 * z */
static v3 imgMat(const double c0[3], const double c1[3], const double c2[3], v3 v) {
    v3 r;
    r.x = c0[0] * v.x + c1[0] * v.y + c2[0] * v.z;
    r.y = c0[1] * v.x + c1[1] * v.y + c2[1] * v.z;
    r.z = c0[2] * v.x + c1[2] * v.y + c2[2] * v.z;
    return r;
}
static double imgClamp(double x, double lo, double hi) { double m = x < lo ? lo : x; return hi < m ? hi : m; } /* glm::clamp */
static v3 imgFilmicACES(v3 in) {
    static const double i0[3] = {0.59719, 0.07600, 0.02840}, i1[3] = {0.35458, 0.90834, 0.13383}, i2[3] = {0.04823, 0.01566, 0.83777};
    static const double o0[3] = {1.60475, -0.10208, -0.00327}, o1[3] = {-0.53108, 1.10813, -0.07276}, o2[3] = {-0.07367, -0.00605, 1.07602};
    v3 v = imgMat(i0, i1, i2, in);
    v3 a = vadds(vmul(v, vadds(v, 0.0245786)), -0.000090537);
    v3 b = vadds(vmul(v, vadds(vscale(v, 0.983729), 0.4329510)), 0.238081);
    v3 c = imgMat(o0, o1, o2, vdiv(a, b));
    c.x = imgClamp(c.x, 0.0, 1.0); c.y = imgClamp(c.y, 0.0, 1.0); c.z = imgClamp(c.z, 0.0, 1.0);
    return c;
}
static v3 imgTonemap(uint32_t tonemapper, int plain, v3 in) { /* image.cpp:27-34 */
    if (plain) return in;
    return tonemapper == MCRT_TONEMAP_ACES ? imgFilmicACES(in) : imgFilmicHable(in);
}

/* common/histogram.cpp:6-41 */
typedef struct { uint64_t* counts; uint64_t num_counts; double bin_size; uint64_t data_size; } ImgHistogram;
static void imgHistogram(ImgHistogram* h, const double* data, uint64_t n, uint64_t num_bins) {
    h->counts = NULL; h->num_counts = 0; h->bin_size = 1.0; h->data_size = n;
    double max = -DBL_MAX;
    for (uint64_t i = 0; i < n; i++) {
        if (data[i] < 0.0) return;
        if (max < data[i]) max = data[i];
    }
    h->counts = (uint64_t*)calloc(num_bins, sizeof(uint64_t));
    h->num_counts = num_bins;
    h->bin_size = max / (double)num_bins;
    for (uint64_t i = 0; i < n; i++) {
        double q = data[i] / h->bin_size;
        uint64_t b = !(q < 9.2233720368547758e18) ? (uint64_t)1 << 63 : (uint64_t)q; /* what x86-64 makes of inf / NaN */
        h->counts[b < num_bins - 1 ? b : num_bins - 1]++;
    }
}
static double imgLevel(const ImgHistogram* h, double count_percentage) {
    uint64_t num = (uint64_t)((double)h->data_size * count_percentage), count = 0;
    double level = 0.0;
    for (uint64_t i = 0; i < h->num_counts; i++) {
        count += h->counts[i];
        if (count >= num) { level = (double)(i + 1) * h->bin_size; break; }
    }
    return level;
}

int oracle_image_save(const double* rgb, uint32_t width, uint32_t height, uint32_t tonemapper, int plain,
                      double exposure_compensation, double gain_compensation, uint8_t* bgr, double* factors) {
    const uint64_t n = (uint64_t)width * height;
    double exposure_factor = 1.0, gain_factor = 1.0;
    if (!plain) {
        double* brightness = (double*)malloc(n * sizeof(double));
        if (!brightness) return -1;
        ImgHistogram h;
        /* getExposure, image.cpp:62-72 (glm::compAdd: T(0) + x + y + z) */
        for (uint64_t i = 0; i < n; i++) brightness[i] = (((0.0 + rgb[3 * i]) + rgb[3 * i + 1]) + rgb[3 * i + 2]) / 3.0;
        imgHistogram(&h, brightness, n, 65536);
        double L = imgLevel(&h, 0.5);
        free(h.counts);
        exposure_factor = (L > 0.0 ? 0.5 / L : 1.0) * pow(2, exposure_compensation); /* :39, :19-22 */
        /* getGain, image.cpp:77-87 */
        for (uint64_t i = 0; i < n; i++) {
            v3 t = imgTonemap(tonemapper, 0, vscale(ld3(rgb + 3 * i), exposure_factor));
            brightness[i] = (((0.0 + t.x) + t.y) + t.z) / 3.0;
        }
        imgHistogram(&h, brightness, n, 65536);
        L = imgLevel(&h, 0.99);
        free(h.counts);
        gain_factor = (L > 0.0 ? 0.99 / L : 1.0) * pow(2, gain_compensation); /* :40 */
        free(brightness);
    }
    const double top = nextafter(256.0, 0.0); /* truncate, pixel-operators.cpp:51-55 */
    for (uint64_t i = 0; i < n; i++) {
        v3 t = vscale(imgTonemap(tonemapper, plain, vscale(ld3(rgb + 3 * i), exposure_factor)), gain_factor);
        double c[3] = {t.x, t.y, t.z};
        for (int k = 0; k < 3; k++) { /* sRGB::gammaCompress, color/srgb.hpp:55-63 */
            double g = c[k] <= 0.0031308 ? 12.92 * c[k] : 1.055 * pow(c[k], 1.0 / 2.4) - 0.055;
            c[k] = imgClamp(g, 0.0, 1.0) * top;
        }
        bgr[3 * i] = (uint8_t)c[2]; bgr[3 * i + 1] = (uint8_t)c[1]; bgr[3 * i + 2] = (uint8_t)c[0];
    }
    if (factors) { factors[0] = exposure_factor; factors[1] = gain_factor; }
    return 0;
}
