// Per-lane integrators (the §8 rows a1, a3, a22-a24) written as ONE-BOUNCE step functions so that the
// persistent kernel can keep every lane of a wavefront busy: a lane whose path ends is re-armed with
// the next sample / next pixel in the same loop iteration instead of idling until the slowest path
// of the wave finishes (mcrt_kernels.hip).
//   PathTracer::sampleRay     integrator/path-tracer/path-tracer.cpp:14-51
//   PhotonMapper::sampleRay   integrator/photon-mapper/photon-mapper.cpp:279-341
//   estimate{Global,Caustic}Radiance  photon-mapper.cpp:343-391
//   LinearOctree::knnSearch   octree/linear-octree.cpp:25-117
#pragma once

#include "mcrt_shade.hpp"

namespace mcrt {

struct PathState {
    Ray ray;
    d3 radiance, throughput;
    LightSample ls;
    Sampler smp;
};

// Optional phase profiler (MCRT_PROFILE_PHASES=1 selects the instrumented kernel): per phase, the
// wave-cycles spent (accumulated by the first active lane) and the lane-cycles spent (every active
// lane), whose ratio is the SIMD lane utilisation of that phase.
enum : int { kPhRegen = 0, kPhTraverse = 1, kPhShade = 2, kPhShadow = 3, kPhSample = 4, kPhLoop = 5, kNumPhases = 6 };
template <bool kProf>
struct PhaseProf {
    MCRT_HD void mark(int) {}
    MCRT_HD unsigned long long now() const { return 0ull; }
    MCRT_HD void span(int, unsigned long long, bool) {}
};
#if defined(__HIPCC__)
template <>
struct PhaseProf<true> {
    unsigned long long wave_cycles[kNumPhases], lane_cycles[kNumPhases];
    unsigned long long t0;
    int cur;
    __device__ void begin() {
        for (int i = 0; i < kNumPhases; i++) wave_cycles[i] = lane_cycles[i] = 0ull;
        t0 = clock64();
        cur = kPhLoop;
    }
    // wave-level form for blocks that only some lanes execute: call now() before and span() after the
    // block from code that ALL lanes run; `participated` = this lane executed the block
    __device__ unsigned long long now() const { return clock64(); }
    __device__ void span(int p, unsigned long long t_begin, bool participated) {
        const unsigned long long dt = clock64() - t_begin;
        if (participated) lane_cycles[p] += dt;
        if (__lane_id() == 0) wave_cycles[p] += dt;
    }
    // called by every active lane when it enters phase p
    __device__ void mark(int p) {
        const unsigned long long t = clock64();
        const unsigned long long dt = t - t0;
        const unsigned long long mask = waveBallot(1);
        lane_cycles[cur] += dt;
        if ((int)__lane_id() == __ffsll((long long)mask) - 1) wave_cycles[cur] += dt;
        t0 = t;
        cur = p;
    }
};
#endif

// Start of sampleRay (path-tracer.cpp:16-19 / photon-mapper.cpp:281-284) for the ray samplePixel built.
MCRT_HD void pathBegin(PathState& st, RefractionHistory& rh, const Ray& camera_ray) {
    st.ray = camera_ray;
    st.radiance = splat(0.0);
    st.throughput = splat(1.0);
    rh.init(camera_ray);
    st.ls.bsdf_pdf = 0.0;
    st.ls.select_probability = 0.0;
    st.ls.light = kNoSurface;
}

// One iteration of the while(true) in PathTracer::sampleRay. Returns true when the path has ended;
// st.radiance then holds sampleRay's return value.
template <bool kCount, bool kAll, bool kProf = false, bool kFlat = false, bool kPreK = false>
MCRT_HD bool pathTracerBounce(PathState& st, RefractionHistory& rh, const SceneViewT<kAll>& sv, const ShadeViewT<kAll>& sh, const LaneStack& stk,
                              TraceCounters& cnt, SobolTab tab, PhaseProf<kProf>* prof = nullptr) {
    if (kProf) prof->mark(kPhTraverse);
    st.smp.shuffle();                                                     // :23
    Hit isect = sceneIntersect<kAll, kCount, false, kFlat, kPreK>(sv, st.ray, stk, cnt);  // :25
    if (kProf) prof->mark(kPhShade);
    if (isect.surface == kNoSurface) {                                    // :27-30
        st.radiance = st.radiance + skyColor(st.ray) * st.throughput;
        return true;
    }
    InteractionT<kAll> ia;
    interactionInit(ia, sh, isect, st.ray, rh.externalIOR(st.ray), st.smp, tab);  // :32
    st.radiance = st.radiance + sampleEmissive(sh, ia, st.ls) * st.throughput;       // :34

    DirectQuery dq;                                                       // :35 Integrator::sampleDirect
    if (sampleDirectSetup(sh, ia, st.ls, dq, st.smp, tab)) {
        if (kProf) prof->mark(kPhShadow);
        Hit shadow = sceneIntersect<kAll, kCount, true, kFlat, kPreK>(sv, dq.shadow_ray, stk, cnt, &dq.sq);
        if (kProf) prof->mark(kPhSample);
        st.radiance = st.radiance + sampleDirectFinish(sh, ia, st.ls, dq, shadow) * st.throughput;
    }
    if (kProf) prof->mark(kPhSample);

    d3 bsdf_absIdotN;
    if (!interactionSampleBSDF(ia, bsdf_absIdotN, st.ls.bsdf_pdf, st.ray, false, st.smp, tab)) return true;  // :37-40
    st.throughput = st.throughput * (bsdf_absIdotN / st.ls.bsdf_pdf);     // :42
    if (absorb(st.ray, st.throughput, st.smp, tab)) return true;          // :44-47
    rh.update(st.ray);                                                 // :49
    return false;
}

// ------------------------------------------------------------------ photon emission pass (§8(f) rank 1)
//   PhotonMapper::PhotonMapper per-emission set-up   photon-mapper.cpp:96-110
//   PhotonMapper::emitPhoton                          photon-mapper.cpp:225-277
//   Photon (FP32 record, polar direction)              photon.hpp:5-38
struct EmitState {
    Ray ray;
    d3 flux;
    Sampler smp;
    uint64_t key;     // light << 48 | emission index << 16 (the bounce number is or-ed in per photon)
    uint32_t bounce;
};

struct PhotonOut {
    bool store, caustic;
    float rec[8];     // flux rgb, position xyz, phi, theta
    uint64_t key;
};

MCRT_HD void encodePhoton(PhotonOut& out, d3 flux, d3 position, d3 direction) {  // Photon ctor, photon.hpp:7-12
    out.rec[0] = (float)flux.x;
    out.rec[1] = (float)flux.y;
    out.rec[2] = (float)flux.z;
    out.rec[3] = (float)position.x;
    out.rec[4] = (float)position.y;
    out.rec[5] = (float)position.z;
    out.rec[7] = (float)refAtan2(sqrt(direction.x * direction.x + direction.y * direction.y), direction.z);  // theta (glibc's atan2, bit for bit)
    out.rec[6] = (float)refAtan2(direction.y, direction.x);                                                   // phi
}

// photon-mapper.cpp:98-110: emission `index` of light number `light` (position in Scene::emissives).
template <bool L>
MCRT_HD void emitBegin(EmitState& es, RefractionHistory& rh, const ShadeViewT<L>& sh, uint32_t light, uint32_t index, d3 photon_flux,
                       uint32_t global_seed, SobolTab tab) {
    es.smp.initiate(global_seed, light);
    es.smp.setIndex(index);
    const double u0 = es.smp.get(0, tab), u1 = es.smp.get(1, tab), u2 = es.smp.get(2, tab), u3 = es.smp.get(3, tab);  // Dim::PM_LIGHT, 4D
    const uint32_t surface = sh.light_surface[light];
    d3 pos = surfSample(sh, surface, u0, u1);
    d3 normal = surfNormal(sh, surface, pos);
    d3 dir = csFrom(orthonormalBasis(normal), cosWeightedHemi<!L>(u2, u3));  // CoordinateSystem::from(v, N)
    pos = pos + normal * kEpsilon;
    es.ray = makeRay(pos, dir, sh.scene_ior);
    es.flux = photon_flux;
    es.key = ((uint64_t)light << 48) | ((uint64_t)index << 16);
    es.bounce = 0;
    rh.init(es.ray);
}

template <bool kAll>
MCRT_HD bool emitAfterHit(EmitState& es, RefractionHistory& rh, const ShadeViewT<kAll>& sh, const Hit& isect, SobolTab tab,
                          double non_caustic_reject, PhotonOut& out);

// One iteration of the while(true) in PhotonMapper::emitPhoton. Returns true when the photon path has
// ended; out.store says whether this bounce deposited a photon (out.caustic: into which map).
template <bool kCount, bool kAll>
MCRT_HD bool emitBounce(EmitState& es, RefractionHistory& rh, const SceneViewT<kAll>& sv, const ShadeViewT<kAll>& sh,
                        const LaneStack& stk, TraceCounters& cnt, SobolTab tab, double non_caustic_reject, PhotonOut& out) {
    es.smp.shuffle();                                                            // :233
    Hit isect = sceneIntersect<kAll, kCount, false>(sv, es.ray, stk, cnt);       // :235
    return emitAfterHit(es, rh, sh, isect, tab, non_caustic_reject, out);
}

// ... the part after Scene::intersect (callers that walk the tree differently trace the ray themselves).
template <bool kAll>
MCRT_HD bool emitAfterHit(EmitState& es, RefractionHistory& rh, const ShadeViewT<kAll>& sh, const Hit& isect, SobolTab tab,
                          double non_caustic_reject, PhotonOut& out) {
    out.store = false;
    out.caustic = false;
    if (isect.surface == kNoSurface) return true;                                // :237-240
    InteractionT<kAll> ia;
    interactionInit(ia, sh, isect, es.ray, rh.externalIOR(es.ray), es.smp, tab);  // :242
    if (!(ia.material->flags & MCRT_MAT_DIRAC_DELTA)) {                          // :245-255
        if (es.ray.dirac_delta) {
            out.store = true;
            out.caustic = true;
            encodePhoton(out, es.flux, ia.position, -es.ray.direction);
        } else if (non_caustic_reject > es.smp.get(2 /* Dim::PM_REJECT */, tab)) {
            out.store = true;
            encodePhoton(out, es.flux / non_caustic_reject, ia.position, -es.ray.direction);
        }
        out.key = es.key | (uint64_t)(es.bounce & 0xFFFFu);
    }
    es.bounce++;
    d3 bsdf_absIdotN;
    double bsdf_pdf;
    if (!interactionSampleBSDF(ia, bsdf_absIdotN, bsdf_pdf, es.ray, true, es.smp, tab)) return true;  // :257-260
    bsdf_absIdotN = bsdf_absIdotN / bsdf_pdf;                                    // :262
    double survive = gmin(compMax(bsdf_absIdotN), 0.95);                         // :267
    if (survive == 0.0 || survive <= es.smp.get(kDimAbsorb, tab)) return true;   // :268-271
    es.flux = es.flux * (bsdf_absIdotN / survive);                               // :273
    rh.update(es.ray);                                                           // :275
    return false;
}

// ------------------------------------------------------------------ photon map
struct PhotonMapView {
    uint32_t num_octants;
    uint64_t num_photons;
    const double* octant_bounds;       // [n][6]
    const uint32_t* octant_start;      // [n] (u32 on the device: maps are < 4G photons per GPU)
    const uint32_t* octant_contained;  // [n]
    const uint32_t* octant_next;       // [n] 0xFFFFFFFF = none
    const uint8_t* octant_leaf;        // [n]
    const float* photons;              // [n][8] flux rgb, position xyz, phi, theta
};

struct KnnEntry {
    double distance2;
    uint32_t index;
};
struct OctantEntry {
    double distance2;
    uint32_t octant;
};

// Per-lane scratch for the search, in global memory, interleaved by lane ([slot][lane]) so that a
// wave touching the same slot coalesces.
struct KnnScratch {
    double* res_d2;      // &res_d2[lane],   stride lanes, k slots
    uint32_t* res_idx;   // &res_idx[lane]
    double* visit_d2;    // &visit_d2[lane], stride lanes, kMaxVisit slots
    uint32_t* visit_oct;
    uint32_t stride;
    // The frontier's capacity per lane (the reference's is an unbounded priority queue, linear-octree.cpp:33): the host sizes the two
    // visit arrays to it and renders a frame / repeats an operator call with a larger one when a search ran out (round 6; until then a
    // full frontier dropped the entry WITHOUT a word - 160 entries, which a best-first descent of a 21-level octree cannot exceed
    // before its first scan but a crowd of inner octants inside the bound can).
    uint32_t max_visit;
    mutable uint32_t overflowed;  // kLaneKnnOverflow once a push found the frontier full (the result is then not to be trusted)
    MCRT_HD KnnEntry res(uint32_t i) const { return KnnEntry{res_d2[(size_t)i * stride], res_idx[(size_t)i * stride]}; }
    MCRT_HD void setRes(uint32_t i, KnnEntry e) const { res_d2[(size_t)i * stride] = e.distance2; res_idx[(size_t)i * stride] = e.index; }
    MCRT_HD OctantEntry visit(uint32_t i) const { return OctantEntry{visit_d2[(size_t)i * stride], visit_oct[(size_t)i * stride]}; }
    MCRT_HD void setVisit(uint32_t i, OctantEntry e) const { visit_d2[(size_t)i * stride] = e.distance2; visit_oct[(size_t)i * stride] = e.octant; }
};
constexpr uint32_t kMaxVisit = 160;  // default frontier capacity: <= 7 new entries per level of an octree's first descent (22 levels)
constexpr uint32_t kMaxVisitLimit = 1u << 15;  // what the host grows it to at most (x 12 bytes x the launch's lanes)
constexpr uint32_t kLaneKnnOverflow = 0x10000u;  // = kKnnOverflowFlag (mcrt_waveknn.hpp): the host tells it from the traversal stacks' overflow counts by the bits above 15

MCRT_HD double boxDistance2(const double* b, d3 p) {  // BoundingBox::distance2, bounding-box.cpp:43-47
    d3 a = ld3(b) - p, c = p - ld3(b + 3);
    d3 d = d3{gmax(gmax(a.x, c.x), 0.0), gmax(gmax(a.y, c.y), 0.0), gmax(gmax(a.z, c.z), 0.0)};
    return dot(d, d);
}
MCRT_HD double boxMaxDistance2(const double* b, d3 p) {  // BoundingBox::max_distance2, bounding-box.cpp:50-54
    d3 a = ld3(b + 3) - p, c = p - ld3(b);
    d3 d = d3{gmax(a.x, c.x), gmax(a.y, c.y), gmax(a.z, c.z)};
    return dot(d, d);
}

// Max-heap on distance2 of the k best photons (the reference's PriorityQueue<SearchResult<Photon>>,
// common/priority-queue.hpp:47-50,103-123). Returns the number of results; the heap root (slot 0)
// holds the farthest of them once `count == k`.
MCRT_HD void knnSiftDown(const KnnScratch& s, uint32_t size, KnnEntry value, uint32_t index) {
    for (;;) {
        uint32_t left = 2 * index + 1, right = left + 1, max_child;
        if (right < size) max_child = left + (s.res(left).distance2 < s.res(right).distance2 ? 1u : 0u);
        else if (left < size) max_child = left;
        else break;
        KnnEntry mc = s.res(max_child);
        if (!(value.distance2 < mc.distance2)) break;
        s.setRes(index, mc);
        index = max_child;
    }
    s.setRes(index, value);
}
// PriorityQueue::make_heap (priority-queue.hpp:57-84), step for step: the array the reference's loops over `photons` then walk.
MCRT_HD void knnMakeHeap(const KnnScratch& s, uint32_t size) {
    if (size <= 1) return;
    const uint32_t last_index = size - 1;
    uint32_t index = (last_index - 1) / 2;
    auto swapIfLess = [&](uint32_t a, uint32_t b) {  // if (H[a] < H[b]) std::swap(H[a], H[b])
        const KnnEntry ea = s.res(a), eb = s.res(b);
        if (ea.distance2 < eb.distance2) {
            s.setRes(a, eb);
            s.setRes(b, ea);
        }
    };
    if (last_index % 2) {
        swapIfLess(index, 2 * index + 1);
        if (index == 0) return;
        index--;
    }
    if (index) {
        const uint32_t lowest_index_with_no_grandchildren = (last_index - 3) / 4 + 1;
        do {
            const uint32_t left = 2 * index + 1;
            const uint32_t max_child = left + (s.res(left).distance2 < s.res(left + 1).distance2 ? 1u : 0u);
            swapIfLess(index, max_child);
        } while (index-- != lowest_index_with_no_grandchildren);
    }
    do {
        knnSiftDown(s, size, s.res(index), index);
    } while (index--);
}
// Min-heap on distance2 of octants still to visit (linear-octree.cpp:37-44; priority-queue.hpp:19-45).
MCRT_HD void visitPush(const KnnScratch& s, uint32_t& size, OctantEntry value) {
    if (size >= s.max_visit) {  // reported, never silent: the host repeats the work with a larger frontier
        s.overflowed = kLaneKnnOverflow;
        return;
    }
    uint32_t index = size++;
    while (index > 0) {
        uint32_t parent = (index - 1) / 2;
        OctantEntry pe = s.visit(parent);
        if (!(value.distance2 < pe.distance2)) break;
        s.setVisit(index, pe);
        index = parent;
    }
    s.setVisit(index, value);
}
MCRT_HD void visitPop(const KnnScratch& s, uint32_t& size) {
    if (size > 1) {
        OctantEntry value = s.visit(--size);
        uint32_t index = 0;
        for (;;) {
            uint32_t left = 2 * index + 1, right = left + 1, c;
            if (right < size) c = left + (s.visit(right).distance2 < s.visit(left).distance2 ? 1u : 0u);
            else if (left < size) c = left;
            else break;
            OctantEntry ce = s.visit(c);
            if (!(ce.distance2 < value.distance2)) break;
            s.setVisit(index, ce);
            index = c;
        }
        s.setVisit(index, value);
    } else {
        size--;
    }
}

// LinearOctree<Photon>::knnSearch (linear-octree.cpp:25-117), per lane: the reference's pruning rules (inclusive <= comparisons), its
// two queues operation for operation (PriorityQueue::push / pop / push_unordered / make_heap / pop_push) - the k-set AND the order
// of the result array are the reference's.
MCRT_HD uint32_t knnSearch(const PhotonMapView& m, d3 p, uint32_t k, const KnnScratch& s, uint32_t& octant_visits) {
    if (m.num_octants == 0) return 0;
    if ((uint64_t)k > m.num_photons) k = (uint32_t)m.num_photons;
    if (k == 0) return 0;
    double max_distance2 = kDblMax;
    uint32_t count = 0, nvisit = 0;
    OctantEntry current{boxDistance2(m.octant_bounds, p), 0u};
    for (;;) {
        const uint32_t oc = current.octant;
        octant_visits++;
        const uint32_t contained = m.octant_contained[oc];
        if (m.octant_leaf[oc] || contained <= k) {
            const uint32_t start = m.octant_start[oc], end = start + contained;
            for (uint32_t i = start; i < end; i++) {
                const float* ph = m.photons + (size_t)i * 8;
                d3 d = p - d3{(double)ph[3], (double)ph[4], (double)ph[5]};  // glm::distance2(data.pos(), p)
                double distance2 = dot(d, d);
                if (distance2 <= max_distance2) {
                    // linear-octree.cpp:58-79, with its heap discipline: the first k - 1 results appended unordered, the heap made at
                    // the k-th (make_heap), then pop_push - so the result ARRAY is the reference's, and a loop over it sums an
                    // estimate's photons in the reference's order (round 5; a heap kept by sifting up held the same set in another order)
                    if (count + 1u < k) {
                        s.setRes(count++, KnnEntry{distance2, i});
                    } else {
                        if (count != k) {
                            s.setRes(count++, KnnEntry{distance2, i});
                            knnMakeHeap(s, count);
                        } else {
                            knnSiftDown(s, count, KnnEntry{distance2, i}, 0);  // pop_push
                        }
                        double top = s.res(0).distance2;
                        if (top < max_distance2) max_distance2 = top;
                    }
                }
            }
        } else {
            uint32_t child = oc + 1;
            while (child != 0xFFFFFFFFu) {
                const double* cb = m.octant_bounds + (size_t)child * 6;
                double distance2 = boxDistance2(cb, p);
                if (distance2 <= max_distance2) {
                    visitPush(s, nvisit, OctantEntry{distance2, child});
                    if (m.octant_contained[child] >= k) {
                        double md = boxMaxDistance2(cb, p);
                        if (md < max_distance2) max_distance2 = md;
                    }
                }
                child = m.octant_next[child];
            }
        }
        if (nvisit == 0) break;
        current = s.visit(0);
        if (current.distance2 > max_distance2) break;
        visitPop(s, nvisit);
    }
    return count;
}

// Photon::dir, photon.hpp:19-27 (float sin/cos overloads: glibc's sincosf, refSinCosF). kExact: the restated sincosf, bit for bit
// (csrc/mcrt_libm.hpp; tests/test_libm.py) - what the per-lane estimates below use, whose sums run in the reference's order.
template <bool kExact = false>
MCRT_HD d3 photonDirection(const float* ph) {
    float phi = ph[6], theta = ph[7];
    float st, ct, sp, cp;
#if defined(MCRT_EXACT_PHOTON_DIR)
    constexpr bool exact = true;
#else
    // Not the default of the wave-cooperative estimates: inside renderKernelPM (128 VGPRs, ~700 spilled) the restated sincosf costs a
    // C5 frame 8.8 % (766 -> 832 ms on the 64 spp probe; as a called function or a loop over the two angles 954 / 987 ms:
    // profiles/r05_ab_c5_bisect.log), and it buys nothing a test can see there - those k terms are summed by a wave reduction, not in
    // the reference's heap order, so the frames are compared at 1e-10 either way. -DMCRT_EXACT_PHOTON_DIR selects it everywhere.
    constexpr bool exact = kExact;
#endif
    if constexpr (exact) {
        refSinCosF(theta, st, ct);
        refSinCosF(phi, sp, cp);
    } else {
        st = sinf(theta); ct = cosf(theta); sp = sinf(phi); cp = cosf(phi);
    }
    double sin_theta = (double)st;
    return d3{sin_theta * (double)cp, sin_theta * (double)sp, (double)ct};
}

struct PhotonViews {
    PhotonMapView global_map, caustic_map;
    uint32_t k_nearest;
    bool direct_visualization;
};

// estimateGlobalRadiance, photon-mapper.cpp:343-363
template <bool L>
MCRT_HD d3 estimateGlobalRadiance(const PhotonViews& pv, const InteractionT<L>& ia, const KnnScratch& s, uint32_t& searches,
                                  uint32_t& octant_visits) {
    searches++;
    const PhotonMapView& m = pv.global_map;
    uint32_t n = knnSearch(m, ia.position, pv.k_nearest, s, octant_visits);
    if (n == 0) return splat(0.0);
    d3 radiance = splat(0.0);
    for (uint32_t i = 0; i < n; i++) {
        const float* ph = m.photons + (size_t)s.res(i).index * 8;
        d3 bsdf_absIdotN;
        double bsdf_pdf;
        if (interactionBSDF(ia, bsdf_absIdotN, photonDirection<true>(ph), bsdf_pdf))
            radiance = radiance + d3{(double)ph[0], (double)ph[1], (double)ph[2]} * bsdf_absIdotN / bsdf_pdf;
    }
    return radiance / (s.res(0).distance2 * kPi);
}

// estimateCausticRadiance (cone filter), photon-mapper.cpp:368-391
template <bool L>
MCRT_HD d3 estimateCausticRadiance(const PhotonViews& pv, const InteractionT<L>& ia, const KnnScratch& s, uint32_t& searches,
                                   uint32_t& octant_visits) {
    searches++;
    const PhotonMapView& m = pv.caustic_map;
    uint32_t n = knnSearch(m, ia.position, pv.k_nearest, s, octant_visits);
    if (n == 0) return splat(0.0);
    double inv_max_squared_radius = 1.0 / s.res(0).distance2;
    d3 radiance = splat(0.0);
    for (uint32_t i = 0; i < n; i++) {
        KnnEntry e = s.res(i);
        const float* ph = m.photons + (size_t)e.index * 8;
        d3 bsdf_absIdotN;
        double bsdf_pdf;
        if (interactionBSDF(ia, bsdf_absIdotN, photonDirection<true>(ph), bsdf_pdf)) {
            double wp = gmax(0.0, 1.0 - sqrt(e.distance2 * inv_max_squared_radius));
            radiance = radiance + (d3{(double)ph[0], (double)ph[1], (double)ph[2]} * bsdf_absIdotN * wp) / bsdf_pdf;
        }
    }
    return 3.0 * radiance * inv_max_squared_radius * kInvPi;
}

// One iteration of the while(true) in PhotonMapper::sampleRay (photon-mapper.cpp:288-340).
template <bool kCount, bool kAll>
MCRT_HD bool photonMapperBounce(PathState& st, RefractionHistory& rh, const SceneViewT<kAll>& sv, const ShadeViewT<kAll>& sh, const PhotonViews& pv,
                                const LaneStack& stk, const KnnScratch& ks, TraceCounters& cnt, uint32_t& searches,
                                uint32_t& octant_visits, SobolTab tab) {
    st.smp.shuffle();
    Hit isect = sceneIntersect<kAll, kCount, false>(sv, st.ray, stk, cnt);
    if (isect.surface == kNoSurface) return true;  // no sky in photon mode, :292-295
    InteractionT<kAll> ia;
    interactionInit(ia, sh, isect, st.ray, rh.externalIOR(st.ray), st.smp, tab);
    st.radiance = st.radiance + sampleEmissive(sh, ia, st.ls) * st.throughput;
    d3 bsdf_absIdotN;
    if (ia.dirac_delta) {  // :301-312
        if (!st.ray.dirac_delta && st.ray.depth != 0) return true;
        if (!interactionSampleBSDF(ia, bsdf_absIdotN, st.ls.bsdf_pdf, st.ray, false, st.smp, tab)) return true;
        st.throughput = st.throughput * (bsdf_absIdotN / st.ls.bsdf_pdf);
    } else {
        st.radiance = st.radiance + estimateCausticRadiance(pv, ia, ks, searches, octant_visits) * st.throughput;  // :315
        if (!pv.direct_visualization && (st.ray.dirac_delta || st.ray.depth == 0)) {  // :317-326
            DirectQuery dq;
            if (sampleDirectSetup(sh, ia, st.ls, dq, st.smp, tab)) {
                Hit shadow = sceneIntersect<kAll, kCount, true>(sv, dq.shadow_ray, stk, cnt, &dq.sq);
                st.radiance = st.radiance + sampleDirectFinish(sh, ia, st.ls, dq, shadow) * st.throughput;
            }
            if (!interactionSampleBSDF(ia, bsdf_absIdotN, st.ls.bsdf_pdf, st.ray, false, st.smp, tab)) return true;
            st.throughput = st.throughput * (bsdf_absIdotN / st.ls.bsdf_pdf);
        } else {  // :327-331
            st.radiance = st.radiance + estimateGlobalRadiance(pv, ia, ks, searches, octant_visits) * st.throughput;
            return true;
        }
    }
    if (absorb(st.ray, st.throughput, st.smp, tab)) return true;
    rh.update(st.ray);
    return false;
}

}  // namespace mcrt
