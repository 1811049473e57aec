// The flat loop's FP64 tests dealt over the wave (round 4; OPTIONAL form of the flat megakernel, option MCRT_FLAT_SHARE=1, off by
// default: built and proven bit-identical on the host emulation of a wavefront, its speed not yet measured on the device).
//
// renderKernel<path tracer, flat> culls the scene's <= 64 primitives per lane in packed FP32 and then runs the reference's FP64
// triangle / sphere test on each lane's survivors (2.4 of 44 per ray on hexagon_room) - in a loop that lasts as long as the wave's
// WORST lane has survivors (5-6 triangle + ~2 sphere iterations): 52 % of the issued lanes idle. Here a wave's (ray, survivor) pairs
// are the work items, as the trace kernel's shared leaf step does it (mcrt_sharedleaf.hpp): every lane offers up to four survivors,
// the offers are numbered by a prefix sum over the wave and the first 64 are a round's items; item lane k reads its owner's ray from
// LDS (every lane stores its ray once per intersection), runs the test, writes {t, u, v}; the owner reads its items' results back
// and keeps the closest - `closer` with its tie rule, so the order of the tests does not matter and the hit is the per-lane loop's,
// bit for bit. Triangles first, then spheres: a round runs one kind of test. ~154 pairs per wave = 3 rounds instead of ~8 iterations,
// each round four dependent LDS round trips (at two waves per SIMD): estimated -3 ... -5 % of a frame.
// Scenes with more than 32 triangles or 32 spheres keep the per-lane loop (one mask word per kind here).
#pragma once

constexpr uint32_t kFlatShareBytes = 64u * 48u + 64u * 24u + 64u * 2u;  // per wave: rays, results, the item -> (owner, primitive) map
struct FlatShare {
    MCRT_LDS_AS double* ray;    // [64][6] start, direction of the lanes' rays
    MCRT_LDS_AS double* res;    // [64][3] t (infinity: no hit), u, v of the round's items
    MCRT_LDS_AS uint16_t* map;  // [64] owner lane | primitive (kind-sorted index) << 8
};
__device__ inline FlatShare flatShareAt(unsigned char* lds, uint32_t base, uint32_t wave) {
    FlatShare F;
    MCRT_LDS_AS unsigned char* p = (MCRT_LDS_AS unsigned char*)lds + base + wave * kFlatShareBytes;
    F.ray = (MCRT_LDS_AS double*)p;
    F.res = (MCRT_LDS_AS double*)(p + 64u * 48u);
    F.map = (MCRT_LDS_AS uint16_t*)(p + 64u * 72u);
    return F;
}
__host__ __device__ inline bool flatShareFits(uint32_t flat_tris, uint32_t num_surfaces) { return flat_tris <= 32u && num_surfaces - flat_tris <= 32u; }

// Scene::intersect without a BVH (scene.cpp:161-173) for the rays of a whole wave; every lane calls, `valid` = the lane has a ray.
template <bool kCount>
__device__ inline Hit flatIntersectShared(const SceneViewT<true>& sv, bool valid, const Ray& ray, const FlatShare& F, TraceCounters& cnt) {
    Hit best;
    hitInit(best, kDblMax);
    const uint32_t lane = laneId();
    const uint32_t nt = sv.flat_tris, ns = sv.num_surfaces;
    uint32_t masks[2] = {0u, 0u};
    if (valid) {
        cnt.rays++;
        const CullRay cr = cullRay(sv, ray.start, ray.direction);
        masks[0] = cullTriangles(sv.flat_pre, sv.pre_tri_pairs, nt, cr);
        masks[1] = cullSpheres(sv.flat_pre + (size_t)sv.pre_tri_pairs * kTriPairFloats, sv.pre_sph_pairs, ns - nt, cr);
        if (kCount) cnt.prim_tests += (uint32_t)__builtin_popcount(masks[0]) + (uint32_t)__builtin_popcount(masks[1]);
        MCRT_LDS_AS double* r = F.ray + lane * 6u;
        r[0] = ray.start.x; r[1] = ray.start.y; r[2] = ray.start.z;
        r[3] = ray.direction.x; r[4] = ray.direction.y; r[5] = ray.direction.z;
    }
    auto below = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    for (int kind = 0; kind < 2; kind++) {
        uint32_t m = masks[kind];
        const uint32_t base = kind == 0 ? 0u : nt;
        while (waveBallot(m != 0u)) {
            const uint32_t have = (uint32_t)__builtin_popcount(m);
            const uint32_t want = have < 4u ? have : 4u;
            const unsigned long long b0 = waveBallot((want & 1u) != 0u), b1 = waveBallot((want & 2u) != 0u), b2 = waveBallot((want & 4u) != 0u);
            const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
            const uint32_t pre = below(b0) + 2u * below(b1) + 4u * below(b2);
            const uint32_t take = pre >= 64u ? 0u : (want < 64u - pre ? want : 64u - pre);
            uint32_t mm = m;
#pragma unroll
            for (uint32_t jj = 0u; jj < 4u; jj++)
                if (jj < take) {
                    const uint32_t i = lowestBit(mm);
                    mm &= mm - 1u;
                    F.map[pre + jj] = (uint16_t)(lane | ((base + i) << 8));
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < (total < 64u ? total : 64u)) {
                const uint32_t e = F.map[lane];
                const uint32_t owner = e & 0xFFu, prim = e >> 8;
                const MCRT_LDS_AS double* r = F.ray + owner * 6u;
                const d3 o = d3{r[0], r[1], r[2]}, d = d3{r[3], r[4], r[5]};
                double t = 0.0, u = 0.0, v = 0.0;
                const bool ok = kind == 0 ? triangleTestFlat(sv.flat_prim + (size_t)prim * kPrimStride, o, d, t, u, v)
                                          : sphereTestFlat(sv.flat_prim + (size_t)prim * kPrimStride, o, d, t);
                MCRT_LDS_AS double* out = F.res + lane * 3u;
                out[0] = ok ? t : INFINITY;
                out[1] = u;
                out[2] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            mm = m;
#pragma unroll
            for (uint32_t jj = 0u; jj < 4u; jj++)
                if (jj < take) {
                    const uint32_t i = lowestBit(mm);
                    mm &= mm - 1u;
                    const MCRT_LDS_AS double* in = F.res + (pre + jj) * 3u;
                    const double tj = in[0];
                    const uint32_t idx = sv.flat_index[base + i];
                    if (tj < INFINITY && closer(tj, idx, best)) {
                        const bool interp = kind == 0 && sv.flat_prim[(size_t)(base + i) * kPrimStride + 9] >= 2.0;
                        best.t = tj;
                        best.u = interp ? in[1] : 0.0;
                        best.v = interp ? in[2] : 0.0;
                        best.interpolate = interp;
                        best.surface = idx;
                    }
                }
            m = mm;
            __builtin_amdgcn_wave_barrier();  // (the next round's map and results stay behind these reads)
        }
    }
    return best;
}

// One iteration of the while(true) in PathTracer::sampleRay (path-tracer.cpp:14-51) for a whole wave: pathTracerBounce
// (mcrt_integrator.hpp) with both intersections served by flatIntersectShared - every lane calls, `live` = the lane has a path.
// Returns true when the lane's path has ended (what it returns for a lane without a path is not used).
template <bool kCount>
__device__ inline bool pathTracerBounceFlatShared(bool live, PathState& st, RefractionHistory& rh, const SceneViewT<true>& sv, const ShadeViewT<true>& sh,
                                                  const FlatShare& F, TraceCounters& cnt, SobolTab tab) {
    if (live) st.smp.shuffle();                                                         // :23
    const Hit isect = flatIntersectShared<kCount>(sv, live, st.ray, F, cnt);            // :25
    InteractionT<true> ia;
    DirectQuery dq;
    bool want_shadow = false, ended = !live;
    if (live) {
        if (isect.surface == kNoSurface) {                                              // :27-30
            st.radiance = st.radiance + skyColor(st.ray) * st.throughput;
            ended = true;
        } else {
            interactionInit(ia, sh, isect, st.ray, rh.externalIOR(st.ray), st.smp, tab);   // :32
            st.radiance = st.radiance + sampleEmissive(sh, ia, st.ls) * st.throughput;     // :34
            want_shadow = sampleDirectSetup(sh, ia, st.ls, dq, st.smp, tab);               // :35 Integrator::sampleDirect
        }
    }
    const Hit shadow = flatIntersectShared<kCount>(sv, want_shadow, dq.shadow_ray, F, cnt);
    if (ended) return true;
    if (want_shadow) st.radiance = st.radiance + sampleDirectFinish(sh, ia, st.ls, dq, shadow) * st.throughput;
    d3 bsdf_absIdotN;
    if (!interactionSampleBSDF(ia, bsdf_absIdotN, st.ls.bsdf_pdf, st.ray, false, st.smp, tab)) return true;  // :37-40
    st.throughput = st.throughput * (bsdf_absIdotN / st.ls.bsdf_pdf);                   // :42
    if (absorb(st.ray, st.throughput, st.smp, tab)) return true;                         // :44-47
    rh.update(st.ray);                                                                   // :49
    return false;
}
