// WAVEFRONT form of the path tracer for scenes whose BVH lives in HBM (everything that does not fit in LDS).
//
// Why: the persistent megakernel (mcrt_lanesm.hpp) keeps a whole path — ray, Interaction, sampler, NEE
// state, traversal state — in the registers of one lane, 256 VGPRs + scratch, 2 waves per SIMD. On a
// 10-100 MB tree every node visit is an L2 / Infinity-Cache / HBM round trip and two waves cannot hide
// it (measured on metal_bunnies: 57 % of wave cycles waiting, VALU 38 % busy at 40 % lane utilisation).
// Here the path state lives in HBM (a pool of N slots, SoA, 448 B per slot) and a bounce is two launches:
//
//   trace kernel   one work item per queued ray (bounce ray or shadow ray of a slot): reads 48-64 B of ray,
//                  walks the BVH with only the traversal state in registers (4+ waves per SIMD), writes the
//                  hit (32 B). Lanes refill from the queue as soon as their ray is finished, so a wave
//                  never waits for its slowest ray.
//   shade kernel   one lane per slot: finishes the previous bounce's next-event estimate with the shadow
//                  hit, shades the new hit (emission, NEE set-up, BSDF sampling, russian roulette), ends the
//                  path and starts the slot's next sample / next pixel when it died, and queues the slot's
//                  next bounce ray and shadow ray. Fully coalesced, no traversal.
//
// A slot runs the samples of one pixel in order and adds them up in order (Film::deposit order, so the
// per-pixel sums are bit-identical to the megakernels'), then takes the next pixel from the frame's work
// counter. The arithmetic of a path is the same functions in the same order as mcrt_lanesm.hpp:
// smShade / smNeeFinish / travBegin / travInnerStep / travLeafStep.
//
// Everything in this header is plain per-slot / per-ray code shared by the gfx950 kernels (mcrt_kernels.hpp)
// and the host emulation used by the CPU tests (tests/emu).
#pragma once

#include "mcrt_film.hpp"
#include "mcrt_lanesm.hpp"

namespace mcrt {

// shard-local row -> image row (include/mcrt.h: rows dealt in groups of shard_rows)
MCRT_HD uint32_t localToGlobalRow(const mcrt_camera_desc& cam, uint32_t ly) {
    if (cam.shard_count <= 1) return ly;
    const uint32_t g = cam.shard_rows ? cam.shard_rows : 1;
    return ((ly / g) * cam.shard_count + cam.shard_index) * g + ly % g;
}

// ---- slot layout: 8-byte words, word w of slot s at pool[w * n + s]
enum : uint32_t {
    kWfRayO = 0,         // 3  st.ray.start      } photon mapper only: the path tracer reads the bounce ray back from the
    kWfRayD = 3,         // 3  st.ray.direction  } entry it queued for the trace launch (kWfSeq, WfRayQueue::prev_ray)
    kWfMediumIor = 6,
    kWfRefrScale = 7,
    kWfRayBits = 8,      // refraction_level (i32) | depth << 32 | diffuse_depth << 48
    kWfFlags = 9,        // flags (u32) | ls.light << 32
    kWfRadiance = 10,    // 3
    kWfThroughput = 13,  // 3
    kWfBsdfPdf = 16,
    kWfSelectProb = 17,
    // The sampler is a function of (pixel, sample index, number of shuffles) — Sampler::restore — so those three travel,
    // not its five words; the unit's last sample follows from the sample index (units are chunk-aligned: wfUnitEnds).
    kWfUnit = 18,        // px (16 bits) | local row (16 bits) << 16 | sample index of the pixel << 32
    kWfSeq = 19,         // st.smp.sequence | queue entry of the slot's bounce ray << 32
    kWfIors = 20,        // 8  RefractionHistory
    kWfNeeBsdf = 28,     // 3  NeePending (its light is st.ls.light: sampleDirectSetup sets both)
    kWfNeePdf = 31,
    kWfNeeAreaCos = 32,
    kWfNeeThroughput = 33,  // 3
    kWfHit0T = 36,       // closest hit of the bounce ray
    kWfHit0U = 37,
    kWfHit0V = 38,
    kWfHit0S = 39,       // surface | interpolate << 32
    kWfHit1T = 40,       // shadow-ray result
    kWfHit1S = 41,
    kWfWords = 42
    // (RefractionHistory entries beyond the kMaxIors a lane keeps in LDS live in their own buffer, WfFrame::iors_deep, [row][slot]:
    // read and written in place, never copied, and as many rows as the deepest path of the frame needs - round 6)
};
enum : uint32_t {
    kWfAlive = 1u,        // the bounce ray in the slot was traced for this iteration
    kWfHavePixel = 2u,
    kWfNeePending = 4u,   // the shadow ray in the slot was traced for this iteration
    kWfDone = 8u,         // no pixels left for this slot
    kWfDirac = 16u,
    kWfRefraction = 32u,
    kWfRhShift = 16,      // RefractionHistory::size in bits 16..31 (bits 8..13 until round 6, when histories could not pass 32 entries)
    kWfEstWait = 1u << 14,  // photon mapper: the hit in the slot waits for its radiance estimates
    kWfEstNeedG = 1u << 15  // ... the global estimate too (the path ends with it)
};

struct WfPool {
    unsigned long long* w;
    uint32_t n;
    MCRT_HD unsigned long long getu(uint32_t word, uint32_t slot) const { return w[(size_t)word * n + slot]; }
    MCRT_HD void setu(uint32_t word, uint32_t slot, unsigned long long v) const { w[(size_t)word * n + slot] = v; }
    MCRT_HD double getd(uint32_t word, uint32_t slot) const { return bitsD(getu(word, slot)); }
    MCRT_HD void setd(uint32_t word, uint32_t slot, double v) const { setu(word, slot, dBits(v)); }
    MCRT_HD d3 get3(uint32_t word, uint32_t slot) const { return d3{getd(word, slot), getd(word + 1, slot), getd(word + 2, slot)}; }
    MCRT_HD void set3(uint32_t word, uint32_t slot, d3 v) const {
        setd(word, slot, v.x);
        setd(word + 1, slot, v.y);
        setd(word + 2, slot, v.z);
    }
};

// One pass of a frame: the local rows [row_base, row_end). Work units are sample chunks of a pixel, exactly as in the
// megakernels (RenderParams, mcrt_kernels.hpp): unit w -> pixel item w >> chunk_shift (8x8 tiles), chunk w & mask; every
// finished sample goes to `samples` ([spp][pass_pixels][3]) and sampleResolveKernel adds them up in sample order. With whole
// pixels as units the slots that drew cheap pixels idle while the expensive pixels run their spp samples: on metal_bunnies
// the last 12 % of a frame's time traced 3 % of its rays.
struct WfFrame {
    mcrt_camera_desc cam;
    uint32_t global_seed, spp, tiles_x;
    uint32_t chunk, chunk_shift;    // samples per unit, log2(units per pixel)
    uint32_t row_base, row_end;
    unsigned long long pass_pixels; // (row_end - row_base) * width
    unsigned long long work_items;  // tiles_x * tile rows of the pass * 64 << chunk_shift
    double* samples;
    FilmView film;                  // type != MCRT_FILM_BOX: samples are splatted into film.blob instead (mcrt_film.hpp)
    // RefractionHistory entries kMaxIors .. kMaxIors + iors_deep_rows - 1 of every slot, [row][slot] (ray.cpp:74-98 is an unbounded
    // vector: a frame whose paths nest deeper than the rows it was given reports it, and the host renders it again with four times
    // the rows - mcrt_render_finish)
    double* iors_deep;
    uint32_t iors_deep_rows;
};

// A unit is the samples [c * chunk, min((c + 1) * chunk, spp)) of a pixel: `sample`, just incremented, is past its unit's end
MCRT_HD bool wfUnitEnds(const WfFrame& fr, uint32_t sample) { return sample >= fr.spp || sample % fr.chunk == 0u; }

// ---- trace side: work item = slot * 2 + port (0 = bounce ray, closest hit; 1 = shadow ray, bounded any-hit).
// The rays themselves travel in QUEUE order (WfRayQueue: eight planes of doubles + the light of a shadow ray, entry w
// next to entry w + 1), so a refill of the trace kernel — the idle lanes of a wave take consecutive entries — reads
// consecutive words: one coalesced round trip after the pop, instead of the item, then scattered slot words.
struct WfRayQueue {
    uint32_t* item;    // [cap] slot * 2 + port
    uint32_t* light;   // [cap] shadow ray: the surface aimed at
    double* ray;       // [8][cap] o.xyz, d.xyz, shadow ray: distance to the light point (a bounce ray's range is [0, max): not stored)
    const double* prev_ray;  // the planes the previous shade launch filled (the two sets alternate): a slot's bounce ray is
                             // read back from its entry there instead of being kept a second time in the pool
    uint64_t cap;
    MCRT_HD void prevRay(uint32_t w, d3& o, d3& d) const {
        o = d3{prev_ray[0 * cap + w], prev_ray[1 * cap + w], prev_ray[2 * cap + w]};
        d = d3{prev_ray[3 * cap + w], prev_ray[4 * cap + w], prev_ray[5 * cap + w]};
    }
    MCRT_HD void put(uint64_t w, uint32_t it, d3 o, d3 d, double dist, uint32_t l) const {
        item[w] = it;
        ray[0 * cap + w] = o.x; ray[1 * cap + w] = o.y; ray[2 * cap + w] = o.z;
        ray[3 * cap + w] = d.x; ray[4 * cap + w] = d.y; ray[5 * cap + w] = d.z;
        if (it & 1u) {
            light[w] = l;
            ray[6 * cap + w] = dist;
        }
    }
    MCRT_HD uint32_t get(uint64_t w, d3& o, d3& d, bool& shadow, ShadowQuery& sq) const {
        const uint32_t it = item[w];
        o = d3{ray[0 * cap + w], ray[1 * cap + w], ray[2 * cap + w]};
        d = d3{ray[3 * cap + w], ray[4 * cap + w], ray[5 * cap + w]};
        // (read for every entry, used by shadow rays: a load that waited for the item word would be a second round trip per refill)
        const double dist = ray[6 * cap + w];
        const uint32_t l = light[w];
        shadow = (it & 1u) != 0u;
        sq.setRange(dist);  // sampleDirectSetup's bounds, from the same distance
        sq.light = l;
        if (!shadow) {
            sq.t_near = 0.0;
            sq.t_far = kDblMax;
            sq.light = kNoSurface;
        }
        return it;
    }
};

MCRT_HD void wfStoreHit(const WfPool& P, uint32_t item, const Hit& h) {
    const uint32_t slot = item >> 1;
    if (item & 1u) {
        P.setd(kWfHit1T, slot, h.t);
        P.setu(kWfHit1S, slot, h.surface);
    } else {
        P.setd(kWfHit0T, slot, h.t);
        P.setd(kWfHit0U, slot, h.u);
        P.setd(kWfHit0V, slot, h.v);
        P.setu(kWfHit0S, slot, (unsigned long long)h.surface | ((unsigned long long)(h.interpolate ? 1u : 0u) << 32));
    }
}

// ---- photon mapper (PhotonMapper::sampleRay, photon-mapper.cpp:279-341) in wavefront form. A non-specular hit needs
// the caustic estimate (and, when it is the path's last hit, the global one) before the bounce can go on: the shade
// launch files an estimate request and leaves the slot waiting; a kNN launch (one query per wave, mcrt_waveknn.hpp, at
// 6 waves per SIMD instead of the megakernel's 2) writes the k photons of every requested search; the next shade launch
// rebuilds the Interaction from the untouched ray / hit / sampler words, sums the photons' contributions per lane
// (estimateCausticRadiance / estimateGlobalRadiance, photon-mapper.cpp:343-391) and finishes the bounce.
struct WfPmView {
    const float* photons[2];     // [n][8] global map, caustic map
    const uint32_t* res_n;       // [2][slots] photons found by the search of (map, slot)
    const double* res_r2;        // [2][slots] largest squared distance among them (photons.top().distance2)
    const uint32_t* res_idx;     // [2][k][slots]
    const double* res_d2;        // [2][k][slots]
    uint32_t k;
    bool direct_visualization;
    // estimates evaluated by the kNN launch itself (wfKnnKernel with staged Interactions, the wave-cooperative evaluation of
    // mcrt_waveknn.hpp): [slots][6] caustic rgb, global rgb; null: the k photons come back and the shade launch sums them per lane
    const double* est;
};

template <bool L>
MCRT_HD d3 wfPhotonEstimate(const WfPmView& pm, uint32_t n_slots, uint32_t slot, int map, const InteractionT<L>& ia) {
    const uint32_t n = pm.res_n[(size_t)map * n_slots + slot];
    if (n == 0) return splat(0.0);
    const double r2 = pm.res_r2[(size_t)map * n_slots + slot];
    const double inv_max_squared_radius = 1.0 / r2;
    const float* photons = pm.photons[map];
    d3 radiance = splat(0.0);
    for (uint32_t i = 0; i < n; i++) {
        const size_t at = ((size_t)map * pm.k + i) * n_slots + slot;
        const float* ph = photons + (size_t)pm.res_idx[at] * 8;
        d3 bsdf_absIdotN;
        double bsdf_pdf;
        if (interactionBSDF(ia, bsdf_absIdotN, photonDirection(ph), bsdf_pdf)) {
            const d3 flux = d3{(double)ph[0], (double)ph[1], (double)ph[2]};
            if (map == 1) {  // caustic: cone filter, photon-mapper.cpp:380-388
                const double wp = gmax(0.0, 1.0 - sqrt(pm.res_d2[at] * inv_max_squared_radius));
                radiance = radiance + (flux * bsdf_absIdotN * wp) / bsdf_pdf;
            } else {
                radiance = radiance + flux * bsdf_absIdotN / bsdf_pdf;
            }
        }
    }
    return map == 1 ? 3.0 * radiance * inv_max_squared_radius * kInvPi : radiance / (r2 * kPi);
}

// ---- shade side. Env supplies the three places where lanes cooperate:
//   bool any(bool)                      true if the predicate holds for any lane of the wave (host: identity)
//   unsigned long long pop(bool need)   next index of the frame's pixel work counter for the lanes that need one
//   uint32_t push(slot, p0, p1, rays...)  queue the slot's bounce ray / shadow ray (given: origin, direction, shadow ray: distance and light) for the next
//                                       trace launch; returns the entry the bounce ray went to
//   void prevRay(entry, o, d)           the bounce ray the previous shade launch queued at `entry`
//   void filmAdd(double*, double)       accumulate into a film splat (any lane, any time; atomic on the GPU)
//   void request(slot, want, global)    photon mapper: queue the slot's caustic (and global) search for the next kNN launch
//   void stage(slot, ia)                photon mapper: leave the hit's Interaction where the kNN launch finds it (any lane; no-op when that launch only searches)
// All but filmAdd are called by every lane of the wave, at the same place.
MCRT_HD unsigned long long wfSlotFlags(const WfPool& P, uint32_t slot, bool valid) { return valid ? P.getu(kWfFlags, slot) : (unsigned long long)kWfDone; }

// fw: wfSlotFlags(P, slot, valid), read by the caller (the kernel asks for it together with its staging loads)
template <bool L, bool kPhoton, class Env>
MCRT_HD void wfShadeSlot(Env& env, const WfPool& P, uint32_t slot, unsigned long long fw, const WfFrame& fr, const ShadeViewT<L>& sh,
                         RefractionHistory& rh, SobolTab tab, uint32_t& paths, const WfPmView* pm = nullptr) {
    const uint32_t flags = (uint32_t)fw;
    const bool was_done = (flags & kWfDone) != 0u;
    bool done = was_done;
    bool alive = (flags & kWfAlive) != 0u, have_pixel = (flags & kWfHavePixel) != 0u;
    const bool nee_was_pending = (flags & kWfNeePending) != 0u;

    // histories deeper than the kMaxIors entries a lane has in LDS go on in the frame's deep rows
    rh.giors = fr.iors_deep;
    rh.glane = slot;
    rh.gstride = P.n;
    rh.lds_depth = kMaxIors;
    rh.max_depth = kMaxIors + (int)fr.iors_deep_rows;

    PathState st;
    NeePending nee;
    nee.pending = false;
    uint32_t px = 0, ly = 0, sample = 0;
    bool need_pixel = false;
    bool want_estimate = false, need_g = false;  // photon mapper: this hit needs its radiance estimates first
    // the shadow ray this call may queue (written into the ray queue by env.push, in queue order)
    d3 sh_o = splat(0.0), sh_d = splat(0.0);
    double sh_dist = 0.0;
    auto loadHit0 = [&]() {
        Hit h;
        h.t = P.getd(kWfHit0T, slot);
        h.u = P.getd(kWfHit0U, slot);
        h.v = P.getd(kWfHit0V, slot);
        const unsigned long long hs = P.getu(kWfHit0S, slot);
        h.surface = (uint32_t)hs;
        h.interpolate = (hs >> 32) != 0ull;
        return h;
    };

    Hit hit0;
    hit0.t = hit0.u = hit0.v = 0.0;
    hit0.surface = kNoSurface;
    hit0.interpolate = false;
    if (!was_done) {
        // ---- load the slot. The launch is bound by the length of its chain of dependent round trips to memory, not by bytes or
        // arithmetic: every word the slot may need is asked for here, before the first of them is used.
        const unsigned long long sq_w = P.getu(kWfSeq, slot);
        const unsigned long long uw = P.getu(kWfUnit, slot);
        const unsigned long long rb = P.getu(kWfRayBits, slot);
        st.ray.medium_ior = P.getd(kWfMediumIor, slot);
        st.ray.refraction_scale = P.getd(kWfRefrScale, slot);
        st.radiance = P.get3(kWfRadiance, slot);
        st.throughput = P.get3(kWfThroughput, slot);
        st.ls.bsdf_pdf = P.getd(kWfBsdfPdf, slot);
        st.ls.select_probability = P.getd(kWfSelectProb, slot);
        st.ls.light = (uint32_t)(fw >> 32);
        const double ior0 = P.getd(kWfIors, slot), ior1 = P.getd(kWfIors + 1u, slot);  // (deeper histories: below)
        hit0 = loadHit0();
        Hit sh_hit;
        sh_hit.t = 0.0;
        sh_hit.u = sh_hit.v = 0.0;
        sh_hit.surface = kNoSurface;
        sh_hit.interpolate = false;
        uint32_t light_material = 0u;
        if (nee_was_pending) {
            nee.light = st.ls.light;
            nee.bsdf_absIdotN = P.get3(kWfNeeBsdf, slot);
            nee.bsdf_pdf = P.getd(kWfNeePdf, slot);
            nee.area_cos = P.getd(kWfNeeAreaCos, slot);
            nee.throughput = P.get3(kWfNeeThroughput, slot);
            sh_hit.t = P.getd(kWfHit1T, slot);
            sh_hit.surface = (uint32_t)P.getu(kWfHit1S, slot);
            light_material = sh.surf_material[st.ls.light];
        }
        if constexpr (kPhoton) {  // a hit that waits for its estimates is not re-queued: the ray stays in the pool
            st.ray.start = P.get3(kWfRayO, slot);
            st.ray.direction = P.get3(kWfRayD, slot);
        } else {
            st.ray.start = st.ray.direction = splat(0.0);
            if (alive) env.prevRay((uint32_t)(sq_w >> 32), st.ray.start, st.ray.direction);
        }
        st.ray.inv_direction = rcp3(st.ray.direction);
        st.ray.refraction_level = (int)(uint32_t)rb;
        st.ray.depth = (uint16_t)(rb >> 32);
        st.ray.diffuse_depth = (uint16_t)(rb >> 48);
        st.ray.dirac_delta = (flags & kWfDirac) != 0u;
        st.ray.refraction = (flags & kWfRefraction) != 0u;
        px = (uint32_t)uw & 0xFFFFu;
        ly = ((uint32_t)uw >> 16) & 0xFFFFu;
        sample = (uint32_t)(uw >> 32);
        st.smp.restore(fr.global_seed, localToGlobalRow(fr.cam, ly) * fr.cam.width + px, sample, (uint32_t)sq_w);
        rh.size = (int)(flags >> kWfRhShift);
        rh.put(0, ior0);
        rh.put(1, ior1);
        if (env.any(rh.size > 2))
            for (int i = 2; i < kMaxIors; i++)
                if (i < rh.size) rh.put(i, P.getd(kWfIors + (uint32_t)i, slot));

        // ---- second half of the previous bounce's Integrator::sampleDirect, now that its shadow ray is back
        if (nee_was_pending) smNeeFinish(st, sh, nee, sh_hit, light_material);

        // ---- this bounce (path-tracer.cpp:27-49 / photon-mapper.cpp:288-340)
        bool ended;
        Ray shadow_ray;
        ShadowQuery shadow_q;
        if constexpr (!kPhoton) {
            if (alive) {
                const Hit h = hit0;
                alive = smShade(st, rh, sh, h, nee, shadow_ray, shadow_q, tab);
                if (alive) st.smp.shuffle();  // top of the next while(true) iteration (path-tracer.cpp:23)
                ended = !alive && !nee.pending;
            } else {
                nee.pending = false;
                ended = have_pixel;  // the path died at its previous bounce; its last NEE has just been added
            }
        } else {
            const bool waiting = (flags & kWfEstWait) != 0u;
            if (alive) {
                const Hit h = hit0;
                bool go_on = false;  // finish the bounce with `ia` (sampling, russian roulette)
                InteractionT<L> ia;
                if (h.surface == kNoSurface) {
                    alive = false;  // no sky in photon mode (photon-mapper.cpp:292-295)
                } else {
                    // (after a wait: the same Interaction again — ray, hit and sampler have not moved)
                    interactionInit(ia, sh, h, st.ray, rh.externalIOR(st.ray), st.smp, tab);
                    if (!waiting) {
                        st.radiance = st.radiance + sampleEmissive(sh, ia, st.ls) * st.throughput;  // :299
                        if (ia.dirac_delta) {                                                        // :301-306
                            if (!st.ray.dirac_delta && st.ray.depth != 0) alive = false;
                            else go_on = true;
                        } else {
                            want_estimate = true;                                                    // :315 (+ :327-330)
                            need_g = !(!pm->direct_visualization && (st.ray.dirac_delta || st.ray.depth == 0));
                            env.stage(slot, ia);
                        }
                    } else {
                        const d3 est_c = pm->est ? ld3(pm->est + (size_t)slot * 6) : wfPhotonEstimate(*pm, P.n, slot, 1, ia);
                        st.radiance = st.radiance + est_c * st.throughput;  // :315
                        if (flags & kWfEstNeedG) {
                            const d3 est_g = pm->est ? ld3(pm->est + (size_t)slot * 6 + 3) : wfPhotonEstimate(*pm, P.n, slot, 0, ia);
                            st.radiance = st.radiance + est_g * st.throughput;  // :330: the path ends here
                            alive = false;
                        } else {
                            go_on = true;
                        }
                    }
                }
                if (go_on) {
                    alive = smContinue(st, rh, sh, ia, !ia.dirac_delta, nee, shadow_ray, shadow_q, tab);
                    if (alive) st.smp.shuffle();  // photon-mapper.cpp:290, next bounce
                }
                ended = !alive && !nee.pending && !want_estimate;
            } else {
                nee.pending = false;
                ended = have_pixel;
            }
        }
        if (nee.pending) {
            sh_o = shadow_ray.start;
            sh_d = shadow_ray.direction;
            sh_dist = shadow_q.dist;
            P.set3(kWfNeeBsdf, slot, nee.bsdf_absIdotN);
            P.setd(kWfNeePdf, slot, nee.bsdf_pdf);
            P.setd(kWfNeeAreaCos, slot, nee.area_cos);
            P.set3(kWfNeeThroughput, slot, nee.throughput);
        }
        if (ended && fr.film.type != MCRT_FILM_BOX) {  // Film::deposit with a reconstruction filter: a splat per sample
            Sampler at_start = st.smp;  // the sample's pixel position: the two draws of camera.cpp:79-80, sampler as it was then
            at_start.setIndex(sample);
            const double fx = (double)px + at_start.get(kDimPixel, tab);
            const double fy = (double)localToGlobalRow(fr.cam, ly) + at_start.get(kDimPixel + 1, tab);  // splats land in any shard's rows
            filmDeposit(fr.film, fx, fy, st.radiance, [&](double* a, double v) { env.filmAdd(a, v); });
            if (wfUnitEnds(fr, ++sample)) have_pixel = false;
        } else if (ended) {  // Film::deposit with the box filter (own pixel, weight 1): the addend, kept per sample
            double* o = fr.samples + ((size_t)sample * fr.pass_pixels + ((size_t)(ly - fr.row_base) * fr.cam.width + px)) * 3;
            o[0] = st.radiance.x * 1.0;
            o[1] = st.radiance.y * 1.0;
            o[2] = st.radiance.z * 1.0;
            if (wfUnitEnds(fr, ++sample)) have_pixel = false;
        }
        need_pixel = !alive && !nee.pending && !have_pixel;
    }

    // ---- a new work unit from the pass's counter (pixels in 8x8 tiles; edge tiles hold positions outside the image)
    while (env.any(need_pixel)) {
        const unsigned long long w = env.pop(need_pixel);
        if (need_pixel) {
            if (w >= fr.work_items) {
                done = true;
                need_pixel = false;
            } else {
                const unsigned long long item = w >> fr.chunk_shift;
                const uint32_t c = (uint32_t)(w - (item << fr.chunk_shift));
                const uint32_t tile = (uint32_t)(item >> 6), in = (uint32_t)(item & 63u);
                const uint32_t lx = (tile % fr.tiles_x) * 8u + (in & 7u);
                const uint32_t y = fr.row_base + (tile / fr.tiles_x) * 8u + (in >> 3);
                const uint32_t first = c * fr.chunk;
                if (lx < fr.cam.width && y < fr.row_end && first < fr.spp) {
                    px = lx;
                    ly = y;
                    have_pixel = true;
                    need_pixel = false;
                    sample = first;
                    st.smp.initiate(fr.global_seed, localToGlobalRow(fr.cam, ly) * fr.cam.width + px);  // camera.cpp:73
                }
            }
        }
    }

    // ---- next sample of the pixel (camera.cpp:77-96)
    if (!was_done && !done && !alive && !nee.pending && have_pixel) {
        st.smp.setIndex(sample);
        pathBegin(st, rh, cameraRay(fr.cam, sh.scene_ior, px, localToGlobalRow(fr.cam, ly), st.smp, tab));
        paths++;
        st.smp.shuffle();  // path-tracer.cpp:23, first bounce
        alive = true;
    }

    const uint32_t entry = env.push(slot, !was_done && !done && alive && !want_estimate, !was_done && nee.pending, st.ray.start, st.ray.direction,
                                    sh_o, sh_d, sh_dist, nee.pending ? nee.light : kNoSurface);

    // ---- store the slot
    if (!was_done) {
        uint32_t nf = (alive ? kWfAlive : 0u) | (have_pixel ? kWfHavePixel : 0u) | (nee.pending ? kWfNeePending : 0u) | (done ? kWfDone : 0u) |
                      (st.ray.dirac_delta ? kWfDirac : 0u) | (st.ray.refraction ? kWfRefraction : 0u) | ((uint32_t)rh.size << kWfRhShift) |
                      (want_estimate ? kWfEstWait : 0u) | (want_estimate && need_g ? kWfEstNeedG : 0u);
        P.setu(kWfFlags, slot, (unsigned long long)nf | ((unsigned long long)st.ls.light << 32));
        if (!done) {
            if constexpr (kPhoton) {
                P.set3(kWfRayO, slot, st.ray.start);
                P.set3(kWfRayD, slot, st.ray.direction);
            }
            P.setd(kWfMediumIor, slot, st.ray.medium_ior);
            P.setd(kWfRefrScale, slot, st.ray.refraction_scale);
            P.setu(kWfRayBits, slot, (unsigned long long)(uint32_t)st.ray.refraction_level | ((unsigned long long)st.ray.depth << 32) |
                                          ((unsigned long long)st.ray.diffuse_depth << 48));
            P.set3(kWfRadiance, slot, st.radiance);
            P.set3(kWfThroughput, slot, st.throughput);
            P.setd(kWfBsdfPdf, slot, st.ls.bsdf_pdf);
            P.setd(kWfSelectProb, slot, st.ls.select_probability);
            P.setu(kWfUnit, slot, (unsigned long long)(px | (ly << 16)) | ((unsigned long long)sample << 32));
            P.setu(kWfSeq, slot, (unsigned long long)st.smp.sequence | ((unsigned long long)entry << 32));
            for (int i = 0; i < kMaxIors; i++)
                if (i < rh.size) P.setd(kWfIors + (uint32_t)i, slot, rh.at(i));
        }
    }
    if constexpr (kPhoton) env.request(slot, want_estimate, need_g);
}

}  // namespace mcrt
