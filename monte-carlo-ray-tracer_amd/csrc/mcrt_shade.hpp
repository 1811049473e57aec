// Per-lane shading math (the §8 rows a9-a20): Interaction, lobe selection, BSDF eval/sample,
// GGX, Fresnel, Oren-Nayar, next-event estimation set-up, russian roulette, refraction history.
// Every function keeps the reference's operation order (citations are path:line under
// /root/reference/source); see mcrt_math.hpp for why.
#pragma once

#include "../../include/mcrt.h"
#include "mcrt_sampler.hpp"
#include "mcrt_scene.hpp"
#include "mcrt_libm.hpp"

namespace mcrt {

// Material FEATURES a translation unit compiles out (round 6). The rough-diffuse (Oren-Nayar), rough-specular (GGX) and conductor-Fresnel
// branches are the register peak of every shading kernel whether or not a scene has such a material: compiled for "none of the three" the
// flat megakernel's 512-lane form spills nothing (13 registers with them), the photon-mapping kernels 100+ fewer, and a frame of
// hexagon_room / water_caustics renders 2-7 % faster with the same bits (profiles/r06_ab_feature_strip*.log). csrc/mcrt_hip_lean.hip
// compiles the default path's kernels once more with MCRT_MAT_FEATURES_OFF = those three bits; launchRender picks them for scenes whose
// materials carry none of the bits (HostLayout / mcrt_upload_scene: the OR of all material flags). Default: nothing compiled out.
#ifndef MCRT_MAT_FEATURES_OFF
#define MCRT_MAT_FEATURES_OFF 0u
#endif
constexpr uint32_t kMatFeaturesOff = (uint32_t)(MCRT_MAT_FEATURES_OFF);
// a material's flag bits as this translation unit sees them
MCRT_HD constexpr uint32_t matFlags(uint32_t flags) { return flags & ~kMatFeaturesOff; }

// Per-surface data needed after the closest hit is known, plus materials and lights.
// L: the arrays are the workgroup's LDS copies (small scenes) instead of global memory.
template <bool L>
struct ShadeViewT {
    cptr<double, L> surf_v;           // [n][9] triangle v0,v1,v2 / sphere origin,radius
    cptr<double, L> surf_normal;      // [n][3] Triangle::normal_
    cptr<double, L> surf_vn;          // [n][9] vertex normals (may be null when no surface interpolates)
    cptr<double, L> surf_area;        // [n]
    cptr<uint32_t, L> surf_material;  // [n]
    cptr<uint8_t, L> surf_kind;       // [n]
    const double* surf_rec;           // [n][16] scenes in memory: normal, material | kind << 32, vertex normals in ONE 128-byte line
                                      // (HostLayout::shade_rec); from the five arrays a hit moves four to six lines
    cptr<mcrt_material, L> materials;
    uint32_t num_lights;
    cptr<uint32_t, L> light_surface;
    cptr<double, L> light_cdf;
    double scene_ior;
};

// ------------------------------------------------------------------ Fresnel (material/fresnel.cpp)
MCRT_HD double fresnelDielectric(double n1, double n2, double cos_theta) {  // :16-27
    double g2 = sq(n2 / n1) + sq(cos_theta) - 1.0;
    if (g2 < 0.0) return 1.0;
    double g = sqrt(g2);
    double g_p_c = g + cos_theta;
    double g_m_c = g - cos_theta;
    return 0.5 * sq(g_m_c / g_p_c) * (1.0 + sq((g_p_c * cos_theta - 1.0) / (g_m_c * cos_theta + 1.0)));
}

MCRT_HD d3 fresnelConductor(double n1, d3 real, d3 imag, double cos_theta) {  // :30-49
    double cos_theta2 = sq(cos_theta);
    double sin_theta2 = 1.0 - cos_theta2;
    d3 r = real / n1, k = imag / n1;
    d3 eta2 = r * r;
    d3 eta_k2 = k * k;
    d3 t0 = eta2 - eta_k2 - sin_theta2;
    d3 a2_p_b2 = sqrt3(t0 * t0 + 4.0 * eta2 * eta_k2);
    d3 t1 = a2_p_b2 + cos_theta2;
    d3 t2 = (2.0 * cos_theta) * sqrt3(0.5 * (a2_p_b2 + t0));
    d3 r_perpendicular = (t1 - t2) / (t1 + t2);
    d3 t3 = cos_theta2 * a2_p_b2 + sq(sin_theta2);
    d3 t4 = t2 * sin_theta2;
    d3 r_parallel = r_perpendicular * (t3 - t4) / (t3 + t4);
    return (r_parallel + r_perpendicular) * 0.5;
}

// ------------------------------------------------------------------ GGX (material/ggx.cpp)
MCRT_HD double ggxD(d3 m, double ax, double ay) {  // :21-24
    return 1.0 / (kPi * ax * ay * sq(sq(m.x / ax) + sq(m.y / ay) + sq(m.z)));
}
MCRT_HD double ggxLambda(d3 wo, double ax, double ay) {  // :31-34
    return (-1.0 + sqrt(1.0 + (sq(ax * wo.x) + sq(ay * wo.y)) / (sq(wo.z)))) / 2.0;
}
MCRT_HD double ggxG1(d3 wo, double ax, double ay) { return 1.0 / (1.0 + ggxLambda(wo, ax, ay)); }  // :36-39
MCRT_HD double ggxG2(d3 wi, d3 wo, double ax, double ay) {                                         // :41-44
    return 1.0 / (1.0 + ggxLambda(wo, ax, ay) + ggxLambda(wi, ax, ay));
}
MCRT_HD double ggxDV(d3 m, d3 wo, double ax, double ay) {  // :26-29
    return ggxG1(wo, ax, ay) * dot(wo, m) * ggxD(m, ax, ay) / wo.z;
}
MCRT_HD double ggxReflection(d3 wi, d3 wo, double ax, double ay, double& pdf) {  // :46-52
    d3 m = normalize(wo + wi);
    pdf = ggxDV(m, wo, ax, ay) / (4.0 * dot(m, wo));
    return ggxD(m, ax, ay) * ggxG2(wi, wo, ax, ay) / (4.0 * wo.z * wi.z);
}
MCRT_HD double ggxTransmission(d3 wi, d3 wo, double n1, double n2, double ax, double ay, double& pdf) {  // :54-65
    d3 m = wo * n1 + wi * n2;
    double m_length2 = dot(m, m);
    m = m / sqrt(m_length2);
    if (n1 < n2) m = -m;
    double dm_dwi = sq(n2) * fabs(dot(wi, m)) / m_length2;
    pdf = ggxDV(m, wo, ax, ay) * dm_dwi;
    return fabs(ggxG2(wi, wo, ax, ay) * ggxD(m, ax, ay) * dot(wo, m) * dm_dwi / (wo.z * wi.z));
}
template <bool kLdsTab = true>
MCRT_HD d3 ggxVisibleMicrofacet(double u, double v, d3 wo, double ax, double ay) {  // :67-88 (Heitz 2018)
    d3 Vh = normalize(d3{ax * wo.x, ay * wo.y, wo.z});
    double len2 = sq(Vh.x) + sq(Vh.y);
    d3 T1 = len2 > 0.0 ? d3{-Vh.y, Vh.x, 0.0} * (1.0 / sqrt(len2)) : d3{1.0, 0.0, 0.0};
    d3 T2 = cross(Vh, T1);
    double r = sqrt(u);
    double phi = v * kTwoPi;
    double sin_phi, cos_phi;
    refSinCos<kLdsTab>(phi, sin_phi, cos_phi);  // glibc's sincos, bit for bit (mcrt_libm.hpp)
    double t1 = r * cos_phi;
    double t2 = r * sin_phi;
    double s = 0.5 * (1.0 + Vh.z);
    t2 = (1.0 - s) * sqrt(1.0 - sq(t1)) + s * t2;
    d3 Nh = t1 * T1 + t2 * T2 + sqrt(gmax(0.0, 1.0 - sq(t1) - sq(t2))) * Vh;
    return normalize(d3{ax * Nh.x, ay * Nh.y, gmax(0.0, Nh.z)});
}

// ------------------------------------------------------------------ Material (material/material.cpp)
template <class M>
MCRT_HD d3 matLambertian(const M& m) { return ld3(m.reflectance) * kInvPi; }  // :76-79
template <class M>
MCRT_HD d3 matOrenNayar(const M& m, d3 wi, d3 wo) {                           // :82-95
    double cos_delta_phi =
        gmin(gmax((wi.x * wo.x + wi.y * wo.y) / sqrt((sq(wi.x) + sq(wi.y)) * (sq(wo.x) + sq(wo.y))), 0.0), 1.0);
    double D = sqrt((1.0 - sq(wi.z)) * (1.0 - sq(wo.z))) / gmax(wi.z, wo.z);
    return matLambertian(m) * (m.A + m.B * cos_delta_phi * D);
}
template <class M>
MCRT_HD d3 matDiffuseReflection(const M& m, d3 wi, d3 wo, double& pdf) {  // :17-27
    if (wi.z < 0.0) {
        pdf = 0.0;
        return splat(0.0);
    }
    pdf = wi.z * kInvPi;
    return (matFlags(m.flags) & MCRT_MAT_ROUGH) ? matOrenNayar(m, wi, wo) : matLambertian(m);
}
template <class M>
MCRT_HD d3 matSpecularReflection(const M& m, d3 wi, d3 wo, double& pdf) {  // :29-45
    if (wi.z < 0.0) {
        pdf = 0.0;
        return splat(0.0);
    }
    if (matFlags(m.flags) & MCRT_MAT_ROUGH_SPECULAR) return ld3(m.specular_reflectance) * ggxReflection(wi, wo, m.a[0], m.a[1], pdf);
    pdf = 1.0;
    return ld3(m.specular_reflectance) / fabs(wi.z);
}
template <class M>
MCRT_HD d3 matSpecularTransmission(const M& m, d3 wi, d3 wo, double n1, double n2, double& pdf,
                                   bool inside, bool flux) {  // :47-69
    if (wi.z > 0.0) {
        pdf = 0.0;
        return splat(0.0);
    }
    d3 btdf = !inside ? ld3(m.transmittance) : splat(1.0);
    if (matFlags(m.flags) & MCRT_MAT_ROUGH_SPECULAR) {
        btdf = btdf * ggxTransmission(wi, wo, n1, n2, m.a[0], m.a[1], pdf);
        if (flux) btdf = btdf * sq(n2 / n1);
    } else {
        pdf = 1.0;
        btdf = btdf * (ld3(m.transmittance) / fabs(wi.z));
        if (!flux) btdf = btdf * sq(n1 / n2);
    }
    return btdf;
}

// ------------------------------------------------------------------ surfaces
template <bool L>
MCRT_HD d3 surfNormal(const ShadeViewT<L>& sh, uint32_t i, d3 pos) {  // triangle.cpp:99-102, sphere.cpp:46-49
    if (sh.surf_kind[i] == MCRT_SURF_SPHERE) {
        cptr<double, L> p = sh.surf_v + (size_t)i * 9;
        return (pos - ld3(p)) / p[3];
    }
    if constexpr (QuadricsIn<L>::value) {
        if (sh.surf_kind[i] == MCRT_SURF_QUADRIC) return quadricNormal(quadricPtr(sh.surf_v[(size_t)i * 9]), pos);  // quadric.cpp:127-130
    }
    return ld3(sh.surf_normal + (size_t)i * 3);
}
template <bool L>
MCRT_HD d3 surfInterpolatedNormal(const ShadeViewT<L>& sh, uint32_t i, double u, double v) {  // triangle.cpp:109-113
    cptr<double, L> n = sh.surf_vn + (size_t)i * 9;
    return normalize((1.0 - u - v) * ld3(n) + u * ld3(n + 3) + v * ld3(n + 6));
}
template <bool L>
MCRT_HD d3 surfSample(const ShadeViewT<L>& sh, uint32_t i, double u, double v) {  // triangle.cpp:93-97, sphere.cpp:37-44
    cptr<double, L> p = sh.surf_v + (size_t)i * 9;
    if (sh.surf_kind[i] == MCRT_SURF_SPHERE) {
        double z = 1.0 - 2.0 * u;
        double r = sqrt(1.0 - sq(z));
        double phi = kTwoPi * v;
        double sin_phi, cos_phi;
        refSinCos<!L>(phi, sin_phi, cos_phi);
        return ld3(p) + p[3] * d3{r * cos_phi, r * sin_phi, z};
    }
    double su = sqrt(u);
    return (1 - su) * ld3(p) + ((1 - v) * su) * ld3(p + 3) + (v * su) * ld3(p + 6);
}

// A surface's record asked for in ONE round trip (kind, the nine vertex words, the face normal, the area) - the kernels that read
// the scene from memory are bound by the number of dependent trips a bounce makes, and kind -> vertices -> normal -> area one after
// the other is four of them. Same arithmetic on the fetched values as surfSample / surfNormal (a triangle pays 32 bytes it does not use
// in surfNormalOf, a sphere 64).
struct SurfRec {
    uint32_t kind;
    double v[9];
    d3 normal;
};
template <bool L>
MCRT_HD SurfRec surfFetch(const ShadeViewT<L>& sh, uint32_t i, bool vertices) {
    SurfRec r;
    r.kind = sh.surf_kind[i];
    cptr<double, L> p = sh.surf_v + (size_t)i * 9;
    for (int k = 0; k < 4; k++) r.v[k] = p[k];
    for (int k = 4; k < 9; k++) r.v[k] = vertices ? p[k] : 0.0;
    r.normal = ld3(sh.surf_normal + (size_t)i * 3);
    return r;
}
template <bool L>
MCRT_HD d3 surfNormalOf(const SurfRec& r, d3 pos) {
    if (r.kind == MCRT_SURF_SPHERE) return (pos - d3{r.v[0], r.v[1], r.v[2]}) / r.v[3];
    if constexpr (QuadricsIn<L>::value) {
        if (r.kind == MCRT_SURF_QUADRIC) return quadricNormal(quadricPtr(r.v[0]), pos);
    }
    return r.normal;
}
template <bool L>
MCRT_HD d3 surfSampleOf(const SurfRec& r, double u, double v) {
    if (r.kind == MCRT_SURF_SPHERE) {
        double z = 1.0 - 2.0 * u;
        double rr = sqrt(1.0 - sq(z));
        double phi = kTwoPi * v;
        double sin_phi, cos_phi;
        refSinCos<!L>(phi, sin_phi, cos_phi);
        return d3{r.v[0], r.v[1], r.v[2]} + r.v[3] * d3{rr * cos_phi, rr * sin_phi, z};
    }
    double su = sqrt(u);
    return (1 - su) * d3{r.v[0], r.v[1], r.v[2]} + ((1 - v) * su) * d3{r.v[3], r.v[4], r.v[5]} + (v * su) * d3{r.v[6], r.v[7], r.v[8]};
}

// ------------------------------------------------------------------ RefractionHistory (ray/ray.cpp:74-98)
// The reference keeps a std::vector<double> of medium IORs (reserve(8)). Here it is a per-lane stack of
// kMaxIors doubles in LDS ([entry][lane]); it is a separate object, not a member of the path state,
// so that the dynamically indexed array does not force the whole path state out of registers.
// kMaxIors entries per lane live in LDS; nesting deeper than that (no scene of the reference comes near it) continues in memory: the
// wavefront pipeline's deep rows (WfFrame::iors_deep, kMaxIorsDeep entries in all to begin with, four times as many on every retry of
// a frame that nests deeper: the reference's vector is unbounded, and so - up to the device's memory - is this), a per-lane region
// for the 1024-lane photon kernel. A megakernel frame that nests deeper than its 8 entries is rendered again through the pipeline.
constexpr int kMaxIors = 8;
constexpr int kMaxIorsDeep = 32;
struct RefractionHistory {
    MCRT_LDS_AS double* iors;  // &lds_iors[lane]; stride = block size
    uint32_t stride;
    int size;
    // The 1024-lane photon-mapping kernel keeps only the first two entries in LDS and the rest in global memory
    // ([kMaxIors - 2][lanes], interleaved by lane): 48 KB of LDS go to the traversal stacks and the estimates' buffers.
    // Everywhere else all kMaxIors entries are in LDS and giors is null.
    // (base and stride are the same for every lane — scalar registers on the GPU — and the lane's column is one 32-bit register: a
    // per-lane pointer cost the lane state machine 3 % of a frame in spilled registers)
    double* giors = nullptr;   // base of the entries in memory, [max_depth - lds_depth][gstride]
    uint32_t gstride = 0;
    uint32_t glane = 0;        // this lane's column
    int lds_depth = kMaxIors;  // entries [0, lds_depth) in LDS, the rest (nesting deeper than that: rare) in giors
    int max_depth = kMaxIors;  // entries in all: lds_depth + the rows giors has
    bool overflow = false;     // a medium was entered at nesting depth max_depth: reported by mcrt_render_finish
    MCRT_HD double at(int i) const { return i < lds_depth ? iors[(uint32_t)i * stride] : giors[(size_t)(i - lds_depth) * gstride + glane]; }
    MCRT_HD void put(int i, double v) {
        if (i < lds_depth) iors[(uint32_t)i * stride] = v;
        else giors[(size_t)(i - lds_depth) * gstride + glane] = v;
    }
    MCRT_HD void init(const Ray& ray) {
        put(0, ray.medium_ior);
        size = 1;
    }
    MCRT_HD void update(const Ray& ray) {
        if (ray.refraction_level > 0) {
            if (ray.refraction_level == size) {
                if (size < max_depth) put(size++, ray.medium_ior);
                else overflow = true;
            } else if (ray.refraction_level < size - 1) {
                size--;
            }
        }
    }
    MCRT_HD double externalIOR(const Ray& ray) const {
        int i = ray.refraction_level - 1;
        i = i < 0 ? 0 : i;
        i = i > size - 1 ? size - 1 : i;
        return at(i);
    }
};

// ------------------------------------------------------------------ Interaction (ray/interaction.cpp)
enum : int { kReflect = 0, kRefract = 1, kDiffuse = 2 };

template <bool L>
struct InteractionT {
    int type;
    double t, n1, n2, T, R;
    cptr<mcrt_material, L> material;
    uint32_t surface;
    d3 position, normal, out;
    m3 shading_cs;
    bool inside, dirac_delta;
    // the members of the incoming ray that later steps read (Interaction::ray, interaction.hpp:39)
    d3 ray_direction;
    double ray_refraction_scale;
    int ray_refraction_level;
    uint16_t ray_depth, ray_diffuse_depth;
    bool ray_dirac_delta;
};

template <bool L>
MCRT_HD void interactionInit(InteractionT<L>& ia, const ShadeViewT<L>& sh, const Hit& isect, const Ray& ray, double external_ior,
                             const Sampler& smp, SobolTab tab) {  // interaction.cpp:12-54
    ia.t = isect.t;
    ia.out = -ray.direction;
    ia.n1 = ray.medium_ior;
    ia.surface = isect.surface;
    ia.ray_direction = ray.direction;
    ia.ray_refraction_scale = ray.refraction_scale;
    ia.ray_refraction_level = ray.refraction_level;
    ia.ray_depth = ray.depth;
    ia.ray_diffuse_depth = ray.diffuse_depth;
    ia.ray_dirac_delta = ray.dirac_delta;
    ia.position = ray.start + ray.direction * ia.t;  // Ray::operator() ray.cpp:69-72
    d3 vn0 = splat(0.0), vn1 = splat(0.0), vn2 = splat(0.0);
    if constexpr (L) {
        ia.material = &sh.materials[sh.surf_material[isect.surface]];
        ia.normal = surfNormal(sh, isect.surface, ia.position);
    } else {  // scene in memory: everything the hit needs in one round trip, the record and (interpolating triangles) the vertex normals
        const double* rp = sh.surf_rec + (size_t)isect.surface * 16;
        const d3 face_normal = ld3(rp);
        const unsigned long long w = dBits(rp[3]);
        if (isect.interpolate) {
            vn0 = ld3(rp + 4);
            vn1 = ld3(rp + 7);
            vn2 = ld3(rp + 10);
        }
        ia.material = &sh.materials[(uint32_t)w];
        const uint32_t kind = (uint32_t)(w >> 32);
        ia.normal = face_normal;
        if (kind != MCRT_SURF_TRIANGLE) {  // (spheres and quadrics read their own data: a second trip for those lanes)
            SurfRec rec;
            rec.kind = kind;
            cptr<double, L> p = sh.surf_v + (size_t)isect.surface * 9;
            for (int k = 0; k < 4; k++) rec.v[k] = p[k];
            for (int k = 4; k < 9; k++) rec.v[k] = 0.0;
            rec.normal = face_normal;
            ia.normal = surfNormalOf<L>(rec, ia.position);
        }
    }
    const uint32_t flags = matFlags(ia.material->flags);

    double cos_theta = dot(ray.direction, ia.normal);
    ia.inside = cos_theta > 0.0;
    ia.n2 = (ia.inside && !(flags & MCRT_MAT_OPAQUE)) ? external_ior : ia.material->ior;

    d3 shading_normal = ia.normal;
    if (isect.interpolate) {
        if constexpr (L) shading_normal = surfInterpolatedNormal(sh, isect.surface, isect.u, isect.v);
        else shading_normal = normalize((1.0 - isect.u - isect.v) * vn0 + isect.u * vn1 + isect.v * vn2);  // (surfInterpolatedNormal)
        if ((cos_theta < 0.0) != (dot(ray.direction, shading_normal) < 0.0)) shading_normal = ia.normal;
    }
    if (cos_theta > 0.0) {
        ia.normal = -ia.normal;
        shading_normal = -shading_normal;
    }
    ia.shading_cs = orthonormalBasis(shading_normal);
    ia.R = fresnelDielectric(ia.n1, ia.n2, dot(shading_normal, ia.out));
    ia.T = ia.material->transparency;
    const bool rough_specular = (flags & MCRT_MAT_ROUGH_SPECULAR) != 0;
    if (rough_specular) ia.R = gmin(gmax(ia.R, 0.1), 0.9);

    // selectType, interaction.cpp:156-183
    if (flags & (MCRT_MAT_PERFECT_MIRROR | MCRT_MAT_COMPLEX_IOR)) {
        ia.type = kReflect;
    } else if (ia.n2 < 1.0) {
        ia.type = kDiffuse;
    } else {
        double p = smp.get(kDimInteraction, tab);
        if (ia.R > p) ia.type = kReflect;
        else if (ia.R + (1.0 - ia.R) * ia.T > p) ia.type = kRefract;
        else ia.type = kDiffuse;
    }
    ia.dirac_delta = ia.type != kDiffuse && !rough_specular;
}

// Interaction::BSDF (local frame), interaction.cpp:84-153
template <bool L>
MCRT_HD d3 interactionBSDFLocal(const InteractionT<L>& ia, d3 wo, d3 wi, double& pdf, bool flux, bool wi_dirac_delta) {
    const auto& m = *ia.material;
    const uint32_t f = matFlags(m.flags);
    const double n1 = ia.n1, n2 = ia.n2;
    double cos_theta = wo.z;
    if (f & MCRT_MAT_ROUGH_SPECULAR) {
        if (wi.z > 0.0) {
            cos_theta = dot(wo, normalize(wo + wi));
        } else {
            d3 mm = normalize(wo * n1 + wi * n2);
            cos_theta = dot(wo, mm);
            if (n1 < n2) cos_theta = -cos_theta;
        }
    }
    if (f & (MCRT_MAT_PERFECT_MIRROR | MCRT_MAT_COMPLEX_IOR)) {
        d3 brdf = matSpecularReflection(m, wi, wo, pdf);
        if (f & MCRT_MAT_COMPLEX_IOR) brdf = brdf * fresnelConductor(n1, ld3(m.ior_real), ld3(m.ior_imag), cos_theta);
        return brdf;
    }
    if (n2 < 1.0) return matDiffuseReflection(m, wi, wo, pdf);

    double F = fresnelDielectric(n1, n2, cos_theta);
    double pdf_s, pdf_d;
    d3 brdf_s = matSpecularReflection(m, wi, wo, pdf_s);
    d3 brdf_d = matDiffuseReflection(m, wi, wo, pdf_d);
    double pdf_t = pdf_s;
    d3 btdf = brdf_s;
    if (F < 1.0) btdf = matSpecularTransmission(m, wi, wo, n1, n2, pdf_t, ia.inside, flux);

    const double R = ia.R, T = ia.T;
    if (wi_dirac_delta) {
        if (ia.type == kReflect) {
            pdf = R;
            return brdf_s * F;
        }
        pdf = T * (1.0 - R);
        return btdf * T * (1.0 - F);
    } else if (!(f & MCRT_MAT_ROUGH_SPECULAR)) {
        pdf = pdf_d * (1.0 - R) * (1.0 - T);
        return brdf_d * (1.0 - F) * (1.0 - T);
    }
    pdf = mix(mix(pdf_d, pdf_t, T), pdf_s, R);
    return mix(mix(brdf_d, btdf, T), brdf_s, F);
}

// Interaction::BSDF (world wi), interaction.cpp:74-82
template <bool L>
MCRT_HD bool interactionBSDF(const InteractionT<L>& ia, d3& bsdf_absIdotN, d3 world_wi, double& pdf) {
    d3 wi = csTo(ia.shading_cs, world_wi);
    d3 wo = csTo(ia.shading_cs, ia.out);
    bsdf_absIdotN = interactionBSDFLocal(ia, wo, wi, pdf, false, false) * fabs(wi.z);
    return pdf > 0.0;
}

template <bool kLdsTab = true>
MCRT_HD d3 cosWeightedHemi(double u, double v) {  // sampling/sampling.hpp:35-44
    double r = sqrt(u);
    double azimuth = v * kTwoPi;
    double sin_a, cos_a;
    refSinCos<kLdsTab>(azimuth, sin_a, cos_a);
    return d3{r * cos_a, r * sin_a, sqrt(1 - u)};
}

template <bool L>
MCRT_HD d3 interactionSpecularNormal(const InteractionT<L>& ia, const Sampler& smp, SobolTab tab) {  // interaction.cpp:185-193
    if (matFlags(ia.material->flags) & MCRT_MAT_ROUGH_SPECULAR) {
        double u0 = smp.get(kDimBsdf, tab), u1 = smp.get(kDimBsdf + 1, tab);
        return csFrom(ia.shading_cs, ggxVisibleMicrofacet<!L>(u0, u1, csTo(ia.shading_cs, ia.out), ia.material->a[0], ia.material->a[1]));
    }
    return ia.shading_cs.c2;
}

// Ray::Ray(const Interaction&), ray/ray.cpp:16-67
template <bool L>
MCRT_HD Ray rayFromInteraction(const InteractionT<L>& ia, const Sampler& smp, SobolTab tab) {
    Ray r;
    r.depth = (uint16_t)(ia.ray_depth + 1);
    r.diffuse_depth = ia.ray_diffuse_depth;
    r.refraction_scale = ia.ray_refraction_scale;
    r.start = ia.position;
    r.refraction_level = ia.ray_refraction_level;
    r.dirac_delta = ia.dirac_delta;
    r.refraction = false;
    if (ia.type == kReflect) {
        d3 sn = interactionSpecularNormal(ia, smp, tab);
        r.direction = ia.ray_direction - sn * dot(sn, ia.ray_direction) * 2.0;  // glm::reflect
        r.medium_ior = ia.n1;
        r.start = r.start + ia.normal * kEpsilon;
    } else if (ia.type == kRefract) {
        d3 sn = interactionSpecularNormal(ia, smp, tab);
        double inv_eta = ia.n1 / ia.n2;
        double cos_theta = dot(sn, ia.ray_direction);
        double k = 1.0 - sq(inv_eta) * (1.0 - sq(cos_theta));
        if (k >= 0.0) {
            r.direction = inv_eta * ia.ray_direction - (inv_eta * cos_theta + sqrt(k)) * sn;
            r.medium_ior = ia.n2;
            r.start = r.start - ia.normal * kEpsilon;
            r.refraction_level += ia.inside ? -1 : 1;
            r.refraction_scale *= sq(1.0 / inv_eta);
            r.refraction = true;
        } else {
            r.direction = ia.ray_direction - sn * cos_theta * 2.0;
            r.medium_ior = ia.n1;
            r.start = r.start + ia.normal * kEpsilon;
        }
    } else {
        r.diffuse_depth++;
        double u0 = smp.get(kDimBsdf, tab), u1 = smp.get(kDimBsdf + 1, tab);
        r.direction = csFrom(ia.shading_cs, cosWeightedHemi<!L>(u0, u1));
        r.medium_ior = ia.n1;
        r.start = r.start + ia.normal * kEpsilon;
    }
    r.inv_direction = rcp3(r.direction);
    return r;
}

// Interaction::sampleBSDF, interaction.cpp:56-72
template <bool L>
MCRT_HD bool interactionSampleBSDF(const InteractionT<L>& ia, d3& bsdf_absIdotN, double& pdf, Ray& new_ray, bool flux,
                                   const Sampler& smp, SobolTab tab) {
    new_ray = rayFromInteraction(ia, smp, tab);
    d3 wi = csTo(ia.shading_cs, new_ray.direction);
    if ((new_ray.refraction && wi.z >= 0.0) || (!new_ray.refraction && wi.z <= 0.0)) return false;
    d3 wo = csTo(ia.shading_cs, ia.out);
    bsdf_absIdotN = interactionBSDFLocal(ia, wo, wi, pdf, flux, new_ray.dirac_delta) * fabs(wi.z);
    return pdf > 0.0;
}

// ------------------------------------------------------------------ Integrator pieces (integrator/integrator.cpp)
struct LightSample {  // integrator.hpp:14-18
    double bsdf_pdf, select_probability;
    uint32_t light;
};

MCRT_HD double powerHeuristic(double a_pdf, double b_pdf) {  // common/util.hpp:85-89
    double a_pdf2 = a_pdf * a_pdf;
    return a_pdf2 / (a_pdf2 + b_pdf * b_pdf);
}

// Scene::selectLight (scene.cpp:225-236) with Sampling::weightedIdx (sampling.hpp:13-27)
template <bool L>
MCRT_HD uint32_t selectLight(const ShadeViewT<L>& sh, double u, double& select_probability) {
    uint32_t left = 0, right = sh.num_lights - 1;
    while (left < right) {
        uint32_t middle = (left + right) / 2;
        if (sh.light_cdf[middle] < u) left = middle + 1;
        else right = middle;
    }
    select_probability = sh.light_cdf[left];
    if (left > 0) select_probability -= sh.light_cdf[left - 1];
    return sh.light_surface[left];
}

// First half of Integrator::sampleDirect (integrator.cpp:31-66): choose the light point and build the
// shadow ray. Returns false when the estimate is zero without tracing (ls.light is still set,
// integrator.cpp:42 / SURVEY appendix A.5).
struct DirectQuery {
    Ray shadow_ray;
    double cos_light_theta;
    double light_area;  // surf_area of the chosen light (scenes in memory: fetched with the light's record)
    ShadowQuery sq;  // bounds for the shadow traversal (mcrt_scene.hpp)
};
template <bool L>
MCRT_HD bool sampleDirectSetup(const ShadeViewT<L>& sh, const InteractionT<L>& ia, LightSample& ls, DirectQuery& q,
                               const Sampler& smp, SobolTab tab) {
    if (sh.num_lights == 0 || (ia.material->flags & MCRT_MAT_DIRAC_DELTA)) {
        ls.light = kNoSurface;
        return false;
    }
    double u0 = smp.get(kDimLight, tab), u1 = smp.get(kDimLight + 1, tab), u2 = smp.get(kDimLight + 2, tab);
    ls.light = selectLight(sh, u2, ls.select_probability);
    d3 light_pos, light_normal;
    d3 shadow_start = ia.position + ia.normal * kEpsilon;
    if constexpr (L) {
        light_pos = surfSample(sh, ls.light, u0, u1);
        q.shadow_ray = makeRayTo(shadow_start, light_pos);
        light_normal = surfNormal(sh, ls.light, light_pos);
        q.light_area = 0.0;  // (read where it is used: one value less to keep across the shadow ray's trace)
    } else {  // scene in memory: the light's record and its area in one round trip
        const SurfRec rec = surfFetch(sh, ls.light, true);
        q.light_area = sh.surf_area[ls.light];
        light_pos = surfSampleOf<L>(rec, u0, u1);
        q.shadow_ray = makeRayTo(shadow_start, light_pos);
        light_normal = surfNormalOf<L>(rec, light_pos);
    }
    q.cos_light_theta = dot(-q.shadow_ray.direction, light_normal);
    if (q.cos_light_theta <= 0.0) return false;
    double cos_theta = dot(q.shadow_ray.direction, ia.normal);
    if (cos_theta <= 0.0) {
        if ((ia.material->flags & MCRT_MAT_OPAQUE) || cos_theta == 0.0) return false;
        shadow_start = ia.position - ia.normal * kEpsilon;
        q.shadow_ray = makeRayTo(shadow_start, light_pos);  // try transmission
    }
    const d3 to_light = light_pos - shadow_start;
    const double dist = sqrt(dot(to_light, to_light));
    q.sq.light = ls.light;
    q.sq.setRange(dist);
    return true;
}
// Second half (integrator.cpp:68-86), given the shadow ray's closest hit.
template <bool L>
MCRT_HD d3 sampleDirectFinish(const ShadeViewT<L>& sh, const InteractionT<L>& ia, const LightSample& ls, const DirectQuery& q,
                              const Hit& shadow_hit) {
    if (shadow_hit.surface == kNoSurface || shadow_hit.surface != ls.light) return splat(0.0);
    double light_pdf = sq(shadow_hit.t) / ((L ? sh.surf_area[ls.light] : q.light_area) * q.cos_light_theta);
    double bsdf_pdf;
    d3 bsdf_absIdotN;
    if (!interactionBSDF(ia, bsdf_absIdotN, q.shadow_ray.direction, bsdf_pdf)) return splat(0.0);
    double mis_weight = powerHeuristic(light_pdf, bsdf_pdf);
    const auto& lm = sh.materials[sh.surf_material[ls.light]];
    return mis_weight * bsdf_absIdotN * ld3(lm.emittance) / (light_pdf * ls.select_probability);
}

// Integrator::sampleEmissive, integrator.cpp:93-110
template <bool L>
MCRT_HD d3 sampleEmissive(const ShadeViewT<L>& sh, const InteractionT<L>& ia, const LightSample& ls) {
    if ((ia.material->flags & MCRT_MAT_EMISSIVE) && !ia.inside) {
        if (ia.ray_depth == 0 || ia.ray_dirac_delta) return ld3(ia.material->emittance);
        if (ls.light == ia.surface) {
            double cos_light_theta = dot(ia.out, ia.normal);
            double light_pdf = sq(ia.t) / (sh.surf_area[ia.surface] * cos_light_theta);
            double mis_weight = powerHeuristic(ls.bsdf_pdf, light_pdf);
            return mis_weight * ld3(ia.material->emittance) / ls.select_probability;
        }
    }
    return splat(0.0);
}

// Integrator::absorb, integrator.cpp:112-129 (min_ray_depth 3, min_priority_ray_depth 16: integrator.hpp:28-29)
MCRT_HD bool absorb(const Ray& ray, d3& throughput, const Sampler& smp, SobolTab tab) {
    double survive = compMax(throughput) * ray.refraction_scale;
    if (survive == 0.0) return true;
    if (ray.diffuse_depth > 3 || ray.depth > 16) {
        survive = gmin(0.95, survive);
        if (survive <= smp.get(kDimAbsorb, tab)) return true;
        throughput = throughput / survive;
    }
    return false;
}

// Scene::skyColor, scene.cpp:219-223
MCRT_HD d3 skyColor(const Ray& ray) {
    double fy = (1.0 + refAsin(dot(d3{0.0, 1.0, 0.0}, ray.direction)) / kPi) / 2.0;  // glibc's asin, bit for bit (mcrt_libm.hpp)
    return mix(d3{1.0, 0.5, 0.0}, d3{0.0, 0.5, 1.0}, fy);
}

// Camera::samplePixel ray generation, camera/camera.cpp:79-95 (sampler already at setIndex(i)).
template <bool kLdsTab = true>
MCRT_HD Ray cameraRay(const mcrt_camera_desc& cam, double scene_ior, uint32_t x, uint32_t y, const Sampler& smp,
                      SobolTab tab) {
    double pixel_size = cam.sensor_width / (double)cam.width;
    double half_x = (double)cam.width * 0.5, half_y = (double)cam.height * 0.5;
    double px = (double)x + smp.get(kDimPixel, tab), py = (double)y + smp.get(kDimPixel + 1, tab);
    double lx = pixel_size * (half_x - px), ly = pixel_size * (half_y - py);
    d3 forward = ld3(cam.forward), left = ld3(cam.left), up = ld3(cam.up), eye = ld3(cam.eye);
    d3 direction = normalize(forward * cam.focal_length + left * lx + up * ly);
    Ray ray = makeRay(eye, direction, scene_ior);
    if (cam.thin_lens) {
        double u0 = smp.get(kDimLens, tab), u1 = smp.get(kDimLens + 1, tab);
        double azimuth = u1 * kTwoPi;  // Sampling::uniformDisk, sampling.hpp:29-33
        double su = sqrt(u0);
        double sin_a, cos_a;
        refSinCos<kLdsTab>(azimuth, sin_a, cos_a);
        double ax = cos_a * su * cam.aperture_radius, ay = sin_a * su * cam.aperture_radius;
        d3 focus_point = ray.start + ray.direction * (cam.focus_distance / dot(ray.direction, forward));
        d3 start = eye + left * ax + up * ay;
        ray = makeRay(start, normalize(focus_point - start), scene_ior);
    }
    return ray;
}

}  // namespace mcrt
