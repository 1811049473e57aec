// Scene-image (*.mcrt) reader/writer: a flat, chunked dump of mcrt_scene_desc / mcrt_camera_desc /
// mcrt_photon_map_desc (include/mcrt.h). Host-only; compiled into libmcrt_hip.so and linked by the
// flattener that runs inside the reference host (oracle/ref_main.cpp).
//
// Layout: "MCRTIMG1" | u32 abi | u32 num_chunks | { char name[24]; u64 nbytes; data; pad to 8 }*
#include "../../include/mcrt.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

struct Chunk {
    std::vector<unsigned char> bytes;
};

const char kMagic[8] = {'M', 'C', 'R', 'T', 'I', 'M', 'G', '1'};
const uint32_t kImageVersion = 1;  // container version (chunks are optional and named; independent of MCRT_ABI_VERSION)

struct ParamKV {
    char key[24];
    uint64_t value;
};

}  // namespace

struct mcrt_image {
    std::map<std::string, Chunk> chunks;
    mcrt_scene_desc scene;
    mcrt_camera_desc camera;
    mcrt_photon_map_desc maps[2];
    bool has_map[2];
    std::map<std::string, uint64_t> params;
};

namespace {

template <class T>
const T* chunkPtr(const mcrt_image* img, const char* name, size_t* count = nullptr) {
    auto it = img->chunks.find(name);
    if (it == img->chunks.end() || it->second.bytes.empty()) {
        if (count) *count = 0;
        return nullptr;
    }
    if (count) *count = it->second.bytes.size() / sizeof(T);
    return reinterpret_cast<const T*>(it->second.bytes.data());
}

bool loadMap(mcrt_image* img, const char* prefix, mcrt_photon_map_desc* m) {
    std::string p(prefix);
    size_t n_oct = 0, n_ph = 0;
    memset(m, 0, sizeof(*m));
    m->octant_bounds = chunkPtr<double>(img, (p + "oct_bounds").c_str(), &n_oct);
    if (!m->octant_bounds) return false;
    m->num_octants = (uint32_t)(n_oct / 6);
    m->octant_start_data = chunkPtr<uint64_t>(img, (p + "oct_start").c_str());
    m->octant_contained_data = chunkPtr<uint64_t>(img, (p + "oct_contained").c_str());
    m->octant_next_sibling = chunkPtr<uint32_t>(img, (p + "oct_next").c_str());
    m->octant_leaf = chunkPtr<uint8_t>(img, (p + "oct_leaf").c_str());
    m->photons = chunkPtr<float>(img, (p + "photons").c_str(), &n_ph);
    m->num_photons = n_ph / 8;
    return m->octant_start_data && m->octant_contained_data && m->octant_next_sibling &&
           m->octant_leaf && m->photons;
}

struct Writer {
    FILE* f;
    uint32_t count = 0;
    bool ok = true;
    void put(const char* name, const void* data, size_t nbytes) {
        if (!data || nbytes == 0) return;
        char nm[24];
        memset(nm, 0, sizeof(nm));
        strncpy(nm, name, sizeof(nm) - 1);
        uint64_t nb = nbytes;
        ok = ok && fwrite(nm, 1, sizeof(nm), f) == sizeof(nm);
        ok = ok && fwrite(&nb, sizeof(nb), 1, f) == 1;
        ok = ok && fwrite(data, 1, nbytes, f) == nbytes;
        static const char zeros[8] = {0};
        size_t pad = (8 - nbytes % 8) % 8;
        if (pad) ok = ok && fwrite(zeros, 1, pad, f) == pad;
        count++;
    }
};

void writeMap(Writer& w, const char* prefix, const mcrt_photon_map_desc* m) {
    if (!m || m->num_octants == 0) return;
    std::string p(prefix);
    size_t n = m->num_octants;
    w.put((p + "oct_bounds").c_str(), m->octant_bounds, n * 6 * sizeof(double));
    w.put((p + "oct_start").c_str(), m->octant_start_data, n * sizeof(uint64_t));
    w.put((p + "oct_contained").c_str(), m->octant_contained_data, n * sizeof(uint64_t));
    w.put((p + "oct_next").c_str(), m->octant_next_sibling, n * sizeof(uint32_t));
    w.put((p + "oct_leaf").c_str(), m->octant_leaf, n * sizeof(uint8_t));
    w.put((p + "photons").c_str(), m->photons, (size_t)m->num_photons * 8 * sizeof(float));
}

}  // namespace

extern "C" {

uint32_t mcrt_abi_version(void) { return MCRT_ABI_VERSION; }

// The file is untrusted input: every chunk size is bounded by the file size, every array is checked against the element
// count its sibling arrays derive (a truncated or mismatched image would otherwise make buildLayout / uploadArray read past
// the host buffers), and nothing thrown by the containers crosses the C boundary.
static int imageLoadImpl(const char* path, mcrt_image** out) {
    FILE* f = fopen(path, "rb");
    if (!f) return MCRT_ERR_IO;
    struct Closer {
        FILE* f;
        ~Closer() { fclose(f); }
    } closer{f};
    if (fseek(f, 0, SEEK_END) != 0) return MCRT_ERR_IO;
    const long file_size = ftell(f);
    if (file_size < 16 || fseek(f, 0, SEEK_SET) != 0) return MCRT_ERR_IO;
    char magic[8];
    uint32_t abi = 0, num_chunks = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kMagic, 8) == 0;
    ok = ok && fread(&abi, 4, 1, f) == 1 && fread(&num_chunks, 4, 1, f) == 1;
    if (!ok || abi != kImageVersion) return MCRT_ERR_IO;
    struct Holder {  // frees the image on every early return
        mcrt_image* p;
        ~Holder() { delete p; }
    } hold{new mcrt_image()};
    mcrt_image* img = hold.p;
    for (uint32_t c = 0; c < num_chunks; c++) {
        char nm[25];
        uint64_t nb = 0;
        memset(nm, 0, sizeof(nm));
        if (fread(nm, 1, 24, f) != 24 || fread(&nb, 8, 1, f) != 1) return MCRT_ERR_IO;
        const long at = ftell(f);
        if (at < 0 || nb > (uint64_t)(file_size - at)) return MCRT_ERR_IO;  // a chunk cannot be larger than what is left of the file
        Chunk& ch = img->chunks[nm];
        ch.bytes.resize((size_t)nb);
        if (nb && fread(ch.bytes.data(), 1, (size_t)nb, f) != nb) return MCRT_ERR_IO;
        const size_t pad = (8 - nb % 8) % 8;
        if (pad && fseek(f, (long)pad, SEEK_CUR) != 0) return MCRT_ERR_IO;
    }
    auto chunkBytes = [&](const char* name) -> uint64_t {
        auto it = img->chunks.find(name);
        return it == img->chunks.end() ? 0ull : (uint64_t)it->second.bytes.size();
    };
    // exact(name, count, elem): the chunk holds exactly count elements; optional chunks may also be absent
    auto exact = [&](const char* name, uint64_t count, uint64_t elem, bool optional = false) {
        const uint64_t b = chunkBytes(name);
        return (optional && b == 0) || b == count * elem;
    };
    mcrt_scene_desc& s = img->scene;
    memset(&s, 0, sizeof(s));
    s.abi_version = MCRT_ABI_VERSION;
    size_t n = 0;
    s.node_bounds = chunkPtr<double>(img, "node_bounds", &n);
    if (n % 6 || n / 6 > 0xFFFFFFFFull) return MCRT_ERR_IO;
    s.num_nodes = (uint32_t)(n / 6);
    s.node_start_surface = chunkPtr<uint32_t>(img, "node_start");
    s.node_num_surfaces = chunkPtr<uint32_t>(img, "node_count");
    s.node_next_sibling = chunkPtr<uint32_t>(img, "node_next");
    s.surf_kind = chunkPtr<uint8_t>(img, "surf_kind", &n);
    if (n == 0 || n > 0xFFFFFFFFull) return MCRT_ERR_IO;
    s.num_surfaces = (uint32_t)n;
    s.surf_interpolate = chunkPtr<uint8_t>(img, "surf_interp");
    s.surf_material = chunkPtr<uint32_t>(img, "surf_material");
    s.surf_area = chunkPtr<double>(img, "surf_area");
    s.surf_v = chunkPtr<double>(img, "surf_v");
    s.surf_e = chunkPtr<double>(img, "surf_e");
    s.surf_vn = chunkPtr<double>(img, "surf_vn");
    s.materials = chunkPtr<mcrt_material>(img, "materials", &n);
    if (n == 0 || chunkBytes("materials") % sizeof(mcrt_material) || n > 0xFFFFFFFFull) return MCRT_ERR_IO;
    s.num_materials = (uint32_t)n;
    s.light_surface = chunkPtr<uint32_t>(img, "light_surface", &n);
    if (chunkBytes("light_surface") % 4 || n > 0xFFFFFFFFull) return MCRT_ERR_IO;
    s.num_lights = (uint32_t)n;
    s.light_cdf = chunkPtr<double>(img, "light_cdf");
    const uint64_t nn = s.num_nodes, ns = s.num_surfaces;
    if (!exact("node_start", nn, 4) || !exact("node_count", nn, 4) || !exact("node_next", nn, 4) || !exact("surf_interp", ns, 1) ||
        !exact("surf_material", ns, 4) || !exact("surf_area", ns, 8) || !exact("surf_v", ns * 9, 8) || !exact("surf_e", ns * 9, 8) ||
        !exact("surf_vn", ns * 9, 8, true) || !exact("light_cdf", s.num_lights, 8))
        return MCRT_ERR_IO;
    const double* sc = chunkPtr<double>(img, "scene_scalars", &n);
    if (sc && n >= 7) {
        s.scene_ior = sc[0];
        for (int i = 0; i < 3; i++) {
            s.bb_min[i] = sc[1 + i];
            s.bb_max[i] = sc[4 + i];
        }
    }
    s.quadrics = chunkPtr<double>(img, "quadrics", &n);
    if (chunkBytes("quadrics") % (22 * 8)) return MCRT_ERR_IO;
    s.num_quadrics = (uint32_t)(n / 22);
    // the camera record has grown over time (film filter fields): take what the file has, the rest stays zero (= box filter)
    memset(&img->camera, 0, sizeof(img->camera));
    {
        auto it = img->chunks.find("camera");
        if (it != img->chunks.end()) {
            if (it->second.bytes.size() < offsetof(mcrt_camera_desc, film_filter)) return MCRT_ERR_IO;  // shorter than the first version of the record
            memcpy(&img->camera, it->second.bytes.data(), std::min(it->second.bytes.size(), sizeof(img->camera)));
            const mcrt_camera_desc& cam = img->camera;
            if (cam.width == 0 || cam.height == 0 || cam.sqrtspp == 0 || (uint64_t)cam.width * cam.height > 0xFFFFFFFFull ||
                cam.sqrtspp > 65535u || cam.film_filter > MCRT_FILM_LANCZOS)
                return MCRT_ERR_IO;
        }
    }
    for (int w = 0; w < 2; w++) {
        const std::string p(w == 0 ? "g_" : "c_");
        img->has_map[w] = loadMap(img, p.c_str(), &img->maps[w]);
        if (chunkBytes((p + "oct_bounds").c_str()) == 0) continue;  // no such map in the file
        const uint64_t no = chunkBytes((p + "oct_bounds").c_str()) / 48;
        if (!img->has_map[w] || chunkBytes((p + "oct_bounds").c_str()) % 48 || no > 0xFFFFFFFFull || !exact((p + "oct_start").c_str(), no, 8) ||
            !exact((p + "oct_contained").c_str(), no, 8) || !exact((p + "oct_next").c_str(), no, 4) || !exact((p + "oct_leaf").c_str(), no, 1) ||
            chunkBytes((p + "photons").c_str()) % 32)
            return MCRT_ERR_IO;
    }
    const ParamKV* kv = chunkPtr<ParamKV>(img, "params", &n);
    for (size_t i = 0; kv && i < n; i++) {
        char key[25];
        memset(key, 0, sizeof(key));
        memcpy(key, kv[i].key, 24);
        img->params[key] = kv[i].value;
    }
    const bool scene_ok = s.surf_kind && s.surf_interpolate && s.surf_material && s.surf_area && s.surf_v && s.surf_e && s.materials &&
                          (s.num_nodes == 0 || (s.node_start_surface && s.node_num_surfaces && s.node_next_sibling)) &&
                          (s.num_lights == 0 || s.light_cdf);
    if (!scene_ok) return MCRT_ERR_IO;
    hold.p = nullptr;
    *out = img;
    return MCRT_OK;
}

int mcrt_image_load(const char* path, mcrt_image** out) {
    if (!path || !out) return MCRT_ERR_INVALID;
    *out = nullptr;
    try {
        return imageLoadImpl(path, out);
    } catch (...) {  // length_error / bad_alloc from the containers
        *out = nullptr;
        return MCRT_ERR_IO;
    }
}

void mcrt_image_free(mcrt_image* img) { delete img; }

const mcrt_scene_desc* mcrt_image_scene(const mcrt_image* img) { return img ? &img->scene : nullptr; }

const mcrt_camera_desc* mcrt_image_camera(const mcrt_image* img) {
    return img ? &img->camera : nullptr;
}

const mcrt_photon_map_desc* mcrt_image_photons(const mcrt_image* img, int which) {
    if (!img || which < 0 || which > 1 || !img->has_map[which]) return nullptr;
    return &img->maps[which];
}

uint64_t mcrt_image_param(const mcrt_image* img, const char* key) {
    if (!img || !key) return 0;
    auto it = img->params.find(key);
    return it == img->params.end() ? 0 : it->second;
}

int mcrt_image_save(const char* path, const mcrt_scene_desc* s, const mcrt_camera_desc* cam,
                    const mcrt_photon_map_desc* global_map, const mcrt_photon_map_desc* caustic_map,
                    const char* const* param_keys, const uint64_t* param_values,
                    uint32_t num_params) {
    if (!path || !s) return MCRT_ERR_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return MCRT_ERR_IO;
    uint32_t abi = kImageVersion, zero = 0;
    fwrite(kMagic, 1, 8, f);
    fwrite(&abi, 4, 1, f);
    fwrite(&zero, 4, 1, f);  // chunk count patched below
    Writer w{f};
    size_t nn = s->num_nodes, ns = s->num_surfaces;
    w.put("node_bounds", s->node_bounds, nn * 6 * sizeof(double));
    w.put("node_start", s->node_start_surface, nn * sizeof(uint32_t));
    w.put("node_count", s->node_num_surfaces, nn * sizeof(uint32_t));
    w.put("node_next", s->node_next_sibling, nn * sizeof(uint32_t));
    w.put("surf_kind", s->surf_kind, ns);
    w.put("surf_interp", s->surf_interpolate, ns);
    w.put("surf_material", s->surf_material, ns * sizeof(uint32_t));
    w.put("surf_area", s->surf_area, ns * sizeof(double));
    w.put("surf_v", s->surf_v, ns * 9 * sizeof(double));
    w.put("surf_e", s->surf_e, ns * 9 * sizeof(double));
    w.put("surf_vn", s->surf_vn, ns * 9 * sizeof(double));
    w.put("materials", s->materials, (size_t)s->num_materials * sizeof(mcrt_material));
    w.put("light_surface", s->light_surface, (size_t)s->num_lights * sizeof(uint32_t));
    w.put("light_cdf", s->light_cdf, (size_t)s->num_lights * sizeof(double));
    double sc[7] = {s->scene_ior, s->bb_min[0], s->bb_min[1], s->bb_min[2],
                    s->bb_max[0], s->bb_max[1], s->bb_max[2]};
    w.put("scene_scalars", sc, sizeof(sc));
    if (s->num_quadrics) w.put("quadrics", s->quadrics, (size_t)s->num_quadrics * 22 * sizeof(double));
    if (cam) w.put("camera", cam, sizeof(*cam));
    writeMap(w, "g_", global_map);
    writeMap(w, "c_", caustic_map);
    if (num_params && param_keys && param_values) {
        std::vector<ParamKV> kv(num_params);
        for (uint32_t i = 0; i < num_params; i++) {
            memset(&kv[i], 0, sizeof(ParamKV));
            strncpy(kv[i].key, param_keys[i], 23);
            kv[i].value = param_values[i];
        }
        w.put("params", kv.data(), kv.size() * sizeof(ParamKV));
    }
    bool ok = w.ok && fseek(f, 12, SEEK_SET) == 0 && fwrite(&w.count, 4, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    return ok ? MCRT_OK : MCRT_ERR_IO;
}

}  // extern "C"
