// obj_loader.cpp -- Scene::load for Wavefront OBJ + MTL (reference src/scene.rs:22-85,
// src/loader/obj.rs:16-436), restated in C++ on the host.  SURVEY.md 8(f) rank 1: the on-disk
// format on the input side of the path; config 1 ("12-triangle OBJ") goes through it.
//
// Behaviour follows the reference statement by statement; where the reference panics
// (unwrap on a malformed number, negative indices, >3 components, missing .mtl) this loader
// returns MIPT_ERR_IO with a message instead of unwinding across the C ABI.
// Deviations, both documented in DESIGN.md:
//  * materials keep INSERTION order (the reference iterates a HashMap, so its material ids are
//    a per-process random permutation -- src/loader/obj.rs:81-90; results are unaffected);
//  * map_* texture lines need an image decoder (the `image` crate, not vendored): PNG (png_decode.cpp), JPEG
//    (jpeg_decode.cpp), TGA / BMP (tga_bmp_decode.cpp) and binary PPM (P6) files are decoded here, chosen by file extension as image::open does;
//    other formats are reported and skipped.
#include "../../include/mipt.h"

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "mipt_internal.h"   // mipt_internal_set_error (mipt_api.cpp) feeds mipt_last_error()
namespace mipt_png {
bool decode(const std::string &path, uint32_t *w, uint32_t *h, std::vector<uint8_t> *rgba, std::string *err);
bool write_rgba(const std::string &path, uint32_t w, uint32_t h, int bits, const void *rgba, std::string *err);
}
namespace mipt_jpeg { bool decode(const std::string &path, uint32_t *w, uint32_t *h, std::vector<uint8_t> *rgba, std::string *err); }
namespace mipt_img {
bool decode_tga(const std::string &path, uint32_t *w, uint32_t *h, std::vector<uint8_t> *rgba, std::string *err);
bool decode_bmp(const std::string &path, uint32_t *w, uint32_t *h, std::vector<uint8_t> *rgba, std::string *err);
}

namespace {

struct ObjTri { size_t pos[3] = {0, 0, 0}, tex[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}; uint32_t material_id = 0; }; // obj.rs:344-350

struct Tex { uint32_t w = 0, h = 0, hash = 0; std::vector<uint8_t> rgba; };

} // namespace

struct MiptObj {
    std::vector<MiptTriangle> tris;
    std::vector<MiptNode> nodes;
    std::vector<MiptMaterial> materials;
    std::vector<std::string> material_names;
    std::vector<const char *> name_ptrs;
    std::vector<Tex> textures;
    std::vector<MiptTexture> tex_desc;
};

namespace {

std::vector<std::string> split_ws(const std::string &s) {         // str::split_whitespace
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) j++;
        if (j > i) out.emplace_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

bool parse_f32(const std::string &s, float *out) {                // str::parse::<f32>
    if (s.empty()) return false;
    char *end = nullptr;
    float v = strtof(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') return false;
    *out = v;
    return true;
}

bool read_lines(const std::string &path, std::vector<std::string> *lines) {   // read_to_string + lines()
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    std::string buf = ss.str(), cur;
    for (char c : buf) {
        if (c == '\n') { if (!cur.empty() && cur.back() == '\r') cur.pop_back(); lines->push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    if (!cur.empty()) lines->push_back(cur);
    return true;
}

std::string resource_path(const std::string &file_path, const std::string &res) {  // obj.rs:319-332
    if (!res.empty() && res[0] == '/') return res;
    size_t slash = file_path.find_last_of('/');
    std::string dir = slash == std::string::npos ? std::string() : file_path.substr(0, slash);
    return dir.empty() ? res : dir + "/" + res;
}

bool starts_with(const std::string &s, const char *p) { return s.compare(0, strlen(p), p) == 0; }

int fail(const std::string &m) { mipt_internal_set_error(m.c_str()); return MIPT_ERR_IO; }

// texture.rs:40-48: djb2 over every 4th pixel, pixel read as a native-endian u32
uint32_t djb2(const std::vector<uint8_t> &rgba) {
    uint32_t hash = 5381;
    for (size_t i = 0; i * 4 + 3 < rgba.size(); i += 4) {
        uint32_t c;
        memcpy(&c, &rgba[i * 4], 4);
        hash = ((hash << 5) + hash) + c;
    }
    return hash;
}

// Texture::load (texture.rs:13-31) for binary PPM only: decode, flip vertically, expand to RGBA8.
bool load_ppm(const std::string &path, Tex *t) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P6") return false;
    auto next_int = [&](int *v) {
        for (;;) {
            int c = f.peek();
            if (c == '#') { std::string skip; std::getline(f, skip); }
            else if (isspace(c)) f.get();
            else break;
        }
        return (bool)(f >> *v);
    };
    int w, h, maxv;
    if (!next_int(&w) || !next_int(&h) || !next_int(&maxv) || w <= 0 || h <= 0 || maxv != 255) return false;
    f.get();
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    f.read((char *)rgb.data(), (std::streamsize)rgb.size());
    if ((size_t)f.gcount() != rgb.size()) return false;
    t->w = (uint32_t)w; t->h = (uint32_t)h;
    t->rgba.resize((size_t)w * h * 4);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *s = &rgb[((size_t)(h - 1 - y) * w + x) * 3];   // flipv()
            uint8_t *d = &t->rgba[((size_t)y * w + x) * 4];
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 255;
        }
    t->hash = djb2(t->rgba);
    return true;
}

// Texture::load for the decoded formats: decode, flipv(), RGBA8 (texture.rs:18)
enum class ImgFmt { Png, Jpeg, Tga, Bmp };
bool load_decoded(const std::string &path, ImgFmt fmt, Tex *t, std::string *err) {
    std::vector<uint8_t> top_down;
    bool ok = false;
    switch (fmt) {
    case ImgFmt::Png: ok = mipt_png::decode(path, &t->w, &t->h, &top_down, err); break;
    case ImgFmt::Jpeg: ok = mipt_jpeg::decode(path, &t->w, &t->h, &top_down, err); break;
    case ImgFmt::Tga: ok = mipt_img::decode_tga(path, &t->w, &t->h, &top_down, err); break;
    case ImgFmt::Bmp: ok = mipt_img::decode_bmp(path, &t->w, &t->h, &top_down, err); break;
    }
    if (!ok) return false;
    t->rgba.resize(top_down.size());
    const size_t row = (size_t)t->w * 4;
    for (uint32_t y = 0; y < t->h; y++) memcpy(&t->rgba[(size_t)y * row], &top_down[(size_t)(t->h - 1 - y) * row], row);
    t->hash = djb2(t->rgba);
    return true;
}

// Texture::load (texture.rs:13-31): the decoder is chosen by extension, as image::open does
bool load_any_texture(const std::string &path, Tex *t, std::string *err) {
    std::string ext;
    const size_t dot = path.find_last_of('.');
    if (dot != std::string::npos)
        for (size_t i = dot + 1; i < path.size(); i++) ext += (char)tolower((unsigned char)path[i]);
    if (ext == "png") return load_decoded(path, ImgFmt::Png, t, err);
    if (ext == "jpg" || ext == "jpeg") return load_decoded(path, ImgFmt::Jpeg, t, err);
    if (ext == "tga") return load_decoded(path, ImgFmt::Tga, t, err);
    if (ext == "bmp") return load_decoded(path, ImgFmt::Bmp, t, err);
    if (load_ppm(path, t)) return true;
    *err = "only PNG, JPEG, TGA, BMP and binary PPM (P6) are decoded in this build";
    return false;
}

// obj.rs:267-309
void load_texture(const std::string &path, MiptObj *obj, uint32_t *slot) {
    Tex t;
    std::string err;
    if (!load_any_texture(path, &t, &err)) {
        // Texture::load returns None when the file is missing (texture.rs:14-17); undecodable files are skipped too
        fprintf(stderr, "[mipt] texture '%s' skipped: %s\n", path.c_str(), err.c_str());
        return;
    }
    for (size_t i = 0; i < obj->textures.size(); i++)
        if (obj->textures[i].hash == t.hash) { *slot = (uint32_t)i; return; }
    obj->textures.push_back(std::move(t));
    *slot = (uint32_t)obj->textures.size() - 1;
}

int load_mtl(MiptObj *obj, const std::string &path) {               // obj.rs:131-265
    std::vector<std::string> lines;
    if (!read_lines(path, &lines)) return fail("could not read .mtl file '" + path + "'");
    size_t li = 0;
    while (li < lines.size()) {
        const std::string &line = lines[li++];
        if (!starts_with(line, "newmtl ")) continue;
        std::string name = line.substr(7);
        MiptMaterial m;
        mipt_material_default(&m);
        while (li < lines.size()) {
            const std::string &l2 = lines[li++];
            std::vector<std::string> tok = split_ws(l2);
            if (tok.empty()) break;                                  // blank line ends the material
            const std::string &p = tok[0];
            auto vec3 = [&](float *dst) -> bool {
                if (tok.size() - 1 > 3) return false;
                for (size_t i = 1; i < tok.size(); i++) if (!parse_f32(tok[i], &dst[i - 1])) return false;
                return true;
            };
            auto scalar = [&](float *dst) -> bool { return tok.size() >= 2 && parse_f32(tok[1], dst); };
            bool ok = true;
            if (p == "Kd") ok = vec3(&m.base_color.x);
            else if (p == "Ks") ok = vec3(&m.specular_tint.x);
            else if (p == "Ke") ok = vec3(&m.emission.x);
            else if (p == "Ni") ok = scalar(&m.ior);
            else if (p == "Pr") ok = scalar(&m.roughness);
            else if (p == "Pm") ok = scalar(&m.metallic);
            else if (p == "Tf") ok = scalar(&m.transmission);
            else if (p == "d") ok = scalar(&m.transparency);
            else if (p == "map_Kd" && tok.size() >= 2) load_texture(resource_path(path, tok[1]), obj, &m.base_color_tex_id);
            else if (p == "map_d" && tok.size() >= 2) load_texture(resource_path(path, tok[1]), obj, &m.transparency_tex_id);
            else if (p == "map_Pr" && tok.size() >= 2) load_texture(resource_path(path, tok[1]), obj, &m.roughness_tex_id);
            else if (p == "map_Pm" && tok.size() >= 2) load_texture(resource_path(path, tok[1]), obj, &m.metallic_tex_id);
            else if (p == "map_Ke" && tok.size() >= 2) load_texture(resource_path(path, tok[1]), obj, &m.emission_tex_id);
            else if (p == "map_Bump" && tok.size() >= 2) load_texture(resource_path(path, tok.back()), obj, &m.normal_tex_id);
            if (!ok) return fail("malformed '" + p + "' line in '" + path + "': " + l2);
        }
        bool replaced = false;                                       // HashMap::insert replaces an existing key
        for (size_t i = 0; i < obj->material_names.size(); i++)
            if (obj->material_names[i] == name) { obj->materials[i] = m; replaced = true; break; }
        if (!replaced) { obj->material_names.push_back(name); obj->materials.push_back(m); }
    }
    return MIPT_OK;
}

bool read_index(const std::string &s, size_t *out) {                 // obj.rs:355-362
    if (s.empty()) return false;
    char *end = nullptr;
    long v = strtol(s.c_str(), &end, 10);
    if (end == s.c_str() || *end != '\0') return false;
    v -= 1;
    if (v < 0) return false;                                         // reference: panic on negative indices
    *out = (size_t)v;
    return true;
}

bool tri_from_groups(const std::string g[3], ObjTri *t) {            // obj.rs:364-400
    for (int gi = 0; gi < 3; gi++) {
        const std::string &grp = g[gi];
        size_t dbl = grp.find("//");
        if (dbl != std::string::npos) {
            if (!read_index(grp.substr(0, dbl), &t->pos[gi])) return false;
            std::string rest = grp.substr(dbl + 2);
            size_t again = rest.find("//");
            if (!read_index(again == std::string::npos ? rest : rest.substr(0, again), &t->nrm[gi])) return false;
        } else if (grp.find('/') != std::string::npos) {
            std::vector<std::string> parts;
            size_t start = 0;
            for (;;) {
                size_t sl = grp.find('/', start);
                parts.push_back(grp.substr(start, sl == std::string::npos ? std::string::npos : sl - start));
                if (sl == std::string::npos) break;
                start = sl + 1;
            }
            if (parts.size() == 2) {
                if (!read_index(parts[0], &t->pos[gi]) || !read_index(parts[1], &t->tex[gi])) return false;
            } else if (parts.size() == 3) {
                if (!read_index(parts[0], &t->pos[gi]) || !read_index(parts[1], &t->tex[gi]) || !read_index(parts[2], &t->nrm[gi])) return false;
            }
        } else {
            if (!read_index(grp, &t->pos[gi])) return false;
        }
    }
    return true;
}

bool tris_from_face(const std::string &s, std::vector<ObjTri> *out) { // obj.rs:352-436
    std::vector<std::string> g = split_ws(s);
    auto emit = [&](size_t a, size_t b, size_t c) {
        std::string grp[3] = {g[a], g[b], g[c]};
        ObjTri t;
        if (!tri_from_groups(grp, &t)) return false;
        out->push_back(t);
        return true;
    };
    if (g.size() == 3) return emit(0, 1, 2);
    if (g.size() == 4) return emit(0, 1, 3) && emit(1, 2, 3);        // quad split, obj.rs:412-419
    if (g.size() >= 5) {                                             // n-gon fan, obj.rs:421-432
        for (size_t i = 0; i + 2 < g.size(); i++) if (!emit(0, i + 1, i + 2)) return false;
        return true;
    }
    return false;
}

} // namespace

extern "C" {

static int obj_load_impl(const char *path_c, MiptObj **out) {
    if (!path_c || !out) return fail("mipt_obj_load: null argument");
    *out = nullptr;
    const std::string path = path_c;
    {   // Scene::load (scene.rs:22-36)
        std::ifstream probe(path);
        if (!probe) return fail("Could not find scene at path: '" + path + "'");
        size_t dot = path.find_last_of('.');
        std::string fmt = dot == std::string::npos ? path : path.substr(dot + 1);
        if (fmt != "obj") return fail("Unsupported scene format '" + fmt + "' at path '" + path + "'");
    }
    std::vector<std::string> lines;
    if (!read_lines(path, &lines)) return fail("could not read '" + path + "'");
    std::unique_ptr<MiptObj> obj(new MiptObj);
    bool has_mtl = false;
    for (const std::string &l : lines) {                             // obj.rs:27-52
        size_t i = 0;
        while (i < l.size() && isspace((unsigned char)l[i])) i++;
        if (l.compare(i, 6, "mtllib") == 0) {
            if (!starts_with(l, "mtllib ")) return fail("malformed mtllib line: " + l);
            int rc = load_mtl(obj.get(), resource_path(path, l.substr(7)));
            if (rc) return rc;
            has_mtl = true;
            break;
        }
    }
    if (!has_mtl) {
        MiptMaterial m;
        mipt_material_default(&m);
        obj->material_names.push_back("default_material");
        obj->materials.push_back(m);
    }
    std::vector<std::array<float, 3>> positions, normals;
    std::vector<std::array<float, 2>> tex_coords;
    std::vector<ObjTri> otris;
    uint32_t active_material = 0;
    for (const std::string &line : lines) {                          // obj.rs:54-104
        std::vector<std::string> tok = split_ws(line);
        if (tok.empty()) continue;
        const std::string &p = tok[0];
        if (p == "v" || p == "vn") {
            std::array<float, 3> d = {0, 0, 0};
            if (tok.size() - 1 > 3) return fail("'" + p + "' line with more than 3 components (the reference panics): " + line);
            for (size_t i = 1; i < tok.size(); i++) if (!parse_f32(tok[i], &d[i - 1])) return fail("bad number in: " + line);
            (p == "v" ? positions : normals).push_back(d);
        } else if (p == "vt") {
            std::array<float, 2> d = {0, 0};
            if (tok.size() - 1 > 2) return fail("'vt' line with more than 2 components (the reference panics): " + line);
            for (size_t i = 1; i < tok.size(); i++) if (!parse_f32(tok[i], &d[i - 1])) return fail("bad number in: " + line);
            tex_coords.push_back(d);
        } else if (p == "usemtl") {
            if (has_mtl) {
                if (!starts_with(line, "usemtl ")) return fail("malformed usemtl line: " + line);
                const std::string name = line.substr(7);
                bool found = false;
                for (size_t i = 0; i < obj->material_names.size(); i++)
                    if (obj->material_names[i] == name) { active_material = (uint32_t)i; found = true; break; }
                if (!found) fprintf(stderr, "[mipt] material '%s' doesn't exist; keeping the active one\n", name.c_str());
            }
        } else if (p == "f") {
            if (!starts_with(line, "f ")) return fail("malformed face line: " + line);
            std::vector<ObjTri> ts;
            if (!tris_from_face(line.substr(2), &ts)) return fail("malformed face (bad, negative or <3 indices): " + line);
            for (ObjTri &t : ts) { t.material_id = active_material; otris.push_back(t); }
        }
    }
    if (normals.empty()) {                                           // flat normals, obj.rs:106-120
        for (size_t i = 0; i < otris.size(); i++) {
            ObjTri &t = otris[i];
            for (int k = 0; k < 3; k++) if (t.pos[k] >= positions.size()) return fail("face references a missing vertex");
            const auto &v1 = positions[t.pos[0]], &v2 = positions[t.pos[1]], &v3 = positions[t.pos[2]];
            const float ux = v2[0] - v1[0], uy = v2[1] - v1[1], uz = v2[2] - v1[2];
            const float vx = v3[0] - v1[0], vy = v3[1] - v1[1], vz = v3[2] - v1[2];
            const float cx = (uy * vz) - (uz * vy), cy = (uz * vx) - (ux * vz), cz = (ux * vy) - (uy * vx);
            const float len = sqrtf((cx * cx) + (cy * cy) + (cz * cz));
            normals.push_back({cx / len, cy / len, cz / len});
            t.nrm[0] = t.nrm[1] = t.nrm[2] = i;
        }
    }
    if (otris.empty()) return fail("'" + path + "' contains no faces (the reference panics in BVH::build)");
    // impl From<OBJ> for Scene (scene.rs:44-85): missing indices read as zeros
    obj->tris.resize(otris.size());
    for (size_t i = 0; i < otris.size(); i++) {
        MiptTriangle &dst = obj->tris[i];
        memset(&dst, 0, sizeof dst);
        for (int k = 0; k < 3; k++) {
            const ObjTri &t = otris[i];
            std::array<float, 3> P = {0, 0, 0}, N = {0, 0, 0};
            std::array<float, 2> T = {0, 0};
            if (t.pos[k] < positions.size()) P = positions[t.pos[k]];
            if (t.tex[k] < tex_coords.size()) T = tex_coords[t.tex[k]];
            if (t.nrm[k] < normals.size()) N = normals[t.nrm[k]];
            dst.vertices[k].position = {P[0], P[1], P[2]}; dst.vertices[k].tex_coord_x = T[0];
            dst.vertices[k].normal = {N[0], N[1], N[2]}; dst.vertices[k].tex_coord_y = T[1];
        }
        dst.material_id = otris[i].material_id;
    }
    obj->nodes.resize(2 * obj->tris.size());
    uint32_t n_nodes = 0;
    int rc = mipt_bvh_build(obj->tris.data(), (uint32_t)obj->tris.size(), obj->nodes.data(), (uint32_t)obj->nodes.size(), &n_nodes, 0);
    if (rc) return fail("BVH::build failed");
    obj->nodes.resize(n_nodes);
    for (const Tex &t : obj->textures) obj->tex_desc.push_back({t.w, t.h, t.rgba.data()});
    for (const std::string &s : obj->material_names) obj->name_ptrs.push_back(s.c_str());
    *out = obj.release();
    return MIPT_OK;
}

int mipt_obj_get(MiptObj *obj, MiptSceneDesc *desc, const char ***material_names) {
    if (!obj || !desc) return fail("mipt_obj_get: null argument");
    desc->tris = obj->tris.data(); desc->n_tris = (uint32_t)obj->tris.size();
    desc->nodes = obj->nodes.data(); desc->n_nodes = (uint32_t)obj->nodes.size();
    desc->materials = obj->materials.data(); desc->n_materials = (uint32_t)obj->materials.size();
    desc->textures = obj->tex_desc.data(); desc->n_textures = (uint32_t)obj->tex_desc.size();
    if (material_names) *material_names = obj->name_ptrs.data();
    return MIPT_OK;
}

void mipt_obj_free(MiptObj *obj) { delete obj; }

struct MiptImage { Tex t; };

static int texture_load_impl(const char *path, MiptImage **out, MiptTexture *desc_out, uint32_t *hash_out) {
    if (!path || !out || !desc_out) return fail("mipt_texture_load: null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(std::string("Could not find texture at path: '") + path + "'");          // texture.rs:14-17
    fclose(f);
    MiptImage *img = new MiptImage();
    std::string err;
    if (!load_any_texture(path, &img->t, &err)) { delete img; return fail(std::string("texture '") + path + "': " + err); }
    desc_out->width = img->t.w; desc_out->height = img->t.h; desc_out->rgba8 = img->t.rgba.data();
    if (hash_out) *hash_out = img->t.hash;
    *out = img;
    return MIPT_OK;
}
void mipt_texture_free(MiptImage *img) { delete img; }

static int image_save_png_impl(const char *path, uint32_t width, uint32_t height, uint32_t bits_per_sample, const void *rgba) {
    if (!path || !rgba) return fail("mipt_image_save_png: null argument");
    std::string err;
    if (!mipt_png::write_rgba(path, width, height, (int)bits_per_sample, rgba, &err)) return fail("Failed to write image data: " + err);   // renderer.rs:79-82
    return MIPT_OK;
}


// No C++ exception may cross the C ABI: allocation failures and parser surprises become status codes.
#define MIPT_NO_THROW(call)                                                                \
    try { return call; }                                                                  \
    catch (const std::bad_alloc &) { return fail("out of host memory"); }                 \
    catch (const std::exception &e) { return fail(std::string("internal error: ") + e.what()); }
int mipt_obj_load(const char *path, MiptObj **out) { MIPT_NO_THROW(obj_load_impl(path, out)) }
int mipt_texture_load(const char *path, MiptImage **out, MiptTexture *desc_out, uint32_t *hash_out) { MIPT_NO_THROW(texture_load_impl(path, out, desc_out, hash_out)) }
int mipt_image_save_png(const char *path, uint32_t width, uint32_t height, uint32_t bits_per_sample, const void *rgba) {
    MIPT_NO_THROW(image_save_png_impl(path, width, height, bits_per_sample, rgba))
}

} // extern "C"
