// obj_loader.cpp -- Scene::load for Wavefront OBJ + MTL (reference src/scene.rs:22-85,
// src/loader/obj.rs:16-436), restated in C++ on the host.  SURVEY.md 8(f) rank 1: the on-disk
// format on the input side of the path; config 1 ("12-triangle OBJ") goes through it.
//
// Behaviour follows the reference statement by statement; where the reference panics
// (unwrap on a malformed number, negative indices, >3 components, missing .mtl) this loader
// returns MIPT_ERR_IO with a message instead of unwinding across the C ABI.
// Deviations:
//  * materials keep INSERTION order (the reference iterates a HashMap, so its material ids are
//    a per-process random permutation -- src/loader/obj.rs:81-90; results are unaffected);
//  * map_* texture lines need an image decoder (the `image` crate, not vendored): PNG (png_decode.cpp), JPEG
//    (jpeg_decode.cpp), TGA / BMP (tga_bmp_decode.cpp) and binary PPM (P6) files are decoded here, chosen by file extension as image::open does;
//    other formats are reported and skipped.
#include "../../include/mipt.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <array>
#include <atomic>
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
#include <thread>
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

// obj.rs:344-350.  Indices as u32, saturated: an index >= 2^32 - 1 names a vertex no array can hold, i.e. "missing" (scene.rs:50-65 reads
// missing indices as zeros) exactly like the usize it stands for.  While a chunk is parsed material_id holds 1 + the chunk's usemtl
// event that governs the face (0 = the material active at the chunk's start).
struct ObjTri { uint32_t pos[3] = {0, 0, 0}, tex[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}; uint32_t material_id = 0; };

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

// ---- the OBJ body: chunked, multi-threaded, allocation-free per line -------------------------------------------------------------
// The file is mapped and cut into chunks at line ends; worker threads parse chunks into private arrays that are concatenated in file
// order afterwards.  Nothing a line means depends on an earlier line except the active material (obj.rs:76-92), which is resolved in
// a sequential pass over the chunks' usemtl events; indices are absolute (negative = relative ones are refused like the reference's
// panic), so faces need no vertex counts.  The arrays are the ones the line-by-line loader produced (tests/test_obj_loader.py compares
// them byte for byte on the 1 M-triangle scene and on every small fixture).
struct Span {
    const char *b, *e;
    size_t size() const { return (size_t)(e - b); }
    std::string str() const { return std::string(b, e); }
};
inline bool is_ws(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }          // isspace in the "C" locale

// str::parse::<f32> as strtof does it (the line loader's parse_f32), with a fast path for plain decimals: mantissa < 2^53 and
// |exponent| <= 22 make `m * 10^k` / `m / 10^k` ONE correctly rounded double operation (both operands exact), and the double rounds
// to the same float as the decimal itself unless it sits exactly on the midpoint of two floats -- detected, and left to strtof
// together with everything unusual (hex, inf, nan, > 19 digits, huge exponents, junk).
bool parse_f32_slow(const char *b, const char *e, float *out) {
    char buf[128];
    std::string big;
    const char *z;
    if ((size_t)(e - b) < sizeof buf) { memcpy(buf, b, (size_t)(e - b)); buf[e - b] = '\0'; z = buf; }
    else { big.assign(b, e); z = big.c_str(); }
    if (!*z || memchr(b, '\0', (size_t)(e - b))) return false;
    char *end = nullptr;
    const float v = strtof(z, &end);
    if (end == z || *end != '\0') return false;
    *out = v;
    return true;
}
inline bool parse_f32_span(const char *b, const char *e, float *out) {
    static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *p = b;
    if (p == e) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; p++; }
    uint64_t m = 0;
    int sig = 0, frac = 0;
    bool any = false, slow = false;
    for (; p < e && *p >= '0' && *p <= '9'; p++) { any = true; if (sig < 19) { m = m * 10 + (uint64_t)(*p - '0'); if (m) sig++; } else slow = true; }
    if (p < e && *p == '.') {
        p++;
        for (; p < e && *p >= '0' && *p <= '9'; p++) { any = true; if (sig < 19) { m = m * 10 + (uint64_t)(*p - '0'); if (m) sig++; frac++; } else slow = true; }
    }
    int ex = 0;
    if (any && p < e && (*p == 'e' || *p == 'E')) {
        const char *q = p + 1;
        bool eneg = false;
        if (q < e && (*q == '+' || *q == '-')) { eneg = *q == '-'; q++; }
        if (q == e || *q < '0' || *q > '9') slow = true;
        for (; q < e && *q >= '0' && *q <= '9'; q++) { if (ex < 100000) ex = ex * 10 + (*q - '0'); }
        if (eneg) ex = -ex;
        p = q;
    }
    if (!any || slow || p != e) return parse_f32_slow(b, e, out);
    if (m == 0) { *out = neg ? -0.0f : 0.0f; return true; }
    const int e10 = ex - frac;
    if (m >= (1ull << 53) || e10 < -22 || e10 > 22) return parse_f32_slow(b, e, out);
    const double d = e10 < 0 ? (double)m / p10[-e10] : (double)m * p10[e10];
    const float f = (float)d;
    if ((double)f != d) {
        const float g = nextafterf(f, d > (double)f ? INFINITY : -INFINITY);
        if (fabs(d - (double)f) == fabs((double)g - d)) return parse_f32_slow(b, e, out);     // on a float midpoint: the decimal decides
    }
    *out = neg ? -f : f;
    return true;
}

// obj.rs:355-362 through strtol's eyes: [+-]digits, all of the token; value - 1 must not be negative
inline bool read_index(const char *b, const char *e, uint32_t *out) {
    const char *p = b;
    if (p == e) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; p++; }
    if (p == e) return false;
    uint64_t v = 0;
    int nd = 0;
    for (; p < e && *p >= '0' && *p <= '9'; p++) { if (nd < 18) { v = v * 10 + (uint64_t)(*p - '0'); nd++; } else v = ~0ull >> 1; }
    if (p != e) return false;
    if (neg || v == 0) return false;                                     // v - 1 < 0: the reference panics on negative (relative) indices
    v -= 1;
    *out = v >= 0xffffffffull ? 0xffffffffu : (uint32_t)v;
    return true;
}
inline const char *find2(const char *b, const char *e, char c) {         // first "cc" in [b, e)
    for (const char *p = b; p + 1 < e; p++) if (p[0] == c && p[1] == c) return p;
    return nullptr;
}
bool tri_from_groups(const Span g[3], ObjTri *t) {                       // obj.rs:364-400
    for (int gi = 0; gi < 3; gi++) {
        const char *b = g[gi].b, *e = g[gi].e;
        if (const char *dbl = find2(b, e, '/')) {
            if (!read_index(b, dbl, &t->pos[gi])) return false;
            const char *rest = dbl + 2, *again = find2(rest, e, '/');
            if (!read_index(rest, again ? again : e, &t->nrm[gi])) return false;
        } else if (const char *s1 = (const char *)memchr(b, '/', (size_t)(e - b))) {
            const char *s2 = (const char *)memchr(s1 + 1, '/', (size_t)(e - s1 - 1));
            if (!s2) {
                if (!read_index(b, s1, &t->pos[gi]) || !read_index(s1 + 1, e, &t->tex[gi])) return false;
            } else if (!memchr(s2 + 1, '/', (size_t)(e - s2 - 1))) {
                if (!read_index(b, s1, &t->pos[gi]) || !read_index(s1 + 1, s2, &t->tex[gi]) || !read_index(s2 + 1, e, &t->nrm[gi])) return false;
            }                                                            // four or more parts: the reference's match has no arm -- indices stay 0
        } else {
            if (!read_index(b, e, &t->pos[gi])) return false;
        }
    }
    return true;
}

struct Chunk {
    const char *b = nullptr, *e = nullptr;
    std::vector<std::array<float, 3>> pos, nrm;
    std::vector<std::array<float, 2>> tex;
    std::vector<ObjTri> tris;
    std::vector<int64_t> events;        // one per usemtl line: the material id it names, -1 = no such material (the active one stays)
    std::vector<std::string> unknown;   // names of those, for the log lines
    bool failed = false;
    std::string err;
};

// one line of the body (obj.rs:54-104); returns false with c->err set on the errors the reference panics on
bool parse_line(Chunk *c, const char *lb, const char *le, bool has_mtl, const std::vector<std::string> &material_names, std::vector<Span> *groups) {
    const char *p = lb;
    while (p < le && is_ws(*p)) p++;
    if (p == le) return true;
    const char *q = p;
    while (q < le && !is_ws(*q)) q++;
    const size_t tl = (size_t)(q - p);
    auto next_tok = [&](const char **tb, const char **te) -> bool {
        while (q < le && is_ws(*q)) q++;
        if (q == le) return false;
        *tb = q;
        while (q < le && !is_ws(*q)) q++;
        *te = q;
        return true;
    };
    const bool is_v = tl == 1 && p[0] == 'v', is_vn = tl == 2 && p[0] == 'v' && p[1] == 'n', is_vt = tl == 2 && p[0] == 'v' && p[1] == 't';
    if (is_v || is_vn || is_vt) {
        float d[3] = {0.0f, 0.0f, 0.0f};
        const int cap = is_vt ? 2 : 3;
        int n = 0;
        const char *tb, *te;
        while (next_tok(&tb, &te)) {
            if (n == cap) {
                c->err = is_vt ? "'vt' line with more than 2 components (the reference panics): " + std::string(lb, le)
                               : "'" + std::string(p, p + tl) + "' line with more than 3 components (the reference panics): " + std::string(lb, le);
                return false;
            }
            if (!parse_f32_span(tb, te, &d[n])) { c->err = "bad number in: " + std::string(lb, le); return false; }
            n++;
        }
        if (is_vt) c->tex.push_back({d[0], d[1]});
        else (is_v ? c->pos : c->nrm).push_back({d[0], d[1], d[2]});
        return true;
    }
    if (tl == 6 && memcmp(p, "usemtl", 6) == 0) {
        if (!has_mtl) return true;
        if ((size_t)(le - lb) < 7 || memcmp(lb, "usemtl ", 7) != 0) { c->err = "malformed usemtl line: " + std::string(lb, le); return false; }
        const size_t nl = (size_t)(le - lb) - 7;
        int64_t id = -1;
        for (size_t i = 0; i < material_names.size(); i++)
            if (material_names[i].size() == nl && memcmp(material_names[i].data(), lb + 7, nl) == 0) { id = (int64_t)i; break; }
        if (id < 0) c->unknown.emplace_back(lb + 7, le);
        c->events.push_back(id);
        return true;
    }
    if (tl == 1 && p[0] == 'f') {
        if ((size_t)(le - lb) < 2 || lb[0] != 'f' || lb[1] != ' ') { c->err = "malformed face line: " + std::string(lb, le); return false; }
        groups->clear();
        q = lb + 2;
        const char *tb, *te;
        while (next_tok(&tb, &te)) groups->push_back(Span{tb, te});
        const std::vector<Span> &g = *groups;
        bool ok = true;
        auto emit = [&](size_t a, size_t b, size_t cc) {
            const Span grp[3] = {g[a], g[b], g[cc]};
            ObjTri t;
            if (!tri_from_groups(grp, &t)) return false;
            t.material_id = (uint32_t)c->events.size();                  // 0 = inherited from before this chunk
            c->tris.push_back(t);
            return true;
        };
        if (g.size() == 3) ok = emit(0, 1, 2);
        else if (g.size() == 4) ok = emit(0, 1, 3) && emit(1, 2, 3);      // quad split, obj.rs:412-419
        else if (g.size() >= 5) { for (size_t i = 0; i + 2 < g.size() && ok; i++) ok = emit(0, i + 1, i + 2); }   // n-gon fan, obj.rs:421-432
        else ok = false;
        if (!ok) { c->err = "malformed face (bad, negative or <3 indices): " + std::string(lb, le); return false; }
        return true;
    }
    return true;
}

void parse_chunk(Chunk *c, bool has_mtl, const std::vector<std::string> &material_names) {
    std::vector<Span> groups;
    const size_t bytes = (size_t)(c->e - c->b);
    c->pos.reserve(bytes / 96); c->tris.reserve(bytes / 160);
    const char *lb = c->b;
    while (lb < c->e) {
        const char *nl = (const char *)memchr(lb, '\n', (size_t)(c->e - lb));
        const char *le = nl ? nl : c->e;
        const char *te = le;
        if (nl && te > lb && te[-1] == '\r') te--;                        // lines() strips "\n" or "\r\n" (a bare CR at the end of the file stays)
        if (!parse_line(c, lb, te, has_mtl, material_names, &groups)) { c->failed = true; return; }
        lb = nl ? nl + 1 : c->e;
    }
}

struct Mapped {                                                          // read-only view of a file
    const char *data = nullptr;
    size_t size = 0;
    bool mapped = false;
    std::string fallback;
    ~Mapped() { if (mapped) munmap((void *)data, size); }
    bool open(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); return false; }
        size = (size_t)st.st_size;
        if (size == 0) { close(fd); data = ""; return true; }
        void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            (void)madvise(m, size, MADV_SEQUENTIAL);
            data = (const char *)m; mapped = true; close(fd); return true;
        }
        fallback.resize(size);                                           // e.g. a pipe or a filesystem without mmap
        size_t got = 0;
        while (got < size) { const ssize_t r = read(fd, &fallback[got], size - got); if (r <= 0) break; got += (size_t)r; }
        close(fd);
        if (got != size) return false;
        data = fallback.data();
        return true;
    }
};

template <class F> void run_parallel(size_t n_items, unsigned threads, F body) {      // body(item) for every item, dynamic schedule
    std::atomic<size_t> next{0};
    auto work = [&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= n_items) return; body(i); } };
    std::vector<std::thread> th;
    try { for (unsigned t = 1; t < threads && t < n_items; t++) th.emplace_back(work); }
    catch (const std::exception &) {}                                    // fewer helpers: the caller works through the rest
    work();
    for (auto &x : th) x.join();
}

} // namespace

extern "C" {

static int obj_load_impl(const char *path_c, MiptObj **out, bool build_bvh) {
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
    Mapped file;
    if (!file.open(path)) return fail("could not read '" + path + "'");
    const char *const fb = file.data, *const fe = file.data + file.size;
    std::unique_ptr<MiptObj> obj(new MiptObj);
    bool has_mtl = false;
    {   // obj.rs:27-52: the FIRST line whose first word starts with "mtllib" names the material library
        const char *p = fb;
        while (p < fe) {
            const char *hit = (const char *)memmem(p, (size_t)(fe - p), "mtllib", 6);
            if (!hit) break;
            const char *lb = hit;
            while (lb > fb && lb[-1] != '\n') lb--;
            bool first_word = true;
            for (const char *c = lb; c < hit; c++) if (!is_ws(*c)) { first_word = false; break; }
            if (!first_word) { p = hit + 6; continue; }
            const char *le = (const char *)memchr(hit, '\n', (size_t)(fe - hit));
            if (!le) le = fe;
            if (le > lb && le[-1] == '\r') le--;
            const std::string l(lb, le);
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
    // ---- the body, in chunks cut at line ends ----
    unsigned threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    if (threads > 16) threads = 16;
    const size_t want_chunks = file.size < ((size_t)1 << 20) ? 1 : (size_t)threads * 4;
    std::vector<Chunk> chunks;
    {
        const char *p = fb;
        const size_t step = file.size / want_chunks + 1;
        while (p < fe) {
            const char *q = (size_t)(fe - p) > step ? p + step : fe;
            if (q < fe) { const char *nl = (const char *)memchr(q, '\n', (size_t)(fe - q)); q = nl ? nl + 1 : fe; }
            Chunk c;
            c.b = p; c.e = q;
            chunks.push_back(std::move(c));
            p = q;
        }
    }
    run_parallel(chunks.size(), threads, [&](size_t i) { parse_chunk(&chunks[i], has_mtl, obj->material_names); });
    for (const Chunk &c : chunks) {                                      // the first error in file order, as a line-by-line reader meets it
        for (const std::string &name : c.unknown) fprintf(stderr, "[mipt] material '%s' doesn't exist; keeping the active one\n", name.c_str());
        if (c.failed) return fail(c.err);
    }
    // ---- concatenate in file order; resolve the active material across chunks (obj.rs:76-92) ----
    std::vector<size_t> o_pos(chunks.size() + 1, 0), o_nrm(chunks.size() + 1, 0), o_tex(chunks.size() + 1, 0), o_tri(chunks.size() + 1, 0);
    std::vector<uint32_t> start_material(chunks.size(), 0);
    {
        uint32_t active = 0;
        for (size_t i = 0; i < chunks.size(); i++) {
            o_pos[i + 1] = o_pos[i] + chunks[i].pos.size(); o_nrm[i + 1] = o_nrm[i] + chunks[i].nrm.size();
            o_tex[i + 1] = o_tex[i] + chunks[i].tex.size(); o_tri[i + 1] = o_tri[i] + chunks[i].tris.size();
            start_material[i] = active;
            for (int64_t &ev : chunks[i].events) { if (ev >= 0) active = (uint32_t)ev; ev = (int64_t)active; }   // event -> the material active after it
        }
    }
    if (o_tri.back() > 0xffffffffull) return fail("'" + path + "' holds more than 2^32 triangles");
    std::vector<std::array<float, 3>> positions(o_pos.back()), normals(o_nrm.back());
    std::vector<std::array<float, 2>> tex_coords(o_tex.back());
    std::vector<ObjTri> otris(o_tri.back());
    run_parallel(chunks.size(), threads, [&](size_t i) {
        Chunk &c = chunks[i];
        if (!c.pos.empty()) memcpy(&positions[o_pos[i]], c.pos.data(), c.pos.size() * sizeof c.pos[0]);
        if (!c.nrm.empty()) memcpy(&normals[o_nrm[i]], c.nrm.data(), c.nrm.size() * sizeof c.nrm[0]);
        if (!c.tex.empty()) memcpy(&tex_coords[o_tex[i]], c.tex.data(), c.tex.size() * sizeof c.tex[0]);
        for (size_t k = 0; k < c.tris.size(); k++) {
            ObjTri t = c.tris[k];
            t.material_id = t.material_id == 0 ? start_material[i] : (uint32_t)c.events[t.material_id - 1];
            otris[o_tri[i] + k] = t;
        }
        std::vector<std::array<float, 3>>().swap(c.pos); std::vector<std::array<float, 3>>().swap(c.nrm);
        std::vector<std::array<float, 2>>().swap(c.tex); std::vector<ObjTri>().swap(c.tris);
    });
    const size_t n_slices = otris.size() < 4096 ? 1 : (size_t)threads * 4;
    auto slice = [&](size_t i, size_t *b2, size_t *e2) { *b2 = otris.size() * i / n_slices; *e2 = otris.size() * (i + 1) / n_slices; };
    if (normals.empty()) {                                           // flat normals, obj.rs:106-120: triangle i gets normal i
        for (const ObjTri &t : otris)
            for (int k = 0; k < 3; k++) if (t.pos[k] >= positions.size()) return fail("face references a missing vertex");
        normals.resize(otris.size());
        run_parallel(n_slices, threads, [&](size_t si) {
            size_t b2, e2;
            slice(si, &b2, &e2);
            for (size_t i = b2; i < e2; i++) {
                ObjTri &t = otris[i];
                const auto &v1 = positions[t.pos[0]], &v2 = positions[t.pos[1]], &v3 = positions[t.pos[2]];
                const float ux = v2[0] - v1[0], uy = v2[1] - v1[1], uz = v2[2] - v1[2];
                const float vx = v3[0] - v1[0], vy = v3[1] - v1[1], vz = v3[2] - v1[2];
                const float cx = (uy * vz) - (uz * vy), cy = (uz * vx) - (ux * vz), cz = (ux * vy) - (uy * vx);
                const float len = sqrtf((cx * cx) + (cy * cy) + (cz * cz));
                normals[i] = {cx / len, cy / len, cz / len};
                t.nrm[0] = t.nrm[1] = t.nrm[2] = i >= 0xffffffffull ? 0xffffffffu : (uint32_t)i;
            }
        });
    }
    if (otris.empty()) return fail("'" + path + "' contains no faces (the reference panics in BVH::build)");
    // impl From<OBJ> for Scene (scene.rs:44-85): missing indices read as zeros
    obj->tris.resize(otris.size());
    run_parallel(n_slices, threads, [&](size_t si) {
        size_t b2, e2;
        slice(si, &b2, &e2);
        for (size_t i = b2; i < e2; i++) {
            MiptTriangle &dst = obj->tris[i];
            memset(&dst, 0, sizeof dst);
            const ObjTri &t = otris[i];
            for (int k = 0; k < 3; k++) {
                std::array<float, 3> P = {0, 0, 0}, N = {0, 0, 0};
                std::array<float, 2> T = {0, 0};
                if (t.pos[k] < positions.size()) P = positions[t.pos[k]];
                if (t.tex[k] < tex_coords.size()) T = tex_coords[t.tex[k]];
                if (t.nrm[k] < normals.size()) N = normals[t.nrm[k]];
                dst.vertices[k].position = {P[0], P[1], P[2]}; dst.vertices[k].tex_coord_x = T[0];
                dst.vertices[k].normal = {N[0], N[1], N[2]}; dst.vertices[k].tex_coord_y = T[1];
            }
            dst.material_id = t.material_id;
        }
    });
    if (build_bvh) {                                                 // BVH::build(&mut scene), scene.rs:80
        obj->nodes.resize(2 * obj->tris.size());
        uint32_t n_nodes = 0;
        int rc = mipt_bvh_build(obj->tris.data(), (uint32_t)obj->tris.size(), obj->nodes.data(), (uint32_t)obj->nodes.size(), &n_nodes, 0);
        if (rc) return fail("BVH::build failed");
        obj->nodes.resize(n_nodes);
    }
    for (const Tex &t : obj->textures) obj->tex_desc.push_back({t.w, t.h, t.rgba.data()});
    for (const std::string &s : obj->material_names) obj->name_ptrs.push_back(s.c_str());
    *out = obj.release();
    return MIPT_OK;
}

int mipt_obj_get(MiptObj *obj, MiptSceneDesc *desc, const char ***material_names) {
    if (!obj || !desc) return fail("mipt_obj_get: null argument");
    desc->tris = obj->tris.data(); desc->n_tris = (uint32_t)obj->tris.size();
    desc->nodes = obj->nodes.empty() ? nullptr : obj->nodes.data(); desc->n_nodes = (uint32_t)obj->nodes.size();   // empty: mipt_obj_load_triangles
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
int mipt_obj_load(const char *path, MiptObj **out) { MIPT_NO_THROW(obj_load_impl(path, out, true)) }
int mipt_obj_load_triangles(const char *path, MiptObj **out) { MIPT_NO_THROW(obj_load_impl(path, out, false)) }
int mipt_texture_load(const char *path, MiptImage **out, MiptTexture *desc_out, uint32_t *hash_out) { MIPT_NO_THROW(texture_load_impl(path, out, desc_out, hash_out)) }
int mipt_image_save_png(const char *path, uint32_t width, uint32_t height, uint32_t bits_per_sample, const void *rgba) {
    MIPT_NO_THROW(image_save_png_impl(path, width, height, bits_per_sample, rgba))
}

} // extern "C"
