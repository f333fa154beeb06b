// obj_writer.cpp -- part of libmipt_diag.so (test infrastructure): writes a triangle array as a Wavefront OBJ body, fast enough for
// the 10 M-triangle stand-in scenes (rust_ray_tracing_amd/synth.py write_obj; the .mtl and the PNG textures are written from Python).
// Every triangle brings its own three v / vt / vn lines and one `f a/a/a b/b/b c/c/c` line; numbers are the shortest decimal that
// reads back as the same f32 (std::to_chars), so OBJ -> mipt_obj_load returns the array bit for bit.  A `usemtl` line is written
// where the material changes and at the start of every block of 32 768 triangles (which also puts usemtl state across the
// loader's chunk boundaries to work).
#include "../../include/mipt.h"
#include "../../include/mipt_diag.h"

#include <atomic>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
inline void put_f32(std::string &s, float v) {
    char buf[48];
    const auto r = std::to_chars(buf, buf + sizeof buf, v);
    s.append(buf, r.ptr);
}
inline void put_u64(std::string &s, unsigned long long v) {
    char buf[24];
    const auto r = std::to_chars(buf, buf + sizeof buf, v);
    s.append(buf, r.ptr);
}
} // namespace

extern "C" int mipt_diag_write_obj(const char *path, const void *tris_v, uint64_t n_tris, const char *mtllib, const char *const *material_names,
                                   uint32_t n_materials) {
    const MiptTriangle *tris = (const MiptTriangle *)tris_v;
    if (!path || !tris) return -1;
    FILE *f = fopen(path, "wb");
    if (!f) return -2;
    if (mtllib) fprintf(f, "# written by mipt_diag_write_obj: %llu triangles\nmtllib %s\n", (unsigned long long)n_tris, mtllib);
    constexpr uint64_t kBlock = 32768;
    const uint64_t n_blocks = (n_tris + kBlock - 1) / kBlock;
    unsigned threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    if (threads > 16) threads = 16;
    std::vector<std::string> out(threads);
    bool ok = true;
    for (uint64_t round = 0; round < n_blocks && ok; round += threads) {
        const unsigned n_now = (unsigned)((n_blocks - round) < threads ? (n_blocks - round) : threads);
        auto work = [&](unsigned t) {
            std::string &s = out[t];
            s.clear();
            const uint64_t b = (round + t) * kBlock, e = b + kBlock < n_tris ? b + kBlock : n_tris;
            uint32_t cur = 0xffffffffu;
            for (uint64_t i = b; i < e; i++) {
                const MiptTriangle &tr = tris[i];
                if (mtllib && material_names && tr.material_id != cur && tr.material_id < n_materials) {
                    s += "usemtl "; s += material_names[tr.material_id]; s += '\n';
                    cur = tr.material_id;
                }
                for (int k = 0; k < 3; k++) { const MiptVertex &v = tr.vertices[k]; s += "v "; put_f32(s, v.position.x); s += ' '; put_f32(s, v.position.y); s += ' '; put_f32(s, v.position.z); s += '\n'; }
                for (int k = 0; k < 3; k++) { const MiptVertex &v = tr.vertices[k]; s += "vt "; put_f32(s, v.tex_coord_x); s += ' '; put_f32(s, v.tex_coord_y); s += '\n'; }
                for (int k = 0; k < 3; k++) { const MiptVertex &v = tr.vertices[k]; s += "vn "; put_f32(s, v.normal.x); s += ' '; put_f32(s, v.normal.y); s += ' '; put_f32(s, v.normal.z); s += '\n'; }
                s += "f";
                for (int k = 0; k < 3; k++) { const unsigned long long id = 3ull * i + (unsigned long long)k + 1ull; s += ' '; put_u64(s, id); s += '/'; put_u64(s, id); s += '/'; put_u64(s, id); }
                s += '\n';
            }
        };
        std::vector<std::thread> th;
        try { for (unsigned t = 1; t < n_now; t++) th.emplace_back(work, t); }
        catch (const std::exception &) { for (auto &x : th) x.join(); th.clear(); for (unsigned t = 1; t < n_now; t++) work(t); }
        work(0);
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < n_now && ok; t++) ok = fwrite(out[t].data(), 1, out[t].size(), f) == out[t].size();
    }
    if (fclose(f) != 0) ok = false;
    return ok ? 0 : -2;
}
