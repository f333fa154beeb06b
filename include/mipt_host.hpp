// mipt_host.hpp -- C++ host-side mirror of the reference's Renderer / Scene interface for the path-tracing hot path,
// header-only over the C ABI of mipt.h.  The reference is Rust (no toolchain in this image), so the host side above the
// ABI is C++: same type and member names, same argument meaning, same error behaviour --
//   Renderer::create  = Renderer::new            (src/renderer.rs:14-48)  -> std::nullopt + the reference's log line
//   Renderer::render                             (src/renderer.rs:50-85)  -> the Vec<u8> of the backend arm
//   Scene::load / Scene::set_camera              (src/scene.rs:22-41)
//   Camera::update_view                          (src/scene.rs:181-194)
//   RendererBackend::{GPU, CPU} + MI355X         (src/renderer/backend.rs:6-10)
#pragma once
#include "mipt.h"

#include <cstdio>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <vector>

namespace mipt {

inline void log_error(const std::string &m) { fprintf(stderr, "[ERROR] %s\n", m.c_str()); }   // src/log.rs:22-29

enum class RendererBackend { GPU, CPU, MI355X };

struct RendererOptions {                                   // src/renderer.rs:96-116 (same defaults)
    size_t samples = 1;
    size_t max_ray_depth = 6;
    std::pair<size_t, size_t> output_image_dimensions{1920, 1080};
    std::optional<std::string> output_image_path;
    RendererBackend backend = RendererBackend::GPU;
    bool is_realtime = true;
    // MI355X arm only: the recommended traversal (best-hit cull with MIPT_CULL_MARGIN_SAFE: the CPU backend's frame bit for bit,
    // 1.7x faster).  MIPT_TRAVERSAL_REFERENCE = the CPU backend's own un-culled traversal (ray.rs:69-81).
    uint32_t traversal = MIPT_TRAVERSAL_CULLED;
    int device_id = 0;
};

struct Camera {                                            // src/scene.rs:169-195
    float pitch = 0.0f, yaw = 0.0f;
    float position[3] = {0.0f, 0.0f, 0.0f};
    MiptCamera uniform{};                                  // look_at + position as uploaded (gpu.rs:480-486)
    void update_view() { mipt_camera_from_pose(position, pitch, yaw, &uniform); }
};

struct Texture {                                           // src/texture.rs:4-10
    uint32_t width = 0, height = 0, hash = 0;
    std::vector<uint8_t> pixel_data;                       // RGBA8, rows as Texture::load stores them (flipv applied)
    static std::optional<Texture> load(const std::string &path) {   // src/texture.rs:13-31
        MiptImage *img = nullptr;
        MiptTexture d{};
        Texture t;
        if (mipt_texture_load(path.c_str(), &img, &d, &t.hash) != MIPT_OK) {
            log_error(mipt_last_error());
            return std::nullopt;
        }
        t.width = d.width; t.height = d.height;
        t.pixel_data.assign(d.rgba8, d.rgba8 + (size_t)d.width * d.height * 4);
        mipt_texture_free(img);
        return t;
    }
};

class Scene {                                              // src/scene.rs:12-19
  public:
    Scene() = default;
    // The device residency (replicas + communicators) is NOT shared between copies: a copy starts without one and creates its own on
    // its first render_node -- two copies rendering from two threads must not meet in one MiptMulti's buffers and streams.
    Scene(const Scene &o) : tris(o.tris), materials(o.materials), textures(o.textures), bvh_nodes(o.bvh_nodes), camera(o.camera) {}
    Scene &operator=(const Scene &o) {
        if (this != &o) { tris = o.tris; materials = o.materials; textures = o.textures; bvh_nodes = o.bvh_nodes; camera = o.camera; release_device(); }
        return *this;
    }
    Scene(Scene &&) = default;
    Scene &operator=(Scene &&) = default;

    std::vector<MiptTriangle> tris;
    std::vector<std::pair<std::string, MiptMaterial>> materials;   // name -> Material, in material-id order
    std::vector<Texture> textures;
    std::vector<MiptNode> bvh_nodes;                       // scene.bvh.nodes
    Camera camera;

    // build_bvh = false: Scene::load without the BVH::build call at scene.rs:80 -- triangles in file order, bvh_nodes empty; the
    // MI355X arm then builds the tree on the GPU (Renderer::render -> mipt_scene_create_from_triangles).
    static std::optional<Scene> load(const std::string &path, bool build_bvh = true) {    // src/scene.rs:22-36
        MiptObj *obj = nullptr;
        if ((build_bvh ? mipt_obj_load(path.c_str(), &obj) : mipt_obj_load_triangles(path.c_str(), &obj)) != MIPT_OK) { log_error(mipt_last_error()); return std::nullopt; }
        MiptSceneDesc d{};
        const char **names = nullptr;
        mipt_obj_get(obj, &d, &names);
        Scene s;
        s.release_device();
        s.tris.assign(d.tris, d.tris + d.n_tris);
        if (d.n_nodes) s.bvh_nodes.assign(d.nodes, d.nodes + d.n_nodes);
        for (uint32_t i = 0; i < d.n_materials; i++) s.materials.emplace_back(names[i], d.materials[i]);
        for (uint32_t i = 0; i < d.n_textures; i++) {
            Texture t;
            t.width = d.textures[i].width; t.height = d.textures[i].height;
            t.pixel_data.assign(d.textures[i].rgba8, d.textures[i].rgba8 + (size_t)t.width * t.height * 4);
            s.textures.push_back(std::move(t));
        }
        mipt_obj_free(obj);
        return s;
    }
    void set_camera(const Camera &c) { camera = c; camera.update_view(); }   // src/scene.rs:38-41

    // Device residency for Renderer::render_node: the per-device replicas, streams and RCCL communicators (MiptMulti) are
    // created on first use and kept -- like the wgpu backend's State, built once in State::new (gpu.rs:96-118) -- so a
    // second frame costs no upload and no ncclCommInitAll.  build_bvh() invalidates it; after editing the public tris / bvh_nodes /
    // materials / textures directly call release_device() (the camera is passed per frame and needs no re-upload).
    void release_device() const { multi_.reset(); multi_devices_ = -1; }
    MiptMulti *node_handle(int n_devices) const {
        if (multi_ && multi_devices_ == n_devices) return multi_.get();
        release_device();
        std::vector<MiptMaterial> mats;
        for (const auto &kv : materials) mats.push_back(kv.second);
        std::vector<MiptTexture> texs;
        for (const Texture &t : textures) texs.push_back({t.width, t.height, t.pixel_data.data()});
        MiptSceneDesc d{tris.data(), (uint32_t)tris.size(), bvh_nodes.data(), (uint32_t)bvh_nodes.size(),
                        mats.data(), (uint32_t)mats.size(), texs.data(), (uint32_t)texs.size()};
        MiptMulti *m = nullptr;
        if ((bvh_nodes.empty() ? mipt_multi_create_from_triangles(&d, nullptr, n_devices, &m) : mipt_multi_create(&d, nullptr, n_devices, &m)) != MIPT_OK) { log_error(mipt_last_error()); return nullptr; }
        multi_ = std::shared_ptr<MiptMulti>(m, [](MiptMulti *p) { mipt_multi_destroy(p); });
        multi_devices_ = n_devices;
        return m;
    }

  private:
    mutable std::shared_ptr<MiptMulti> multi_;
    mutable int multi_devices_ = -1;

  public:
    void build_bvh(uint32_t threads = 0) {                 // BVH::build (src/bvh.rs:13-54)
        bvh_nodes.resize(tris.empty() ? 1 : 2 * tris.size());
        uint32_t n = 0;
        if (mipt_bvh_build(tris.data(), (uint32_t)tris.size(), bvh_nodes.data(), (uint32_t)bvh_nodes.size(), &n, threads) != MIPT_OK) n = 0;
        bvh_nodes.resize(n);
        release_device();                                  // the replicas hold the old tree and triangle order
    }
};

class Renderer {                                           // src/renderer.rs:8-85
  public:
    RendererOptions options;

    static std::optional<Renderer> create(const RendererOptions &o) {      // Renderer::new, renderer.rs:14-48
        if (o.output_image_dimensions.first == 0 || o.output_image_dimensions.second == 0) { log_error("Width and height must be greater than 0"); return std::nullopt; }
        if (o.max_ray_depth == 0) { log_error("Max ray depth must be greater than 0"); return std::nullopt; }
        if (o.samples == 0) { log_error("Sample count must be greater than 0"); return std::nullopt; }
        if (!o.output_image_path && !o.is_realtime) { log_error("Output image path must be Some if realtime mode is disabled"); return std::nullopt; }
        if (o.backend != RendererBackend::GPU && o.is_realtime) { log_error("Only the GPU backend is supported for realtime mode"); return std::nullopt; }
        Renderer r;
        r.options = o;
        return r;
    }

    // The offline arm of Renderer::render (renderer.rs:55-64): returns the backend's Vec<u8> (w*h*4 RGBA8 for the
    // MI355X arm, exactly what cpu::render_scene returns).  Empty on error (message logged).
    std::vector<uint8_t> render(const Scene &scene) const {
        if (options.backend != RendererBackend::MI355X) { log_error("this build only provides RendererBackend::MI355X"); return {}; }
        std::vector<MiptMaterial> mats;
        for (const auto &kv : scene.materials) mats.push_back(kv.second);
        std::vector<MiptTexture> texs;
        for (const Texture &t : scene.textures) texs.push_back({t.width, t.height, t.pixel_data.data()});
        MiptSceneDesc d{scene.tris.data(), (uint32_t)scene.tris.size(), scene.bvh_nodes.data(), (uint32_t)scene.bvh_nodes.size(),
                        mats.data(), (uint32_t)mats.size(), texs.data(), (uint32_t)texs.size()};
        MiptScene *h = nullptr;
        // no host-built tree (Scene::load(path, false)): BVH::build + layout on the GPU, identical tree and bytes
        const int rc_create = scene.bvh_nodes.empty() ? mipt_scene_create_from_triangles(&d, options.device_id, &h) : mipt_scene_create(&d, options.device_id, &h);
        if (rc_create != MIPT_OK) { log_error(mipt_last_error()); return {}; }
        MiptOptions o{};
        o.width = (uint32_t)options.output_image_dimensions.first; o.height = (uint32_t)options.output_image_dimensions.second;
        o.samples = (uint32_t)options.samples; o.max_ray_depth = (uint32_t)options.max_ray_depth;
        o.traversal = options.traversal; o.cull_margin = MIPT_CULL_MARGIN_SAFE;
        std::vector<uint8_t> out((size_t)o.width * o.height * 4);
        const int rc = mipt_render(h, &scene.camera.uniform, &o, nullptr, out.data(), nullptr);
        mipt_scene_destroy(h);
        if (rc != MIPT_OK) { log_error(mipt_last_error()); return {}; }
        if (options.output_image_path) {                                   // renderer.rs:66-83 (as Rgba8: SURVEY T12)
            if (mipt_image_save_png(options.output_image_path->c_str(), o.width, o.height, 8, out.data()) == MIPT_OK)
                std::fprintf(stderr, "[INFO] Succesfully wrote image data to '%s'\n", options.output_image_path->c_str());
            else
                log_error(mipt_last_error());
        }
        return out;
    }

    // The same arm over EVERY GPU of the node from this single-threaded host (mipt_render_multi: scene replicas, RCCL
    // communicators and the one gather / sum-reduce per frame live inside the library).  mode = MIPT_MULTI_TILES reproduces
    // render()'s bytes exactly; MIPT_MULTI_SAMPLES uses the wgpu shader's per-sample seeds (rt_compute.wgsl:102).
    std::vector<uint8_t> render_node(const Scene &scene, uint32_t mode = MIPT_MULTI_TILES, int n_devices = 0) const {
        if (options.backend != RendererBackend::MI355X) { log_error("this build only provides RendererBackend::MI355X"); return {}; }
        MiptMulti *m = scene.node_handle(n_devices);                       // cached in the Scene: created on the first frame only
        if (!m) return {};
        MiptOptions o{};
        o.width = (uint32_t)options.output_image_dimensions.first; o.height = (uint32_t)options.output_image_dimensions.second;
        o.samples = (uint32_t)options.samples; o.max_ray_depth = (uint32_t)options.max_ray_depth;
        o.traversal = options.traversal; o.cull_margin = MIPT_CULL_MARGIN_SAFE;
        std::vector<uint8_t> out((size_t)o.width * o.height * 4);
        const int rc = mipt_render_multi(m, &scene.camera.uniform, &o, mode, nullptr, out.data(), nullptr);
        if (rc != MIPT_OK) { log_error(mipt_last_error()); return {}; }
        return out;
    }
};

} // namespace mipt
