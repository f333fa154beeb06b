// C++ host-mirror checks.  `test_host cpu <cornell.obj>` needs no GPU; `test_host gpu <cornell.obj> <expected.rgba>` renders
// BASELINE config 1 through Renderer::render and compares with the bytes the oracle produced.
#include "mipt_host.hpp"

#include <cstring>
#include <fstream>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    using namespace mipt;
    CHECK(argc >= 3);
    const std::string mode = argv[1], obj = argv[2];
    // Renderer::new validation (renderer.rs:15-34)
    RendererOptions o;
    o.is_realtime = false; o.backend = RendererBackend::MI355X; o.output_image_path = argc >= 5 ? argv[4] : "/tmp/mipt_test_host_out.png";
    { auto b = o; b.output_image_dimensions = {0, 5}; CHECK(!Renderer::create(b)); }
    { auto b = o; b.max_ray_depth = 0; CHECK(!Renderer::create(b)); }
    { auto b = o; b.samples = 0; CHECK(!Renderer::create(b)); }
    { auto b = o; b.output_image_path.reset(); CHECK(!Renderer::create(b)); }
    { auto b = o; b.is_realtime = true; CHECK(!Renderer::create(b)); }
    { RendererOptions d; CHECK(d.samples == 1 && d.max_ray_depth == 6 && d.output_image_dimensions.first == 1920 && d.is_realtime); }
    CHECK(!Scene::load("/nonexistent/scene.obj"));
    CHECK(!Texture::load("/nonexistent/tex.png"));                                       // texture.rs:14-17
    auto scene = Scene::load(obj);
    CHECK(scene && scene->tris.size() == 12 && scene->materials.size() == 4 && scene->materials[3].first == "light");
    CHECK(scene->bvh_nodes.size() % 2 == 1 && scene->bvh_nodes.size() <= 23);
    Camera cam;
    cam.position[0] = 3.2f;
    scene->set_camera(cam);
    CHECK(scene->camera.uniform.look_at[2][0] == -1.0f && scene->camera.uniform.position.x == 3.2f);   // looks down -X
    {   // image output (renderer.rs:66-83) through the ABI: 8- and 16-bit RGBA, read back by Texture::load
        const uint32_t w = 5, h = 3;
        std::vector<uint8_t> p8(w * h * 4);
        std::vector<uint16_t> p16(w * h * 4);
        for (size_t i = 0; i < p8.size(); i++) { p8[i] = (uint8_t)(i * 37 + 11); p16[i] = (uint16_t)(i * 2654435761u >> 7); }
        const std::string base = obj.substr(0, obj.find_last_of('/'));
        CHECK(mipt_image_save_png((base + "/w8.png").c_str(), w, h, 8, p8.data()) == MIPT_OK);
        CHECK(mipt_image_save_png((base + "/w16.png").c_str(), w, h, 16, p16.data()) == MIPT_OK);
        CHECK(mipt_image_save_png("/nonexistent-dir/x.png", w, h, 8, p8.data()) != MIPT_OK);
        CHECK(mipt_image_save_png((base + "/bad.png").c_str(), w, h, 12, p8.data()) != MIPT_OK);
        auto t8 = Texture::load(base + "/w8.png");
        CHECK(t8 && t8->width == w && t8->height == h);
        for (uint32_t y = 0; y < h; y++) CHECK(memcmp(&t8->pixel_data[(size_t)(h - 1 - y) * w * 4], &p8[(size_t)y * w * 4], w * 4) == 0);   // flipv
        auto t16 = Texture::load(base + "/w16.png");
        CHECK(t16 && t16->width == w);
        for (uint32_t y = 0; y < h; y++) for (uint32_t i = 0; i < w * 4; i++)
            CHECK(t16->pixel_data[(size_t)(h - 1 - y) * w * 4 + i] == (uint8_t)((p16[(size_t)y * w * 4 + i] + 128u) / 257u));
    }
    if (mode == "cpu") { printf("host mirror (cpu) ok\n"); return 0; }
    CHECK(argc >= 4);
    o.samples = 4; o.max_ray_depth = 64; o.output_image_dimensions = {256, 256};
    auto r = Renderer::create(o);
    CHECK(r);
    const std::vector<uint8_t> px = r->render(*scene);
    CHECK(px.size() == 256u * 256u * 4u);
    std::ifstream f(argv[3], std::ios::binary);
    std::vector<uint8_t> want((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    CHECK(want.size() == px.size() && memcmp(want.data(), px.data(), px.size()) == 0);
    const std::vector<uint8_t> px_node = r->render_node(*scene);                     // all GPUs of the node, one call (RCCL inside)
    CHECK(px_node.size() == px.size() && memcmp(px_node.data(), px.data(), px.size()) == 0);
    MiptMulti *kept = scene->node_handle(0);                                         // replicas + communicators are cached in the Scene ...
    const std::vector<uint8_t> px_node2 = r->render_node(*scene);                    // ... so a second frame re-uses them
    CHECK(scene->node_handle(0) == kept && px_node2 == px_node);
    scene->release_device();
    {   // the same file without the host BVH build: the tree is built on the GPU, the bytes are the same
        auto raw = Scene::load(obj, false);
        CHECK(raw && raw->tris.size() == 12 && raw->bvh_nodes.empty());
        raw->set_camera(cam);
        CHECK(r->render(*raw) == px);
        CHECK(r->render_node(*raw) == px);
    }
    printf("host mirror (gpu) ok: config 1 RGBA8 identical to the oracle\n");
    return 0;
}
