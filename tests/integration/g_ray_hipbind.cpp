// TEST INFRASTRUCTURE: the reference's g_ray (/root/reference/src/g_ray.cpp:7-27) with the HIP renderer bound in.  The host IS the
// reference here -- ContextLoader::load parses the scene and builds its objects, the Film is the reference's, writeImage is the
// reference's -- and the integrator is libgoblin_hip.so behind the C ABI.  Built only where /root/reference exists
// (oracle/Makefile `hipbind` -> oracle/_ref/g_ray_hipbind, which travels to the GPU box like ref_harness); run by
// tests/test_gpu_integration.py.
//
//   g_ray_hipbind <scene.json> [--sampler native|stream] [--dump-film <file.f32>] [--dump-desc <file.bin>]
//
// --dump-film: the Film's accumulators after render(), W*H*4 floats {r, g, b, weight} (Film::mPixels)
// --dump-desc: the flattened description's arrays and structs, for the CPU test that compares them with libgoblin_host.so's
//              (no GPU needed: with --dump-desc alone nothing is rendered)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "GoblinContextLoader.h"
#include "GoblinFilm.h"
#include "GoblinRenderContext.h"
#include "GoblinPathtracer.h"
#include "json.hpp"

#include "GoblinHipPathtracer.h"
#include "flatten_scene.h"

using namespace Goblin;

namespace {
// the "render_setting" block as createRenderer reads it (GoblinContextLoader.cpp:67-92, GoblinPathtracer.cpp:210-217)
bool read_setting(const std::string& file, gbl_render_setting* s, ParamSet* ps) {
    std::ifstream in(file);
    if (!in.is_open()) return false;
    nlohmann::json doc = nlohmann::json::parse(in);
    memset(s, 0, sizeof(*s));
    s->integrator = GBL_INTEGRATOR_PATH;
    s->sample_per_pixel = 1;
    s->max_ray_depth = 5;
    s->bssrdf_sample_num = 4;
    s->ao_sample_num = 25;
    s->thread_num = 1;
    auto it = doc.find("render_setting");
    if (it != doc.end()) {
        const nlohmann::json& r = it.value();
        if (r.contains("render_method") && r["render_method"].get<std::string>() != "path_tracing") return false;
        if (r.contains("sample_per_pixel")) s->sample_per_pixel = r["sample_per_pixel"].get<int>();
        if (r.contains("max_ray_depth")) s->max_ray_depth = r["max_ray_depth"].get<int>();
        if (r.contains("bssrdf_sample_num")) s->bssrdf_sample_num = r["bssrdf_sample_num"].get<int>();
    }
    ps->setInt("sample_per_pixel", s->sample_per_pixel);
    ps->setInt("max_ray_depth", s->max_ray_depth);
    ps->setInt("bssrdf_sample_num", s->bssrdf_sample_num);
    return true;
}

template <class T>
void put(FILE* f, const char* tag, const T* p, size_t n) {
    const uint64_t bytes = n * sizeof(T);
    char name[16] = {0};
    strncpy(name, tag, 15);
    fwrite(name, 1, 16, f);
    fwrite(&bytes, 8, 1, f);
    if (bytes) fwrite(p, 1, bytes, f);
}
}   // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "usage: g_ray_hipbind scene.json [--sampler native|stream] [--dump-film f] [--dump-desc f]" << std::endl;
        return 2;
    }
    const std::string file = argv[1];
    std::string dump_film, dump_desc;
    uint32_t sample_mode = GBL_SAMPLES_NATIVE;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--sampler" && i + 1 < argc) sample_mode = std::string(argv[++i]) == "stream" ? GBL_SAMPLES_STREAM : GBL_SAMPLES_NATIVE;
        else if (a == "--dump-film" && i + 1 < argc) dump_film = argv[++i];
        else if (a == "--dump-desc" && i + 1 < argc) dump_desc = argv[++i];
    }
    RenderContext* ctx = ContextLoader::load(file);      // the reference's loader, unmodified: JSON, OBJ, SceneCache, Scene
    if (!ctx) return 1;
    gbl_render_setting setting;
    ParamSet ps;
    if (!read_setting(file, &setting, &ps)) {
        std::cerr << "g_ray_hipbind: render_method must be path_tracing" << std::endl;
        return 1;
    }
    FlatScene flat;
    try {
        flattenSceneForHip(ctx->mScene, setting, &flat);
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    if (!dump_desc.empty()) {
        FILE* f = fopen(dump_desc.c_str(), "wb");
        if (!f) return 1;
        const gbl_scene_desc& d = flat.desc;
        put(f, "positions", d.positions, 3 * size_t(d.num_vertices));
        put(f, "normals", d.normals, 3 * size_t(d.num_vertices));
        put(f, "uvs", d.uvs, 2 * size_t(d.num_vertices));
        put(f, "indices", d.indices, 3 * size_t(d.num_triangles));
        put(f, "meshes", d.meshes, d.num_meshes);
        put(f, "materials", d.materials, d.num_materials);
        put(f, "instances", d.instances, d.num_instances);
        put(f, "lights", d.lights, d.num_lights);
        put(f, "camera", &d.camera, 1);
        put(f, "film", &d.film, 1);
        put(f, "setting", &d.setting, 1);
        fclose(f);
        if (dump_film.empty()) return 0;
    }
    try {
        // RenderContext::mRenderer is public (GoblinRenderContext.h:15): the same swap createRenderer would make for
        // "render_method": "hip_path_tracing"
        HipPathTracer* hip = new HipPathTracer(ps, flat.desc, sample_mode);
        ctx->mRenderer = RendererPtr(hip);
        ctx->render();                                    // preprocess + render: trace on the device, mergeTile, writeImage
        std::cout << "{\"paths\": " << hip->pathsTraced() << "}" << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "g_ray_hipbind: " << e.what() << std::endl;
        return 1;
    }
    if (!dump_film.empty()) {
        Film* film = ctx->mScene->getCamera()->getFilm();
        const int w = film->getXResolution(), h = film->getYResolution();
        FILE* f = fopen(dump_film.c_str(), "wb");
        if (!f) return 1;
        for (int i = 0; i < w * h; ++i) {
            const Pixel& p = film->mPixels[i];
            const float v[4] = {p.color.r, p.color.g, p.color.b, p.weight};
            fwrite(v, sizeof(float), 4, f);
        }
        fclose(f);
    }
    delete ctx;
    return 0;
}
