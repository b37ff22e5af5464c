// TEST INFRASTRUCTURE -- builds only where /root/reference exists (this
// container); never shipped, never on the product path.
//
// Capture harness around the REAL reference.  It is an out-of-tree translation
// unit that #includes the reference's own headers from /root/reference/src and
// links the reference's own objects (see oracle/Makefile: the reference sources
// are compiled where they lie, nothing is copied).  It is compiled with
// -fno-access-control so it can reach Renderer::querySampleQuota,
// Renderer::getSampleRanges and Film::mPixels, which lets it run the render
// loop of Renderer::render (GoblinRenderer.cpp:99-126) WITHOUT the final
// Film::writeImage -- the EXR writer (GoblinImageIO.cpp, off the hot path) is
// the one reference file this build leaves out.
//
// Modes
//   film  <scene.json> <prefix> [threads]    render, dump Film accumulators
//   li    <scene.json> <prefix> <stride> <max_records>
//                                            thread_num=1 render; every
//                                            stride-th Li call logs
//                                            {Sample floats, Li rgba}
//   kat   <scene.json> <prefix>              known-answer tables (camera rays,
//                                            filter table, transforms, lights)
//   time  <scene.json> [threads]             render only, print seconds
// Outputs: <prefix>.film.f32 (W*H*4: r,g,b,weight), <prefix>.samples.f32,
//          <prefix>.li.f32, <prefix>.kat.txt, and a one-line JSON on stdout.
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "GoblinAO.h"
#include "GoblinCamera.h"
#include "GoblinContextLoader.h"
#include "GoblinFilm.h"
#include "GoblinFilter.h"
#include "GoblinLight.h"
#include "GoblinPathtracer.h"
#include "GoblinRenderContext.h"
#include "GoblinImageIO.h"
#include "GoblinRenderer.h"
#include "GoblinSampler.h"
#include "GoblinScene.h"
#include "GoblinTexture.h"
#include "GoblinThreadLocalStorage.h"
#include "GoblinThreadPool.h"
#include "GoblinWhitted.h"

using namespace Goblin;

namespace {

struct Recorder {
    size_t stride = 1, max_records = 0, calls = 0;
    std::vector<float> samples, li;
    size_t dims = 0;
    void record(const Sample& s, const Color& c) {
        size_t call = calls++;
        if (max_records == 0 || call % stride != 0 || li.size() / 4 >= max_records) return;
        size_t quota = 0;
        for (uint32_t n : s.n1D) quota += n;
        for (uint32_t n : s.n2D) quota += 2 * n;
        dims = 4 + quota;
        samples.push_back(s.imageX);
        samples.push_back(s.imageY);
        samples.push_back(s.lensU1);
        samples.push_back(s.lensU2);
        // u1D[0] is the start of one contiguous quota buffer (Sample::allocateQuota)
        const float* q = (s.u1D != nullptr) ? s.u1D[0] : nullptr;
        for (size_t i = 0; i < quota; ++i) samples.push_back(q[i]);
        li.push_back(c.r);
        li.push_back(c.g);
        li.push_back(c.b);
        li.push_back(c.a);
    }
};

Recorder g_rec;

struct ProbePT : public PathTracer {
    ProbePT(int spp, int threads, int depth, int bssrdf) : PathTracer(spp, threads, depth, bssrdf) {}
    Color Li(const ScenePtr& scene, const RayDifferential& ray, const Sample& sample, const RNG& rng,
             RenderingTLS* tls) const {
        Color c = PathTracer::Li(scene, ray, sample, rng, tls);
        g_rec.record(sample, c);
        return c;
    }
};

struct ProbeAO : public AORenderer {
    ProbeAO(int spp, int threads, int n) : AORenderer(spp, threads, n) {}
    Color Li(const ScenePtr& scene, const RayDifferential& ray, const Sample& sample, const RNG& rng,
             RenderingTLS* tls) const {
        Color c = AORenderer::Li(scene, ray, sample, rng, tls);
        g_rec.record(sample, c);
        return c;
    }
};

// The Whitted renderer calls Li recursively (specularReflect / specularRefract, GoblinRenderer.cpp:598-648): only the
// camera ray's call (depth 0) is a record.
struct ProbeWhitted : public WhittedRenderer {
    ProbeWhitted(int spp, int threads, int depth, int bssrdf) : WhittedRenderer(spp, threads, depth, bssrdf) {}
    Color Li(const ScenePtr& scene, const RayDifferential& ray, const Sample& sample, const RNG& rng,
             RenderingTLS* tls) const {
        Color c = WhittedRenderer::Li(scene, ray, sample, rng, tls);
        if (ray.depth == 0) g_rec.record(sample, c);
        return c;
    }
};

// Renderer::render (GoblinRenderer.cpp:99-126) minus drawDebugData/writeImage.
double run_render(Renderer* renderer, const ScenePtr& scene, int threads) {
    const CameraPtr camera = scene->getCamera();
    Film* film = camera->getFilm();
    SampleQuota quota;
    renderer->querySampleQuota(scene, &quota);
    std::vector<SampleRange> ranges;
    renderer->getSampleRanges(film, ranges);
    std::vector<Task*> tasks;
    RenderProgress progress(static_cast<int>(ranges.size()) + 1);  // +1: never prints "Render Complete"
    for (size_t i = 0; i < ranges.size(); ++i) {
        tasks.push_back(new RenderTask(renderer, camera, scene, ranges[i], quota, renderer->mSamplePerPixel, &progress));
    }
    auto t0 = std::chrono::steady_clock::now();
    {
        RenderingTLSManager tls(film);
        ThreadPool pool(threads, &tls);
        pool.enqueue(tasks);
        pool.waitForAll();
    }
    auto t1 = std::chrono::steady_clock::now();
    for (Task* t : tasks) delete t;
    return std::chrono::duration<double>(t1 - t0).count();
}

void write_f32(const std::string& path, const std::vector<float>& v) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) {
        fprintf(stderr, "can't write %s\n", path.c_str());
        exit(2);
    }
    fwrite(v.data(), sizeof(float), v.size(), f);
    fclose(f);
}

std::vector<float> film_buffer(Film* film) {
    int w = film->getXResolution(), h = film->getYResolution();
    std::vector<float> out(static_cast<size_t>(w) * h * 4);
    for (int i = 0; i < w * h; ++i) {
        const Pixel& p = film->mPixels[i];
        out[4 * i + 0] = p.color.r;
        out[4 * i + 1] = p.color.g;
        out[4 * i + 2] = p.color.b;
        out[4 * i + 3] = p.weight;
    }
    return out;
}

void kat(RenderContext* ctx, const std::string& prefix) {
    FILE* f = fopen((prefix + ".kat.txt").c_str(), "w");
    ScenePtr scene = ctx->mScene;
    CameraPtr cam = scene->getCamera();
    Film* film = cam->getFilm();
    // filter table (GoblinFilm.cpp:10-27)
    const FilterTable& ft = film->getFilterTable();
    fprintf(f, "filter_table %d\n", FILTER_TABLE_WIDTH * FILTER_TABLE_WIDTH);
    for (int i = 0; i < FILTER_TABLE_WIDTH * FILTER_TABLE_WIDTH; ++i) fprintf(f, "%.9g\n", ft.mTable[i]);
    SampleRange sr;
    film->getSampleRange(sr);
    fprintf(f, "sample_range %d %d %d %d\n", sr.xStart, sr.xEnd, sr.yStart, sr.yEnd);
    // camera rays on a lattice of image positions (GoblinCamera.cpp:97-148)
    fprintf(f, "camera_rays 25\n");
    for (int j = 0; j < 5; ++j) {
        for (int i = 0; i < 5; ++i) {
            Sample s;
            s.imageX = sr.xStart + (sr.xEnd - sr.xStart) * (i + 0.37f) / 5.0f;
            s.imageY = sr.yStart + (sr.yEnd - sr.yStart) * (j + 0.61f) / 5.0f;
            s.lensU1 = 0.25f;
            s.lensU2 = 0.75f;
            RayDifferential r;
            float w = cam->generateRay(s, &r);
            fprintf(f, "%.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", s.imageX, s.imageY, r.o.x, r.o.y,
                    r.o.z, r.d.x, r.d.y, r.d.z, r.mint, r.maxt > 1e30f ? -1.0f : r.maxt, w);
        }
    }
    // light power distribution (GoblinScene.cpp:21-27)
    const std::vector<Light*>& lights = scene->getLights();
    fprintf(f, "light_power %zu\n", lights.size());
    for (Light* l : lights) {
        Color p = l->power(*scene);
        fprintf(f, "%.9g %.9g %.9g %.9g\n", p.r, p.g, p.b, p.luminance());
    }
    // closest-hit probes through the full two-level BVH for the lattice rays
    fprintf(f, "hits 25\n");
    for (int j = 0; j < 5; ++j) {
        for (int i = 0; i < 5; ++i) {
            Sample s;
            s.imageX = sr.xStart + (sr.xEnd - sr.xStart) * (i + 0.37f) / 5.0f;
            s.imageY = sr.yStart + (sr.yEnd - sr.yStart) * (j + 0.61f) / 5.0f;
            s.lensU1 = s.lensU2 = 0.5f;
            RayDifferential r;
            cam->generateRay(s, &r);
            float eps = 0.0f;
            Intersection isect;
            bool hit = scene->intersect(r, &eps, &isect);
            if (hit) {
                const Fragment& fr = isect.fragment;
                Matrix3 w2s = fr.getWorldToShade();
                fprintf(f, "1 %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", r.maxt, eps,
                        fr.getPosition().x, fr.getPosition().y, fr.getPosition().z, fr.getNormal().x, fr.getNormal().y,
                        fr.getNormal().z, w2s[0][0], w2s[0][1], w2s[0][2]);
            } else {
                fprintf(f, "0 0 0 0 0 0 0 0 0 0 0 0\n");
            }
        }
    }
    fclose(f);
}

}  // namespace

// image <in.f32> <w> <h> <prefix> <bloom radius> <bloom weight>: the reference's image output path on a given float4 image
// (GoblinImageIO.cpp): bloom, toneMapping, writeImage (.exr: three HALF channels through tinyexr; .ppm with tone
// mapping) and loadImage of the .exr it wrote.  imageload <file.exr> <out.f32>: loadImage alone.
static int image_mode(int argc, char** argv) {
    std::string mode = argv[1];
    if (mode == "imageload") {
        int w = 0, h = 0;
        Color* c = loadImage(argv[2], &w, &h);
        if (!c) return 1;
        std::vector<float> v(reinterpret_cast<float*>(c), reinterpret_cast<float*>(c) + 4 * static_cast<size_t>(w) * h);
        write_f32(argv[3], v);
        printf("{\"mode\": \"imageload\", \"width\": %d, \"height\": %d}\n", w, h);
        return 0;
    }
    if (mode == "mipmap") {   // mipmap <file.exr> <out.f32>: MIPMap<Color>'s levels, level 0 first, each row-major float4
        int w = 0, h = 0;
        Color* c = loadImage(argv[2], &w, &h);
        if (!c) return 1;
        Color* copy = new Color[static_cast<size_t>(w) * h];
        memcpy(copy, c, sizeof(Color) * static_cast<size_t>(w) * h);
        MIPMap<Color> mip(copy, w, h);
        std::vector<float> all;
        for (int l = 0; l < mip.getLevelsNum(); ++l) {
            const ImageBuffer<Color>* b = mip.getImageBuffer(l);
            const float* p = reinterpret_cast<const float*>(b->image);
            all.insert(all.end(), p, p + 4 * static_cast<size_t>(b->width) * b->height);
        }
        write_f32(argv[3], all);
        printf("{\"mode\": \"mipmap\", \"width\": %d, \"height\": %d, \"levels\": %d}\n", mip.getWidth(), mip.getHeight(), mip.getLevelsNum());
        return 0;
    }
    if (mode == "miplookup") {   // miplookup <file.exr> <queries.f32> <out.f32> <filter 0-3> <address 0-2> <max anisotropy>: MIPMap<Color>::lookup
        if (argc < 8) return 2;      // of n queries {s, t, dsdx, dtdx, dsdy, dtdy} -> n float4
        int w = 0, h = 0;
        Color* c = loadImage(argv[2], &w, &h);
        if (!c) return 1;
        Color* copy = new Color[static_cast<size_t>(w) * h];   // (MIPMap takes ownership with delete[]; loadImage's buffer is malloc'ed)
        memcpy(copy, c, sizeof(Color) * static_cast<size_t>(w) * h);
        MIPMap<Color> mip(copy, w, h, static_cast<float>(atof(argv[7])));
        FILE* f = fopen(argv[3], "rb");
        if (!f) return 1;
        fseek(f, 0, SEEK_END);
        const size_t n = static_cast<size_t>(ftell(f)) / (6 * sizeof(float));
        fseek(f, 0, SEEK_SET);
        std::vector<float> q(6 * n), out(4 * n);
        if (fread(q.data(), sizeof(float), q.size(), f) != q.size()) return 1;
        fclose(f);
        const FilterType filters[] = {FilterNone, FilterBilinear, FilterTrilinear, FilterEWA};
        const AddressMode modes[] = {AddressRepeat, AddressClamp, AddressBorder};
        for (size_t i = 0; i < n; ++i) {
            TextureCoordinate tc;
            tc.st = Vector2(q[6 * i], q[6 * i + 1]);
            tc.dsdx = q[6 * i + 2], tc.dtdx = q[6 * i + 3], tc.dsdy = q[6 * i + 4], tc.dtdy = q[6 * i + 5];
            const Color r = mip.lookup(tc, filters[atoi(argv[5])], modes[atoi(argv[6])]);
            out[4 * i] = r.r, out[4 * i + 1] = r.g, out[4 * i + 2] = r.b, out[4 * i + 3] = r.a;
        }
        write_f32(argv[4], out);
        printf("{\"mode\": \"miplookup\", \"n\": %zu}\n", n);
        return 0;
    }
    if (argc < 8) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]);
    const std::string prefix = argv[5];
    const float radius = static_cast<float>(atof(argv[6])), weight = static_cast<float>(atof(argv[7]));
    std::vector<Color> img(static_cast<size_t>(w) * h);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(img.data(), sizeof(Color), img.size(), f) != img.size()) return 1;
    fclose(f);
    auto dump = [&](const std::vector<Color>& c, const std::string& path) {
        std::vector<float> v(reinterpret_cast<const float*>(c.data()), reinterpret_cast<const float*>(c.data()) + 4 * c.size());
        write_f32(path, v);
    };
    std::vector<Color> a = img;
    bloom(a.data(), w, h, radius, weight);
    dump(a, prefix + ".bloom.f32");
    std::vector<Color> b = img;
    toneMapping(b.data(), w, h);
    dump(b, prefix + ".tone.f32");
    std::vector<Color> c = img;
    if (!writeImage(prefix + ".exr", c.data(), w, h, false)) return 1;
    std::vector<Color> d = img;
    if (!writeImage(prefix + ".ppm", d.data(), w, h, true)) return 1;
    int lw = 0, lh = 0;
    Color* back = loadImage(prefix + ".exr", &lw, &lh);
    if (!back || lw != w || lh != h) return 1;
    std::vector<Color> e(back, back + static_cast<size_t>(w) * h);
    dump(e, prefix + ".load.f32");
    printf("{\"mode\": \"image\", \"width\": %d, \"height\": %d}\n", w, h);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s film|li|kat|time <scene.json> ... | image <in.f32> <w> <h> <prefix> <radius> <weight> | imageload <file.exr> <out.f32>\n", argv[0]);
        return 2;
    }
    if (!strcmp(argv[1], "image") || !strcmp(argv[1], "imageload") || !strcmp(argv[1], "mipmap") || !strcmp(argv[1], "miplookup")) return image_mode(argc, argv);
    std::string mode = argv[1], scene_path = argv[2];
    // the loader echoes every parsed parameter to stdout; keep stdout for our JSON line
    FILE* real_stdout = fdopen(dup(fileno(stdout)), "w");
    if (!freopen("/dev/null", "w", stdout)) return 2;
    ContextLoader loader;
    RenderContext* ctx = loader.load(scene_path);
    if (!ctx) {
        fprintf(stderr, "load failed: %s\n", scene_path.c_str());
        return 1;
    }
    ScenePtr scene = ctx->mScene;
    Film* film = scene->getCamera()->getFilm();
    Renderer* base = ctx->mRenderer.get();
    int spp = base->mSamplePerPixel;
    bool is_ao = dynamic_cast<AORenderer*>(base) != nullptr;
    bool is_pt = dynamic_cast<PathTracer*>(base) != nullptr;
    bool is_wh = dynamic_cast<WhittedRenderer*>(base) != nullptr;
    if (!is_ao && !is_pt && !is_wh) {
        fprintf(stderr, "scene selects a renderer outside the hot path\n");
        return 1;
    }

    if (mode == "kat") {
        kat(ctx, argv[3]);
        fprintf(real_stdout, "{\"mode\": \"kat\"}\n");
        return 0;
    }

    int threads = 1;
    std::string prefix;
    if (mode == "film") {
        prefix = argv[3];
        if (argc > 4) threads = atoi(argv[4]);
    } else if (mode == "li") {
        prefix = argv[3];
        g_rec.stride = static_cast<size_t>(atoll(argv[4]));
        g_rec.max_records = static_cast<size_t>(atoll(argv[5]));
    } else if (mode == "time") {
        if (argc > 3) threads = atoi(argv[3]);
    } else {
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }

    Renderer* renderer = base;
    Renderer* probe = nullptr;
    if (mode == "li") {
        if (is_pt) {
            PathTracer* pt = static_cast<PathTracer*>(base);
            probe = new ProbePT(spp, 1, pt->mMaxRayDepth, pt->mBssrdfSampleNum);
        } else if (is_wh) {
            WhittedRenderer* wh = static_cast<WhittedRenderer*>(base);
            probe = new ProbeWhitted(spp, 1, wh->mMaxRayDepth, wh->mBssrdfSampleNum);
        } else {
            AORenderer* ao = static_cast<AORenderer*>(base);
            probe = new ProbeAO(spp, 1, ao->mAOSampleNum);
        }
        renderer = probe;
    }
    double seconds = run_render(renderer, scene, threads);

    SampleRange sr;
    film->getSampleRange(sr);
    int sq = roundToSquare(spp);
    unsigned long long paths = 1ull * (sr.xEnd - sr.xStart) * (sr.yEnd - sr.yStart) * sq;
    if (mode == "film" || mode == "li") write_f32(prefix + ".film.f32", film_buffer(film));
    if (mode == "li") {
        write_f32(prefix + ".samples.f32", g_rec.samples);
        write_f32(prefix + ".li.f32", g_rec.li);
    }
    fprintf(real_stdout,
            "{\"mode\": \"%s\", \"xres\": %d, \"yres\": %d, \"window\": [%d, %d, %d, %d], \"spp\": %d, "
            "\"paths\": %llu, \"threads\": %d, \"seconds\": %.6f, \"mpaths_per_s\": %.6f, \"records\": %zu, "
            "\"dims\": %zu, \"li_calls\": %zu}\n",
            mode.c_str(), film->getXResolution(), film->getYResolution(), sr.xStart, sr.xEnd, sr.yStart, sr.yEnd, sq,
            paths, threads, seconds, paths / seconds * 1e-6, g_rec.li.size() / 4, g_rec.dims, g_rec.calls);
    fflush(real_stdout);
    // Skip destructors: ~Scene would run ImageTexture cache cleanup from the TU we did not build.
    _exit(0);
}
