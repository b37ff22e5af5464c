// The binding a Goblin maintainer adds to the reference tree (INTEGRATION.md 1 quotes this file): a Renderer whose render()
// hands the scene description to libgoblin_hip.so through the C ABI (include/goblin_hip.h) and gives the Film back to the
// reference's own Film::mergeTile / Film::writeImage.  TEST INFRASTRUCTURE in this repository: compiled only against
// /root/reference/src where that exists (oracle/Makefile `hipbind`), never part of the shipped libraries.
//
// Replaces Renderer::render (GoblinRenderer.cpp:99-126); selected the way createRenderer selects BDPT / SPPM / the light
// tracer (GoblinContextLoader.cpp:67-92), which override render() wholesale as this does.
#ifndef GOBLIN_HIP_PATHTRACER_H
#define GOBLIN_HIP_PATHTRACER_H
#include <algorithm>
#include <stdexcept>
#include <vector>

#include <hip/hip_runtime_api.h>
#include "GoblinCamera.h"
#include "GoblinFilm.h"
#include "GoblinParamSet.h"
#include "GoblinRenderer.h"
#include "GoblinScene.h"
#include "goblin_hip.h"          // this repository's include/

namespace Goblin {
class HipPathTracer : public Renderer {
public:
    // desc: the flattened scene (flattenSceneForHip, tests/integration/flatten_scene.h); sample_mode: GBL_SAMPLES_NATIVE (fast)
    // or GBL_SAMPLES_STREAM (this class's own mt19937 sample stream -> the Film g_ray writes for the same file)
    HipPathTracer(const ParamSet& setting, const gbl_scene_desc& desc, uint32_t sample_mode = GBL_SAMPLES_NATIVE)
        : Renderer(setting.getInt("sample_per_pixel", 1), 1),
          mDepth(std::max(1, setting.getInt("max_ray_depth", 5))),            // GoblinPathtracer.cpp:210-217
          mBssrdf(setting.getInt("bssrdf_sample_num", 4)), mSampleMode(sample_mode) {
        if (gbl_create(&desc, /*device*/ 0, &mCtx) != GBL_OK)
            throw std::runtime_error(gbl_last_error(nullptr));                  // no CPU fallback
    }
    ~HipPathTracer() { gbl_destroy(mCtx); }

    void render(const ScenePtr& scene) override {
        Film* film = scene->getCamera()->getFilm();
        const int w = film->getXResolution(), h = film->getYResolution();
        float* accum = nullptr;                                                 // W*H float4 {sum w*L, sum w}
        if (hipMalloc(reinterpret_cast<void**>(&accum), sizeof(float) * 4 * w * h) != hipSuccess)
            throw std::runtime_error("hipMalloc(film accumulators) failed");
        hipMemset(accum, 0, sizeof(float) * 4 * w * h);
        gbl_render_params p = {};
        p.integrator = GBL_INTEGRATOR_PATH;
        p.sample_per_pixel = mSamplePerPixel;
        p.max_ray_depth = mDepth;
        p.bssrdf_sample_num = mBssrdf;
        p.sample_mode = mSampleMode;                                            // window {0,0,0,0} = Film::getSampleRange
        gbl_stats stats;
        if (gbl_render(mCtx, &p, accum, &stats) != GBL_OK) {
            hipFree(accum);
            throw std::runtime_error(gbl_last_error(mCtx));
        }
        // hand the accumulators back as ONE ImageTile, the way a worker thread's tile is merged
        // (GoblinThreadLocalStorage.h:69-75, GoblinFilm.cpp:140-153)
        std::vector<float> host(4 * size_t(w) * h);
        hipMemcpy(host.data(), accum, host.size() * sizeof(float), hipMemcpyDeviceToHost);
        hipFree(accum);
        ImageRect rect; film->getImageRect(rect);
        ImageTile tile(rect, film->getFilterTable());
        Pixel* px = const_cast<Pixel*>(tile.getTileBuffer());
        for (int y = rect.yStart; y < rect.yStart + rect.yCount; ++y)
            for (int x = rect.xStart; x < rect.xStart + rect.xCount; ++x) {
                const float* a = &host[4 * (size_t(y) * w + x)];
                Pixel& q = px[rect.pixelToOffset(x, y)];
                q.color = Color(a[0], a[1], a[2]);
                q.weight = a[3];
            }
        film->mergeTile(tile);
        film->writeImage();                                                     // unchanged: normalise, EXR
        mPaths = stats.paths;
    }
    Color Li(const ScenePtr&, const RayDifferential&, const Sample&, const RNG&, RenderingTLS*) const override {
        return Color::Black;                                                    // never called: render() is overridden
    }
    unsigned long long pathsTraced() const { return mPaths; }
private:
    void querySampleQuota(const ScenePtr&, SampleQuota*) override {}
    gbl_ctx* mCtx = nullptr;
    int mDepth, mBssrdf;
    uint32_t mSampleMode;
    unsigned long long mPaths = 0;
};
}
#endif
