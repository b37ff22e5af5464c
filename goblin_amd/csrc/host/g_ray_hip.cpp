// g_ray_hip <scene.json> [--device N] [--seed S] [--sampler native|stream] [--out file.{exr,ppm,pfm}]
//
// --sampler stream renders with the reference's own sample stream (GBL_SAMPLES_STREAM): the image is the one the
// reference binary writes for this scene file, up to float summation order; native (default) is the fast sampler.
//
// Stand-alone host with the call shape of the reference's g_ray
// (/root/reference/src/g_ray.cpp:7-27): load the scene, render it, and run
// Film::writeImage's tail (GoblinFilm.cpp:164-198): normalise, bloom, write the
// film's "file" (default <scene>.exr, HALF B/G/R).  Everything goes through the C
// ABI of include/goblin_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/goblin_hip.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "Usage: g_ray_hip scene_file.json [--device N] [--seed S] [--sampler native|stream] [--out image.{exr,ppm,pfm}]\n");
        return 0;
    }
    std::string scene_path = argv[1], out_path;
    int device = 0;
    unsigned long long seed = 0;
    bool stream_sampler = false;
    for (int i = 2; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--device")) device = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--seed")) seed = strtoull(argv[i + 1], nullptr, 10);
        else if (!strcmp(argv[i], "--out")) out_path = argv[i + 1];
        else if (!strcmp(argv[i], "--sampler")) stream_sampler = !strcmp(argv[i + 1], "stream");
    }
    gbl_host_scene* hs = nullptr;
    if (gbl_host_load_file(scene_path.c_str(), &hs) != GBL_OK) {
        fprintf(stderr, "load failed: %s\n", gbl_host_last_error());
        return 1;
    }
    const gbl_scene_desc* desc = gbl_host_desc(hs);
    if (out_path.empty()) out_path = gbl_host_output_path(hs);
    gbl_ctx* ctx = nullptr;
    if (gbl_create(desc, device, &ctx) != GBL_OK) {
        fprintf(stderr, "gbl_create failed: %s\n", gbl_last_error(nullptr));
        return 1;
    }
    gbl_info info;
    gbl_get_info(ctx, &info);
    size_t npix = static_cast<size_t>(info.xres) * info.yres;
    float *accum = nullptr, *rgb = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&accum), npix * 4 * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&rgb), npix * 3 * sizeof(float)) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    (void)hipMemset(accum, 0, npix * 4 * sizeof(float));
    gbl_render_params p;
    memset(&p, 0, sizeof(p));
    p.integrator = desc->setting.integrator;
    p.sample_per_pixel = desc->setting.sample_per_pixel;
    p.max_ray_depth = desc->setting.max_ray_depth;
    p.ao_sample_num = desc->setting.ao_sample_num;
    p.bssrdf_sample_num = desc->setting.bssrdf_sample_num;
    p.sample_mode = GBL_SAMPLES_NATIVE;
    p.seed = seed;
    gbl_stats st;
    auto t0 = std::chrono::steady_clock::now();
    if (stream_sampler) {
        // bands of whole tile rows, sized so a band's per-sample buffers stay near 1 GiB
        p.sample_mode = GBL_SAMPLES_STREAM;
        const int spp = gbl_host_round_to_square(p.sample_per_pixel);
        const long long row_samples = static_cast<long long>(info.window[1] - info.window[0]) * spp;
        int band = static_cast<int>(std::max<long long>(8, ((1ll << 26) / std::max<long long>(1, row_samples)) / 8 * 8));
        gbl_stats total;
        memset(&total, 0, sizeof(total));
        for (int y = info.window[2]; y < info.window[3]; y += band) {
            p.window[0] = info.window[0];
            p.window[1] = info.window[1];
            p.window[2] = y;
            p.window[3] = std::min(y + band, info.window[3]);
            if (gbl_render(ctx, &p, accum, &st) != GBL_OK) {
                fprintf(stderr, "gbl_render failed: %s\n", gbl_last_error(ctx));
                return 1;
            }
            total.paths += st.paths;
            total.kernel_ms += st.kernel_ms;
        }
        st = total;
    } else if (gbl_render(ctx, &p, accum, &st) != GBL_OK) {
        fprintf(stderr, "gbl_render failed: %s\n", gbl_last_error(ctx));
        return 1;
    }
    gbl_film_resolve(ctx, accum, rgb, nullptr);
    (void)hipDeviceSynchronize();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<float> host(npix * 3);
    (void)hipMemcpy(host.data(), rgb, host.size() * sizeof(float), hipMemcpyDeviceToHost);
    if (desc->film.bloom_radius > 0.0f && desc->film.bloom_weight > 0.0f)
        gbl_host_bloom(host.data(), info.xres, info.yres, desc->film.bloom_radius, desc->film.bloom_weight);
    if (gbl_host_write_image(out_path.c_str(), host.data(), info.xres, info.yres, static_cast<int32_t>(desc->film.tone_mapping)) != GBL_OK) {
        fprintf(stderr, "write failed: %s\n", gbl_host_last_error());
        return 1;
    }
    printf("Render Complete!\n%llu paths in %.3f s (kernel %.3f ms, %.1f Mpaths/s)\nwrite image to : %s\n",
           static_cast<unsigned long long>(st.paths), sec, st.kernel_ms, st.paths / (st.kernel_ms * 1e3), out_path.c_str());
    (void)hipFree(accum);
    (void)hipFree(rgb);
    gbl_destroy(ctx);
    gbl_host_free(hs);
    return 0;
}
