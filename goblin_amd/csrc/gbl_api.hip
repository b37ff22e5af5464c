// libgoblin_hip.so -- C ABI of the device integrator (include/goblin_hip.h).
//
// gbl_create   packs the scene on the host (scene_prep.cpp) and uploads it once.
// gbl_render   launches the persistent render kernel over a sample sub-window and
//              accumulates into the caller's device film.
// There is no CPU fallback anywhere in this library: every entry point that
// computes something needs a HIP device and fails with GBL_ERR_DEVICE otherwise.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/goblin_hip.h"
#include "device_scene.h"
#include "kernels/render_kernels.h"
#include "scene_prep.h"

namespace {
thread_local std::string g_create_error;
}

struct gbl_ctx {
    int device = 0;
    std::string error;
    std::vector<void*> allocations;
    DevScene scene;
    gbl_info info;
    uint32_t* work_counter = nullptr;
    unsigned long long* stats = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cus = 256;
    void* rccl = nullptr;
    void* rccl_allreduce = nullptr;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return GBL_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)

template <class T>
gbl_status upload(gbl_ctx* ctx, const std::vector<T>& v, const T** out) {
    size_t bytes = std::max<size_t>(1, v.size()) * sizeof(T);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        ctx->error = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? GBL_ERR_OOM : GBL_ERR_DEVICE;
    }
    ctx->allocations.push_back(p);
    ctx->info.scene_bytes += bytes;
    if (!v.empty()) HIP_TRY(ctx, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return GBL_OK;
}

int round_to_square(int n, int* root) {
    int s = static_cast<int>(std::ceil(std::sqrt(static_cast<float>(n))));
    *root = s;
    return s * s;
}

uint32_t host_mix(uint32_t a, uint32_t b) {   // same integer hash as kernels/sampler.h nat_mix
    uint32_t h = (a ^ 0x9E3779B9u) * 0x85EBCA6Bu;
    h ^= b + 0x7F4A7C15u + (h << 6) + (h >> 2);
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

}  // namespace

extern "C" {

int gbl_abi_version(void) { return GBL_ABI_VERSION; }

const char* gbl_last_error(const gbl_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

gbl_status gbl_create(const gbl_scene_desc* desc, int device, gbl_ctx** out) {
    if (!desc || !out) {
        g_create_error = "null argument";
        return GBL_ERR_INVALID;
    }
    *out = nullptr;
    PackedScene packed;
    std::string err;
    gbl_status st = pack_scene(desc, &packed, &err);
    if (st != GBL_OK) {
        g_create_error = err;
        return st;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device < 0 || device >= count) {
        g_create_error = "no HIP device " + std::to_string(device) + " (" +
                         (e != hipSuccess ? hipGetErrorString(e) : "device count " + std::to_string(count)) +
                         "); the device integrator has no CPU fallback";
        return GBL_ERR_DEVICE;
    }
    gbl_ctx* ctx = new gbl_ctx();
    ctx->device = device;
    memset(&ctx->scene, 0, sizeof(ctx->scene));
    memset(&ctx->info, 0, sizeof(ctx->info));
    auto bail = [&](gbl_status s) {
        g_create_error = ctx->error;
        gbl_destroy(ctx);
        return s;
    };
    if (hipSetDevice(device) != hipSuccess) {
        ctx->error = "hipSetDevice failed";
        return bail(GBL_ERR_DEVICE);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    DevScene& sc = ctx->scene;
    std::vector<float> ftab(packed.filter_table, packed.filter_table + 256);
    if ((st = upload(ctx, packed.nodes, &sc.nodes)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.tris, &sc.tris)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.tri_shade, &sc.tri_shade)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.normals, &sc.normals)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.uvs, &sc.uvs)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.instances, &sc.instances)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.materials, &sc.materials)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.lights, &sc.lights)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.light_tris, &sc.light_tris)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.light_cdf, &sc.light_cdf)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.light_pick_pdf, &sc.light_pick_pdf)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, ftab, &sc.filter_table)) != GBL_OK) return bail(st);
    sc.tlas_root = packed.tlas_root;
    sc.num_instances = static_cast<int32_t>(packed.instances.size());
    sc.num_lights = static_cast<int32_t>(packed.lights.size());
    sc.stack_entries = packed.stack_entries;
    sc.camera = packed.camera;
    sc.film = packed.film;
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(uint32_t)) != hipSuccess) {
        ctx->error = "hipMalloc(work counter) failed";
        return bail(GBL_ERR_OOM);
    }
    ctx->allocations.push_back(p);
    ctx->work_counter = static_cast<uint32_t*>(p);
    if (hipMalloc(&p, 8 * sizeof(unsigned long long)) != hipSuccess) {
        ctx->error = "hipMalloc(stats) failed";
        return bail(GBL_ERR_OOM);
    }
    ctx->allocations.push_back(p);
    ctx->stats = static_cast<unsigned long long*>(p);
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        ctx->error = "hipEventCreate failed";
        return bail(GBL_ERR_DEVICE);
    }
    ctx->info.xres = packed.film.xres;
    ctx->info.yres = packed.film.yres;
    memcpy(ctx->info.window, packed.film.window, sizeof(ctx->info.window));
    ctx->info.blas_nodes = packed.blas_nodes;
    ctx->info.tlas_nodes = packed.tlas_nodes;
    ctx->info.triangles = packed.tris.size();
    ctx->info.instances = packed.instances.size();
    *out = ctx;
    return GBL_OK;
}

void gbl_destroy(gbl_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (void* p : ctx->allocations) (void)hipFree(p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->rccl) dlclose(ctx->rccl);
    delete ctx;
}

gbl_status gbl_get_info(const gbl_ctx* ctx, gbl_info* out) {
    if (!ctx || !out) return GBL_ERR_INVALID;
    *out = ctx->info;
    return GBL_OK;
}

gbl_status gbl_render(gbl_ctx* ctx, const gbl_render_params* p, float* film_accum, gbl_stats* stats) {
    if (!ctx) return GBL_ERR_INVALID;
    if (!p || !film_accum) {
        ctx->error = "null argument";
        return GBL_ERR_INVALID;
    }
    const DevScene& sc = ctx->scene;
    RenderArgs ra;
    memset(&ra, 0, sizeof(ra));
    if (p->integrator != GBL_INTEGRATOR_PATH && p->integrator != GBL_INTEGRATOR_AO) {
        ctx->error = "unknown integrator";
        return GBL_ERR_INVALID;
    }
    if (p->sample_per_pixel < 1 || p->max_ray_depth < 1) {
        ctx->error = "sample_per_pixel and max_ray_depth must be >= 1";
        return GBL_ERR_INVALID;
    }
    ra.integrator = static_cast<int32_t>(p->integrator);
    ra.spp = round_to_square(p->sample_per_pixel, &ra.root);
    ra.max_depth = p->max_ray_depth;
    int tmp;
    int ao_n = round_to_square(std::max(1, p->ao_sample_num), &tmp);
    ra.ao_n = ao_n;
    if (p->integrator == GBL_INTEGRATOR_AO) {
        int r2;
        ra.dims = 4 + 2 * round_to_square(ao_n, &r2);
        ra.off2_base = 4;
    } else {
        int r1, r2;
        int n1 = round_to_square(std::max(1, p->bssrdf_sample_num), &r1);
        int n2 = round_to_square(n1, &r2);
        ra.dims = 4 + 7 * ra.max_depth + 4 * n1 + 4 * n2;
        ra.off2_base = 4 + 3 * ra.max_depth + 4 * n1;
    }
    const int32_t* full = sc.film.window;
    bool whole = p->window[0] == 0 && p->window[1] == 0 && p->window[2] == 0 && p->window[3] == 0;
    for (int i = 0; i < 4; ++i) ra.window[i] = whole ? full[i] : p->window[i];
    if (ra.window[0] < full[0] || ra.window[1] > full[1] || ra.window[2] < full[2] || ra.window[3] > full[3] ||
        ra.window[0] > ra.window[1] || ra.window[2] > ra.window[3]) {
        ctx->error = "render window lies outside the film's sample window";
        return GBL_ERR_INVALID;
    }
    if (p->sample_mode == GBL_SAMPLES_REPLAY && !p->replay_samples) {
        ctx->error = "replay mode needs replay_samples";
        return GBL_ERR_INVALID;
    }
    if (p->sample_mode != GBL_SAMPLES_REPLAY && p->sample_mode != GBL_SAMPLES_NATIVE) {
        ctx->error = "unknown sample_mode";
        return GBL_ERR_INVALID;
    }
    uint64_t npix = static_cast<uint64_t>(ra.window[1] - ra.window[0]) * (ra.window[3] - ra.window[2]);
    if (npix * ra.spp >= (1ull << 32)) {
        ctx->error = "more than 2^32 paths in one call: split the window";
        return GBL_ERR_INVALID;
    }
    ra.tiles_x = (ra.window[1] - ra.window[0] + GBL_TILE - 1) / GBL_TILE;
    ra.tiles_y = (ra.window[3] - ra.window[2] + GBL_TILE - 1) / GBL_TILE;
    ra.shard_count = std::max(1, p->tile_shard_count);
    ra.shard_index = p->tile_shard_count > 1 ? p->tile_shard_index : 0;
    if (ra.shard_index < 0 || ra.shard_index >= ra.shard_count) {
        ctx->error = "tile_shard_index out of range";
        return GBL_ERR_INVALID;
    }
    int total_tiles = ra.tiles_x * ra.tiles_y;
    ra.local_tiles = total_tiles > ra.shard_index ? (total_tiles - ra.shard_index + ra.shard_count - 1) / ra.shard_count : 0;
    // Work granularity: a work item is one tile x one chunk of its samples.  Start
    // at <= 64 samples per item (4096 paths) and keep halving while the launch
    // would have fewer than ~16 items per resident workgroup (tail effect),
    // down to 4 samples (256 paths) per item.
    int chunks = 1;
    while (ra.spp / chunks > 64 && ra.spp % (chunks * 2) == 0) chunks *= 2;
    const uint64_t want_items = 16ull * ctx->num_cus * 4;
    while (static_cast<uint64_t>(ra.local_tiles) * chunks < want_items && ra.spp / chunks > 4 && ra.spp % (chunks * 2) == 0)
        chunks *= 2;
    ra.chunks = chunks;
    ra.chunk_spp = ra.spp / chunks;
    ra.seed_key = host_mix(static_cast<uint32_t>(p->seed), static_cast<uint32_t>(p->seed >> 32));
    ra.russian_roulette = p->russian_roulette;
    ra.replay = p->replay_samples;
    ra.li_out = p->li_out;
    ra.film = film_accum;
    ra.work_counter = ctx->work_counter;
    ra.stats = ctx->stats;
    uint64_t n_items = static_cast<uint64_t>(ra.local_tiles) * ra.chunks;
    if (n_items == 0) {
        if (stats) memset(stats, 0, sizeof(*stats));
        return GBL_OK;
    }

    hipStream_t stream = static_cast<hipStream_t>(p->stream);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->work_counter, 0, sizeof(uint32_t), stream));
    const bool want_stats = p->collect_stats != 0;
    if (want_stats) HIP_TRY(ctx, hipMemsetAsync(ctx->stats, 0, 8 * sizeof(unsigned long long), stream));
    const int tp = GBL_TILE + 2 * sc.film.halo;
    size_t lds = sizeof(float) * (4 * tp * tp + 256) + 4 * sizeof(uint32_t) +
                 static_cast<size_t>(sc.stack_entries) * GBL_BLOCK * sizeof(uint32_t);
    if (lds > 160 * 1024) {
        ctx->error = "scene needs " + std::to_string(lds) + " bytes of LDS per workgroup (BVH too deep)";
        return GBL_ERR_UNSUPPORTED;
    }
    // persistent grid: enough workgroups to fill every CU at the occupancy LDS allows, never more than items
    int per_cu = static_cast<int>(std::min<size_t>(4, (160 * 1024) / lds));
    per_cu = std::max(1, per_cu);
    uint64_t grid64 = std::min<uint64_t>(n_items, static_cast<uint64_t>(ctx->num_cus) * per_cu);
    dim3 grid(static_cast<unsigned>(grid64)), block(GBL_BLOCK);
    const bool replay = p->sample_mode == GBL_SAMPLES_REPLAY;
    void (*kernel)(DevScene, RenderArgs) = nullptr;
    if (p->integrator == GBL_INTEGRATOR_PATH) {
        kernel = replay ? (want_stats ? path_trace_kernel<true, true> : path_trace_kernel<true, false>)
                        : (want_stats ? path_trace_kernel<false, true> : path_trace_kernel<false, false>);
    } else {
        kernel = replay ? (want_stats ? ao_kernel<true, true> : ao_kernel<true, false>)
                        : (want_stats ? ao_kernel<false, true> : ao_kernel<false, false>);
    }
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds)));
    if (stats) HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, sc, ra);
    HIP_TRY(ctx, hipGetLastError());
    if (stats) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        memset(stats, 0, sizeof(*stats));
        stats->kernel_ms = ms;
        uint64_t shard_pixels = 0;
        for (int t = ra.shard_index; t < total_tiles; t += ra.shard_count) {
            int tx = t % ra.tiles_x, ty = t / ra.tiles_x;
            int tw = std::min(GBL_TILE, ra.window[1] - (ra.window[0] + GBL_TILE * tx));
            int th = std::min(GBL_TILE, ra.window[3] - (ra.window[2] + GBL_TILE * ty));
            shard_pixels += static_cast<uint64_t>(tw) * th;
        }
        stats->paths = shard_pixels * ra.spp;
        if (want_stats) {
            unsigned long long h[8];
            HIP_TRY(ctx, hipMemcpy(h, ctx->stats, sizeof(h), hipMemcpyDeviceToHost));
            stats->paths = h[0];
            stats->extension_rays = h[1];
            stats->shadow_rays = h[2];
            stats->nodes = h[3];
            stats->tris = h[4];
            stats->splats = h[5];
            stats->dims = h[6];
        }
    }
    return GBL_OK;
}

gbl_status gbl_film_resolve(gbl_ctx* ctx, const float* film_accum, float* rgb_out, void* stream) {
    if (!ctx || !film_accum || !rgb_out) return GBL_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int n = ctx->info.xres * ctx->info.yres;
    hipLaunchKernelGGL(film_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), film_accum,
                       rgb_out, n);
    HIP_TRY(ctx, hipGetLastError());
    return GBL_OK;
}

// ncclAllReduce(sum, float) over the film, resolved from librccl at first use so
// single-GPU users never load RCCL.
gbl_status gbl_film_allreduce(gbl_ctx* ctx, void* rccl_comm, float* film_accum, void* stream) {
    if (!ctx || !rccl_comm || !film_accum) return GBL_ERR_INVALID;
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    if (!ctx->rccl_allreduce) {
        ctx->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!ctx->rccl) ctx->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!ctx->rccl) {
            ctx->error = std::string("cannot load librccl: ") + dlerror();
            return GBL_ERR_DEVICE;
        }
        ctx->rccl_allreduce = dlsym(ctx->rccl, "ncclAllReduce");
        if (!ctx->rccl_allreduce) {
            ctx->error = "librccl has no ncclAllReduce";
            return GBL_ERR_DEVICE;
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t count = static_cast<size_t>(ctx->info.xres) * ctx->info.yres * 4;
    const int kNcclFloat32 = 7, kNcclSum = 0;
    int rc = reinterpret_cast<allreduce_fn>(ctx->rccl_allreduce)(film_accum, film_accum, count, kNcclFloat32, kNcclSum, rccl_comm,
                                                                 static_cast<hipStream_t>(stream));
    if (rc != 0) {
        ctx->error = "ncclAllReduce failed with code " + std::to_string(rc);
        return GBL_ERR_DEVICE;
    }
    return GBL_OK;
}

}  // extern "C"
