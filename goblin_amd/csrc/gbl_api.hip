// libgoblin_hip.so -- C ABI of the device integrator (include/goblin_hip.h).
//
// gbl_create   packs the scene on the host (scene_prep.cpp) and uploads it once.
// gbl_render   launches the persistent render kernel over a sample sub-window and
//              accumulates into the caller's device film.
// There is no CPU fallback anywhere in this library: every entry point that
// computes something needs a HIP device and fails with GBL_ERR_DEVICE otherwise.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/goblin_hip.h"
#include "device_scene.h"
#include "abi_guard.h"
#include "gbl_internal.h"
#include "kernels/stream.h"     // StreamLayout: the host sizes the stream sampler's scratch
#include "kernels/trace.h"      // GBL_WF_STACK_LDS
#include "scene_prep.h"

namespace {
thread_local std::string g_create_error;
}

namespace {

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

// hipMalloc `capacity` elements and copy the first `count` from the host
template <class T>
gbl_status upload_raw(gbl_ctx* ctx, const T* src, size_t count, size_t capacity, const T** out) {
    size_t bytes = std::max<size_t>(1, capacity) * sizeof(T);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        ctx->error = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? GBL_ERR_OOM : GBL_ERR_DEVICE;
    }
    ctx->allocations.push_back(p);
    ctx->info.scene_bytes += bytes;
    if (count) HIP_TRY(ctx, hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
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


// ---------------------------------------------------------------------------
#ifndef GBL_WF_POOL_LOG2
#define GBL_WF_POOL_LOG2 23   // 8 Mi path slots in flight (1.9 GB of path state): 2^21 -> 2^23 is -9 % on config 3, -10 % on config 4
#endif
#ifndef GBL_WF_SHADOW_WGS
#define GBL_WF_SHADOW_WGS 1   // workgroups per CU of the concurrent shadow-ray trace launch
#endif
// Wavefront schedule: host side of kernels/wavefront.h
// ---------------------------------------------------------------------------
template <class T>
gbl_status wf_alloc(gbl_ctx* ctx, T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) {
        ctx->error = std::string("hipMalloc(wavefront pool): ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? GBL_ERR_OOM : GBL_ERR_DEVICE;
    }
    ctx->allocations.push_back(p);
    *out = static_cast<T*>(p);
    return GBL_OK;
}

gbl_status wf_ensure_pool(gbl_ctx* ctx) {
    if (ctx->wf_pool) return GBL_OK;
    uint32_t pool_log2 = GBL_WF_POOL_LOG2;
    if (const char* e = getenv("GBL_WF_POOL_LOG2")) pool_log2 = static_cast<uint32_t>(std::min(26, std::max(16, atoi(e))));
    const uint32_t pool = 1u << pool_log2;   // path slots (~160 B each)
    WfArgs& w = ctx->wf;
    memset(&w, 0, sizeof(w));
    gbl_status st;
#define WF_A(field, n) if ((st = wf_alloc(ctx, &w.field, (n))) != GBL_OK) return st
    WF_A(ray_o, pool); WF_A(ray_d, pool); WF_A(hit, pool); WF_A(hit_inst, pool);
    WF_A(s_thr, pool); WF_A(s_li, pool); WF_A(s_ld, pool); WF_A(s_f, pool); WF_A(s_id, pool); WF_A(s_vis, pool);
    WF_A(ext_q, pool); WF_A(ext_count, pool / 64);
    WF_A(sh_d, pool); WF_A(sh_slot, pool); WF_A(sh_count, pool / 64);
    WF_A(live_flags, 8);
    WF_A(wave_next, 2 * (pool / 64)); WF_A(steal_next, 1);
    if (ctx->scene.has_masks) {   // see WfArgs
        WF_A(hit2, pool); WF_A(hit2_inst, pool); WF_A(mis_tr, pool); WF_A(sh_f, pool); WF_A(sh_L, pool); WF_A(sh_c, pool);
    }
#undef WF_A
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->wf_host_flags), 8 * sizeof(uint32_t)) != hipSuccess) {
        ctx->error = "hipHostMalloc(wavefront flags) failed";
        return GBL_ERR_OOM;
    }
    ctx->wf_pool = pool;
    return GBL_OK;
}

// Budget for the per-sample radiance buffer: a quarter of the device's memory (72 GB of the MI355X's 288 GB), so that
// BASELINE's largest frame (config 3: 1028^2 px x 1024 spp x 16 B = 17.3 GB) is one pass.
uint64_t li_budget_bytes(gbl_ctx* ctx) {
    if (ctx->li_budget == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) total_b = 8ull << 30;
        ctx->li_budget = std::max<uint64_t>(1ull << 30, static_cast<uint64_t>(total_b) / 4);
        if (const char* e = getenv("GBL_LI_BUDGET_MB")) ctx->li_budget = std::max<uint64_t>(1ull << 20, strtoull(e, nullptr, 10) << 20);
    }
    return ctx->li_budget;
}

// Quad-per-ray queries (kernels/quadtrace.h): what the lean kernels of the native sampler run (-9 ... -14 % on the BASELINE scenes);
// GBL_MK_QUAD=0 selects their one-ray-per-lane builds instead (bit-identity tests, A/B measurements).
static bool quad_wanted() {
    const char* e = getenv("GBL_MK_QUAD");
    return e == nullptr || e[0] != '0';
}

// Random numbers one camera sample's transmittance + Lv may draw (GBL_SAMPLES_STREAM sizes a pixel's tail with it): 9 per light
// sample and the pick for the homogeneous region; for a heterogeneous one the jitters and up to 5 per point of the march,
// whose length is bounded by the region's longest world-space diagonal over the step (kernels/render_kernels.h
// stream_medium_phase) -- plus room for the generator state that phase sets aside.
static uint64_t medium_draws_per_sample(const DevScene& sc) {
    if (sc.volume.on == 0u) return 0;
    if (sc.volume.hetero == 0u) return 9ull * static_cast<uint64_t>(std::max(0, sc.volume.sample_num)) + 1;
    const DevVolume& v = sc.volume;
    double diag = 0.0;
    for (int s = 0; s < 4; ++s) {
        const double e[3] = {double(v.hi[0] - v.lo[0]), (s & 1 ? -1.0 : 1.0) * double(v.hi[1] - v.lo[1]), (s & 2 ? -1.0 : 1.0) * double(v.hi[2] - v.lo[2])};
        double w[3];
        for (int r = 0; r < 3; ++r) w[r] = v.m[4 * r] * e[0] + v.m[4 * r + 1] * e[1] + v.m[4 * r + 2] * e[2];
        diag = std::max(diag, std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
    }
    const double steps = std::floor(diag / std::max(1e-6, double(v.step))) + 3.0;
    return 2ull + 5ull * static_cast<uint64_t>(std::min(steps, 1.0e6)) + 700ull;
}

// Stack levels of the wavefront trace kernels beyond the LDS part: one column per thread of the largest persistent
// trace grid (8 workgroups per CU).  Re-made when an instance edit deepens the TLAS.
gbl_status wf_ensure_spill(gbl_ctx* ctx, int min_levels = 0) {
    const int deep = std::max(min_levels, ctx->scene.stack_entries > GBL_WF_STACK_LDS ? ctx->scene.stack_entries - GBL_WF_STACK_LDS : 1);
    if (ctx->wf_spill && deep <= ctx->wf_spill_levels) return GBL_OK;
    if (ctx->wf_spill) (void)hipFree(ctx->wf_spill);
    ctx->wf_spill = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->wf_spill), static_cast<size_t>(deep) * ctx->num_cus * 8 * GBL_BLOCK * sizeof(uint32_t));
    if (e != hipSuccess) {
        ctx->error = std::string("hipMalloc(trace stack backing): ") + hipGetErrorString(e);
        return GBL_ERR_OOM;
    }
    ctx->wf_spill_levels = deep;
    return GBL_OK;
}

// per-sample radiance scratch (16 B per sample), grown on demand
gbl_status ensure_li(gbl_ctx* ctx, size_t entries) {
    if (entries <= ctx->wf_li_entries) return GBL_OK;
    if (ctx->wf_li) (void)hipFree(ctx->wf_li);
    ctx->wf_li = nullptr;
    ctx->wf_li_entries = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->wf_li), entries * sizeof(float4));
    if (e != hipSuccess) {
        ctx->error = std::string("hipMalloc(per-sample radiance): ") + hipGetErrorString(e);
        return GBL_ERR_OOM;
    }
    ctx->wf_li_entries = entries;
    return GBL_OK;
}

// GBL_SAMPLES_STREAM: per-workgroup sample-generation scratch and the per-sample image positions the splat reads
gbl_status ensure_stream_buffers(gbl_ctx* ctx, uint64_t words_per_wg, uint64_t workgroups, uint64_t samples, RenderArgs* ra) {
    ra->stream_stride = words_per_wg;
    const uint64_t need = words_per_wg * sizeof(uint32_t) * workgroups;
    if (need > ctx->stream_scratch_bytes) {
        if (ctx->stream_scratch) (void)hipFree(ctx->stream_scratch);
        ctx->stream_scratch = nullptr;
        ctx->stream_scratch_bytes = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->stream_scratch), need);
        if (e != hipSuccess) {
            ctx->error = std::string("hipMalloc(stream scratch): ") + hipGetErrorString(e);
            return GBL_ERR_OOM;
        }
        ctx->stream_scratch_bytes = need;
    }
    ra->stream_scratch = ctx->stream_scratch;
    const uint64_t xy_bytes = samples * 2 * sizeof(float);
    if (xy_bytes > ctx->stream_xy_bytes) {
        if (ctx->stream_xy) (void)hipFree(ctx->stream_xy);
        ctx->stream_xy = nullptr;
        ctx->stream_xy_bytes = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->stream_xy), xy_bytes);
        if (e != hipSuccess) {
            ctx->error = std::string("hipMalloc(stream image positions): ") + hipGetErrorString(e);
            return GBL_ERR_OOM;
        }
        ctx->stream_xy_bytes = xy_bytes;
    }
    ra->image_xy = ctx->stream_xy;
    return GBL_OK;
}

gbl_status render_wavefront(gbl_ctx* ctx, const RenderArgs& ra, const gbl_render_params* p, hipStream_t stream, bool want_stats,
                            bool replay) {
    (void)p;
    const DevScene& sc = ctx->scene;
    gbl_status st = wf_ensure_pool(ctx);
    if (st != GBL_OK) return st;
    if ((st = wf_ensure_spill(ctx)) != GBL_OK) return st;
    const uint64_t window_pixels = static_cast<uint64_t>(ra.window[1] - ra.window[0]) * (ra.window[3] - ra.window[2]);
    // samples per pass: bound the per-sample radiance buffer (16 B per sample) to ~2 GiB
    int pass_spp = ra.spp;
    if (!ra.li_out) {
        const uint64_t budget = li_budget_bytes(ctx) / 16;
        while (window_pixels * pass_spp > budget && pass_spp % 2 == 0 && pass_spp > 1) pass_spp /= 2;
    }
    if (static_cast<uint64_t>(ra.local_tiles) * 64 * pass_spp >= (1ull << 32) || window_pixels * pass_spp >= (1ull << 32)) {
        ctx->error = "too many paths per pass for 32-bit path ids: split the window";
        return GBL_ERR_INVALID;
    }
    WfArgs wa = ctx->wf;
    wa.stack_spill = ctx->wf_spill;
    if (ra.li_out) {
        wa.li_buf = reinterpret_cast<float4*>(ra.li_out);   // single pass: li_buf is the caller's buffer, in its order
    } else {
        if ((st = ensure_li(ctx, static_cast<size_t>(window_pixels) * pass_spp)) != GBL_OK) return st;
        wa.li_buf = ctx->wf_li;
    }
    const uint32_t total = static_cast<uint32_t>(static_cast<uint64_t>(ra.local_tiles) * 64 * pass_spp);
    uint32_t pool = std::min<uint32_t>(ctx->wf_pool, (total + GBL_BLOCK - 1) / GBL_BLOCK * GBL_BLOCK);
    wa.pool_size = pool;
    wa.total_paths = total;
    wa.pass_spp = pass_spp;
    {   // ids are dealt in blocks of 64, round-robin over the pool/64 shade-waves
        // ... three quarters of them up front; the rest is the reserve the waves that finish early draw on (wf_shade)
        const uint32_t waves = pool / 64, blocks = (total + 63) / 64;
        uint32_t reserve_pct = 25;
        if (const char* e = getenv("GBL_WF_RESERVE")) reserve_pct = static_cast<uint32_t>(std::min(100, std::max(0, atoi(e))));   // measurement aid
        wa.static_blocks = static_cast<uint32_t>(static_cast<uint64_t>(blocks / waves) * (100 - reserve_pct) / 100);
        if (reserve_pct == 0) wa.static_blocks = (blocks + waves - 1) / waves;   // (everything dealt up front: round 3's scheme)
        wa.total_blocks = blocks;
    }
    size_t lds_stack = static_cast<size_t>(std::min<int>(sc.stack_entries, GBL_WF_STACK_LDS)) * GBL_BLOCK * sizeof(uint32_t);
    // the trace kernels keep the top of the tree in LDS behind their stacks (trace.h HotSplitStack): GBL_WF_HOT nodes
    RenderArgs ra_trace = ra;
    ra_trace.hot_word = static_cast<uint32_t>(lds_stack / sizeof(uint32_t));
    ra_trace.hot_count = 0;
    {
        uint32_t want = GBL_WF_HOT_NODES;
        if (const char* e = getenv("GBL_WF_HOT")) want = static_cast<uint32_t>(std::max(0, atoi(e)));
        ra_trace.hot_count = std::min<uint32_t>(want, sc.hot_nodes);
        lds_stack += ra_trace.hot_count * sizeof(DevNode);
    }
    const int tp = GBL_TILE + 2 * sc.film.halo;
    const size_t lds_tile = sizeof(float) * (4 * tp * tp + 256);
    // EXT kernels carry the analytic shapes / directional light / non-pinhole cameras; plain scenes run the lean
    // build.  Instrumented launches always use the EXT build (same work, same counters).
    const bool ext = sc.extended != 0;
    const bool masks = sc.has_masks != 0;
    // native sampler, lean build: no tie rule (trace.h)
    const bool lean_native = !ext && !masks && !want_stats && !replay && p->exact_ties == 0;
    gbl_wf_kernel k_ext = gbl_kernel_wf_trace(false, want_stats, ext || masks || want_stats, masks, !lean_native);
    gbl_wf_kernel k_shd = gbl_kernel_wf_trace(true, want_stats, ext || masks || want_stats, masks, !lean_native);
    // persistent trace grids: exactly the resident workgroups (regions are assigned statically, so a
    // workgroup that has to wait for a free CU would serialise its share), never more waves than regions
    int occ_ext = 0, occ_shd = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_ext, reinterpret_cast<const void*>(k_ext), GBL_BLOCK, lds_stack) != hipSuccess || occ_ext < 1) occ_ext = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_shd, reinterpret_cast<const void*>(k_shd), GBL_BLOCK, lds_stack) != hipSuccess || occ_shd < 1) occ_shd = 1;
    occ_ext = std::min(occ_ext, 8);   // stack_spill is sized for 8 workgroups per CU
    occ_shd = std::min(occ_shd, 8);
    // The shadow rays of iteration k and the extension rays of iteration k+1 are both known once wf_shade(k) has run
    // and do not depend on each other, so they trace CONCURRENTLY: the shadow launch goes to a second stream with
    // GBL_WF_SHADOW_WGS workgroups per CU, the extension launch keeps the rest of the occupancy (both grids are
    // persistent, so together they must not exceed what is resident).  One launch tail per iteration instead of two.
    const bool overlap = occ_ext > GBL_WF_SHADOW_WGS && !getenv("GBL_WF_NO_OVERLAP");
    if (overlap) {
        occ_ext -= GBL_WF_SHADOW_WGS;
        occ_shd = GBL_WF_SHADOW_WGS;
        if (!ctx->wf_aux) {
            HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->wf_aux, hipStreamNonBlocking));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->wf_ev_shade, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->wf_ev_shadow, hipEventDisableTiming));
        }
    }
    const uint64_t max_wgs = (pool / 64 + 3) / 4;
    unsigned ext_wgs = static_cast<unsigned>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(ctx->num_cus) * occ_ext, max_wgs)));
    unsigned shd_wgs = static_cast<unsigned>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(ctx->num_cus) * occ_shd, max_wgs)));
    // the two trace launches may run at the same time: disjoint columns of the stack backing (ext <= 7 and shd = 1
    // workgroups per CU of the 8 the backing is sized for).  Without the overlap they are serialised on one stream and
    // each may take up to 8 per CU, so they share the columns.
    uint32_t* const spill_ext = ctx->wf_spill;
    uint32_t* const spill_shd = overlap ? ctx->wf_spill + static_cast<size_t>(ctx->wf_spill_levels) * ext_wgs * GBL_BLOCK : ctx->wf_spill;
    if (static_cast<uint64_t>(overlap ? ext_wgs + shd_wgs : std::max(ext_wgs, shd_wgs)) > static_cast<uint64_t>(ctx->num_cus) * 8) {
        ctx->error = "wavefront trace grids exceed the stack backing";
        return GBL_ERR_DEVICE;
    }
    dim3 block(GBL_BLOCK), grid_ext(ext_wgs), grid_shd(shd_wgs), grid_shade(pool / GBL_BLOCK);
    gbl_wf_kernel k_shade = gbl_kernel_wf_shade(replay, want_stats, ext || want_stats);
    gbl_wf_kernel k_splat = gbl_kernel_wf_splat(replay, want_stats);
    if (!k_ext || !k_shd || !k_shade || !k_splat) {
        ctx->error = "wavefront kernel variant not built";
        return GBL_ERR_UNSUPPORTED;
    }
    if (lds_stack > 64 * 1024) {
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ext), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds_stack)));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_shd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds_stack)));
    }
    for (int k0 = 0; k0 < ra.spp; k0 += pass_spp) {
        wa.pass_k0 = k0;
        HIP_TRY(ctx, hipMemsetAsync(wa.wave_next, 0, 2 * (pool / 64) * sizeof(uint32_t), stream));
        HIP_TRY(ctx, hipMemsetAsync(wa.steal_next, 0, sizeof(uint32_t), stream));
        wa.init = 1;
        wa.flag_index = 7;
        hipLaunchKernelGGL(k_shade, grid_shade, block, 0, stream, sc, ra, wa);
        wa.init = 0;
        // slot s traces ceil((total - s) / pool) paths, each at least one iteration
        uint64_t iter = 0, min_iters = total / pool;
        bool done = false, shadow_pending = false;
        while (!done) {
            HIP_TRY(ctx, hipMemsetAsync(wa.live_flags, 0, 8 * sizeof(uint32_t), stream));
            int batch = iter + 4 <= min_iters ? static_cast<int>(std::min<uint64_t>(min_iters - iter, 64)) : 4;
            for (int b = 0; b < batch; ++b) {
                wa.flag_index = b & 7;
                wa.stack_spill = spill_ext;
                hipLaunchKernelGGL(k_ext, grid_ext, block, lds_stack, stream, sc, ra_trace, wa);
                if (overlap && shadow_pending) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->wf_ev_shadow, 0));   // wf_shade reads s_ld, rewrites the shadow queue
                hipLaunchKernelGGL(k_shade, grid_shade, block, 0, stream, sc, ra, wa);
                wa.stack_spill = spill_shd;
                if (overlap) {
                    HIP_TRY(ctx, hipEventRecord(ctx->wf_ev_shade, stream));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->wf_aux, ctx->wf_ev_shade, 0));
                    hipLaunchKernelGGL(k_shd, grid_shd, block, lds_stack, ctx->wf_aux, sc, ra_trace, wa);
                    HIP_TRY(ctx, hipEventRecord(ctx->wf_ev_shadow, ctx->wf_aux));
                    shadow_pending = true;
                } else {
                    hipLaunchKernelGGL(k_shd, grid_shd, block, lds_stack, stream, sc, ra_trace, wa);
                }
                ++iter;
            }
            if (iter >= min_iters) {
                HIP_TRY(ctx, hipMemcpyAsync(ctx->wf_host_flags, wa.live_flags, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(ctx, hipStreamSynchronize(stream));
                if (ctx->wf_host_flags[(batch - 1) & 7] == 0) done = true;
            }
            if (iter > (1u << 20)) {
                ctx->error = "wavefront loop did not terminate";
                return GBL_ERR_DEVICE;
            }
        }
        if (overlap && shadow_pending) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->wf_ev_shadow, 0));   // rejoin before the pool is reused
        hipLaunchKernelGGL(k_splat, dim3(ra.local_tiles), block, lds_tile, stream, sc, ra, wa);
        HIP_TRY(ctx, hipGetLastError());
    }
    return GBL_OK;
}

}  // namespace

extern "C" {

int gbl_abi_version(void) { return GBL_ABI_VERSION; }

const char* gbl_last_error(const gbl_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

gbl_status gbl_create(const gbl_scene_desc* desc, int device, gbl_ctx** out) {
    const char* e = getenv("GBL_BVH_BUILD");
    return gbl_create_ex(desc, device, (e && !strcmp(e, "device")) ? GBL_CREATE_DEVICE_BVH : 0u, out);
}

static gbl_status gbl_create_ex_impl(const gbl_scene_desc* desc, int device, uint32_t flags, gbl_ctx** out) {
    const bool device_bvh = (flags & GBL_CREATE_DEVICE_BVH) != 0;
    if (!desc || !out) {
        g_create_error = "null argument";
        return GBL_ERR_INVALID;
    }
    *out = nullptr;
    PackedScene packed;
    std::string err;
    auto t_pack0 = std::chrono::steady_clock::now();
    gbl_status st = pack_scene(desc, &packed, &err, device_bvh);
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
    if (!device_bvh) {
        if ((st = upload(ctx, packed.nodes, &sc.nodes)) != GBL_OK) return bail(st);
        if ((st = upload(ctx, packed.tris, &sc.tris)) != GBL_OK) return bail(st);
    } else {
        // nodes = [TLAS (from the host) | mesh 0 | mesh 1 | ...], tris = every mesh's triangles in Morton order
        size_t node_cap = packed.nodes.size(), tri_cap = 0;
        for (uint32_t m = 0; m < desc->num_meshes; ++m)
            if (desc->meshes[m].shape == GBL_SHAPE_MESH) {
                node_cap += desc->meshes[m].tri_count;
                tri_cap += desc->meshes[m].tri_count;
            }
        // device buffers are allocated at their final size; only the TLAS nodes and the raw geometry cross PCIe
        if ((st = upload_raw(ctx, packed.nodes.data(), packed.nodes.size(), node_cap, &sc.nodes)) != GBL_OK) return bail(st);
        if ((st = upload_raw(ctx, static_cast<const DevTri*>(nullptr), 0, tri_cap, &sc.tris)) != GBL_OK) return bail(st);
        const float* d_pos = nullptr;
        const uint32_t* d_idx = nullptr;
        const size_t n_pos = 3 * static_cast<size_t>(desc->num_vertices), n_idx = 3 * static_cast<size_t>(desc->num_triangles);
        if ((st = upload_raw(ctx, desc->positions, n_pos, n_pos, &d_pos)) != GBL_OK) return bail(st);
        if ((st = upload_raw(ctx, desc->indices, n_idx, n_idx, &d_idx)) != GBL_OK) return bail(st);
        std::vector<int32_t> mesh_root(desc->num_meshes, 0);
        int32_t node_base = static_cast<int32_t>(packed.nodes.size());
        uint32_t tri_base = 0;
        int max_depth = 0;
        for (uint32_t m = 0; m < desc->num_meshes; ++m) {
            const gbl_mesh& gm = desc->meshes[m];
            if (gm.shape != GBL_SHAPE_MESH) continue;
            uint32_t used = 0;
            int depth = 0;
            st = gbl_build_blas_device(ctx, d_pos + 3 * static_cast<size_t>(gm.vertex_offset), d_idx + 3 * static_cast<size_t>(gm.tri_offset), gm.tri_count,
                                   &packed.mesh_lo[3 * m], &packed.mesh_hi[3 * m], const_cast<DevNode*>(sc.nodes), node_base,
                                   const_cast<DevTri*>(sc.tris), tri_base, gm.tri_offset, (gm.has_normal ? 1u : 0u) | (gm.has_uv ? 2u : 0u), &mesh_root[m], &used, &depth);
            if (st != GBL_OK) return bail(st);
            node_base += static_cast<int32_t>(used);
            tri_base += gm.tri_count;
            max_depth = std::max(max_depth, depth);
            packed.mesh_stack_need[m] = 3 * depth;
        }
        for (size_t i = 0; i < packed.instances.size(); ++i)
            if (packed.instances[i].shape == 0u) packed.instances[i].root = mesh_root[packed.instances[i].mesh];
        for (uint32_t m = 0; m < desc->num_meshes; ++m)
            if (desc->meshes[m].shape == GBL_SHAPE_MESH) packed.mesh_root[m] = mesh_root[m];
        packed.blas_max_depth = max_depth;
        packed.blas_nodes = static_cast<uint64_t>(node_base) - packed.nodes.size();
        {   // device-built trees: the per-level bound of each mesh's BLAS under the exact TLAS sum
            const std::vector<DevNode> tl(packed.nodes.begin() + packed.tlas_base, packed.nodes.begin() + packed.tlas_base + static_cast<std::ptrdiff_t>(packed.tlas_nodes));
            packed.stack_entries = scene_stack_entries(tl, packed.tlas_base, packed.tlas_root, packed.instances, packed.mesh_stack_need);
        }
        packed.tris.resize(tri_cap);   // for gbl_info only
    }
    ctx->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pack0).count();
    if ((st = upload(ctx, packed.tri_shade, &sc.tri_shade)) != GBL_OK) return bail(st);
    if (!device_bvh) {
        if ((st = upload(ctx, packed.tri_bounds_leaf, &sc.tri_bounds)) != GBL_OK) return bail(st);
    } else {
        const DevTriBound* by_id = nullptr;
        if ((st = upload(ctx, packed.tri_bounds, &by_id)) != GBL_OK) return bail(st);
        if ((st = upload_raw(ctx, static_cast<const DevTriBound*>(nullptr), 0, packed.tris.size(), &sc.tri_bounds)) != GBL_OK) return bail(st);
        gbl_launch_tri_bounds_gather(sc.tris, by_id, const_cast<DevTriBound*>(sc.tri_bounds), static_cast<uint32_t>(packed.tris.size()));
        if (hipError_t le = hipGetLastError(); le != hipSuccess) {
            ctx->error = std::string("triangle bound gather: ") + hipGetErrorString(le);
            return bail(GBL_ERR_DEVICE);
        }
    }
    if ((st = upload(ctx, packed.tri_order, &sc.tri_order)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.instance_bounds, &sc.instance_bounds)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.positions, &sc.positions)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.normals, &sc.normals)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.uvs, &sc.uvs)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.instances, &sc.instances)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.materials, &sc.materials)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.textures, &sc.textures)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.lights, &sc.lights)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.light_tris, &sc.light_tris)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.light_cdf, &sc.light_cdf)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.light_pick_pdf, &sc.light_pick_pdf)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, ftab, &sc.filter_table)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.images, &sc.images)) != GBL_OK) return bail(st);
    if ((st = upload_raw(ctx, desc->texels, desc->num_texels, desc->num_texels, &sc.texels)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.ewa_lut, &sc.ewa_lut)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.ibl_dist, &sc.ibl_dist)) != GBL_OK) return bail(st);
    if ((st = upload(ctx, packed.vol_density, &sc.vol_density)) != GBL_OK) return bail(st);
    sc.has_ibl = packed.has_ibl;
    sc.hot_nodes = packed.hot_nodes;
    ctx->has_images = desc->num_images > 0;
    sc.tlas_root = packed.tlas_root;
    sc.num_instances = static_cast<int32_t>(packed.instances.size());
    sc.num_lights = static_cast<int32_t>(packed.lights.size());
    for (const DevLight& l : packed.lights) ctx->h_light_slots.push_back(l.wh_n);
    sc.stack_entries = packed.stack_entries;
    sc.extended = packed.extended;
    sc.has_masks = packed.has_masks;
    sc.has_bssrdf = packed.has_bssrdf;
    sc.wh_slots = packed.wh_slots;
    sc.volume = packed.volume;
    sc.camera = packed.camera;
    sc.film = packed.film;
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(uint32_t)) != hipSuccess) {
        ctx->error = "hipMalloc(work counter) failed";
        return bail(GBL_ERR_OOM);
    }
    ctx->allocations.push_back(p);
    ctx->work_counter = static_cast<uint32_t*>(p);
    if (hipMalloc(&p, 32 * sizeof(unsigned long long)) != hipSuccess) {
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
    ctx->h_instances.assign(desc->instances, desc->instances + desc->num_instances);
    ctx->h_meshes.assign(desc->meshes, desc->meshes + desc->num_meshes);
    ctx->h_materials.assign(desc->materials, desc->materials + desc->num_materials);
    ctx->mesh_lo = packed.mesh_lo;
    ctx->mesh_hi = packed.mesh_hi;
    ctx->mesh_root = packed.mesh_root;
    ctx->tlas_base = packed.tlas_base;
    ctx->tlas_capacity = packed.tlas_capacity;
    ctx->blas_depth = packed.blas_max_depth;
    ctx->mesh_stack_need = packed.mesh_stack_need;
    if (getenv("GBL_PROBE"))
        fprintf(stderr, "probe: traversal stack entries %d (per-level bound %d: TLAS depth %d, BLAS depth %d)\n", packed.stack_entries,
                3 * (packed.tlas_depth + packed.blas_max_depth) + 2, packed.tlas_depth, packed.blas_max_depth);
    for (uint32_t i = 0; i < desc->num_lights; ++i)
        if (desc->lights[i].type == GBL_LIGHT_DIRECTIONAL || desc->lights[i].type == GBL_LIGHT_IBL) ctx->has_directional = true;   // lights sized by the scene bound
    ctx->info.build_ms = ctx->build_ms;
    ctx->info.blas_depth = packed.blas_max_depth;
    ctx->info.tlas_depth = packed.tlas_depth;
    ctx->info.instanced_triangles = 0;
    for (uint32_t i = 0; i < desc->num_instances; ++i) ctx->info.instanced_triangles += desc->meshes[desc->instances[i].mesh].tri_count;
    *out = ctx;
    return GBL_OK;
}
gbl_status gbl_create_ex(const gbl_scene_desc* desc, int device, uint32_t flags, gbl_ctx** out) {
    return gbl_guard([&] { return gbl_create_ex_impl(desc, device, flags, out); }, [&](const std::string& what) { g_create_error = what; });
}

static gbl_status gbl_update_instances_impl(gbl_ctx* ctx, uint32_t first, uint32_t count, const gbl_trs* to_world) {
    if (!ctx) return GBL_ERR_INVALID;
    if (!to_world || static_cast<uint64_t>(first) + count > ctx->h_instances.size()) {
        ctx->error = "gbl_update_instances: instance range out of bounds";
        return GBL_ERR_INVALID;
    }
    if (ctx->has_directional) {
        ctx->error = "gbl_update_instances: a directional or image based light's power (and the image based light's sampling sphere) "
                     "depends on the scene bound (GoblinLight.cpp:203-210, 590-629); re-create the context instead";
        return GBL_ERR_UNSUPPORTED;
    }
    for (uint32_t i = 0; i < count; ++i)
        if (ctx->h_instances[first + i].area_light >= 0) {
            ctx->error = "gbl_update_instances: instance " + std::to_string(first + i) + " carries an area light, whose own transform would "
                         "have to move with it; re-create the context instead";
            return GBL_ERR_UNSUPPORTED;
        }
    std::vector<gbl_instance> edited = ctx->h_instances;
    for (uint32_t i = 0; i < count; ++i) edited[first + i].to_world = to_world[i];
    std::vector<DevInstance> inst;
    std::vector<DevNode> tlas;
    int32_t root = 0;
    int depth = 0;
    float lo[3], hi[3];
    std::string err;
    std::vector<DevInstanceBound> bounds;
    gbl_status st = build_tlas(edited.data(), static_cast<uint32_t>(edited.size()), ctx->h_meshes.data(), ctx->h_materials.data(), ctx->mesh_lo.data(),
                               ctx->mesh_hi.data(), ctx->mesh_root.data(), ctx->tlas_base, &inst, &tlas, &root, &depth, lo, hi, &err, &bounds);
    if (st != GBL_OK) {
        ctx->error = err;
        return st;
    }
    if (tlas.size() > ctx->tlas_capacity) {
        ctx->error = "gbl_update_instances: rebuilt TLAS does not fit its reserved nodes";
        return GBL_ERR_DEVICE;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());   // no render may be reading the old TLAS
    DevScene& sc = ctx->scene;
    if (!inst.empty())
        HIP_TRY(ctx, hipMemcpy(const_cast<DevInstance*>(sc.instances), inst.data(), inst.size() * sizeof(DevInstance), hipMemcpyHostToDevice));
    if (!bounds.empty())
        HIP_TRY(ctx, hipMemcpy(const_cast<DevInstanceBound*>(sc.instance_bounds), bounds.data(), bounds.size() * sizeof(DevInstanceBound), hipMemcpyHostToDevice));
    if (!tlas.empty())
        HIP_TRY(ctx, hipMemcpy(const_cast<DevNode*>(sc.nodes) + ctx->tlas_base, tlas.data(), tlas.size() * sizeof(DevNode), hipMemcpyHostToDevice));
    sc.tlas_root = root;
    sc.stack_entries = scene_stack_entries(tlas, ctx->tlas_base, root, inst, ctx->mesh_stack_need);   // (the wavefront stack backing is re-checked at render time)
    ctx->info.tlas_depth = depth;
    ctx->info.tlas_nodes = tlas.size();
    ctx->h_instances.swap(edited);
    ctx->auto_rays_per_path.clear();   // the edited scene's paths may be longer or shorter: AUTO measures again
    return GBL_OK;
}
gbl_status gbl_update_instances(gbl_ctx* ctx, uint32_t first, uint32_t count, const gbl_trs* to_world) {
    return gbl_guard([&] { return gbl_update_instances_impl(ctx, first, count, to_world); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

void gbl_destroy(gbl_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (void* p : ctx->allocations) (void)hipFree(p);
    if (ctx->wf_li) (void)hipFree(ctx->wf_li);
    if (ctx->prim_buf) (void)hipFree(ctx->prim_buf);
    if (ctx->prim_items) (void)hipFree(ctx->prim_items);
    if (ctx->sss_buf) (void)hipFree(ctx->sss_buf);
    if (ctx->vol_buf) (void)hipFree(ctx->vol_buf);
    if (ctx->stream_seeds) (void)hipFree(ctx->stream_seeds);
    if (ctx->stream_scratch) (void)hipFree(ctx->stream_scratch);
    if (ctx->stream_xy) (void)hipFree(ctx->stream_xy);
    if (ctx->wf_spill) (void)hipFree(ctx->wf_spill);
    if (ctx->wf_ev_shade) (void)hipEventDestroy(ctx->wf_ev_shade);
    if (ctx->wf_ev_shadow) (void)hipEventDestroy(ctx->wf_ev_shadow);
    if (ctx->wf_aux) (void)hipStreamDestroy(ctx->wf_aux);
    if (ctx->wf_host_flags) (void)hipHostFree(ctx->wf_host_flags);
    for (int i = 0; i < gbl_ctx::kTimingRing; ++i)
        for (int k = 0; k < 3; ++k)
            if (ctx->t_ev[i][k]) (void)hipEventDestroy(ctx->t_ev[i][k]);
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

}   // extern "C"

namespace {
// The first n values of libc rand() in a process that never called srand() -- what the reference seeds its per-tile
// generators with (RNGImp::RNGImp, GoblinUtils.cpp:19-20).  glibc's default is the TYPE_3 additive feedback generator
// over 31 words, r[i] = r[i-3] + r[i-31], seeded with 1 through the Park-Miller step and run 310 times before the
// first output, which drops the low bit.
std::vector<uint32_t> glibc_rand_sequence(size_t n) {
    std::vector<uint32_t> st(344 + n);
    int32_t r = 1;
    st[0] = 1u;
    for (int i = 1; i < 31; ++i) {
        const int64_t hi = r / 127773, lo = r % 127773;
        int64_t w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r = static_cast<int32_t>(w);
        st[i] = static_cast<uint32_t>(r);
    }
    for (int i = 31; i < 34; ++i) st[i] = st[i - 31];
    for (size_t i = 34; i < st.size(); ++i) st[i] = st[i - 31] + st[i - 3];
    std::vector<uint32_t> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = st[344 + i] >> 1;
    return out;
}
}   // namespace

extern "C" {

static gbl_status gbl_render_impl(gbl_ctx* ctx, const gbl_render_params* p, float* film_accum, gbl_stats* stats) {
    if (!ctx) return GBL_ERR_INVALID;
    if (!p || !film_accum) {
        ctx->error = "null argument";
        return GBL_ERR_INVALID;
    }
    const DevScene& sc = ctx->scene;
    RenderArgs ra;
    memset(&ra, 0, sizeof(ra));
    if (p->integrator != GBL_INTEGRATOR_PATH && p->integrator != GBL_INTEGRATOR_AO && p->integrator != GBL_INTEGRATOR_WHITTED) {
        ctx->error = "unknown integrator";
        return GBL_ERR_INVALID;
    }
    if (p->sample_per_pixel < 1 || p->max_ray_depth < 1) {
        ctx->error = "sample_per_pixel and max_ray_depth must be >= 1";
        return GBL_ERR_INVALID;
    }
    if (p->schedule > GBL_SCHEDULE_WAVEFRONT) {
        ctx->error = "unknown schedule " + std::to_string(p->schedule);
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
        ra.bssrdf_n = n1;
        ra.bssrdf_n2 = n2;
        ra.sss_off1 = static_cast<uint32_t>(4 + 3 * ra.max_depth);
        ra.sss_off2 = static_cast<uint32_t>(ra.off2_base + 4 * ra.max_depth);
        ra.sss_pat1 = 3u * static_cast<uint32_t>(ra.max_depth);
        ra.sss_pat2 = 2u * static_cast<uint32_t>(ra.max_depth);
        if (p->integrator == GBL_INTEGRATOR_WHITTED) {   // per-light patterns instead of per-bounce ones (kernels/whitted.h)
            ra.dims = 4 + 6 * sc.wh_slots + 1 + 4 * n1 + 4 * n2;
            ra.off2_base = 4 + 2 * sc.wh_slots + 1 + 4 * n1;
            ra.sss_off1 = static_cast<uint32_t>(4 + 2 * sc.wh_slots + 1);
            ra.sss_off2 = static_cast<uint32_t>(ra.off2_base + 4 * sc.wh_slots);
            ra.sss_pat1 = 2u * static_cast<uint32_t>(sc.num_lights) + 1u;   // after the per-light ls / bs patterns and pickLight
            ra.sss_pat2 = 2u * static_cast<uint32_t>(sc.num_lights);
        }
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
    if (p->sample_mode != GBL_SAMPLES_REPLAY && p->sample_mode != GBL_SAMPLES_NATIVE && p->sample_mode != GBL_SAMPLES_STREAM) {
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
    if (const char* e = getenv("GBL_CHUNK_SPP")) {   // measurement aid: samples of a pixel per work item
        const int c = atoi(e);
        if (c >= 1 && ra.spp % c == 0) chunks = ra.spp / c;
    }
    const bool stream_mode = p->sample_mode == GBL_SAMPLES_STREAM;
    if (stream_mode) chunks = 1;   // a work item is a whole tile, walked pixel by pixel (kernels/stream.h)
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

    // GBL_SCHEDULE_AUTO for the path tracer goes by how long the scene's paths are (see the schedule paragraph below): measured
    // once per context, max_ray_depth and Russian-roulette setting by a pilot -- one instrumented sample per pixel over every 4th
    // tile, native sampler, into a scratch film -- before anything of this call is queued.  A call too small to fill the
    // wavefront pool takes the megakernel whatever the pilot would say, so it does not pay for one.
    const uint64_t call_paths = static_cast<uint64_t>(ra.window[1] - ra.window[0]) * static_cast<uint64_t>(ra.window[3] - ra.window[2]) * ra.spp /
                                static_cast<uint64_t>(ra.shard_count);
    float pilot_rays_per_path = 0.0f;
    if (p->schedule == GBL_SCHEDULE_AUTO && p->integrator == GBL_INTEGRATOR_PATH && p->sample_mode != GBL_SAMPLES_STREAM && sc.has_masks == 0 &&
        call_paths >= GBL_AUTO_WAVEFRONT_PATHS) {
        const int key = p->max_ray_depth * 2 + (p->russian_roulette != 0 ? 1 : 0);
        auto it = ctx->auto_rays_per_path.find(key);
        if (it == ctx->auto_rays_per_path.end()) {
            float* scratch = nullptr;
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            const hipError_t me = hipMalloc(reinterpret_cast<void**>(&scratch), static_cast<size_t>(ctx->info.xres) * ctx->info.yres * 4 * sizeof(float));
            if (me != hipSuccess) {
                ctx->error = std::string("hipMalloc(AUTO pilot film): ") + hipGetErrorString(me);
                return GBL_ERR_OOM;
            }
            gbl_render_params pilot;
            memset(&pilot, 0, sizeof(pilot));
            pilot.integrator = GBL_INTEGRATOR_PATH;
            pilot.sample_per_pixel = 1;
            pilot.max_ray_depth = p->max_ray_depth;
            pilot.ao_sample_num = p->ao_sample_num;
            pilot.bssrdf_sample_num = p->bssrdf_sample_num;
            pilot.tile_shard_index = 0;
            pilot.tile_shard_count = 4;
            pilot.sample_mode = GBL_SAMPLES_NATIVE;
            pilot.seed = 0x9011057ull;
            pilot.russian_roulette = p->russian_roulette;   // roulette shortens the paths AUTO goes by
            pilot.collect_stats = 1;
            pilot.schedule = GBL_SCHEDULE_MEGAKERNEL;
            pilot.stream = p->stream;
            gbl_stats ps;
            const gbl_status pst = gbl_render_impl(ctx, &pilot, scratch, &ps);
            (void)hipFree(scratch);
            if (pst != GBL_OK) return pst;
            const float rpp = ps.paths ? static_cast<float>(static_cast<double>(ps.extension_rays + ps.shadow_rays) / static_cast<double>(ps.paths)) : 0.0f;
            it = ctx->auto_rays_per_path.emplace(key, rpp).first;
        }
        pilot_rays_per_path = it->second;
    }
    hipStream_t stream = static_cast<hipStream_t>(p->stream);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->work_counter, 0, sizeof(uint32_t), stream));
    const bool want_stats = p->collect_stats != 0;
    if (want_stats) HIP_TRY(ctx, hipMemsetAsync(ctx->stats, 0, 32 * sizeof(unsigned long long), stream));
    const int tp = GBL_TILE + 2 * sc.film.halo;
    size_t lds = sizeof(float) * (4 * tp * tp + 256) + 4 * sizeof(uint32_t) +
                 static_cast<size_t>(sc.stack_entries) * GBL_BLOCK * sizeof(uint32_t);
    if (lds > 160 * 1024) {
        ctx->error = "scene needs " + std::to_string(lds) + " bytes of LDS per workgroup (BVH too deep)";
        return GBL_ERR_UNSUPPORTED;
    }
    const bool replay = p->sample_mode == GBL_SAMPLES_REPLAY || stream_mode;
    if (stream_mode) {
        if (p->schedule == GBL_SCHEDULE_WAVEFRONT) {
            ctx->error = "GBL_SAMPLES_STREAM runs on the megakernel schedule";
            return GBL_ERR_UNSUPPORTED;
        }
        // the tiles rendered must be tiles of the reference's own tiling of the full sample window
        if ((ra.window[0] - full[0]) % GBL_TILE != 0 || (ra.window[2] - full[2]) % GBL_TILE != 0 ||
            (ra.window[1] != full[1] && (ra.window[1] - full[0]) % GBL_TILE != 0) ||
            (ra.window[3] != full[3] && (ra.window[3] - full[2]) % GBL_TILE != 0)) {
            ctx->error = "GBL_SAMPLES_STREAM: the window must consist of whole 8x8 tiles of the full sample window";
            return GBL_ERR_INVALID;
        }
        StreamLayout L = stream_layout(ra.spp, ra.root, ra.max_depth, ra.bssrdf_n, ra.bssrdf_n2,
                                       p->integrator == GBL_INTEGRATOR_AO ? ra.ao_n : 0);
        if (p->integrator == GBL_INTEGRATOR_WHITTED &&
            !stream_layout_whitted(L, ra.spp, ra.root, ra.bssrdf_n, ra.bssrdf_n2, sc.num_lights, [&](int i) { return ctx->h_light_slots[i]; })) {
            ctx->error = "GBL_SAMPLES_STREAM under the Whitted integrator covers up to " + std::to_string(GBL_STREAM_MAX_RUNS - 2) + " lights";
            return GBL_ERR_UNSUPPORTED;
        }
        if (static_cast<uint64_t>(sc.stack_entries) * GBL_BLOCK < L.S) {
            ctx->error = "GBL_SAMPLES_STREAM: sample_per_pixel too large for the shuffle scratch";
            return GBL_ERR_UNSUPPORTED;
        }
        lds += GBL_STREAM_LDS_WORDS * sizeof(uint32_t);
        {
            // the shuffles run one column per lane in the LDS region of the (idle) traversal stacks; widening that region
            // to 40 KB lets 40 columns of 256 samples go at once (65 columns at config 2: two rounds instead of three)
            // and still leaves three workgroups per CU
            const size_t stack_bytes = static_cast<size_t>(sc.stack_entries) * GBL_BLOCK * sizeof(uint32_t);
            const size_t want = std::max<size_t>(stack_bytes, 40 * 1024);
            if (lds - stack_bytes + want <= 52 * 1024) {
                lds += want - stack_bytes;
                ra.stream_lperm_words = static_cast<uint32_t>(want / sizeof(uint32_t));
            } else {
                ra.stream_lperm_words = static_cast<uint32_t>(stack_bytes / sizeof(uint32_t));
            }
        }
        const int ftx = (full[1] - full[0] + GBL_TILE - 1) / GBL_TILE, fty = (full[3] - full[2] + GBL_TILE - 1) / GBL_TILE;
        ra.full_tiles_x = ftx;
        if (!ctx->stream_seeds) {
            const std::vector<uint32_t> seeds = glibc_rand_sequence(static_cast<size_t>(ftx) * fty);
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->stream_seeds), seeds.size() * sizeof(uint32_t)));
            HIP_TRY(ctx, hipMemcpy(ctx->stream_seeds, seeds.data(), seeds.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        ra.tile_seeds = ctx->stream_seeds;
    }
    // schedule.  The persistent megakernel keeps the path state in registers and regenerates paths in place; the wavefront
    // formulation moves it through a 2^23-slot pool in HBM (~400 B per slot and iteration) to trace at five waves per SIMD with
    // compacted queues.  Which pays is a matter of how much of a path is incoherent traversal: measured (tools/auto_check.py,
    // 512^2 x 64 spp, wavefront / megakernel time) 1.6 on bunny.json at any depth (3.4 ... 3.5 rays per path), 1.15 ... 1.19 on the
    // 15-bunny grid (3.5 ... 3.8), 1.04 / 0.98 / 0.96 / 0.92 / 0.91 on the Cornell box at max_ray_depth 4 / 6 / 8 / 12 / 16 (4.9 / 6.6 /
    // 7.9 / 9.6 / 10.7 rays per path: a closed box, its paths never leave), 1.35 ... 1.73 on the feature scenes (2.9 ... 4.0); at full
    // size 291 against 280 ms on BASELINE configs[3] and 3.39 against 4.62 s on configs[2], where the pool is refilled 130
    // times.  AUTO = wavefront when the pilot above sees GBL_AUTO_WAVEFRONT_RAYS_PER_PATH rays per path or more and the call
    // (this rank's tiles x spp) holds at least GBL_AUTO_WAVEFRONT_PATHS camera samples to fill the pool with; megakernel
    // otherwise, for mask scenes (the wavefront kernels run the filtered MIS query and the attenuation walks inline: 35.3
    // against 20.2 ms on masked.json), for AO, Whitted and the stream sampler.  (Until round 3 AUTO went by instanced triangles
    // and by whether the megakernel's LDS stacks would leave three workgroups per CU; the megakernel has since gained 6 ... 9 %
    // and wins the grid at every size.)
    const bool wf_capable = p->integrator == GBL_INTEGRATOR_PATH;
    const bool auto_wavefront = pilot_rays_per_path >= GBL_AUTO_WAVEFRONT_RAYS_PER_PATH && call_paths >= GBL_AUTO_WAVEFRONT_PATHS;
    bool wavefront = wf_capable && !stream_mode && (p->schedule == GBL_SCHEDULE_WAVEFRONT ||
                                    (p->schedule == GBL_SCHEDULE_AUTO && !sc.has_masks && auto_wavefront));
    if (p->schedule == GBL_SCHEDULE_WAVEFRONT && !wavefront) {
        ctx->error = "the wavefront schedule covers the path tracer only";
        return GBL_ERR_UNSUPPORTED;
    }
    int per_cu = static_cast<int>(std::min<size_t>(8, (160 * 1024) / lds));
    per_cu = std::max(1, per_cu);
    if (stats) HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    hipEvent_t* tev = ctx->t_ev[ctx->t_calls % gbl_ctx::kTimingRing];
    for (int k = 0; k < 3; ++k)
        if (!tev[k]) HIP_TRY(ctx, hipEventCreate(&tev[k]));
    HIP_TRY(ctx, hipEventRecord(tev[0], stream));
    // the participating medium's {tr, Lv} of every camera sample (kernels/volume.h); the splat applies them.  Under
    // GBL_SAMPLES_STREAM the integrator kernels do it per pixel instead: the medium's draws follow each sample's Li draws
    // in the tile's stream (stream_medium_phase)
    if (sc.volume.on != 0u && !stream_mode) {
        const uint64_t entries = npix * ra.spp;
        if (entries * 32 > li_budget_bytes(ctx)) {
            ctx->error = "a scene with a participating medium keeps 32 bytes per camera sample: render this window in smaller pieces";
            return GBL_ERR_UNSUPPORTED;
        }
        if (entries > ctx->vol_entries) {
            if (ctx->vol_buf) (void)hipFree(ctx->vol_buf);
            ctx->vol_buf = nullptr;
            ctx->vol_entries = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->vol_buf), entries * 2 * sizeof(float4));
            if (e != hipSuccess) {
                ctx->error = std::string("hipMalloc(medium terms): ") + hipGetErrorString(e);
                return GBL_ERR_OOM;
            }
            ctx->vol_entries = entries;
        }
        ra.vol = ctx->vol_buf;
        gbl_render_kernel k_vol = gbl_kernel_vol(replay);
        const size_t lds_vol = static_cast<size_t>(sc.stack_entries) * GBL_BLOCK * sizeof(uint32_t);
        if (lds_vol > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_vol), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds_vol)));
        const uint64_t total = static_cast<uint64_t>(ra.local_tiles) * 64 * ra.spp;
        const uint64_t blocks = std::min<uint64_t>((total + GBL_BLOCK - 1) / GBL_BLOCK, static_cast<uint64_t>(ctx->num_cus) * 8);
        hipLaunchKernelGGL(k_vol, dim3(static_cast<unsigned>(blocks)), dim3(GBL_BLOCK), lds_vol, stream, sc, ra);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (p->integrator == GBL_INTEGRATOR_WHITTED) {
        // WhittedRenderer: one lane per camera sample with the recursion's frames in scratch (kernels/whitted.h), then the
        // shared splat kernel
        if (p->schedule == GBL_SCHEDULE_WAVEFRONT) {
            ctx->error = "the Whitted integrator has one kernel of its own: there is no wavefront schedule for it";
            return GBL_ERR_UNSUPPORTED;
        }
        if (ra.max_depth > GBL_WHITTED_MAX_DEPTH) {
            ctx->error = "max_ray_depth above " + std::to_string(GBL_WHITTED_MAX_DEPTH) + " is outside the Whitted kernel's frame stack";
            return GBL_ERR_UNSUPPORTED;
        }
        const uint64_t entries = npix * ra.spp;
        float4* li = reinterpret_cast<float4*>(ra.li_out);
        if (!li) {
            if (entries * 16 > li_budget_bytes(ctx)) {
                ctx->error = "the Whitted integrator keeps 16 bytes per camera sample of the call: render this window in smaller pieces";
                return GBL_ERR_UNSUPPORTED;
            }
            gbl_status lst = ensure_li(ctx, entries);
            if (lst != GBL_OK) return lst;
            li = ctx->wf_li;
        }
        if (stats) HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
        hipEvent_t* wev = ctx->t_ev[ctx->t_calls % gbl_ctx::kTimingRing];
        for (int k = 0; k < 3; ++k)
            if (!wev[k]) HIP_TRY(ctx, hipEventCreate(&wev[k]));
        HIP_TRY(ctx, hipEventRecord(wev[0], stream));
        gbl_li_kernel k_wh = stream_mode ? gbl_kernel_whitted_stream() : gbl_kernel_whitted(replay);
        size_t lds_wh = static_cast<size_t>(sc.stack_entries) * GBL_BLOCK * sizeof(uint32_t);
        if (stream_mode) {
            const size_t want = std::max<size_t>(lds_wh, 40 * 1024);
            ra.stream_lperm_words = static_cast<uint32_t>(want / sizeof(uint32_t));
            lds_wh = want + (4 + GBL_STREAM_LDS_WORDS) * sizeof(uint32_t);
        }
        if (lds_wh > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_wh), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds_wh)));
        const uint64_t total = static_cast<uint64_t>(ra.local_tiles) * 64 * ra.spp;
        uint64_t blocks = std::min<uint64_t>((total + GBL_BLOCK - 1) / GBL_BLOCK, static_cast<uint64_t>(ctx->num_cus) * 8);
        if (stream_mode) {   // one workgroup per tile in flight, each with its sample-generation scratch
            blocks = std::min<uint64_t>(static_cast<uint64_t>(ra.local_tiles), static_cast<uint64_t>(ctx->num_cus) * 4);
            StreamLayout L;
            stream_layout_whitted(L, ra.spp, ra.root, ra.bssrdf_n, ra.bssrdf_n2, sc.num_lights, [&](int i) { return ctx->h_light_slots[i]; });
            // a pixel's tail in the stream when a medium is present: per Li evaluation 6 floats per (light, slot) and 6 for the
            // two specular children, up to 2^(depth+1) - 1 evaluations per sample, then 9 per light sample of the medium.  The
            // phase walks the pixel in chunks, so the scratch only has to hold one sample's medium draws; give it the
            // worst case when that is small, 4 MiB per workgroup otherwise
            uint32_t tail = 0;
            if (sc.volume.on != 0u) {
                uint64_t slots = 0;
                for (int i = 0; i < sc.num_lights; ++i) slots += ctx->h_light_slots[i];
                const uint64_t med = medium_draws_per_sample(sc);
                const uint64_t worst = ((2ull << std::min(ra.max_depth, 20)) - 1) * (6 * slots + 6) + med;
                tail = static_cast<uint32_t>(std::max<uint64_t>(med, std::min<uint64_t>(worst, (1ull << 20) / L.S)));
                if (const char* e = getenv("GBL_STREAM_TAIL")) tail = static_cast<uint32_t>(std::max<uint64_t>(med, strtoull(e, nullptr, 10)));   // tests: force the chunked walk
            }
            ra.stream_tail_cap = L.S * tail;
            gbl_status sst = ensure_stream_buffers(ctx, stream_scratch_words(L, tail), blocks, entries, &ra);
            if (sst != GBL_OK) return sst;
        }
        hipLaunchKernelGGL(k_wh, dim3(static_cast<unsigned>(blocks)), dim3(GBL_BLOCK), lds_wh, stream, sc, ra, li);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(wev[1], stream));
        {
            WfArgs wa;
            memset(&wa, 0, sizeof(wa));
            wa.li_buf = li;
            wa.pass_k0 = 0;
            wa.pass_spp = ra.spp;
            // replay records of another quota: the splat only reads their image positions, at the Whitted record stride
            gbl_wf_kernel k_splat = gbl_kernel_wf_splat(replay, want_stats);
            const size_t lds_tile = sizeof(float) * (4 * tp * tp + 256);
            hipLaunchKernelGGL(k_splat, dim3(ra.local_tiles), dim3(GBL_BLOCK), lds_tile, stream, sc, ra, wa);
            HIP_TRY(ctx, hipGetLastError());
        }
        if (ra.vol && ra.li_out) {   // the caller's per-sample output carries what the tile received: tr * Li + Lv
            gbl_launch_vol_combine(reinterpret_cast<float4*>(ra.li_out), reinterpret_cast<const float4*>(ra.vol), entries, stream);
            HIP_TRY(ctx, hipGetLastError());
        }
        HIP_TRY(ctx, hipEventRecord(wev[2], stream));
        ctx->t_calls += 1;
        if (stats) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
            HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
            float ms = 0.0f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            memset(stats, 0, sizeof(*stats));
            stats->kernel_ms = ms;
            stats->schedule = GBL_SCHEDULE_MEGAKERNEL;
            stats->paths = npix * ra.spp;   // (the Whitted kernel is not instrumented: no ray counters)
        }
        return GBL_OK;
    }
    if (sc.has_bssrdf != 0 && p->integrator == GBL_INTEGRATOR_PATH && !stream_mode) {
        // Lsubsurface of every camera sample, ahead of the path kernels that add it at the first hit (kernels/subsurface.h)
        const uint64_t entries = npix * ra.spp;
        if (entries * 16 > li_budget_bytes(ctx)) {
            ctx->error = "a scene with subsurface materials keeps 16 bytes per camera sample: render this window in smaller pieces";
            return GBL_ERR_UNSUPPORTED;
        }
        if (entries > ctx->sss_entries) {
            if (ctx->sss_buf) (void)hipFree(ctx->sss_buf);
            ctx->sss_buf = nullptr;
            ctx->sss_entries = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->sss_buf), entries * sizeof(float4));
            if (e != hipSuccess) {
                ctx->error = std::string("hipMalloc(subsurface term): ") + hipGetErrorString(e);
                return GBL_ERR_OOM;
            }
            ctx->sss_entries = entries;
        }
        ra.sss = reinterpret_cast<const float*>(ctx->sss_buf);
        gbl_li_kernel k_sss = gbl_kernel_sss(replay);
        const size_t lds_sss = static_cast<size_t>(sc.stack_entries) * GBL_BLOCK * sizeof(uint32_t);
        if (lds_sss > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sss), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds_sss)));
        const uint64_t total = static_cast<uint64_t>(ra.local_tiles) * 64 * ra.spp;
        const uint64_t blocks = std::min<uint64_t>((total + GBL_BLOCK - 1) / GBL_BLOCK, static_cast<uint64_t>(ctx->num_cus) * 8);
        hipLaunchKernelGGL(k_sss, dim3(static_cast<unsigned>(blocks)), dim3(GBL_BLOCK), lds_sss, stream, sc, ra, ctx->sss_buf);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (wavefront) {
        gbl_status wst = render_wavefront(ctx, ra, p, stream, want_stats, replay);
        if (wst != GBL_OK) return wst;
        HIP_TRY(ctx, hipEventRecord(tev[1], stream));
    } else {
        // persistent grid: enough workgroups to fill every CU at the occupancy LDS allows, never more than items
        uint64_t grid64 = std::min<uint64_t>(n_items, static_cast<uint64_t>(ctx->num_cus) * per_cu);
        dim3 grid(static_cast<unsigned>(grid64)), block(GBL_BLOCK);
        const bool ext = sc.extended != 0;   // see render_wavefront
        void (*kernel)(DevScene, RenderArgs) = nullptr;
        if (stream_mode) {
            kernel = p->integrator == GBL_INTEGRATOR_AO ? gbl_kernel_ao_stream(ext || want_stats)
                                                        : gbl_kernel_path_stream(want_stats, ext || want_stats);
            const StreamLayout L = stream_layout(ra.spp, ra.root, ra.max_depth, ra.bssrdf_n, ra.bssrdf_n2,
                                                 p->integrator == GBL_INTEGRATOR_AO ? ra.ao_n : 0);
            // a sample's tail in the stream: up to 6 discarded floats per bounce, 9 per light sample of the medium
            const uint32_t med = static_cast<uint32_t>(medium_draws_per_sample(sc));
            const uint32_t tail = (p->integrator == GBL_INTEGRATOR_AO ? 0u : 6u * static_cast<uint32_t>(ra.max_depth)) + med;
            uint32_t tail_words = tail;
            if (const char* e = getenv("GBL_STREAM_TAIL"))   // tests: force the medium phase's chunked walk
                if (sc.volume.on) tail_words = std::max<uint32_t>(med, static_cast<uint32_t>(strtoul(e, nullptr, 10)));
            ra.stream_tail_cap = L.S * tail_words;
            gbl_status sst = ensure_stream_buffers(ctx, stream_scratch_words(L, tail_words), grid64, npix * ra.spp, &ra);
            if (sst != GBL_OK) return sst;
        } else if (p->integrator == GBL_INTEGRATOR_PATH) {
            kernel = gbl_kernel_path(replay, want_stats, ext || want_stats, p->exact_ties != 0);
        } else {
            kernel = gbl_kernel_ao(replay, want_stats, ext || want_stats, p->exact_ties != 0);
        }
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds)));
        // Keep the per-sample radiance (16 B each) and filter it into the film with the register-accumulating
        // splat kernel afterwards, unless that buffer would not fit the budget below (then the kernel splats
        // through its LDS tile as it goes).  64.6 -> ~53 ms on the 68 M-path frame.
        bool defer = false;
        {
            const uint64_t entries = static_cast<uint64_t>(ra.window[1] - ra.window[0]) * (ra.window[3] - ra.window[2]) * ra.spp;
            if (ra.li_out) {
                ra.li_defer = ra.li_out;
                defer = true;
            } else if (entries * 16 <= li_budget_bytes(ctx) && entries < (1ull << 32)) {
                gbl_status lst = ensure_li(ctx, entries);
                if (lst != GBL_OK) return lst;
                ra.li_defer = reinterpret_cast<float*>(ctx->wf_li);
                defer = true;
            }
        }
        if (sc.volume.on != 0u && !defer) {
            ctx->error = "a scene with a participating medium needs the per-sample radiance buffer: render this window in smaller pieces";
            return GBL_ERR_UNSUPPORTED;
        }
        // kernels/quadtrace.h: sparse interior steps run four lanes per ray; per-sample radiance only, the quads' records take
        // the LDS film tile's place.  Its LDS need differs from the film-tile formula checked above: checked again here, and a
        // scene whose stacks only fit the one-ray-per-lane kernel keeps that one.  The lean kernels of the native sampler only
        // (with or without exact_ties): the EXT builds are slower under it, replay and instrumented renders are not timed.
        if (defer && !ext && (stream_mode || !replay) && !want_stats && quad_wanted()) {
            // (stream mode: the shuffles' LDS region, at least the stacks', and the generator's state come on top)
            const size_t stack_bytes = stream_mode ? static_cast<size_t>(ra.stream_lperm_words) * sizeof(uint32_t)
                                                   : static_cast<size_t>(sc.stack_entries) * GBL_BLOCK * sizeof(uint32_t);
            size_t lds_quad = (gbl_quad_lds_words() + 4 + (stream_mode ? GBL_STREAM_LDS_WORDS : 0)) * sizeof(uint32_t) + stack_bytes;
            // the top of the tree in LDS (trace.h HotLdsStack): as many nodes of the breadth-first prefix as fit into what the
            // stacks leave of the LDS share of the workgroups per CU they allow anyway (granules of 1280 bytes, 128 per CU)
            ra.hot_count = 0;
            ra.hot_word = static_cast<uint32_t>(lds_quad / sizeof(uint32_t));
            if (!stream_mode && sc.hot_nodes > 0 && lds_quad <= 160 * 1024) {
                const size_t gran = 1280, granules = (lds_quad + gran - 1) / gran;
                const size_t wgs = std::max<size_t>(1, std::min<size_t>(GBL_PT_WAVES * 4 * 64 / GBL_BLOCK, 128 / granules));
                size_t room = (128 / wgs) * gran - lds_quad;
                if (const char* e = getenv("GBL_HOT_LDS")) room = static_cast<size_t>(std::max(0, atoi(e))) * sizeof(DevNode);   // measurement aid: any size
                ra.hot_count = static_cast<uint32_t>(std::min<size_t>(sc.hot_nodes, room / sizeof(DevNode)));
                if (lds_quad + ra.hot_count * sizeof(DevNode) > 160 * 1024) ra.hot_count = 0;
                lds_quad += ra.hot_count * sizeof(DevNode);
            }
            gbl_render_kernel k_quad = stream_mode ? (p->integrator == GBL_INTEGRATOR_PATH ? gbl_kernel_path_stream_quad() : nullptr)
                                       : p->integrator == GBL_INTEGRATOR_AO ? gbl_kernel_ao_quad(p->exact_ties != 0)
                                       : (p->integrator == GBL_INTEGRATOR_PATH ? gbl_kernel_path_quad(p->exact_ties != 0) : nullptr);
            if (k_quad && lds_quad <= 160 * 1024) {
                kernel = k_quad;
                lds = lds_quad;
                if (lds > 64 * 1024)
                    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
                // The primary pass (kernels/packet.h): the camera rays of the call traced as packets, one wave per pixel and 64 of its
                // samples, ahead of the path kernel, which then starts every path at its first hit (a camera ray whose answer depends
                // on the visiting order -- an exact tie; under exact_ties also a hit the reference might not reach -- is flagged and
                // traced by the path kernel itself, so the radiance is bit for bit what it is without the pass).  Native sampler's quad
                // path kernels; 20 bytes per camera sample, within the per-sample radiance buffer's budget; GBL_PRIMARY=0 turns it
                // off (A/B, bit-identity test).
                const char* pe = getenv("GBL_PRIMARY");
                const uint64_t entries = npix * ra.spp;
                if (!stream_mode && p->integrator == GBL_INTEGRATOR_PATH && sc.num_lights > 0 && !(pe && pe[0] == '0') &&
                    sc.stack_entries <= 64 && entries * 20 <= li_budget_bytes(ctx)) {
                    if (entries > ctx->prim_entries) {
                        if (ctx->prim_buf) (void)hipFree(ctx->prim_buf);
                        ctx->prim_buf = nullptr;
                        ctx->prim_entries = 0;
                        const hipError_t pe2 = hipMalloc(&ctx->prim_buf, entries * 20);
                        if (pe2 != hipSuccess) {
                            ctx->error = std::string("hipMalloc(primary hits): ") + hipGetErrorString(pe2);
                            return GBL_ERR_OOM;
                        }
                        ctx->prim_entries = entries;
                    }
                    {   // one word per work item of the path kernel
                        const uint64_t items = static_cast<uint64_t>(ra.local_tiles) * ra.chunks;
                        if (items > ctx->prim_items_cap) {
                            if (ctx->prim_items) (void)hipFree(ctx->prim_items);
                            ctx->prim_items = nullptr;
                            ctx->prim_items_cap = 0;
                            const hipError_t ie = hipMalloc(reinterpret_cast<void**>(&ctx->prim_items), items * sizeof(uint32_t));
                            if (ie != hipSuccess) {
                                ctx->error = std::string("hipMalloc(primary items): ") + hipGetErrorString(ie);
                                return GBL_ERR_OOM;
                            }
                            ctx->prim_items_cap = items;
                        }
                        HIP_TRY(ctx, hipMemsetAsync(ctx->prim_items, 0, items * sizeof(uint32_t), stream));
                        ra.prim_items = ctx->prim_items;
                    }
                    float4* ph = static_cast<float4*>(ctx->prim_buf);
                    int32_t* pi = reinterpret_cast<int32_t*>(ph + entries);
                    unsigned prim_wgs = 64u;   // workgroups per CU of the pass's grid-stride launch: 5 / 8 / 16 / 32 / 64 / 128 / 2048 -> 2.92 / 2.76 / 2.54 / 2.45 / 2.43 / 2.42 / 2.53 ms on configs[1]
                    if (const char* e = getenv("GBL_PRIMARY_WGS")) prim_wgs = static_cast<unsigned>(std::min(4096, std::max(1, atoi(e))));   // measurement aid
                    gbl_launch_primary(sc, ra, p->exact_ties != 0, ph, pi, static_cast<unsigned>(ctx->num_cus) * prim_wgs, stream);
                    HIP_TRY(ctx, hipGetLastError());
                    ra.prim_hit = reinterpret_cast<const float*>(ph);
                    ra.prim_inst = pi;
                    kernel = gbl_kernel_path_quad_primary(p->exact_ties != 0);   // the same kernel, its paths starting at those hits
                    if (lds > 64 * 1024)
                        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
                }
            }
        }
        if (stream_mode && !defer) {
            ctx->error = "GBL_SAMPLES_STREAM keeps 16 bytes per camera sample of the call: render this window in smaller pieces";
            return GBL_ERR_UNSUPPORTED;
        }
        {
            // the persistent grid is what is resident: registers may allow fewer workgroups per CU than LDS does (EXT builds)
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kernel), GBL_BLOCK, lds) == hipSuccess && occ >= 1) {
                grid64 = std::max<uint64_t>(1, std::min<uint64_t>(grid64, static_cast<uint64_t>(ctx->num_cus) * occ));
                if (!stream_mode) grid = dim3(static_cast<unsigned>(grid64));   // (stream mode sized its scratch for the original grid)
            }
        }
        const bool phase_clock = getenv("GBL_PHASE_CLOCK") != nullptr && !want_stats;   // measurement builds (-DGBL_PHASE_CLOCK, tools/phase_clock.py)
        if (phase_clock) HIP_TRY(ctx, hipMemsetAsync(ctx->stats, 0, 32 * sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, sc, ra);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(tev[1], stream));
        if (phase_clock) {
            unsigned long long h[32];
            HIP_TRY(ctx, hipStreamSynchronize(stream));
            HIP_TRY(ctx, hipMemcpy(h, ctx->stats, sizeof(h), hipMemcpyDeviceToHost));
            if (h[25] + h[26] + h[27] + h[28] + h[29]) {   // -DGBL_STREAM_TM: the stream sampler's phases in an un-instrumented build
                const double tot = static_cast<double>(h[25] + h[26] + h[27] + h[28] + h[29]);
                fprintf(stderr, "stream phases (share of the workgroups' time): emit %.1f%% permute %.1f%% assemble %.1f%% paths %.1f%% skip %.1f%%\n",
                        100 * h[25] / tot, 100 * h[26] / tot, 100 * h[27] / tot, 100 * h[28] / tot, 100 * h[29] / tot);
            }
            if (h[8]) {
                const double k = static_cast<double>(h[8]);
                fprintf(stderr, "phase clock (share of the waves' ticks): closest-hit query %.1f%% = dense %.1f%% + migrate %.1f%% + quad %.1f%% | any-hit query %.1f%% = dense "
                        "%.1f%% + migrate %.1f%% + quad %.1f%% | rest (shading, regeneration, item fetch) %.1f%% | dense iterations %llu, quad iterations %llu, "
                        "wave ticks %llu\n", 100 * h[0] / k, 100 * h[1] / k, 100 * h[2] / k, 100 * h[3] / k, 100 * h[4] / k, 100 * h[5] / k, 100 * h[6] / k, 100 * h[7] / k,
                        100 * (k - h[0] - h[4]) / k, h[9], h[10], h[8]);
                fprintf(stderr, "phase clock, dense loop: interior blocks %llu (%.0f ticks, %.1f lanes each, %.1f%% of the kernel), leaf / instance blocks %llu (%.0f ticks, %.1f lanes, %.1f%%)\n",
                        h[13], h[13] ? double(h[11]) / h[13] : 0.0, h[13] ? double(h[15]) / h[13] : 0.0, 100 * h[11] / k, h[14], h[14] ? double(h[12]) / h[14] : 0.0,
                        h[14] ? double(h[16]) / h[14] : 0.0, 100 * h[12] / k);
                fprintf(stderr, "phase clock, quad loop: interior iterations %llu (%.0f ticks each, %.1f%%), leaf %llu (%.0f ticks, %.1f%%), transitions / exit %llu (%.0f ticks, %.1f%%); %.2f rays per iteration\n",
                        h[20], h[20] ? double(h[17]) / h[20] : 0.0, 100 * h[17] / k, h[21], h[21] ? double(h[18]) / h[21] : 0.0, 100 * h[18] / k, h[22],
                        h[22] ? double(h[19]) / h[22] : 0.0, 100 * h[19] / k, h[10] ? double(h[23]) / h[10] : 0.0);
            }
        }
        if (defer) {
            WfArgs wa;
            memset(&wa, 0, sizeof(wa));
            wa.li_buf = reinterpret_cast<float4*>(ra.li_defer);
            wa.pass_k0 = 0;
            wa.pass_spp = ra.spp;
            gbl_wf_kernel k_splat = gbl_kernel_wf_splat(replay, want_stats);
            const size_t lds_tile = sizeof(float) * (4 * tp * tp + 256);
            hipLaunchKernelGGL(k_splat, dim3(ra.local_tiles), block, lds_tile, stream, sc, ra, wa);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    if (ra.vol && ra.li_out) {   // the caller's per-sample output carries what the tile received: tr * Li + Lv
        const uint64_t n_li = npix * ra.spp;
        gbl_launch_vol_combine(reinterpret_cast<float4*>(ra.li_out), reinterpret_cast<const float4*>(ra.vol), n_li, stream);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(tev[2], stream));
    ctx->t_calls += 1;
    if (stats) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        memset(stats, 0, sizeof(*stats));
        stats->kernel_ms = ms;
        stats->schedule = wavefront ? GBL_SCHEDULE_WAVEFRONT : GBL_SCHEDULE_MEGAKERNEL;
        uint64_t shard_pixels = 0;
        for (int t = ra.shard_index; t < total_tiles; t += ra.shard_count) {
            int tx = t % ra.tiles_x, ty = t / ra.tiles_x;
            int tw = std::min(GBL_TILE, ra.window[1] - (ra.window[0] + GBL_TILE * tx));
            int th = std::min(GBL_TILE, ra.window[3] - (ra.window[2] + GBL_TILE * ty));
            shard_pixels += static_cast<uint64_t>(tw) * th;
        }
        stats->paths = shard_pixels * ra.spp;
        if (want_stats) {
            unsigned long long h[32];
            HIP_TRY(ctx, hipMemcpy(h, ctx->stats, sizeof(h), hipMemcpyDeviceToHost));
            if (getenv("GBL_PROBE"))
                fprintf(stderr, "probe: interior lane-steps %llu wave-steps %llu (util %.3f) | leaf/other lane %llu wave %llu (util %.3f)\n", h[7], h[8],
                        h[8] ? h[7] / (64.0 * h[8]) : 0.0, h[9], h[10], h[10] ? h[9] / (64.0 * h[10]) : 0.0);
            if (getenv("GBL_PROBE_RAW")) {
                fprintf(stderr, "probe raw: hist");
                for (int i = 0; i < 7; ++i) fprintf(stderr, " %llu", h[11 + i]);
                fprintf(stderr, " | hist_steps");
                for (int i = 0; i < 7; ++i) fprintf(stderr, " %llu", h[18 + i]);
                fprintf(stderr, "\n");
            }
            if (getenv("GBL_PROBE") && h[25] + h[26] + h[27] + h[28] + h[29]) {
                const double tot = static_cast<double>(h[25] + h[26] + h[27] + h[28] + h[29]);
                fprintf(stderr, "probe: stream sampler phases (share of the workgroups' time): emit %.1f%% permute %.1f%% assemble %.1f%% paths %.1f%% skip %.1f%%\n",
                        100.0 * h[25] / tot, 100.0 * h[26] / tot, 100.0 * h[27] / tot, 100.0 * h[28] / tot, 100.0 * h[29] / tot);
            }
            if (getenv("GBL_PROBE") && h[11] + h[12] + h[13] + h[14] + h[15] + h[16] + h[17]) {
                const char* names[7] = {"<=3", "4-7", "8-15", "16-31", "32-63", "64-127", ">=128"};
                unsigned long long rays = 0, steps = 0;
                for (int i = 0; i < 7; ++i) {
                    rays += h[11 + i];
                    steps += h[18 + i];
                }
                fprintf(stderr, "probe: closest-hit rays by interior steps (share of rays / share of steps):");
                for (int i = 0; i < 7; ++i)
                    fprintf(stderr, " %s %.1f%%/%.1f%%", names[i], 100.0 * h[11 + i] / rays, 100.0 * h[18 + i] / std::max(1ull, steps));
                fprintf(stderr, "\n");
            }
            // (the AO kernel of the stream sampler has no instrumented build, gbl_kernel_ao_stream: its launch leaves the device
            //  counters at zero -- report the path count computed above and no ray counters rather than zeros for both)
            const bool main_instrumented = !(stream_mode && p->integrator == GBL_INTEGRATOR_AO);
            if (main_instrumented) stats->paths = h[0];
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
gbl_status gbl_render(gbl_ctx* ctx, const gbl_render_params* p, float* film_accum, gbl_stats* stats) {
    return gbl_guard([&] { return gbl_render_impl(ctx, p, film_accum, stats); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

int gbl_get_timings(gbl_ctx* ctx, int n, gbl_timing* out) {
    if (!ctx || !out || n <= 0) return 0;
    int have = static_cast<int>(std::min<unsigned long long>(ctx->t_calls, gbl_ctx::kTimingRing));
    n = std::min(n, have);
    for (int i = 0; i < n; ++i) {
        hipEvent_t* ev = ctx->t_ev[(ctx->t_calls - 1 - i) % gbl_ctx::kTimingRing];
        float a = 0.0f, b = 0.0f;
        if (hipEventSynchronize(ev[2]) != hipSuccess || hipEventElapsedTime(&a, ev[0], ev[1]) != hipSuccess ||
            hipEventElapsedTime(&b, ev[0], ev[2]) != hipSuccess)
            return i;
        out[i].main_kernel_ms = a;
        out[i].total_ms = b;
    }
    return n;
}

static gbl_status gbl_film_resolve_impl(gbl_ctx* ctx, const float* film_accum, float* rgb_out, void* stream) {
    if (!ctx || !film_accum || !rgb_out) return GBL_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int n = ctx->info.xres * ctx->info.yres;
    gbl_launch_film_resolve(film_accum, rgb_out, n, static_cast<hipStream_t>(stream));
    HIP_TRY(ctx, hipGetLastError());
    return GBL_OK;
}
gbl_status gbl_film_resolve(gbl_ctx* ctx, const float* film_accum, float* rgb_out, void* stream) {
    return gbl_guard([&] { return gbl_film_resolve_impl(ctx, film_accum, rgb_out, stream); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

// ncclAllReduce(sum, float) over the film, resolved from librccl at first use so
// single-GPU users never load RCCL.
static gbl_status gbl_film_allreduce_impl(gbl_ctx* ctx, void* rccl_comm, float* film_accum, void* stream) {
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
gbl_status gbl_film_allreduce(gbl_ctx* ctx, void* rccl_comm, float* film_accum, void* stream) {
    return gbl_guard([&] { return gbl_film_allreduce_impl(ctx, rccl_comm, film_accum, stream); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

}  // extern "C"
