// libgoblin_hip.so, kernel unit: the persistent megakernel whose sparse interior steps put four lanes on each ray
// (kernels/quadtrace.h), under the native and replay samplers.
#include "gbl_internal.h"
#include "kernels/render_kernels.h"
#include "kernels/packet.h"

// (the lean kernels of the native sampler only: the EXT builds are slower under the quad queries, the replay / instrumented builds
//  run one ray per lane -- gbl_api.hip.  exact_ties: the reference's tie rule and reachability test kept, trace.h TIES)
gbl_render_kernel gbl_kernel_path_quad(bool exact_ties) {
    return exact_ties ? path_trace_kernel<GBL_SRC_NATIVE, false, false, true, true> : path_trace_kernel<GBL_SRC_NATIVE, false, false, true>;
}
gbl_render_kernel gbl_kernel_path_quad_primary(bool exact_ties) {
    return exact_ties ? path_trace_kernel<GBL_SRC_NATIVE, false, false, true, true, true> : path_trace_kernel<GBL_SRC_NATIVE, false, false, true, false, true>;
}
gbl_render_kernel gbl_kernel_ao_quad(bool exact_ties) {
    return exact_ties ? ao_kernel<GBL_SRC_NATIVE, false, false, true, true> : ao_kernel<GBL_SRC_NATIVE, false, false, true>;
}
uint32_t gbl_quad_lds_words(void) { return GBL_QUAD_LDS_WORDS; }

// The primary pass: every camera ray of the call, one wave per (pixel, 64 samples of it), traced as a packet (kernels/packet.h) --
// the 64 rays start at the camera and pass through one pixel, so they meet the same nodes in the same order.  Writes the hit of
// sample `out_index` (the path kernel's own numbering: pixel-major over the render window, spp samples per pixel) to
// RenderArgs::prim_hit / prim_inst; the lean quad path kernel then starts every path at its first hit instead of tracing the camera
// ray among the scattered ones.  Camera samples are made exactly as the path kernel's regeneration makes them (native sampler).
// Replaces, for the first segment of every path, Scene::intersect as PathTracer::Li calls it (GoblinPathtracer.cpp:58-60).
// EXACT (gbl_render_params.exact_ties): a hit the reference's own traversal might not have returned -- one its box tests pass by, or
// one of a tie (trace.h trace_needs_redo, the same end-of-query check the path kernel's queries make) -- is left to the path kernel,
// which traces that camera ray under the reference's rule.
template <bool EXACT>
__global__ __launch_bounds__(GBL_BLOCK) void primary_kernel(DevScene sc, RenderArgs ra, float4* prim_hit, int32_t* prim_inst) {
    __shared__ uint32_t pk_stack[(GBL_BLOCK / 64) * GBL_PACKET_STACK_WORDS];
    gbl_lds_u32* const wstack = gbl_as_lds(pk_stack + (threadIdx.x >> 6) * GBL_PACKET_STACK_WORDS);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunks = (static_cast<uint32_t>(ra.spp) + 63u) / 64u;
    const uint64_t n_tasks = static_cast<uint64_t>(ra.local_tiles) * 64u * chunks;   // (owned tile, pixel of the tile, 64-sample chunk)
    const uint64_t waves = static_cast<uint64_t>(gridDim.x) * (GBL_BLOCK / 64);
    const int sub_w = ra.window[1] - ra.window[0];
    const int full_w = sc.film.window[1] - sc.film.window[0];
    for (uint64_t task = static_cast<uint64_t>(blockIdx.x) * (GBL_BLOCK / 64) + (threadIdx.x >> 6); task < n_tasks; task += waves) {
        const uint32_t c = static_cast<uint32_t>(task % chunks);
        const uint64_t pt = task / chunks;
        const uint32_t pix = static_cast<uint32_t>(pt % 64u), lt = static_cast<uint32_t>(pt / 64u);
        const uint32_t tile = ra.shard_index + lt * ra.shard_count;
        const int tx = tile % ra.tiles_x, ty = tile / ra.tiles_x;
        const int px = ra.window[0] + GBL_TILE * tx + static_cast<int>(pix & 7u), py = ra.window[2] + GBL_TILE * ty + static_cast<int>(pix >> 3);
        if (px >= ra.window[1] || py >= ra.window[3]) continue;   // (wave-uniform: an edge tile's clipped pixels)
        const uint32_t k = c * 64u + lane;
        const bool live = k < static_cast<uint32_t>(ra.spp);
        SampleSource src;
        src.spp = ra.spp;
        src.root = ra.root;
        src.rec = nullptr;
        src.k = live ? k : 0u;
        src.pixel_key = nat_mix(ra.seed_key, static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0])));
        float u, v;
        src.native_2d(0u, 1u, 0u, false, &u, &v);
        F3 o, d;
        float mint;
        camera_ray<false>(sc.camera, px + u, py + v, 0.0f, 0.0f, &o, &d, &mint);
        Hit h;
        bool tied;
        packet_closest(sc, live, o, d, mint, wstack, h, tied);
        if (EXACT && live) tied = trace_needs_redo(sc, false, h.inst >= 0, h, tied, o, d, mint, INFINITY);
        if (live) {
            const size_t out_index = static_cast<size_t>(static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0]))) * ra.spp + k;
            if (h.inst >= 0 && !tied) prim_hit[out_index] = make_float4(h.t, h.b1, h.b2, __uint_as_float(h.tri));
            prim_inst[out_index] = tied ? GBL_PRIM_TIED : h.inst;   // (a miss: GBL_PRIM_MISS)
            if (h.inst < 0 && !tied) {
                // PathTracer::Li of a camera ray that left the scene: Black (:58-66, no image based light in the lean builds) ...
                reinterpret_cast<float4*>(ra.li_defer)[out_index] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            } else {
                // ... and the path kernel's work item (owned tile x chunk of samples) of every other sample has something to do
                ra.prim_items[lt * ra.chunks + k / static_cast<uint32_t>(ra.chunk_spp)] = 1u;
            }
        }
    }
}
void gbl_launch_primary(const DevScene& sc, const RenderArgs& ra, bool exact_ties, float4* prim_hit, int32_t* prim_inst, unsigned blocks, hipStream_t stream) {
    if (exact_ties)
        hipLaunchKernelGGL(primary_kernel<true>, dim3(blocks), dim3(GBL_BLOCK), 0, stream, sc, ra, prim_hit, prim_inst);
    else
        hipLaunchKernelGGL(primary_kernel<false>, dim3(blocks), dim3(GBL_BLOCK), 0, stream, sc, ra, prim_hit, prim_inst);
}
