// vol_kernel: the participating medium's {tr, Lv} of every camera sample as a pass of its own (see medium.h).
#pragma once
#include "medium.h"
#include "render_kernels.h"

// One lane per camera sample: the medium's {tr, Lv} into ra.vol[2 * out_index ..] (native / replay samplers).
template <bool REPLAY, bool STATS>
__global__ __launch_bounds__(GBL_BLOCK) void vol_kernel(DevScene sc, RenderArgs ra) {
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsStack stk = {gbl_as_lds(reinterpret_cast<uint32_t*>(smem) + threadIdx.x)};
    LaneCounters cnt = {};
    const uint64_t per_tile = 64ull * static_cast<uint64_t>(ra.spp);
    const uint64_t total = static_cast<uint64_t>(ra.local_tiles) * per_tile;
    const int sub_w = ra.window[1] - ra.window[0];
    const int full_w = sc.film.window[1] - sc.film.window[0];
    for (uint64_t id = static_cast<uint64_t>(blockIdx.x) * GBL_BLOCK + threadIdx.x; id < total; id += static_cast<uint64_t>(gridDim.x) * GBL_BLOCK) {
        const uint32_t lt = static_cast<uint32_t>(id / per_tile), r = static_cast<uint32_t>(id % per_tile);
        const uint32_t pix = r / static_cast<uint32_t>(ra.spp), k = r % static_cast<uint32_t>(ra.spp);
        const uint32_t tile = ra.shard_index + lt * ra.shard_count;
        const int tx = tile % ra.tiles_x, ty = tile / ra.tiles_x;
        const int px = ra.window[0] + GBL_TILE * tx + static_cast<int>(pix % 8u), py = ra.window[2] + GBL_TILE * ty + static_cast<int>(pix / 8u);
        if (px >= ra.window[1] || py >= ra.window[3]) continue;
        const uint32_t out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + k;
        SampleSource src;
        src.spp = ra.spp;
        src.root = ra.root;
        src.rec = nullptr;
        src.pixel_key = 0;
        src.k = k;
        float image_x, image_y, lens_u1 = 0.0f, lens_u2 = 0.0f;
        if (REPLAY) {
            src.rec = ra.replay + static_cast<size_t>(out_index) * ra.dims;
            image_x = src.rec[0];
            image_y = src.rec[1];
            lens_u1 = src.rec[2];
            lens_u2 = src.rec[3];
        } else {
            const uint32_t pixel = static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0]));
            src.pixel_key = nat_mix(ra.seed_key, pixel);
            float u, v;
            src.native_2d(0u, 1u, 0u, false, &u, &v);
            image_x = px + u;
            image_y = py + v;
            if (sc.camera.lens_radius != 0.0f) src.native_2d(1u, 1u, 0u, true, &lens_u1, &lens_u2);
        }
        F3 o, d;
        float mint;
        camera_ray<true>(sc.camera, image_x, image_y, lens_u1, lens_u2, &o, &d, &mint);
        // the camera ray as Li leaves it: clipped at the first surface -- unless Li returned before its first query
        // (PathTracer::Li without lights, GoblinPathtracer.cpp:53-56)
        float maxt = INFINITY;
        if (!(ra.integrator == 0 && sc.num_lights == 0)) {
            Hit hit;
            LaneCounters scratch = {};   // the integrator's own query: not counted twice
            if (trace<false, false, true>(sc, o, d, mint, INFINITY, stk, hit, scratch)) maxt = hit.t;
        }
        VolRand rnd = vol_rand_hashed(image_x, image_y);
        const F3 tr = vol_transmittance(sc, o, d, mint, maxt, rnd);   // (Renderer::transmittance comes first: the heterogeneous region's jitter is the sample's first draw)
        const F3 Lv = volume_lv<STATS>(sc, o, d, mint, maxt, rnd, stk, cnt);
        float4* q = reinterpret_cast<float4*>(ra.vol) + 2 * static_cast<size_t>(out_index);
        q[0] = make_float4(tr.x, tr.y, tr.z, 0.0f);
        q[1] = make_float4(Lv.x, Lv.y, Lv.z, 0.0f);
    }
    if (STATS) accumulate_stats(ra, cnt, 0);
}
