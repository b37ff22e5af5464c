// sss_kernel: Renderer::Lsubsurface of every camera sample as a pass ahead of the path kernels (see bssrdf.h).
#pragma once
#include "bssrdf.h"
#include "render_kernels.h"

// One lane per camera sample of the launch's tiles (ids enumerate owned tile, pixel in tile, sample: consecutive lanes
// are samples of one pixel, so the primary rays of a wave are coherent).  out[pixel * spp + k], pixel-major over the
// render window like li_out.
#ifndef GBL_SSS_WAVES
#define GBL_SSS_WAVES 2   // 256 registers (left alone: 301 with 45 accumulation registers, one wave per SIMD)
#endif
template <bool REPLAY, bool STATS>
__global__ __launch_bounds__(GBL_BLOCK, GBL_SSS_WAVES) void sss_kernel(DevScene sc, RenderArgs ra, float4* out) {
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
        F3 L = f3(0, 0, 0);
        Hit hit;
        LaneCounters scratch = {};   // the camera ray is the path kernel's query; it is not counted twice
        if (sc.num_lights > 0 && trace<false, false, true>(sc, o, d, mint, INFINITY, stk, hit, scratch)) {
            const int material = sc.instances[hit.inst].material;
            if (sc.materials[material].type == GBL_MAT_SUBSURFACE) {
                Frag fr;
                TexFrag tf;
                make_fragment<true>(sc, hit, o, d, fr, &tf);
                DevMaterial mo;
                sss_resolve(sc, material, fr, tf, mo);   // Lsubsurface runs before computeUVDifferential (GoblinPathtracer.cpp:69, :77)
                const F3 wo = -d;
                const F3 single = l_bssrdf_single<REPLAY, STATS>(sc, ra, src, fr, mo, material, wo, stk, cnt);
                const F3 multi = l_bssrdf_diffusion<REPLAY, STATS>(sc, ra, src, fr, tf, mo, material, wo, stk, cnt);
                L = single + multi;
            }
        }
        out[out_index] = make_float4(L.x, L.y, L.z, 0.0f);
    }
    if (STATS) accumulate_stats(ra, cnt, 0);
}
