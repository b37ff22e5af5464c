// The render kernels: persistent workgroups pull (tile, sample-chunk) work items
// from a global counter; inside an item the lanes of a wave keep themselves busy
// by regenerating camera paths from an LDS counter as soon as their previous
// path ends (wave ballot + prefix popcount).  Radiance is splatted through the
// reconstruction filter into an LDS film tile (8x8 pixels + halo) with LDS float
// atomics and flushed to the HBM film once per item.
//
// Replaces RenderTask::run + PathTracer::Li / AORenderer::Li + ImageTile::addSample
// (GoblinRenderer.cpp:29-52, GoblinPathtracer.cpp:50-179, GoblinAO.cpp:12-37,
// GoblinFilm.cpp:61-90) and PerspectiveCamera::generateRay (GoblinCamera.cpp:97-148).
//
// Per bounce the reference issues up to five BVH traversals (shadow, two
// notOpaque attenuation walks that cannot hit anything in a mask-free scene, the
// MIS ray and the identical extension ray).  Here a bounce is exactly one
// any-hit query and one closest-hit query: the MIS lookup and the path
// extension are the same ray, so its hit is used for both.
#pragma once
#include <type_traits>
#include "../device_scene.h"
#include "sampler.h"
#include "shade.h"
#include "stream.h"
#include "bssrdf.h"
#include "medium.h"
#include "trace.h"
#include "quadtrace.h"
#include "vecmath.h"

struct ItemInfo {
    int px0, py0, tw, th;   // tile origin and clipped size in pixels
    int k0;                 // first sample index of this chunk
    int paths;              // tw * th * chunk_spp
};

__device__ __forceinline__ ItemInfo decode_item(const RenderArgs& ra, uint32_t item) {
    ItemInfo it;
    uint32_t tile = ra.shard_index + (item / ra.chunks) * ra.shard_count, chunk = item % ra.chunks;
    int tx = tile % ra.tiles_x, ty = tile / ra.tiles_x;
    it.px0 = ra.window[0] + GBL_TILE * tx;
    it.py0 = ra.window[2] + GBL_TILE * ty;
    it.tw = min(GBL_TILE, ra.window[1] - it.px0);
    it.th = min(GBL_TILE, ra.window[3] - it.py0);
    it.k0 = chunk * ra.chunk_spp;
    it.paths = it.tw * it.th * ra.chunk_spp;
    return it;
}

// Quaternion * Vector3 (GoblinQuaternion.cpp:86-92)
__device__ __forceinline__ F3 quat_rotate(const float q[4], F3 v) {
    F3 qv = f3(q[1], q[2], q[3]);
    F3 uv = cross(qv, v);
    F3 uuv = cross(qv, uv);
    uv = uv * (2.0f * q[0]);
    uuv = uuv * 2.0f;
    return v + uv + uuv;
}

// Camera::generateRay: PerspectiveCamera pinhole / thin lens (GoblinCamera.cpp:97-148) and, in EXT builds,
// OrthographicCamera (:298-326).  lens_u* are Sample::lensU1/2 (only read by the thin lens).
template <bool EXT>
__device__ __forceinline__ void camera_ray(const DevCamera& c, float image_x, float image_y, float lens_u1, float lens_u2, F3* o, F3* d,
                                           float* mint) {
    float xndc = +2.0f * image_x * c.inv_xres - 1.0f;
    float yndc = -2.0f * image_y * c.inv_yres + 1.0f;
    const F3 pos = f3(c.pos[0], c.pos[1], c.pos[2]);
    *mint = 1e-3f;
    if (EXT && c.type == 1u) {
        float xv = 0.5f * c.film_w * xndc;
        float yv = 0.5f * c.film_h * yndc;
        *o = pos + quat_rotate(c.q, f3(xv, yv, 0.0f));
        *d = quat_rotate(c.q, f3(0.0f, 0.0f, 1.0f));
        *mint = 0.0f;
        return;
    }
    float xv = xndc / c.proj00;
    float yv = yndc / c.proj11;
    F3 view = f3(xv, yv, 1.0f);
    if (EXT && c.lens_radius != 0.0f) {
        float ft = c.focal_distance / view.z;
        F3 p_focus = view * ft;
        float lx, ly;
        uniform_sample_disk(lens_u1, lens_u2, &lx, &ly);
        F3 view_origin = f3(c.lens_radius * lx, c.lens_radius * ly, 0.0f);
        *o = quat_rotate(c.q, view_origin) + pos;
        *d = quat_rotate(c.q, normalize(p_focus - view_origin));
        return;
    }
    *o = pos;
    *d = quat_rotate(c.q, normalize(view));
}

// The camera ray's auxiliary rays (RayDifferential::dx/dy, one pixel to the right / below), GoblinCamera.cpp:97-148,
// 298-326.  Only texture filtering at the primary hit reads them.
__device__ __forceinline__ void camera_differentials(const DevCamera& c, float image_x, float image_y, float lens_u1, float lens_u2,
                                                     F3* dxo, F3* dxd, F3* dyo, F3* dyd) {
    float xndc = +2.0f * image_x * c.inv_xres - 1.0f;
    float yndc = -2.0f * image_y * c.inv_yres + 1.0f;
    float dxndc = +2.0f * (image_x + 1.0f) * c.inv_xres - 1.0f;
    float dyndc = -2.0f * (image_y + 1.0f) * c.inv_yres + 1.0f;
    const F3 pos = f3(c.pos[0], c.pos[1], c.pos[2]);
    if (c.type == 1u) {
        float xv = 0.5f * c.film_w * xndc, yv = 0.5f * c.film_h * yndc;
        float dxv = 0.5f * c.film_w * dxndc, dyv = 0.5f * c.film_h * dyndc;
        *dxo = pos + quat_rotate(c.q, f3(dxv, yv, 0.0f));
        *dyo = pos + quat_rotate(c.q, f3(xv, dyv, 0.0f));
        *dxd = *dyd = quat_rotate(c.q, f3(0.0f, 0.0f, 1.0f));
        return;
    }
    float xv = xndc / c.proj00, yv = yndc / c.proj11;
    F3 dx_view = f3(dxndc / c.proj00, yv, 1.0f), dy_view = f3(xv, dyndc / c.proj11, 1.0f);
    if (c.lens_radius != 0.0f) {
        float ft = c.focal_distance / 1.0f;
        float lx, ly;
        uniform_sample_disk(lens_u1, lens_u2, &lx, &ly);
        F3 vo = f3(c.lens_radius * lx, c.lens_radius * ly, 0.0f);
        *dxo = *dyo = quat_rotate(c.q, vo) + pos;
        *dxd = quat_rotate(c.q, normalize(dx_view * ft - vo));
        *dyd = quat_rotate(c.q, normalize(dy_view * ft - vo));
        return;
    }
    *dxo = *dyo = pos;
    *dxd = quat_rotate(c.q, normalize(dx_view));
    *dyd = quat_rotate(c.q, normalize(dy_view));
}

// TexFrag differentials of a hit: the camera ray's for the primary hit, none afterwards (RayDifferential(p, wi, eps)
// carries no auxiliary rays, GoblinRay.h:48-51).
template <bool REPLAY>
__device__ __forceinline__ void hit_differentials(const DevScene& sc, const SampleSource& src, bool primary, float image_x, float image_y,
                                                  const Frag& fr, TexFrag& tf) {
    F3 dxo = f3(0, 0, 0), dxd = dxo, dyo = dxo, dyd = dxo;
    if (primary) {
        float lens_u1 = 0.0f, lens_u2 = 0.0f;
        if (sc.camera.lens_radius != 0.0f) {
            if (REPLAY) {
                lens_u1 = src.rec[2];
                lens_u2 = src.rec[3];
            } else {
                src.native_2d(1u, 1u, 0u, true, &lens_u1, &lens_u2);
            }
        }
        camera_differentials(sc.camera, image_x, image_y, lens_u1, lens_u2, &dxo, &dxd, &dyo, &dyd);
    }
    uv_differential(fr, tf, primary, dxo, dxd, dyo, dyd);
}

// ImageTile::addSample into the LDS tile.  tile origin (tx0, ty0), row pitch tp pixels.
template <bool STATS>
__device__ __forceinline__ void splat(const DevFilm& film, const float* ftab, float* tile, int tx0, int ty0, int tp, float image_x,
                                      float image_y, F3 L, LaneCounters& cnt) {
    if (L.x != L.x || L.y != L.y || L.z != L.z) return;   // NaN sample: dropped (GoblinFilm.cpp:62-66)
    float dx = image_x - 0.5f, dy = image_y - 0.5f;
    int x0 = static_cast<int>(ceilf(dx - film.wx)), x1 = static_cast<int>(floorf(dx + film.wx));
    int y0 = static_cast<int>(ceilf(dy - film.wy)), y1 = static_cast<int>(floorf(dy + film.wy));
    x0 = max(x0, film.xstart);
    x1 = min(x1, film.xstart + film.xcount - 1);
    y0 = max(y0, film.ystart);
    y1 = min(y1, film.ystart + film.ycount - 1);
    // The footprint of a sample generated inside this tile always lies inside the
    // LDS tile (halo = ceil(w + 0.5)).  A replayed record may carry any image
    // position: clip so a foreign record can never write outside the tile.
    x0 = max(x0, tx0);
    x1 = min(x1, tx0 + tp - 1);
    y0 = max(y0, ty0);
    y1 = min(y1, ty0 + tp - 1);
    for (int y = y0; y <= y1; ++y) {
        int iy = min(static_cast<int>(floorf(fabsf(16 * (y - dy) / film.wy))), 15);
        for (int x = x0; x <= x1; ++x) {
            int ix = min(static_cast<int>(floorf(fabsf(16 * (x - dx) / film.wx))), 15);
            float w = ftab[iy * 16 + ix];
            float* px = tile + 4 * ((y - ty0) * tp + (x - tx0));
            atomicAdd(px + 0, w * L.x);
            atomicAdd(px + 1, w * L.y);
            atomicAdd(px + 2, w * L.z);
            atomicAdd(px + 3, w);
            if (STATS) cnt.splats += 1;
        }
    }
}

__device__ __forceinline__ void flush_tile(const DevFilm& film, float* tile, int tx0, int ty0, int tp, float* out) {
    for (int i = threadIdx.x; i < tp * tp; i += GBL_BLOCK) {
        int x = tx0 + i % tp, y = ty0 + i / tp;
        float4 v = reinterpret_cast<float4*>(tile)[i];
        if (x >= 0 && y >= 0 && x < film.xres && y < film.yres && (v.w != 0.0f || v.x != 0.0f || v.y != 0.0f || v.z != 0.0f)) {
            float* px = out + 4 * (static_cast<size_t>(y) * film.xres + x);
            atomicAdd(px + 0, v.x);
            atomicAdd(px + 1, v.y);
            atomicAdd(px + 2, v.z);
            atomicAdd(px + 3, v.w);
        }
    }
}

// Grab the next path index of this work item for every idle lane of the wave:
// one LDS atomic per wave, prefix popcount for the lane's offset.
// (Round 3 measured a barrier-free alternative for the quad kernels -- two work items in flight per workgroup as 64-bit
//  {item, next path} LDS words, refilled under a per-slot lock while the other slot keeps feeding the waves, no drain and no
//  barrier between items: Cornell 512^2 x 64 spp 69.1 -> 63.4 ms, the grid unchanged, configs[1] 42.1 -> 44.6 ms.  Items that
//  start together keep the four waves of a workgroup on neighbouring pixels of one tile, which the short paths of configs[1]
//  are worth more than their drains; the long-path scenes that gain run the wavefront schedule anyway.  Not kept.
//  Nor were per-CU item lists: blocks b, b + 256, b + 512 of the persistent grid share a CU (read from HW_REG_HW_ID by a self test since removed),
//  so a list per blockIdx % 256 puts a CU's twelve waves on the same tiles -- no gain on the Cornell box, the grid or AO, and
//  configs[1] 42.8 -> 60.5 ms from the static split's imbalance (bunny tiles against background tiles).)
__device__ __forceinline__ int wave_fetch(bool want, uint32_t* next_path) {
    unsigned long long mask = __ballot(want);
    if (mask == 0ull) return -1;
    int lane = threadIdx.x & 63;
    int leader = __ffsll(static_cast<long long>(mask)) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(next_path, static_cast<uint32_t>(__popcll(mask)));
    base = __shfl(base, leader);
    uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    return want ? static_cast<int>(base + rank) : -1;
}

__device__ __forceinline__ void accumulate_stats(const RenderArgs& ra, const LaneCounters& c, uint32_t paths) {
    unsigned long long v[25] = {paths, c.ext, c.shadow, c.nodes, c.tris, c.splats, c.dims, c.int_lane, c.int_wave, c.oth_lane, c.oth_wave};
    for (int i = 0; i < 7; ++i) {
        v[11 + i] = c.hist[i];
        v[18 + i] = c.hist_steps[i];
    }
    for (int i = 0; i < 25; ++i) {
        unsigned long long x = v[i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(ra.stats + i, x);
    }
}

// PathTracer::evalAttenuation (GoblinPathtracer.cpp:21-48): the product of (1 - alpha) * transparentColor over the
// mask surfaces the segment [mint, maxt] of the ray crosses, found one notOpaque closest-hit query at a time.
template <bool STATS, class STK>
__device__ __forceinline__ F3 eval_attenuation(const DevScene& sc, F3 o, F3 d, float mint, float maxt, const STK& stk, LaneCounters& cnt) {
    F3 thr = f3(1.0f, 1.0f, 1.0f);
    for (int guard = 0; guard < 1024; ++guard) {   // every step moves mint past a hit; the cap only bounds a degenerate scene
        Hit h;
        if (!trace<false, STATS, true>(sc, o, d, mint, maxt, stk, h, cnt, GBL_FILTER_MASK)) break;
        Frag fr;
        TexFrag tf;
        make_fragment<true>(sc, h, o, d, fr, &tf);
        uv_differential(fr, tf, false, o, o, o, o);
        const DevMaterial& m = sc.materials[sc.instances[h.inst].material];
        // a subsurface material does not match the BSDFnullptr request (matchType, GoblinMaterial.cpp:733-736): Black
        if (m.type == GBL_MAT_SUBSURFACE) return f3(0.0f, 0.0f, 0.0f);
        float alpha = m.tex_exponent >= 0 ? tex_eval<GBL_TEX_MAX_DEPTH>(sc, m.tex_exponent, fr, tf).x : m.exponent;
        F3 tcolor = m.tex_color >= 0 ? tex_eval<GBL_TEX_MAX_DEPTH>(sc, m.tex_color, fr, tf) : f3(m.color[0], m.color[1], m.color[2]);
        thr = thr * ((1.0f - alpha) * tcolor);
        if (is_black(thr)) break;
        mint = h.t + fr.eps;   // currentRay.mint = currentRay.maxt + epsilon
    }
    return thr;
}

// ---------------------------------------------------------------------------
// Path state carried by a lane between iterations of the persistent loop.
// ---------------------------------------------------------------------------
struct PathState {
    F3 o, d;            // ray to trace next (primary or extension)
    float mint;
    F3 throughput, Li;
    // deferred terms of the bounce whose extension ray is in flight
    F3 Ld, f;           // direct light gathered so far; bsdf value of the sampled direction
    float cosw, fw, bsdf_pdf, pick_pdf;   // |wi.n|, MIS weight, pdfs
    int light;          // light picked at the previous vertex
    int bounce;         // -1: the ray in flight is the camera ray
    uint32_t path;      // index of this path inside the work item
    bool punch;         // EXT: the ray in flight left a mask surface through its alpha (sampledType == BSDFnullptr)
    bool first;         // EXT: PathTracer::Li's firstBounce -- no real bounce yet, only punch-throughs (GoblinPathtracer.cpp:75,176)
};

// GBL_SAMPLES_STREAM with a participating medium.  RenderTask::run continues each sample with transmittance(ray) and
// Lv(ray, rng) (GoblinRenderer.cpp:43-46), so the medium's draws sit in the tile's stream right after that sample's Li
// draws.  Per sample of the pixel the integrator leaves {n: the floats its Li discarded, t: the camera ray's maxt as Li
// left it}; this phase counts the medium's draws (9 per light sample when the clipped camera ray crosses the region, the
// pick alone without lights), walks the samples in stream order in chunks that fit the scratch -- skipping the first
// sample's Li draws, emitting the rest of the chunk, letting every sample read its own slice -- and folds
// 1 * (tr * Li + Lv) into li[].  Ends with the stream behind the pixel's last draw.  Workgroup-uniform.
struct StreamVol {
    uint32_t* n;     // [S]
    float* t;        // [S]
    uint32_t* off;   // [S]
    uint32_t* raw;   // [RenderArgs::stream_tail_cap]
};
__device__ inline StreamVol stream_vol_scratch(const StreamCtx& scx, const StreamLayout& slay) {
    StreamVol v;
    v.n = reinterpret_cast<uint32_t*>(scx.recs + static_cast<size_t>(slay.S) * slay.dims);
    v.t = reinterpret_cast<float*>(v.n + slay.S);
    v.off = v.n + 2 * slay.S;
    v.raw = v.n + 3 * slay.S;
    return v;
}
template <bool STATS, class STK>
__device__ void stream_medium_phase(const DevScene& sc, const RenderArgs& ra, StreamCtx& scx, const StreamLayout& slay, const StreamVol& sv,
                                    uint32_t* ctrl, float4* li, const STK& stk, LaneCounters& cnt) {
    __syncthreads();
    if (sc.volume.hetero != 0u) {
        // A heterogeneous region: how many numbers a sample's transmittance and Lv draw depends on what the march meets
        // (4 per step, a 5th when the light sample is unoccluded), and sample k + 1 starts where sample k stopped -- so the
        // pixel's samples are taken ONE AFTER THE OTHER, by one lane, each reading the stream from where the previous one
        // left it.  Per chunk of samples: remember the generator's state, emit an upper bound of the chunk's draws (a
        // sample's march is at most floor((t1 - t0) / step) + 2 points long), let lane 0 walk the chunk and count what it
        // really drew, then put the generator back and advance it by exactly that.  (Slow -- a tile runs on one lane here --
        // and only ever used where the reference's own Film is wanted.)
        const uint32_t cap = ra.stream_tail_cap - (GBL_MT_N + 16u);
        uint32_t* save = sv.raw + cap;
        for (uint32_t k = threadIdx.x; k < slay.S; k += GBL_BLOCK) {
            const float* rec = scx.recs + static_cast<size_t>(k) * ra.dims;
            F3 o, d;
            float mint;
            camera_ray<true>(sc.camera, rec[0], rec[1], rec[2], rec[3], &o, &d, &mint);
            float t0, t1;
            uint32_t bound = 0u;
            if (vol_intersect(sc.volume, o, d, mint, sv.t[k], &t0, &t1)) {
                bound = 1u;   // transmittance's jitter
                if (!((t1 - t0) < 1e-5f)) bound += 1u + 5u * (static_cast<uint32_t>(fminf(floorf((t1 - t0) / sc.volume.step), 1.0e6f)) + 2u);
            }
            sv.off[k] = bound;
        }
        uint32_t k0 = 0;
        while (k0 < slay.S) {
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t run = sv.off[k0], k1 = k0 + 1;
                while (k1 < slay.S && run + sv.n[k1] + sv.off[k1] <= cap) {
                    run += sv.n[k1] + sv.off[k1];
                    k1 += 1;
                }
                ctrl[1] = sv.n[k0];
                ctrl[2] = min(run, cap);
                ctrl[3] = k1;
            }
            __syncthreads();
            const uint32_t skip = ctrl[1], total = ctrl[2], k1 = ctrl[3];
            __syncthreads();
            stream_emit(scx, nullptr, skip);
            mt_save(scx, save);
            stream_emit(scx, sv.raw, total);
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t pos = 0;
                for (uint32_t k = k0; k < k1; ++k) {
                    if (k > k0) pos += sv.n[k];
                    const float* rec = scx.recs + static_cast<size_t>(k) * ra.dims;
                    F3 o, d;
                    float mint;
                    camera_ray<true>(sc.camera, rec[0], rec[1], rec[2], rec[3], &o, &d, &mint);
                    VolRand rnd;
                    rnd.raw = sv.raw + pos;
                    rnd.key = 0u;
                    rnd.i = 0u;
                    const F3 tr = vol_transmittance(sc, o, d, mint, sv.t[k], rnd);
                    const F3 Lv = volume_lv<STATS>(sc, o, d, mint, sv.t[k], rnd, stk, cnt);
                    const float4 L = li[k];
                    li[k] = make_float4(1.0f * (tr.x * L.x + Lv.x), 1.0f * (tr.y * L.y + Lv.y), 1.0f * (tr.z * L.z + Lv.z), L.w);
                    pos += rnd.i;
                }
                ctrl[2] = pos;
            }
            __syncthreads();
            const uint32_t consumed = ctrl[2];
            __syncthreads();
            mt_restore(scx, save);
            stream_emit(scx, nullptr, consumed);
            k0 = k1;
        }
        __syncthreads();
        return;
    }
    const uint32_t per = static_cast<uint32_t>(max(0, sc.volume.sample_num)) * (sc.num_lights > 0 ? 9u : 1u);
    for (uint32_t k = threadIdx.x; k < slay.S; k += GBL_BLOCK) {
        const float* rec = scx.recs + static_cast<size_t>(k) * ra.dims;
        F3 o, d;
        float mint;
        camera_ray<true>(sc.camera, rec[0], rec[1], rec[2], rec[3], &o, &d, &mint);
        float t0, t1;
        const bool crosses = vol_intersect(sc.volume, o, d, mint, sv.t[k], &t0, &t1) && !((t1 - t0) < 1e-5f);
        sv.off[k] = crosses ? per : 0u;
    }
    uint32_t k0 = 0;
    while (k0 < slay.S) {
        __syncthreads();
        if (threadIdx.x == 0) {
            // samples [k0, k1): the first one's Li draws are skipped, not stored
            uint32_t run = sv.off[k0], k1 = k0 + 1;
            sv.off[k0] = 0u;
            while (k1 < slay.S && run + sv.n[k1] + sv.off[k1] <= ra.stream_tail_cap) {
                const uint32_t mine = sv.off[k1];
                sv.off[k1] = run + sv.n[k1];   // where sample k1's medium draws start
                run += sv.n[k1] + mine;
                k1 += 1;
            }
            ctrl[1] = sv.n[k0];
            ctrl[2] = run;
            ctrl[3] = k1;
        }
        __syncthreads();
        const uint32_t skip = ctrl[1], total = ctrl[2], k1 = ctrl[3];
        __syncthreads();
        stream_emit(scx, nullptr, skip);
        stream_emit(scx, sv.raw, total);
        for (uint32_t k = k0 + threadIdx.x; k < k1; k += GBL_BLOCK) {
            const float* rec = scx.recs + static_cast<size_t>(k) * ra.dims;
            F3 o, d;
            float mint;
            camera_ray<true>(sc.camera, rec[0], rec[1], rec[2], rec[3], &o, &d, &mint);
            VolRand rnd;
            rnd.raw = sv.raw + sv.off[k];
            rnd.key = 0u;
            rnd.i = 0u;
            const F3 tr = vol_transmittance(sc, o, d, mint, sv.t[k], rnd);
            const F3 Lv = volume_lv<STATS>(sc, o, d, mint, sv.t[k], rnd, stk, cnt);
            const float4 L = li[k];
            li[k] = make_float4(1.0f * (tr.x * L.x + Lv.x), 1.0f * (tr.y * L.y + Lv.y), 1.0f * (tr.z * L.z + Lv.z), L.w);
        }
        k0 = k1;
    }
    __syncthreads();
}

// SAMPLER: where a path's Sample record comes from -- GBL_SRC_NATIVE (counter-based law, kernels/sampler.h), GBL_SRC_REPLAY
// (caller's records) or GBL_SRC_STREAM (the reference's own records, generated per pixel from the tile's mt19937,
// kernels/stream.h; a work item is then a whole tile, walked pixel by pixel).
// QUAD: both queries run as wave-wide calls whose sparse interior steps put four lanes on each ray (kernels/quadtrace.h);
// the kernel writes per-sample radiance only (ra.li_defer), so the LDS film tile's place holds the quads' records.
#define GBL_SRC_NATIVE 0
#define GBL_SRC_REPLAY 1
#define GBL_SRC_STREAM 2
// EXACT: the native sampler's lean kernels leave the reference's exact-t tie rule and reachability test out (trace.h: +16 % on
// configs[1] for the few rays per 10^6 they decide); gbl_render_params.exact_ties selects the instantiations that follow them.
// Every other build does anyway: replay, stream, instrumented, and the EXT builds of the feature scenes.
#ifdef GBL_STREAM_TM   // measurement builds: the stream sampler's phase clock in every instantiation (tools/stream_probe.py)
#define GBL_STREAM_TM_ON true
#else
#define GBL_STREAM_TM_ON false
#endif
template <int SAMPLER, bool STATS, bool EXT, bool QUAD = false, bool EXACT = false, bool PRIM = false>
// (EXT builds carry the analytic shapes, texture graphs, image lookups (out-of-line calls), masks, the BSSRDF and medium hooks:
//  held to the lean build's 168 registers they spilled 300-1200 of them; two waves per SIMD (256 registers) hold them)
__global__ __launch_bounds__(GBL_BLOCK, EXT ? GBL_EXT_WAVES : GBL_PT_WAVES) void path_trace_kernel(DevScene sc, RenderArgs ra) {
    constexpr bool REPLAY = SAMPLER != GBL_SRC_NATIVE, STREAM = SAMPLER == GBL_SRC_STREAM, TIES = REPLAY || STATS || EXACT || EXT;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tp = GBL_TILE + 2 * sc.film.halo;
    float* tile = reinterpret_cast<float*>(smem);
    float* ftab = tile + 4 * tp * tp;
    uint32_t* ctrl = QUAD ? reinterpret_cast<uint32_t*>(smem) + GBL_QUAD_LDS_WORDS : reinterpret_cast<uint32_t*>(ftab + 256);
    uint32_t* stack = ctrl + 4 + (STREAM ? GBL_STREAM_LDS_WORDS : 0);
    static_assert(!(QUAD && (EXT || STATS)), "quad-per-ray steps are built for the lean kernels");
    // QUAD: LDS = 16 records per wave | ctrl | stacks
    gbl_lds_u32* const quad_slab = gbl_as_lds(reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * 16 * GBL_QUAD_REC_WORDS);
    gbl_lds_u32* const quad_stack = gbl_as_lds(stack + (threadIdx.x & ~63u));
    typename std::conditional<QUAD, HotLdsStack, LdsStack>::type stk;
    stk.p = gbl_as_lds(stack + threadIdx.x);
    if constexpr (QUAD) {   // the top of the tree, once per workgroup (trace.h HotLdsStack); the item loop's first barrier publishes it
        uint4* hot = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(smem) + ra.hot_word);
        for (uint32_t i = threadIdx.x; i < 4u * ra.hot_count; i += GBL_BLOCK) hot[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
        stk.hot = (const gbl_lds_u4*)hot;
        stk.hot_count = ra.hot_count;
    }
    if (!QUAD)
        for (int i = threadIdx.x; i < 256; i += GBL_BLOCK) ftab[i] = sc.filter_table[i];
    StreamCtx scx = {};
    StreamLayout slay = {};
    if constexpr (STREAM) {
        slay = stream_layout(ra.spp, ra.root, ra.max_depth, ra.bssrdf_n, ra.bssrdf_n2);
        slay.path_quota = 1u;   // only the columns this integrator reads are shuffled and placed (stream.h stream_col_needed)
        slay.used_bounces = static_cast<uint32_t>(max(0, ra.max_depth - 1));
        slay.use_lens = (EXT && sc.camera.lens_radius != 0.0f) ? 1u : 0u;
        slay.use_bssrdf = (EXT && sc.has_bssrdf != 0) ? 1u : 0u;
        scx.mt = ctrl + 4;
        scx.lperm = stack;
        scx.lperm_words = ra.stream_lperm_words;
        scx.raw = ra.stream_scratch + static_cast<size_t>(blockIdx.x) * ra.stream_stride;
        scx.cols = reinterpret_cast<DevStreamCol*>(scx.raw + slay.NF + slay.NU);
        scx.recs = reinterpret_cast<float*>(scx.cols + slay.ncols);
    }
    StreamVol svol = {};   // STREAM with a participating medium (stream_medium_phase)
    if constexpr (STREAM) svol = stream_vol_scratch(scx, slay);
    if constexpr (STREAM) stream_columns(scx, slay);

    LaneCounters cnt = {};
#ifdef GBL_PHASE_CLOCK
    const unsigned long long pc_k0 = __builtin_amdgcn_s_memtime();
#endif
    uint32_t paths_done = 0;
    const uint32_t n_items = static_cast<uint32_t>(ra.local_tiles) * ra.chunks;
    const int sub_w = ra.window[1] - ra.window[0];
    const int full_w = sc.film.window[1] - sc.film.window[0];
    unsigned long long stream_tm[5] = {0ull, 0ull, 0ull, 0ull, 0ull};   // instrumented STREAM builds: phase ticks (emit, permute, assemble, paths, skip)

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            ctrl[0] = atomicAdd(ra.work_counter, 1u);
            ctrl[1] = 0u;
        }
        if (!QUAD && !ra.li_defer)
            for (int i = threadIdx.x; i < 4 * tp * tp; i += GBL_BLOCK) tile[i] = 0.0f;
        __syncthreads();
        const uint32_t item = ctrl[0];
        if (item >= n_items) break;
        if (PRIM && ra.prim_items[item] == 0u) continue;   // (uniform) every camera ray of the item left the scene: the pass has written its Blacks
        const ItemInfo tile_item = decode_item(ra, item);
        const int tx0 = tile_item.px0 - sc.film.halo, ty0 = tile_item.py0 - sc.film.halo;
        if constexpr (STREAM) {
            // RNGImp of this tile's RenderTask: seeded with the tile's rand() value (row-major over the FULL sample window)
            const int ftx = (tile_item.px0 - sc.film.window[0]) / GBL_TILE, fty = (tile_item.py0 - sc.film.window[2]) / GBL_TILE;
            mt_seed(scx, ra.tile_seeds[fty * ra.full_tiles_x + ftx]);
        }
        uint32_t stream_draws = 0;   // STREAM: BSDFSample(rng) floats this lane's paths discarded
        uint32_t path_draws = 0;     // ... and the path in flight alone
        float prim_t = INFINITY;     // the camera ray's maxt as Li leaves it (the medium integrates up to there)
        unsigned long long stream_t0 = 0ull;
        const int n_sub = STREAM ? tile_item.tw * tile_item.th : 1;
        for (int sub = 0; sub < n_sub; ++sub) {
        ItemInfo it = tile_item;
        if constexpr (STREAM) {
            it.px0 = tile_item.px0 + sub % tile_item.tw;
            it.py0 = tile_item.py0 + sub / tile_item.tw;
            it.tw = it.th = 1;
            it.paths = ra.spp;
            stream_generate_pixel(scx, slay, it.px0, it.py0, (STATS || GBL_STREAM_TM_ON) ? stream_tm : nullptr);
            if (threadIdx.x == 0) ctrl[1] = ctrl[2] = 0u;
            stream_draws = 0;
            __syncthreads();
            if (STATS || GBL_STREAM_TM_ON) stream_t0 = wall_clock64();
        }

        PathState ps;
        bool active = false;
        bool exhausted = false;
        SampleSource src;
        src.spp = ra.spp;
        src.root = ra.root;
        src.rec = nullptr;
        src.pixel_key = 0;
        src.k = 0;
        float image_x = 0.0f, image_y = 0.0f;
        uint32_t out_index = 0;

        for (;;) {
            // ---- regeneration: idle lanes start a new camera path
            // (PRIM: below, after the bounce in flight is closed -- a path starts at the hit the primary pass found for it)
            int fetched = PRIM ? -1 : wave_fetch(!active && !exhausted, ctrl + 1);
            if (!PRIM && !active && !exhausted) {
                if (fetched >= 0 && fetched < it.paths) {
                    int pix = fetched / ra.chunk_spp;
                    src.k = static_cast<uint32_t>(it.k0 + fetched % ra.chunk_spp);
                    int px = it.px0 + pix % it.tw, py = it.py0 + pix / it.tw;
                    out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + src.k;
                    if (REPLAY) {
                        src.rec = STREAM ? scx.recs + static_cast<size_t>(src.k) * ra.dims : ra.replay + static_cast<size_t>(out_index) * ra.dims;
                        image_x = src.rec[0];
                        image_y = src.rec[1];
                        if (STREAM) {
                            ra.image_xy[2 * static_cast<size_t>(out_index)] = image_x;
                            ra.image_xy[2 * static_cast<size_t>(out_index) + 1] = image_y;
                        }
                    } else {
                        uint32_t pixel = static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0]));
                        src.pixel_key = nat_mix(ra.seed_key, pixel);
                        float u, v;
                        src.native_2d(0u, 1u, 0u, false, &u, &v);
                        image_x = px + u;
                        image_y = py + v;
                    }
                    float lens_u1 = 0.0f, lens_u2 = 0.0f;
                    if (EXT && sc.camera.lens_radius != 0.0f) {
                        if (REPLAY) {
                            lens_u1 = src.rec[2];
                            lens_u2 = src.rec[3];
                        } else {
                            src.native_2d(1u, 1u, 0u, true, &lens_u1, &lens_u2);
                        }
                    }
                    camera_ray<EXT>(sc.camera, image_x, image_y, lens_u1, lens_u2, &ps.o, &ps.d, &ps.mint);
                    ps.throughput = f3(1.0f, 1.0f, 1.0f);
                    ps.Li = f3(0.0f, 0.0f, 0.0f);
                    ps.bounce = -1;
                    ps.punch = false;
                    ps.first = true;
                    ps.path = fetched;
                    path_draws = 0;
                    prim_t = INFINITY;
                    active = true;
                    if (STATS) cnt.dims += 2;
                } else {
                    exhausted = true;
                }
            }
            if (!PRIM && __ballot(active) == 0ull) break;

            bool finished = false;
            Hit hit;
            bool got = false;
            if constexpr (QUAD) {
                const bool want = active && sc.num_lights != 0;
                if (!PRIM || __ballot(want) != 0ull)
                    got = trace_quad<false, STATS, EXT, TIES>(sc, want, ps.o, ps.d, ps.mint, INFINITY, stk, quad_slab, quad_stack, hit, cnt);
                if (active && !want) finished = true;
                if (STATS && want) cnt.ext += 1;
            } else if (active) {
                if (sc.num_lights == 0) {
                    finished = true;   // PathTracer::Li returns Black without lights (:53-56)
                } else {
                    got = trace<false, STATS, EXT, TIES>(sc, ps.o, ps.d, ps.mint, INFINITY, stk, hit, cnt);
                    if (STATS) cnt.ext += 1;
                }
            }
            bool vis = active;
            Frag fr;
            TexFrag tf;
            if (vis && !finished) {
                if (got) {
                    make_fragment<EXT>(sc, hit, ps.o, ps.d, fr, &tf);
                    if (EXT && sc.materials[sc.instances[hit.inst].material].has_tex != 0u)
                        hit_differentials<REPLAY>(sc, src, ps.bounce < 0, image_x, image_y, fr, tf);
                }
                // The MIS query sees opaque surfaces only (isOpaque, GoblinPathtracer.cpp:148) and is attenuated by the
                // masks in front of them; the extension query above sees everything.  They only differ when the
                // closest surface is a mask.
                int mis_inst = got ? hit.inst : -1;
                F3 mis_n = fr.n, mis_tr = f3(1.0f, 1.0f, 1.0f);
                if (EXT && sc.has_masks != 0 && got && ps.bounce >= 0 && !ps.punch && sc.instances[hit.inst].is_mask != 0u) {
                    Hit ho;
                    const bool go = trace<false, STATS, EXT>(sc, ps.o, ps.d, ps.mint, INFINITY, stk, ho, cnt, GBL_FILTER_OPAQUE);
                    mis_tr = eval_attenuation<STATS>(sc, ps.o, ps.d, ps.mint, go ? ho.t : INFINITY, stk, cnt);
                    mis_inst = go ? ho.inst : -1;
                    if (go) {
                        Frag fo;
                        make_fragment<EXT>(sc, ho, ps.o, ps.d, fo);
                        mis_n = fo.n;
                    }
                }
                if (ps.bounce < 0) {
                    if (STREAM && got) prim_t = hit.t;
                    if (!got) {
                        if (EXT) {   // Li += scene->evalEnvironmentLight(ray), GoblinPathtracer.cpp:61-65: Black without an image based light
                            const F3 le = environment_le<EXT>(sc, ps.d);
                            ps.Li = f3(ps.Li.x + le.x, ps.Li.y + le.y, ps.Li.z + le.z);
                        }
                        finished = true;
                    } else {
                        F3 le = hit_Le(sc, hit.inst, fr.n, -ps.d);
                        ps.Li = f3(ps.Li.x + le.x, ps.Li.y + le.y, ps.Li.z + le.z);
                        if (EXT && sc.has_bssrdf != 0) {   // Li += Lsubsurface, GoblinPathtracer.cpp:69
                            if constexpr (STREAM) {
                                // the records only exist while the tile walk is on this pixel: evaluate in place (bssrdf.h)
                                const int material = sc.instances[hit.inst].material;
                                if (sc.materials[material].type == GBL_MAT_SUBSURFACE) {
                                    TexFrag ts = tf;   // zero differentials for these lookups; the bounce keeps its own
                                    DevMaterial mo;
                                    sss_resolve(sc, material, fr, ts, mo);
                                    const F3 single = l_bssrdf_single<true, STATS>(sc, ra, src, fr, mo, material, -ps.d, stk, cnt);
                                    const F3 multi = l_bssrdf_diffusion<true, STATS>(sc, ra, src, fr, ts, mo, material, -ps.d, stk, cnt);
                                    const F3 ss = single + multi;
                                    ps.Li = f3(ps.Li.x + ss.x, ps.Li.y + ss.y, ps.Li.z + ss.z);
                                }
                            } else {   // computed ahead by sss_kernel (subsurface.h)
                                const float4 ss = reinterpret_cast<const float4*>(ra.sss)[out_index];
                                ps.Li = f3(ps.Li.x + ss.x, ps.Li.y + ss.y, ps.Li.z + ss.z);
                            }
                        }
                        ps.bounce = 0;
                    }
                } else if (EXT && ps.punch) {
                    // the bounce that punched through a mask contributes no direct light (`continue`, :122-136)
                    ps.punch = false;
                    ps.bounce += 1;
                    if (!got) {
                        if (ps.first) {   // "primary ray need to evaluate image based lighting in this case" (:125-131)
                            const F3 le = ps.throughput * environment_le<EXT>(sc, ps.d);
                            ps.Li = f3(ps.Li.x + le.x, ps.Li.y + le.y, ps.Li.z + le.z);
                        }
                        finished = true;
                    }
                } else {
                    ps.first = false;
                    // close the previous bounce: MIS term for the sampled direction, then Li and throughput
                    if (mis_inst >= 0 && sc.instances[mis_inst].area_light == ps.light) {
                        F3 le = hit_Le(sc, mis_inst, mis_n, -ps.d);
                        if (!is_black(le)) {
                            // Ld += f * tr * Li * absdot(wi, n) * fWeight / bsdfPdf   (tr == 1 without masks)
                            F3 term = EXT ? div(ps.f * mis_tr * le * ps.cosw * ps.fw, ps.bsdf_pdf) : div(ps.f * le * ps.cosw * ps.fw, ps.bsdf_pdf);
                            ps.Ld = f3(ps.Ld.x + term.x, ps.Ld.y + term.y, ps.Ld.z + term.z);
                        }
                    } else if (EXT && mis_inst < 0 && sc.has_ibl != 0) {
                        // the sampled direction left the scene: Ld += f * tr * light->Le(r) * fWeight / bsdfPdf (:157-161; no cosine there)
                        const F3 le = light_le_escaped<EXT>(sc, sc.lights[ps.light], ps.d);
                        const F3 term = div(ps.f * mis_tr * le * ps.fw, ps.bsdf_pdf);
                        ps.Ld = f3(ps.Ld.x + term.x, ps.Ld.y + term.y, ps.Ld.z + term.z);
                    }
                    F3 add = div(ps.throughput * ps.Ld, ps.pick_pdf);
                    ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
                    F3 scale = div(ps.f * ps.cosw, ps.bsdf_pdf);
                    ps.throughput = ps.throughput * scale;
                    ps.bounce += 1;
                    if (!got) finished = true;
                }
                if (!finished && ps.bounce >= ra.max_depth - 1) finished = true;
            }
            if constexpr (PRIM) {
                // The lanes whose path ended at this bounce hand in their Li and start the next path HERE, at the first hit the primary
                // pass (kernels_quad.hip primary_kernel) found for its camera ray: the new path shades its first vertex in this same
                // iteration, so a path costs one extension query less, and a camera ray that left the scene costs a Black and nothing else
                // (PathTracer::Li, GoblinPathtracer.cpp:58-66).  Requires lights (the host only takes this kernel with num_lights > 0).
                if (vis && finished) {
                    reinterpret_cast<float4*>(ra.li_defer)[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                    active = false;
                    finished = false;
                    paths_done += 1;
                }
                for (;;) {
                    const bool idle = !active && !exhausted;
                    if (__ballot(idle) == 0ull) break;
                    fetched = wave_fetch(idle, ctrl + 1);
                    if (!idle) continue;
                    if (fetched < 0 || fetched >= it.paths) {
                        exhausted = true;
                        continue;
                    }
                    const int pix = fetched / ra.chunk_spp;
                    src.k = static_cast<uint32_t>(it.k0 + fetched % ra.chunk_spp);
                    const int px = it.px0 + pix % it.tw, py = it.py0 + pix / it.tw;
                    out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + src.k;
                    hit.inst = ra.prim_inst[out_index];
                    if (hit.inst == GBL_PRIM_MISS) {   // (its Black is in li already: primary_kernel)
                        paths_done += 1;
                        continue;
                    }
                    const uint32_t pixel = static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0]));
                    src.pixel_key = nat_mix(ra.seed_key, pixel);
                    float u, v;
                    src.native_2d(0u, 1u, 0u, false, &u, &v);
                    image_x = px + u;
                    image_y = py + v;
                    camera_ray<EXT>(sc.camera, image_x, image_y, 0.0f, 0.0f, &ps.o, &ps.d, &ps.mint);
                    ps.throughput = f3(1.0f, 1.0f, 1.0f);
                    ps.punch = false;
                    ps.first = true;
                    ps.path = fetched;
                    active = true;
                    if (hit.inst == GBL_PRIM_TIED) {
                        // the packet met two triangles at exactly this ray's closest distance: which of them a ray on its own keeps
                        // depends on its own visiting order, so it is traced here like any other ray (next iteration's query)
                        ps.Li = f3(0.0f, 0.0f, 0.0f);
                        ps.bounce = -1;
                        continue;
                    }
                    const float4 h = reinterpret_cast<const float4*>(ra.prim_hit)[out_index];
                    hit.t = h.x;
                    hit.b1 = h.y;
                    hit.b2 = h.z;
                    hit.tri = __float_as_uint(h.w);
                    make_fragment<EXT>(sc, hit, ps.o, ps.d, fr, &tf);
                    const F3 le = hit_Le(sc, hit.inst, fr.n, -ps.d);
                    ps.Li = f3(0.0f + le.x, 0.0f + le.y, 0.0f + le.z);
                    ps.bounce = 0;
                    finished = ra.max_depth <= 1;
                }
                if (__ballot(active) == 0ull) break;
                vis = active && ps.bounce >= 0;   // (a tied camera ray has nothing to shade yet)
            }

            // ---- shade the vertex: light sample (shadow ray below) and BSDF sample
            bool need_shadow = false;
            F3 shadow_d = f3(0, 0, 1), contrib = f3(0, 0, 0);
            float shadow_maxt = 0.0f;
            F3 wo = -ps.d;
            const DevMaterial* mat = nullptr;
            ResolvedMat rmat;       // EXT: the hit material with its textures evaluated / its mask unwrapped
            F3 l_f = f3(0, 0, 0), l_L = f3(0, 0, 0);
            float l_cos = 0.0f, l_w = 1.0f, l_pdf = 1.0f;
            bool l_area = false;
            float u_bsdf_c = 0.0f, u_bsdf_1 = 0.0f, u_bsdf_2 = 0.0f;
            if (vis && !finished) {
                const int b = ps.bounce;
                float u_light_c, u_light_1, u_light_2, u_pick;
                if (REPLAY) {
                    const float* r1 = src.rec + 4 + 3 * b;
                    const float* r2 = src.rec + ra.off2_base + 4 * b;
                    u_light_c = r1[0]; u_bsdf_c = r1[1]; u_pick = r1[2];
                    u_light_1 = r2[0]; u_light_2 = r2[1]; u_bsdf_1 = r2[2]; u_bsdf_2 = r2[3];
                } else {
                    u_light_c = src.native_1d(3u * b + 0u);
                    u_bsdf_c = src.native_1d(3u * b + 1u);
                    u_pick = src.native_1d(3u * b + 2u);
                    src.native_2d(0x10000u + 2u * b, 1u, 0u, true, &u_light_1, &u_light_2);
                    src.native_2d(0x10000u + 2u * b + 1u, 1u, 0u, true, &u_bsdf_1, &u_bsdf_2);
                }
                if (STATS) cnt.dims += 7;
                // Scene::sampleLight: CDF1D::sampleDiscrete over the power distribution
                int li = 0;
                for (int i = 1; i <= sc.num_lights; ++i)
                    if (sc.light_cdf[i] < u_pick) li = i;
                if (li >= sc.num_lights) li = sc.num_lights - 1;
                ps.light = li;
                ps.pick_pdf = sc.light_pick_pdf[li];
                ps.Ld = f3(0, 0, 0);
                mat = sc.materials + sc.instances[hit.inst].material;
                if (EXT) resolve_hit_material(sc, sc.instances[hit.inst].material, fr, tf, rmat);
                const DevLight& light = sc.lights[li];
                LightSampleOut ls;
                light_sample<EXT>(sc, light, fr.p, fr.eps, u_light_c, u_light_1, u_light_2, ls);
                if (!is_black(ls.L) && ls.pdf > 0.0f) {
                    F3 f = EXT ? rmat_bsdf(rmat, fr.n, wo, ls.wi) : mat_bsdf(*mat, fr.n, wo, ls.wi);
                    if (!is_black(f)) {
                        need_shadow = true;
                        shadow_d = ls.wi;
                        shadow_maxt = ls.maxt;
                        float lw = 1.0f;
                        if (light_is_delta<EXT>(light)) {
                            contrib = div(f * ls.L * absdot(fr.n, ls.wi), ls.pdf);
                        } else {
                            float bp = EXT ? rmat_pdf(rmat, fr.n, wo, ls.wi) : mat_pdf(*mat, fr.n, wo, ls.wi);
                            lw = power_heuristic(ls.pdf, bp);
                            contrib = div(f * ls.L * absdot(fr.n, ls.wi) * lw, ls.pdf);
                        }
                        if (EXT) {   // kept for the attenuated form f * tr * L * |n.wi| (* lWeight) / lightPdf below
                            l_f = f;
                            l_L = ls.L;
                            l_cos = absdot(fr.n, ls.wi);
                            l_w = lw;
                            l_pdf = ls.pdf;
                            l_area = !light_is_delta<EXT>(light);
                        }
                    }
                }
            }
            // ---- shadow query (any-hit)
            bool quad_occluded = false;
            if constexpr (QUAD) {
                if (__ballot(need_shadow) != 0ull) {
                    Hit dummy;
                    quad_occluded = trace_quad<true, STATS, EXT, TIES>(sc, need_shadow, fr.p, shadow_d, fr.eps, shadow_maxt, stk, quad_slab, quad_stack, dummy, cnt,
                                                                     (EXT && sc.has_masks != 0) ? GBL_FILTER_OPAQUE : GBL_FILTER_NONE);
                }
            }
            if (need_shadow) {
                Hit dummy;
                const bool masks = EXT && sc.has_masks != 0;
                bool occluded = QUAD ? quad_occluded
                                   : trace<true, STATS, EXT, TIES>(sc, fr.p, shadow_d, fr.eps, shadow_maxt, stk, dummy, cnt,
                                                             masks ? GBL_FILTER_OPAQUE : GBL_FILTER_NONE);
                if (STATS) cnt.shadow += 1;
                if (!occluded && masks) {
                    F3 tr = eval_attenuation<STATS>(sc, fr.p, shadow_d, fr.eps, shadow_maxt, stk, cnt);
                    contrib = l_area ? div(l_f * tr * l_L * l_cos * l_w, l_pdf) : div(l_f * tr * l_L * l_cos, l_pdf);
                }
                if (!occluded) ps.Ld = f3(ps.Ld.x + contrib.x, ps.Ld.y + contrib.y, ps.Ld.z + contrib.z);
                if (STREAM && !occluded) {   // evalAttenuation(scene, shadowRay, BSDFSample(rng)), :101-103
                    stream_draws += 3;
                    path_draws += 3;
                }
            }
            // ---- BSDF sample: the next ray
            if (vis && !finished) {
                F3 wi;
                float pdf;
                bool specular;
                bool null_sampled = false;
                F3 f = EXT ? rmat_sample(rmat, fr, wo, u_bsdf_c, u_bsdf_1, u_bsdf_2, &wi, &pdf, &specular, &null_sampled)
                           : mat_sample(*mat, fr, wo, u_bsdf_c, u_bsdf_1, u_bsdf_2, &wi, &pdf, &specular);
                if (EXT && null_sampled && !is_black(f) && pdf > 0.0f) {
                    // punch through the mask: throughput *= f / pdf, no cosine, and this bounce's Ld is dropped (:122-136)
                    ps.throughput = ps.throughput * div(f, pdf);
                    ps.o = fr.p;
                    ps.d = wi;
                    ps.mint = fr.eps;
                    ps.punch = true;
                } else if (!is_black(f) && pdf > 0.0f) {
                    float fw = 1.0f;
                    if (!specular) fw = power_heuristic(pdf, light_pdf<EXT>(sc, sc.lights[ps.light], fr.p, wi));
                    if (STREAM) {   // evalAttenuation(scene, r, BSDFSample(rng)) on either branch, :150 / :159
                        stream_draws += 3;
                        path_draws += 3;
                    }
                    ps.f = f;
                    ps.fw = fw;
                    ps.bsdf_pdf = pdf;
                    ps.cosw = absdot(wi, fr.n);
                    ps.o = fr.p;
                    ps.d = wi;
                    ps.mint = fr.eps;
                    if (ra.russian_roulette && !REPLAY && ps.bounce >= 2) {
                        // build-side extension, off in every parity mode
                        F3 tn = ps.throughput * div(ps.f * ps.cosw, ps.bsdf_pdf);
                        float q = fminf(0.95f, fmaxf(tn.x, fmaxf(tn.y, tn.z)));
                        float u = nat_u01(nat_mix(nat_mix(src.pixel_key, 0xBADC0DEu + ps.bounce), src.k));
                        if (!(u < q)) {
                            // terminate after accounting this vertex's direct light
                            F3 add = div(ps.throughput * ps.Ld, ps.pick_pdf);
                            ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
                            finished = true;
                        } else {
                            // survivors carry 1/q on what depends on the extension ray -- the BSDF-sampled light term, the
                            // environment term of an escaping ray and the next throughput, all proportional to f -- NOT on this
                            // vertex's light-sampled term, which is collected whether or not the path survives
                            // (restated in oracle/goblin_oracle.cpp russian_roulette: checked sample by sample)
                            ps.f = ps.f * (1.0f / q);
                        }
                    }
                } else {
                    // Li += throughput * Ld / pickLightPdf; break   (:163-167)
                    F3 add = div(ps.throughput * ps.Ld, ps.pick_pdf);
                    ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
                    finished = true;
                }
            }
            // ---- path end: splat and free the lane
            if (vis && finished) {
                // RenderTask::run: w * (tr * L + Lv), w = 1, tr = 1, Lv = 0
                if (QUAD || ra.li_defer) {
                    reinterpret_cast<float4*>(ra.li_defer)[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                } else {
                    splat<STATS>(sc.film, ftab, tile, tx0, ty0, tp, image_x, image_y, ps.Li, cnt);
                    if (ra.li_out) reinterpret_cast<float4*>(ra.li_out)[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                }
                if (STREAM && EXT && sc.volume.on != 0u) {
                    svol.n[src.k] = path_draws;
                    svol.t[src.k] = prim_t;
                }
                active = false;
                paths_done += 1;
            }
        }
        if constexpr (STREAM && EXT) {
            if (sc.volume.on != 0u) {
                const size_t oi = static_cast<size_t>(static_cast<uint32_t>((it.py0 - ra.window[2]) * sub_w + (it.px0 - ra.window[0]))) * ra.spp;
                stream_medium_phase<STATS>(sc, ra, scx, slay, svol, ctrl, reinterpret_cast<float4*>(ra.li_defer) + oi, stk, cnt);
                continue;   // next pixel: the stream already stands behind this one's last draw
            }
        }
        if constexpr (STREAM) {
            // the stream moves past what this pixel's Li evaluations drew before the next pixel's records are taken from it
            if (stream_draws) atomicAdd(ctrl + 2, stream_draws);
            __syncthreads();
            const uint32_t drawn = ctrl[2];
            __syncthreads();
            if (STATS || GBL_STREAM_TM_ON) {
                const unsigned long long t1 = wall_clock64();
                stream_tm[3] += t1 - stream_t0;
                stream_t0 = t1;
            }
            stream_emit(scx, nullptr, drawn);
            if (STATS || GBL_STREAM_TM_ON) stream_tm[4] += wall_clock64() - stream_t0;
        }
        }   // sub
        __syncthreads();
        if (!ra.li_defer) flush_tile(sc.film, tile, tx0, ty0, tp, ra.film);
    }
    if (STATS) accumulate_stats(ra, cnt, paths_done);
    if ((STATS || GBL_STREAM_TM_ON) && STREAM && threadIdx.x == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(ra.stats + 25 + i, stream_tm[i]);
#ifdef GBL_PHASE_CLOCK
    if (QUAD && !STATS && (threadIdx.x & 63) == 0) {   // measurement build: one lane per wave reports its phase ticks
        cnt.pc[8] = __builtin_amdgcn_s_memtime() - pc_k0;
        for (int i = 0; i < 24; ++i) atomicAdd(ra.stats + i, cnt.pc[i]);
    }
#endif
}

// ---------------------------------------------------------------------------
// Ambient occlusion (AORenderer::Li): one closest hit, N uniform-hemisphere
// any-hit rays, unoccluded fraction as grey radiance.
// ---------------------------------------------------------------------------
// QUAD: the camera ray and the occlusion rays run as wave-wide queries whose last <= 16 rays migrate to quads of lanes
// (kernels/quadtrace.h); per-sample radiance only (ra.li_defer), LDS = quads' records | ctrl | stacks.
template <int SAMPLER, bool STATS, bool EXT, bool QUAD = false, bool EXACT = false>
__global__ __launch_bounds__(GBL_BLOCK, EXT ? GBL_EXT_WAVES : GBL_PT_WAVES) void ao_kernel(DevScene sc, RenderArgs ra) {
    constexpr bool REPLAY = SAMPLER != GBL_SRC_NATIVE, STREAM = SAMPLER == GBL_SRC_STREAM, TIES = REPLAY || STATS || EXACT || EXT;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tp = GBL_TILE + 2 * sc.film.halo;
    float* tile = reinterpret_cast<float*>(smem);
    float* ftab = tile + 4 * tp * tp;
    uint32_t* ctrl = QUAD ? reinterpret_cast<uint32_t*>(smem) + GBL_QUAD_LDS_WORDS : reinterpret_cast<uint32_t*>(ftab + 256);
    uint32_t* stack = ctrl + 4 + (STREAM ? GBL_STREAM_LDS_WORDS : 0);
    static_assert(!(QUAD && (STREAM || EXT || STATS)), "quad-per-ray steps are built for the lean AO kernel");
    typename std::conditional<QUAD, HotLdsStack, LdsStack>::type stk;
    stk.p = gbl_as_lds(stack + threadIdx.x);
    if constexpr (QUAD) {   // see path_trace_kernel
        uint4* hot = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(smem) + ra.hot_word);
        for (uint32_t i = threadIdx.x; i < 4u * ra.hot_count; i += GBL_BLOCK) hot[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
        stk.hot = (const gbl_lds_u4*)hot;
        stk.hot_count = ra.hot_count;
    }
    gbl_lds_u32* const quad_slab = gbl_as_lds(reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * 16 * GBL_QUAD_REC_WORDS);
    gbl_lds_u32* const quad_stack = gbl_as_lds(stack + (threadIdx.x & ~63u));
    if (!QUAD)
        for (int i = threadIdx.x; i < 256; i += GBL_BLOCK) ftab[i] = sc.filter_table[i];
    StreamCtx scx = {};
    StreamLayout slay = {};
    if constexpr (STREAM) {   // see path_trace_kernel; AORenderer::Li draws nothing from the tile's generator itself
        slay = stream_layout(ra.spp, ra.root, 0, 0, 0, ra.ao_n);
        scx.mt = ctrl + 4;
        scx.lperm = stack;
        scx.lperm_words = ra.stream_lperm_words;
        scx.raw = ra.stream_scratch + static_cast<size_t>(blockIdx.x) * ra.stream_stride;
        scx.cols = reinterpret_cast<DevStreamCol*>(scx.raw + slay.NF + slay.NU);
        scx.recs = reinterpret_cast<float*>(scx.cols + slay.ncols);
    }
    StreamVol svol = {};   // STREAM with a participating medium (stream_medium_phase)
    if constexpr (STREAM) svol = stream_vol_scratch(scx, slay);
    if constexpr (STREAM) stream_columns(scx, slay);

    LaneCounters cnt = {};
    uint32_t paths_done = 0;
    const uint32_t n_items = static_cast<uint32_t>(ra.local_tiles) * ra.chunks;
    const int sub_w = ra.window[1] - ra.window[0];
    const int full_w = sc.film.window[1] - sc.film.window[0];

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            ctrl[0] = atomicAdd(ra.work_counter, 1u);
            ctrl[1] = 0u;
        }
        if (!QUAD && !ra.li_defer)
            for (int i = threadIdx.x; i < 4 * tp * tp; i += GBL_BLOCK) tile[i] = 0.0f;
        __syncthreads();
        const uint32_t item = ctrl[0];
        if (item >= n_items) break;
        const ItemInfo tile_item = decode_item(ra, item);
        const int tx0 = tile_item.px0 - sc.film.halo, ty0 = tile_item.py0 - sc.film.halo;
        if constexpr (STREAM) {
            const int ftx = (tile_item.px0 - sc.film.window[0]) / GBL_TILE, fty = (tile_item.py0 - sc.film.window[2]) / GBL_TILE;
            mt_seed(scx, ra.tile_seeds[fty * ra.full_tiles_x + ftx]);
        }
        const int n_sub = STREAM ? tile_item.tw * tile_item.th : 1;
        for (int sub = 0; sub < n_sub; ++sub) {
        ItemInfo it = tile_item;
        if constexpr (STREAM) {
            it.px0 = tile_item.px0 + sub % tile_item.tw;
            it.py0 = tile_item.py0 + sub / tile_item.tw;
            it.tw = it.th = 1;
            it.paths = ra.spp;
            stream_generate_pixel(scx, slay, it.px0, it.py0);
            if (threadIdx.x == 0) ctrl[1] = 0u;
            __syncthreads();
        }

        for (;;) {
            int fetched = wave_fetch(true, ctrl + 1);
            bool valid = fetched >= 0 && fetched < it.paths;
            if (__ballot(valid) == 0ull) break;
            if constexpr (QUAD) {
                // the same sample, with the queries hoisted out of the per-lane branches
                SampleSource src;
                src.spp = ra.spp;
                src.root = ra.root;
                src.rec = nullptr;
                src.pixel_key = 0;
                src.k = 0;
                uint32_t out_index = 0;
                F3 o = f3(0, 0, 0), d = f3(0, 0, 1);
                float cam_mint = 0.0f;
                if (valid) {
                    const int pix = fetched / ra.chunk_spp;
                    src.k = static_cast<uint32_t>(it.k0 + fetched % ra.chunk_spp);
                    const int px = it.px0 + pix % it.tw, py = it.py0 + pix / it.tw;
                    out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + src.k;
                    float image_x, image_y;
                    if (REPLAY) {
                        src.rec = ra.replay + static_cast<size_t>(out_index) * ra.dims;
                        image_x = src.rec[0];
                        image_y = src.rec[1];
                    } else {
                        const uint32_t pixel = static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0]));
                        src.pixel_key = nat_mix(ra.seed_key, pixel);
                        float u, v;
                        src.native_2d(0u, 1u, 0u, false, &u, &v);
                        image_x = px + u;
                        image_y = py + v;
                    }
                    camera_ray<EXT>(sc.camera, image_x, image_y, 0.0f, 0.0f, &o, &d, &cam_mint);
                }
                Hit hit;
                const bool got = trace_quad<false, STATS, EXT, TIES>(sc, valid, o, d, cam_mint, INFINITY, stk, quad_slab, quad_stack, hit, cnt);
                const bool want = valid && got;
                Frag fr;
                fr.p = f3(0, 0, 0);
                fr.eps = 0.0f;
                if (want) make_fragment<EXT>(sc, hit, o, d, fr);
                uint32_t occluded = 0;
                // (The wave's ao_n x 64 occlusion rays as one pool dealt to whichever lanes are idle -- 16 at a time, the owner's
                //  fragment read across lanes, counts in LDS, only the pool's last rays through the quads -- keeps 3/4 of the lanes
                //  busy instead of 59 % and is bit-identical, but a ray's direction (counter hashes, glibc's sin / cos) is then made
                //  by a quarter of the lanes four times as often: 33.2 against 27.1 ms at 1024^2 x 16 spp.  Not kept.)
                if (__ballot(want) != 0ull) {
                    for (int i = 0; i < ra.ao_n; ++i) {
                        F3 dir = f3(0, 0, 1);
                        if (want) {
                            float u1, u2;
                            if (REPLAY) {
                                u1 = src.rec[4 + 2 * i];
                                u2 = src.rec[4 + 2 * i + 1];
                            } else {
                                src.native_2d(0x10000u, static_cast<uint32_t>(ra.ao_n), static_cast<uint32_t>(i), true, &u1, &u2);
                            }
                            dir = shade_to_world(fr, uniform_sample_hemisphere(u1, u2));
                        }
                        Hit dummy;
                        const bool occ = trace_quad<true, STATS, EXT, TIES>(sc, want, fr.p, dir, fr.eps, INFINITY, stk, quad_slab, quad_stack, dummy, cnt);
                        if (want && occ) occluded += 1;
                    }
                }
                if (valid) {
                    F3 L = f3(0, 0, 0);
                    if (got) {
                        const float g = static_cast<float>(static_cast<uint32_t>(ra.ao_n) - occluded) / static_cast<float>(static_cast<uint32_t>(ra.ao_n));
                        L = f3(g, g, g);
                    }
                    reinterpret_cast<float4*>(ra.li_defer)[out_index] = make_float4(L.x, L.y, L.z, 1.0f);
                    paths_done += 1;
                }
                continue;
            }
            if (valid) {
            int pix = fetched / ra.chunk_spp;
            SampleSource src;
            src.spp = ra.spp;
            src.root = ra.root;
            src.rec = nullptr;
            src.pixel_key = 0;
            src.k = static_cast<uint32_t>(it.k0 + fetched % ra.chunk_spp);
            int px = it.px0 + pix % it.tw, py = it.py0 + pix / it.tw;
            uint32_t out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + src.k;
            float image_x, image_y;
            if (REPLAY) {
                src.rec = STREAM ? scx.recs + static_cast<size_t>(src.k) * ra.dims : ra.replay + static_cast<size_t>(out_index) * ra.dims;
                image_x = src.rec[0];
                image_y = src.rec[1];
                if (STREAM) {
                    ra.image_xy[2 * static_cast<size_t>(out_index)] = image_x;
                    ra.image_xy[2 * static_cast<size_t>(out_index) + 1] = image_y;
                }
            } else {
                uint32_t pixel = static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0]));
                src.pixel_key = nat_mix(ra.seed_key, pixel);
                float u, v;
                src.native_2d(0u, 1u, 0u, false, &u, &v);
                image_x = px + u;
                image_y = py + v;
            }
            F3 o, d;
            float cam_mint, lens_u1 = 0.0f, lens_u2 = 0.0f;
            if (EXT && sc.camera.lens_radius != 0.0f) {
                if (REPLAY) {
                    lens_u1 = src.rec[2];
                    lens_u2 = src.rec[3];
                } else {
                    src.native_2d(1u, 1u, 0u, true, &lens_u1, &lens_u2);
                }
            }
            camera_ray<EXT>(sc.camera, image_x, image_y, lens_u1, lens_u2, &o, &d, &cam_mint);
            Hit hit;
            F3 L = f3(0, 0, 0);
            bool got = trace<false, STATS, EXT, TIES>(sc, o, d, cam_mint, INFINITY, stk, hit, cnt);   // lean native build: no tie rule (trace.h)
            if (STATS) {
                cnt.ext += 1;
                cnt.dims += 2;
            }
            if (got) {
                Frag fr;
                make_fragment<EXT>(sc, hit, o, d, fr);
                // (Chaining a lane's N rays inside one wave-level loop with batched restarts was measured: 38.6 ms against
                // 33.8 ms for this plain loop on bunny 1024^2 x 16 spp x 25 rays -- the samples of a pixel are coherent.)
                uint32_t occluded = 0;
                for (int i = 0; i < ra.ao_n; ++i) {
                    float u1, u2;
                    if (REPLAY) {
                        u1 = src.rec[4 + 2 * i];
                        u2 = src.rec[4 + 2 * i + 1];
                    } else {
                        src.native_2d(0x10000u, static_cast<uint32_t>(ra.ao_n), static_cast<uint32_t>(i), true, &u1, &u2);
                    }
                    F3 dir = shade_to_world(fr, uniform_sample_hemisphere(u1, u2));
                    Hit dummy;
                    if (trace<true, STATS, EXT, TIES>(sc, fr.p, dir, fr.eps, INFINITY, stk, dummy, cnt)) occluded += 1;
                    if (STATS) cnt.shadow += 1;
                }
                if (STATS) cnt.dims += 2 * ra.ao_n;
                float g = static_cast<float>(static_cast<uint32_t>(ra.ao_n) - occluded) / static_cast<float>(static_cast<uint32_t>(ra.ao_n));
                L = f3(g, g, g);
            }
            if (ra.li_defer) {
                reinterpret_cast<float4*>(ra.li_defer)[out_index] = make_float4(L.x, L.y, L.z, 1.0f);
            } else {
                splat<STATS>(sc.film, ftab, tile, tx0, ty0, tp, image_x, image_y, L, cnt);
                if (ra.li_out) reinterpret_cast<float4*>(ra.li_out)[out_index] = make_float4(L.x, L.y, L.z, 1.0f);
            }
            if (STREAM && EXT && sc.volume.on != 0u) {   // AORenderer::Li draws nothing; scene->intersect clipped the camera ray
                svol.n[src.k] = 0u;
                svol.t[src.k] = got ? hit.t : INFINITY;
            }
            paths_done += 1;
            }   // valid
        }
        if constexpr (STREAM && EXT) {
            if (sc.volume.on != 0u) {
                const size_t oi = static_cast<size_t>(static_cast<uint32_t>((it.py0 - ra.window[2]) * sub_w + (it.px0 - ra.window[0]))) * ra.spp;
                stream_medium_phase<STATS>(sc, ra, scx, slay, svol, ctrl, reinterpret_cast<float4*>(ra.li_defer) + oi, stk, cnt);
            }
        }
        if constexpr (STREAM) __syncthreads();   // every record of this pixel has been read before the next overwrites them
        }   // sub
        __syncthreads();
        if (!QUAD && !ra.li_defer) flush_tile(sc.film, tile, tx0, ty0, tp, ra.film);
    }
    if (STATS) accumulate_stats(ra, cnt, paths_done);
}
