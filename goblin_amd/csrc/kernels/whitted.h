// WhittedRenderer (GoblinWhitted.cpp:13-44): per hit, emission + Renderer::multiSampleLd over EVERY light
// (GoblinRenderer.cpp:474-500; estimateLd :502-567 with type = BSDFAll & ~BSDFSpecular) + the specular tree
// (specularReflect / specularRefract, :598-648) down to ray depth max_ray_depth.
//
// The reference recurses: L += f * Li(child) * |wi.n| / pdf.  To reproduce that expression's float order the device
// keeps the recursion's frames explicitly (one per ray depth, in scratch) and combines a child's radiance into its
// parent on the way back up, reflection first, then refraction -- instead of carrying a path weight downwards, which
// would distribute the multiplications differently.  Every level re-reads the SAME Sample slots (the reference passes
// `sample` unchanged into the recursion).
//
// Quota (WhittedRenderer::querySampleQuota, GoblinWhitted.cpp:46-70): per light i a LightSampleIndex and a
// BSDFSampleIndex of n_i = roundToSquare(getSamplesNum()) slots, one pick 1D (never read), the BSSRDF block.
// Record offsets: 1D  4 + 2 P_i (+ n_i for the bsdf component), P_i = sum of n_j before light i;
//                 2D  4 + 2 N + 1 + 4 nb + 4 P_i (+ 2 n_i), N = sum of all n_j.
// Mask and subsurface materials are rejected for this integrator by gbl_render.
#pragma once
#include "render_kernels.h"


struct WhSlots {   // the Sample values of one (light, slot) pair
    float ls_comp, ls_u1, ls_u2, bs_comp, bs_u1, bs_u2;
};
template <bool REPLAY>
__device__ __forceinline__ WhSlots wh_slots(const DevScene& sc, const RenderArgs& ra, const SampleSource& src, int li, uint32_t slot) {
    const DevLight& l = sc.lights[li];
    const uint32_t n = l.wh_n, P = l.wh_prefix;
    WhSlots w;
    if (REPLAY) {
        const float* r1 = src.rec + 4 + 2 * P;
        const float* r2 = src.rec + ra.off2_base + 4 * P;
        w.ls_comp = r1[slot];
        w.bs_comp = r1[n + slot];
        w.ls_u1 = r2[2 * slot];
        w.ls_u2 = r2[2 * slot + 1];
        w.bs_u1 = r2[2 * n + 2 * slot];
        w.bs_u2 = r2[2 * n + 2 * slot + 1];
    } else {
        const uint32_t i = static_cast<uint32_t>(li);
        w.ls_comp = src.native_1d_n(2u * i, n, slot);
        w.bs_comp = src.native_1d_n(2u * i + 1u, n, slot);
        src.native_2d_slot(0x10000u + 2u * i, n, slot, &w.ls_u1, &w.ls_u2);
        src.native_2d_slot(0x10000u + 2u * i + 1u, n, slot, &w.bs_u1, &w.bs_u2);
    }
    return w;
}

// Renderer::estimateLd for a non-specular request (GoblinRenderer.cpp:502-567)
template <bool STATS, class STK>
__device__ __forceinline__ F3 wh_estimate_ld(const DevScene& sc, const Frag& fr, const ResolvedMat& rmat, F3 wo, int li, const WhSlots& w,
                                             const STK& stk, LaneCounters& cnt) {
    const DevLight& light = sc.lights[li];
    F3 Ld = f3(0, 0, 0);
    LightSampleOut ls;
    light_sample<true>(sc, light, fr.p, fr.eps, w.ls_comp, w.ls_u1, w.ls_u2, ls);
    if (!is_black(ls.L) && ls.pdf > 0.0f) {
        const F3 f = rmat_bsdf(rmat, fr.n, wo, ls.wi);
        if (!is_black(f)) {
            Hit dummy;
            if (STATS) cnt.shadow += 1;
            if (!trace<true, STATS, true>(sc, fr.p, ls.wi, fr.eps, ls.maxt, stk, dummy, cnt)) {
                if (light_is_delta<true>(light)) return div(f * ls.L * absdot(fr.n, ls.wi), ls.pdf);   // isDelta(): no MIS
                const float bp = rmat_pdf(rmat, fr.n, wo, ls.wi);
                const float lw = power_heuristic(ls.pdf, bp);
                const F3 t = div(f * ls.L * absdot(fr.n, ls.wi) * lw, ls.pdf);
                Ld = f3(Ld.x + t.x, Ld.y + t.y, Ld.z + t.z);
            }
        }
    }
    // sampleBSDF(..., BSDFAll & ~BSDFSpecular): Transparent and Mirror do not match (pdf 0), nor does SubsurfaceMaterial whose
    // BSDFAll type only matches that very request (GoblinMaterial.cpp:732-736); Lambert / Blinn sample as always
    // A MaskMaterial forwards the request to the wrapped material when uComponent < alpha and answers it itself with the
    // BSDFnullptr lobe otherwise (:757-784): straight through, non-specular, so it still looks for the light behind.
    if ((rmat.m.type == GBL_MAT_TRANSPARENT || rmat.m.type == GBL_MAT_MIRROR || rmat.m.type == GBL_MAT_SUBSURFACE) &&
        !(rmat.is_mask && !(w.bs_comp < rmat.alpha)))
        return Ld;
    F3 wi;
    float pdf;
    bool specular, null_sampled;
    const F3 f = rmat_sample(rmat, fr, wo, w.bs_comp, w.bs_u1, w.bs_u2, &wi, &pdf, &specular, &null_sampled);
    if (!is_black(f) && pdf > 0.0f) {
        float fw = 1.0f;
        if (!specular) {
            const float lp = light_pdf<true>(sc, light, fr.p, wi);
            if (lp == 0.0f) return Ld;
            fw = power_heuristic(pdf, lp);
        }
        Hit lh;
        if (STATS) cnt.ext += 1;
        if (trace<false, STATS, true>(sc, fr.p, wi, fr.eps, INFINITY, stk, lh, cnt)) {
            if (sc.instances[lh.inst].area_light == li) {
                Frag lf;
                make_fragment<true>(sc, lh, fr.p, wi, lf);
                const F3 le = hit_Le(sc, lh.inst, lf.n, -wi);
                if (!is_black(le)) {
                    const F3 t = div(f * le * absdot(wi, fr.n) * fw, pdf);
                    Ld = f3(Ld.x + t.x, Ld.y + t.y, Ld.z + t.z);
                }
            }
        } else if (sc.has_ibl != 0) {   // the radiance contribution from IBL: Ld += f * light->Le(r) * fWeight / bsdfPdf (:558-561)
            const F3 le = light_le_escaped<true>(sc, light, wi);
            const F3 t = div(f * le * fw, pdf);
            Ld = f3(Ld.x + t.x, Ld.y + t.y, Ld.z + t.z);
        }
    }
    return Ld;
}

// One frame of the recursion: the hit's radiance so far and the two specular children still to be folded in.
struct WhFrame {
    F3 acc;                       // Le + multiSampleLd (+ the reflection term once it has returned)
    F3 p;
    float eps;
    F3 n;
    F3 refl_f, refl_wi;           // reflection child: f, wi; pdf is 1 for both lobes
    F3 refr_f, refr_wi;
    uint32_t flags;               // bit 0 reflection child valid, bit 1 refraction child valid, bit 2 the reflection child is in flight
};

// L = Black; L += f * Lr * |wi.n| / pdf   (specularReflect / specularRefract), then Li += L
__device__ __forceinline__ F3 wh_fold(F3 acc, F3 f, F3 Lr, F3 wi, F3 n) {
    const F3 t = div(f * Lr * absdot(wi, n), 1.0f);
    const F3 L = f3(0.0f + t.x, 0.0f + t.y, 0.0f + t.z);
    return f3(acc.x + L.x, acc.y + L.y, acc.z + L.z);
}

template <bool REPLAY, bool STATS, class STK>
__device__ F3 whitted_li(const DevScene& sc, const RenderArgs& ra, const SampleSource& src, F3 o, F3 d, float mint, float image_x, float image_y,
                         const STK& stk, LaneCounters& cnt, uint32_t* draws = nullptr, float* prim_t = nullptr) {
    WhFrame frames[GBL_WHITTED_MAX_DEPTH + 1];
    int depth = 0;
    F3 ret = f3(0, 0, 0);
    for (;;) {
        // ---- WhittedRenderer::Li for the ray (o, d, mint) at `depth`
        Hit hit;
        bool descend = false;
        ret = f3(0, 0, 0);
        if (STATS) cnt.ext += 1;
        if (trace<false, STATS, true>(sc, o, d, mint, INFINITY, stk, hit, cnt)) {
            if (prim_t && depth == 0) *prim_t = hit.t;   // the camera ray's maxt after scene->intersect
            Frag fr;
            TexFrag tf;
            make_fragment<true>(sc, hit, o, d, fr, &tf);
            const int material = sc.instances[hit.inst].material;
            if (sc.materials[material].has_tex != 0u) hit_differentials<REPLAY>(sc, src, depth == 0, image_x, image_y, fr, tf);
            const F3 wo = -d;
            const F3 le = hit_Le(sc, hit.inst, fr.n, wo);
            F3 Li = f3(0.0f + le.x, 0.0f + le.y, 0.0f + le.z);
            if (sc.has_bssrdf != 0 && sc.num_lights > 0 && sc.materials[material].type == GBL_MAT_SUBSURFACE) {
                // Li += Lsubsurface(...) with the differentials in place, at every level, always from the camera sample's
                // one BSSRDF block (GoblinWhitted.cpp:25-27)
                DevMaterial mo;
                sss_resolve<true>(sc, material, fr, tf, mo);
                const F3 single = l_bssrdf_single<REPLAY, STATS>(sc, ra, src, fr, mo, material, wo, stk, cnt);
                const F3 multi = l_bssrdf_diffusion<REPLAY, STATS>(sc, ra, src, fr, tf, mo, material, wo, stk, cnt);
                const F3 ss = single + multi;
                Li = f3(Li.x + ss.x, Li.y + ss.y, Li.z + ss.z);
            }
            ResolvedMat rmat;
            resolve_hit_material(sc, material, fr, tf, rmat);
            // multiSampleLd: every light, its samplesNum slots averaged
            F3 total = f3(0, 0, 0);
            for (int li = 0; li < sc.num_lights; ++li) {
                const uint32_t n = sc.lights[li].wh_n;
                F3 Ld = f3(0, 0, 0);
                for (uint32_t s = 0; s < n; ++s) {
                    const WhSlots w = wh_slots<REPLAY>(sc, ra, src, li, s);
                    if (STATS) cnt.dims += 6;
                    if (draws) *draws += 6;   // LightSample ls(rng); BSDFSample bs(rng); before the Sample's values replace them (:487-488)
                    const F3 e = wh_estimate_ld<STATS>(sc, fr, rmat, wo, li, w, stk, cnt);
                    Ld = f3(Ld.x + e.x, Ld.y + e.y, Ld.z + e.z);
                }
                Ld = div(Ld, static_cast<float>(n));
                total = f3(total.x + Ld.x, total.y + Ld.y, total.z + Ld.z);
            }
            Li = f3(Li.x + total.x, Li.y + total.y, Li.z + total.z);
            ret = Li;
            if (depth < ra.max_depth) {
                if (draws) *draws += 6;   // BSDFSample(rng) in specularReflect and in specularRefract (:612, :638)
                // the two specular requests: Mirror answers the reflection one, Transparent each with the matching lobe at pdf 1
                WhFrame& F = frames[depth];
                F.flags = 0u;
                const DevMaterial& m = rmat.m;
                const F3 n = fr.n;
                if (m.type == GBL_MAT_MIRROR) {
                    const float cosi = dot(n, wo);
                    if (cosi > 0.0f) {
                        const float fres = fresnel_conductor(cosi, m.index, m.k);
                        F.refl_wi = 2 * cosi * n - wo;
                        F.refl_f = f3(m.color[0], m.color[1], m.color[2]) * (fres / cosi);
                        if (!is_black(F.refl_f) && absdot(F.refl_wi, n) != 0.0f) F.flags |= 1u;
                    }
                } else if (m.type == GBL_MAT_TRANSPARENT) {
                    const float cosi = dot(n, wo);
                    const bool entering = cosi > 0.0f;
                    const F3 nn = entering ? n : -n;
                    const float ci = entering ? cosi : -cosi;
                    const float ei = entering ? 1.0f : m.index, et = entering ? m.index : 1.0f;
                    const float fr_refl = fresnel_dielectric(ci, ei, et);
                    F.refl_wi = 2 * ci * nn - wo;
                    F.refl_f = f3(m.color[0], m.color[1], m.color[2]) * (fr_refl / ci);
                    if (!is_black(F.refl_f) && absdot(F.refl_wi, n) != 0.0f) F.flags |= 1u;
                    const float ro_et = entering ? 1.0f : m.index, ro_ei = entering ? m.index : 1.0f;
                    const float f2 = fresnel_dielectric(ci, ro_et, ro_ei);
                    if (f2 != 1.0f) {
                        const float eta = ro_et / ro_ei;
                        F.refr_wi = normalize(nn * (eta * ci - sqrtf(fmaxf(0.0f, 1.0f - eta * eta * (1.0f - ci * ci)))) - eta * wo);
                        const float refract = eta * eta * (1.0f - f2) / absdot(F.refr_wi, nn);
                        F.refr_f = f3(m.color2[0], m.color2[1], m.color2[2]) * refract;
                        if (!is_black(F.refr_f) && absdot(F.refr_wi, n) != 0.0f) F.flags |= 2u;
                    }
                }
                if (rmat.is_mask) {   // neither request has the BSDFnullptr bit: the wrapped material's answer times alpha, at its pdf (:781-783)
                    if (F.flags & 1u) {
                        F.refl_f = rmat.alpha * F.refl_f;
                        if (is_black(F.refl_f)) F.flags &= ~1u;
                    }
                    if (F.flags & 2u) {
                        F.refr_f = rmat.alpha * F.refr_f;
                        if (is_black(F.refr_f)) F.flags &= ~2u;
                    }
                }
                if (F.flags != 0u) {
                    F.acc = Li;
                    F.p = fr.p;
                    F.eps = fr.eps;
                    F.n = n;
                    o = fr.p;
                    mint = fr.eps;
                    if (F.flags & 1u) {
                        F.flags |= 4u;
                        d = F.refl_wi;
                    } else {
                        d = F.refr_wi;
                    }
                    depth += 1;
                    descend = true;
                }
            }
        } else {
            // get image based lighting if the ray didn't hit anything: Li += scene->evalEnvironmentLight(ray) (GoblinWhitted.cpp:40-43)
            const F3 le = environment_le<true>(sc, d);
            ret = f3(0.0f + le.x, 0.0f + le.y, 0.0f + le.z);
        }
        if (descend) continue;
        // ---- hand `ret` to the parents until one of them has another child to trace
        bool resumed = false;
        while (depth > 0) {
            depth -= 1;
            WhFrame& F = frames[depth];
            if (F.flags & 4u) {   // the reflection child came back
                F.acc = wh_fold(F.acc, F.refl_f, ret, F.refl_wi, F.n);
                F.flags &= ~4u;
                if (F.flags & 2u) {
                    o = F.p;
                    mint = F.eps;
                    d = F.refr_wi;
                    depth += 1;
                    resumed = true;
                    break;
                }
                ret = F.acc;
            } else {              // the refraction child came back
                F.acc = wh_fold(F.acc, F.refr_f, ret, F.refr_wi, F.n);
                ret = F.acc;
            }
        }
        if (!resumed) return ret;
    }
}

// Two waves per SIMD (256 registers; the recursion frames live in scratch either way): left to itself the compiler takes 335 (79 of
// them accumulation registers as spill space) and one wave per SIMD -- whitted.json 512^2 x 64 spp: 50.1 ms; held to 256: 31.5 ms.
#ifndef GBL_WHITTED_WAVES
#define GBL_WHITTED_WAVES 2
#endif
// One lane per camera sample (ids enumerate owned tile, pixel in tile, sample), per-sample radiance into `out`
// (pixel-major like li_out); wf_splat filters it into the film.
template <bool REPLAY, bool STATS>
__global__ __launch_bounds__(GBL_BLOCK, GBL_WHITTED_WAVES) void whitted_kernel(DevScene sc, RenderArgs ra, float4* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsStack stk = {gbl_as_lds(reinterpret_cast<uint32_t*>(smem) + threadIdx.x)};
    LaneCounters cnt = {};
    uint32_t paths_done = 0;
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
        if (STATS) cnt.dims += 2;
        F3 o, d;
        float mint;
        camera_ray<true>(sc.camera, image_x, image_y, lens_u1, lens_u2, &o, &d, &mint);
        const F3 L = whitted_li<REPLAY, STATS>(sc, ra, src, o, d, mint, image_x, image_y, stk, cnt);
        out[out_index] = make_float4(L.x, L.y, L.z, 1.0f);
        paths_done += 1;
    }
    if (STATS) accumulate_stats(ra, cnt, paths_done);
}

// The Whitted renderer under GBL_SAMPLES_STREAM: the tile walk of path_trace_kernel<.., STREAM> (kernels/stream.h) around
// whitted_li -- one workgroup per tile, per pixel the reference's records generated from the tile's mt19937, the S
// samples traced, and the stream moved past the 6 floats per (light, slot) and 6 per specular level they discarded.
template <bool STATS>
__global__ __launch_bounds__(GBL_BLOCK, GBL_WHITTED_WAVES) void whitted_stream_kernel(DevScene sc, RenderArgs ra, float4* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* ctrl = reinterpret_cast<uint32_t*>(smem);
    uint32_t* stack = ctrl + 4 + GBL_STREAM_LDS_WORDS;
    const LdsStack stk = {gbl_as_lds(stack + threadIdx.x)};
    StreamLayout slay;
    stream_layout_whitted(slay, ra.spp, ra.root, ra.bssrdf_n, ra.bssrdf_n2, sc.num_lights, [&](int i) { return sc.lights[i].wh_n; });
    StreamCtx scx;
    scx.mt = ctrl + 4;
    scx.pos = GBL_MT_N;
    scx.which = 0u;
    scx.lperm = stack;
    scx.lperm_words = ra.stream_lperm_words;
    scx.raw = ra.stream_scratch + static_cast<size_t>(blockIdx.x) * ra.stream_stride;
    scx.cols = reinterpret_cast<DevStreamCol*>(scx.raw + slay.NF + slay.NU);
    scx.recs = reinterpret_cast<float*>(scx.cols + slay.ncols);
    const StreamVol svol = stream_vol_scratch(scx, slay);   // with a participating medium (stream_medium_phase)
    stream_columns(scx, slay);
    LaneCounters cnt = {};
    uint32_t paths_done = 0;
    const uint32_t n_items = static_cast<uint32_t>(ra.local_tiles);
    const int sub_w = ra.window[1] - ra.window[0];
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) ctrl[0] = atomicAdd(ra.work_counter, 1u);
        __syncthreads();
        const uint32_t item = ctrl[0];
        if (item >= n_items) break;
        const ItemInfo tile_item = decode_item(ra, item);
        const int ftx = (tile_item.px0 - sc.film.window[0]) / GBL_TILE, fty = (tile_item.py0 - sc.film.window[2]) / GBL_TILE;
        mt_seed(scx, ra.tile_seeds[fty * ra.full_tiles_x + ftx]);
        for (int sub = 0; sub < tile_item.tw * tile_item.th; ++sub) {
            const int px = tile_item.px0 + sub % tile_item.tw, py = tile_item.py0 + sub / tile_item.tw;
            stream_generate_pixel(scx, slay, px, py);
            if (threadIdx.x == 0) ctrl[2] = 0u;
            __syncthreads();
            uint32_t draws = 0;
            for (uint32_t k = threadIdx.x; k < static_cast<uint32_t>(ra.spp); k += GBL_BLOCK) {
                SampleSource src;
                src.spp = ra.spp;
                src.root = ra.root;
                src.pixel_key = 0;
                src.k = k;
                src.rec = scx.recs + static_cast<size_t>(k) * ra.dims;
                const uint32_t out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + k;
                const float image_x = src.rec[0], image_y = src.rec[1];
                ra.image_xy[2 * static_cast<size_t>(out_index)] = image_x;
                ra.image_xy[2 * static_cast<size_t>(out_index) + 1] = image_y;
                if (STATS) cnt.dims += 2;
                F3 o, d;
                float mint;
                camera_ray<true>(sc.camera, image_x, image_y, src.rec[2], src.rec[3], &o, &d, &mint);
                const uint32_t before = draws;
                float prim_t = INFINITY;
                const F3 L = whitted_li<true, STATS>(sc, ra, src, o, d, mint, image_x, image_y, stk, cnt, &draws, &prim_t);
                out[out_index] = make_float4(L.x, L.y, L.z, 1.0f);
                if (sc.volume.on != 0u) {
                    svol.n[k] = draws - before;
                    svol.t[k] = prim_t;
                }
                paths_done += 1;
            }
            if (sc.volume.on != 0u) {
                const size_t oi = static_cast<size_t>(static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0]))) * ra.spp;
                stream_medium_phase<STATS>(sc, ra, scx, slay, svol, ctrl, out + oi, stk, cnt);
                continue;   // next pixel: the stream already stands behind this one's last draw
            }
            if (draws) atomicAdd(ctrl + 2, draws);
            __syncthreads();
            const uint32_t drawn = ctrl[2];
            __syncthreads();
            stream_emit(scx, nullptr, drawn);
        }
    }
    if (STATS) accumulate_stats(ra, cnt, paths_done);
}
