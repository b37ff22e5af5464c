// The persistent megakernel with the two rays of a vertex traced as ONE job per lane.
//
// path_trace_kernel (render_kernels.h) runs, per bounce: extension query -> close + shade -> shadow query -> BSDF sample;
// every query is a wave-wide loop that ends when its slowest lane does, so a bounce waits for the slowest lane twice.
// The shadow ray and the next extension ray of a vertex are both known once the vertex is shaded and do not depend on
// each other (the shadow result only enters Ld when the bounce is CLOSED, after the extension ray's hit is known,
// GoblinPathtracer.cpp:96-113 / :140-163).  Here a lane traces them back to back inside one traversal loop -- shadow ray
// (any hit), then extension ray (closest hit) -- and the wave waits once per bounce, for max(shadow + extension) instead
// of max(shadow) + max(extension).  Same arithmetic, statement for statement; per-sample radiance is bit-identical.
//
// Mask scenes (filtered queries, attenuation walks between the two rays) and GBL_SAMPLES_STREAM keep path_trace_kernel.
#pragma once
#include "render_kernels.h"

// One lane's job: [shadow ray (o, sd, mint, smaxt), any hit] then [extension ray (o, ed, mint, inf), closest hit].
//
// The traversal is trace.h's, step for step (same node test, same Moller-Trumbore, same tie rule, same visiting order),
// written flat for this loop: the lanes of a wave are out of step with each other here -- the second ray starts whenever
// the first ends -- so the loop runs ONE kind of step per iteration, the kind most lanes wait for:
//   INTERIOR  fetch a 4-wide node, test, descend / pop            (trav_interior)
//   LEAF      test the triangles (or the shape) of a BLAS leaf, pop
//   ENTER     a TLAS leaf: move the ray into the instance, push the sentinel
//   SWITCH    the shadow ray is done: start the extension ray (its reciprocal direction was formed before the loop)
// and everything that is mere bookkeeping -- popping the instance sentinel back to the world ray, noticing the exit
// marker -- is folded into the pop that found it instead of costing the wave an iteration of its own.
#ifndef PAIR_SWITCH_MIN
#define PAIR_SWITCH_MIN 1
#endif
template <bool STATS, bool EXT, bool TIES, class STK>
__device__ __forceinline__ void trace_pair(const DevScene& sc, bool has_shadow, bool has_ext, F3 o, float mint, F3 sd, float smaxt, F3 ed,
                                           const STK& stk, bool* occluded, Hit* hit, LaneCounters& cnt) {
    TravState st;
    bool any = has_shadow;
    bool busy = has_shadow || has_ext;
    bool sw = false;   // shadow ray done, extension ray to start
    bool occ = false;
    RaySpace ext;      // the extension ray's space, formed here at full occupancy
    ray_space(ext, o, ed);
    trav_begin(sc, st, o, any ? sd : ed, mint, any ? smaxt : INFINITY, stk);
    if (STATS && busy) {
        if (any) cnt.shadow += 1; else cnt.ext += 1;
    }
    if (busy && st.cur == GBL_STACK_EXIT) busy = false;   // empty scene
    uint32_t steps = 0;
    for (;;) {
        const bool k_int = busy && !sw && trav_at_interior(st);
        const bool k_ref = busy && !sw && !k_int;             // a leaf reference: instance (TLAS) or triangles / shape (BLAS)
        const bool k_enter = k_ref && st.inst < 0;
        const bool k_leaf = k_ref && st.inst >= 0;
        const int n_int = __popcll(__ballot(k_int)), n_leaf = __popcll(__ballot(k_leaf)), n_enter = __popcll(__ballot(k_enter)),
                  n_sw = __popcll(__ballot(sw));
        if ((n_int | n_leaf | n_enter | n_sw) == 0) break;
        bool popped = false;
        if (n_int >= n_leaf && n_int >= n_enter && n_int >= n_sw) {
            if (k_int) {
                trav_interior<STATS, true>(sc, st, stk, cnt);   // descends into the nearest child or pops
                if (STATS) ++steps;
                popped = true;   // (cur may be a popped entry: run the fix-ups)
            }
        } else if (n_leaf >= n_enter && n_leaf >= n_sw) {
            if (k_leaf) {
                if (STATS) probe(cnt.oth_lane, cnt.oth_wave);
                const uint32_t ref = ~static_cast<uint32_t>(st.cur);
                const uint32_t first = ref >> 2, count = (ref & 3u) + 1u;
                bool accepted_any = false;
                if (EXT && first >= GBL_SHAPE_FIRST_DISK) {   // Model::intersect of an intersectable geometry (GoblinModel.cpp:46-54)
                    const float radius = sc.instances[st.inst].radius;
                    float t;
                    if (STATS) cnt.tris += 1;
                    const bool got = first == GBL_SHAPE_FIRST_SPHERE ? sphere_test(radius, st.r.o, st.r.d, st.mint, st.maxt, &t)
                                                                     : disk_test(radius, st.r.o, st.r.d, st.mint, st.maxt, &t);
                    if (got) {
                        accepted_any = true;
                        if (!any) {
                            st.maxt = t;
                            st.hit.t = t;
                            st.hit.inst = st.inst;
                            st.hit.tri = 0;
                            st.hit.b1 = st.hit.b2 = 0.0f;
                        }
                    }
                } else {
                    for (uint32_t i = 0; i < count; ++i) {
                        float t, b1, b2;
                        if (STATS) cnt.tris += 1;
                        bool take = tri_test(sc.tris + first + i, st.r.o, st.r.d, st.mint, st.maxt, &t, &b1, &b2) && !(any && accepted_any);
                        if (take) accepted_any = true;
#ifndef GBL_NO_TIE_RULE
                        if (TIES && take && !any && t == st.hit.t && st.hit.inst == st.inst && sc.tri_order != nullptr &&
                            !tie_goes_to(sc, st.hit.tri, first + i, st.r.o, st.r.d, st.mint, t))
                            take = false;
#endif
                        if (take && !any) {
                            st.maxt = t;
                            st.hit.t = t;
                            st.hit.inst = st.inst;
                            st.hit.tri = first + i;
                            st.hit.b1 = b1;
                            st.hit.b2 = b2;
                        }
                    }
                }
                if (any && accepted_any) {   // Scene::occluded: the first accepted primitive ends the query
                    occ = true;
                    st.cur = GBL_STACK_EXIT;
                } else {
                    st.cur = static_cast<int>(stk.load(--st.sp));
                    popped = true;
                }
            }
        } else if (n_enter >= n_sw) {
            if (k_enter) {
                if (STATS) probe(cnt.oth_lane, cnt.oth_wave);
                const uint32_t ref = ~static_cast<uint32_t>(st.cur);
                const DevInstance* ip = sc.instances + (ref >> 2);
                st.inst = static_cast<int>(ref >> 2);
                ray_space(st.r, xf_point(ip->inv, st.world.o), xf_vector(ip->inv, st.world.d));
                stk.store(st.sp++, GBL_STACK_SENTINEL);
                st.cur = ip->root;
            }
        } else {
            if (sw) {   // the extension ray of this job
                if (STATS) {
                    probe(cnt.oth_lane, cnt.oth_wave);
                    cnt.ext += 1;
                }
                st.world = ext;
                st.r = ext;
                st.maxt = INFINITY;
                st.sp = 0;
                stk.store(st.sp++, GBL_STACK_EXIT);
                st.cur = sc.tlas_root;
                st.inst = -1;
                any = false;
                sw = false;
            }
        }
        // ---- bookkeeping folded into the pop: the instance sentinel, then the exit marker
        if (popped && st.cur == GBL_STACK_SENTINEL) {
            st.r = st.world;
            st.inst = -1;
            st.cur = static_cast<int>(stk.load(--st.sp));
        }
        if (busy && !sw && st.cur == GBL_STACK_EXIT) {
            if (any && has_ext) sw = true; else busy = false;
        }
    }
    *occluded = occ;
    *hit = st.hit;
    if (!has_ext) hit->inst = -1;
    if (STATS && has_ext) {
        int b = steps <= 3 ? 0 : min(6, 30 - __clz(static_cast<int>(steps)));
        cnt.hist[b] += 1;
        cnt.hist_steps[b] += steps;
    }
}

template <bool REPLAY, bool STATS, bool EXT>
// (EXT builds carry the analytic shapes, texture graphs, image lookups (out-of-line calls), masks, the BSSRDF and medium hooks:
//  held to the lean build's 168 registers they spilled 300-1200 of them; two waves per SIMD (256 registers) hold them)
__global__ __launch_bounds__(GBL_BLOCK, EXT ? GBL_EXT_WAVES : GBL_PT_WAVES) void pair_trace_kernel(DevScene sc, RenderArgs ra) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* ctrl = reinterpret_cast<uint32_t*>(smem);
    uint32_t* stack = ctrl + 4;
    const LdsStack stk = {gbl_as_lds(stack + threadIdx.x)};
    constexpr bool TIES = REPLAY || STATS;   // the lean native build leaves the tie rule out (trace.h)

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
        __syncthreads();
        const uint32_t item = ctrl[0];
        if (item >= n_items) break;
        const ItemInfo it = decode_item(ra, item);

        PathState ps;
        ps.bounce = -1;
        ps.light = 0;
        ps.pick_pdf = 1.0f;
        bool active = false, exhausted = false;
        bool need_shadow = false, has_ext = false;
        F3 shadow_d = f3(0, 0, 1), contrib = f3(0, 0, 0);
        float shadow_maxt = 0.0f;
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
            int fetched = wave_fetch(!active && !exhausted, ctrl + 1);
            if (!active && !exhausted) {
                if (fetched >= 0 && fetched < it.paths) {
                    int pix = fetched / ra.chunk_spp;
                    src.k = static_cast<uint32_t>(it.k0 + fetched % ra.chunk_spp);
                    int px = it.px0 + pix % it.tw, py = it.py0 + pix / it.tw;
                    out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + src.k;
                    if (REPLAY) {
                        src.rec = ra.replay + static_cast<size_t>(out_index) * ra.dims;
                        image_x = src.rec[0];
                        image_y = src.rec[1];
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
                    ps.f = f3(0.0f, 0.0f, 0.0f);
                    ps.cosw = ps.fw = 0.0f;
                    ps.bsdf_pdf = 1.0f;
                    ps.bounce = -1;
                    need_shadow = false;
                    has_ext = true;
                    contrib = f3(0, 0, 0);
                    active = true;
                    if (STATS) cnt.dims += 2;
                } else {
                    exhausted = true;
                }
            }
            if (__ballot(active) == 0ull) break;

            // ---- the vertex's two rays, one job per lane
            bool finished = false, occluded = false;
            Hit hit;
            hit.inst = -1;
            if (active && sc.num_lights == 0) finished = true;   // PathTracer::Li returns Black without lights (:53-56)
            trace_pair<STATS, EXT, TIES>(sc, active && !finished && need_shadow, active && !finished && has_ext, ps.o, ps.mint, shadow_d, shadow_maxt,
                                         ps.d, stk, &occluded, &hit, cnt);
            const bool got = hit.inst >= 0;

            // ---- close the bounce whose rays were just traced
            Frag fr;
            TexFrag tf;
            if (active && !finished) {
                if (got) {
                    make_fragment<EXT>(sc, hit, ps.o, ps.d, fr, &tf);
                    if (EXT && sc.materials[sc.instances[hit.inst].material].has_tex != 0u)
                        hit_differentials<REPLAY>(sc, src, ps.bounce < 0, image_x, image_y, fr, tf);
                }
                if (ps.bounce < 0) {
                    if (!got) {
                        if (EXT) {   // Li += scene->evalEnvironmentLight(ray), GoblinPathtracer.cpp:61-65
                            const F3 le = environment_le<EXT>(sc, ps.d);
                            ps.Li = f3(ps.Li.x + le.x, ps.Li.y + le.y, ps.Li.z + le.z);
                        }
                        finished = true;
                    } else {
                        F3 le = hit_Le(sc, hit.inst, fr.n, -ps.d);
                        ps.Li = f3(ps.Li.x + le.x, ps.Li.y + le.y, ps.Li.z + le.z);
                        if (EXT && sc.has_bssrdf != 0) {   // Li += Lsubsurface, GoblinPathtracer.cpp:69 (computed ahead by sss_kernel)
                            const float4 ss = reinterpret_cast<const float4*>(ra.sss)[out_index];
                            ps.Li = f3(ps.Li.x + ss.x, ps.Li.y + ss.y, ps.Li.z + ss.z);
                        }
                        ps.bounce = 0;
                    }
                } else {
                    // the light sample's term joins Ld when its shadow ray found nothing (Ld starts at 0, :88-113) ...
                    ps.Ld = (need_shadow && !occluded) ? f3(0.0f + contrib.x, 0.0f + contrib.y, 0.0f + contrib.z) : f3(0.0f, 0.0f, 0.0f);
                    // ... then the MIS term of the sampled direction, Li and throughput
                    if (got && sc.instances[hit.inst].area_light == ps.light) {
                        F3 le = hit_Le(sc, hit.inst, fr.n, -ps.d);
                        if (!is_black(le)) {
                            // Ld += f * tr * Li * absdot(wi, n) * fWeight / bsdfPdf   (tr == 1 without masks)
                            F3 term = div(ps.f * le * ps.cosw * ps.fw, ps.bsdf_pdf);
                            ps.Ld = f3(ps.Ld.x + term.x, ps.Ld.y + term.y, ps.Ld.z + term.z);
                        }
                    } else if (EXT && !got && sc.has_ibl != 0) {
                        // the sampled direction left the scene: Ld += f * tr * light->Le(r) * fWeight / bsdfPdf (:157-161).  A job that only
                        // carried its shadow ray (f == 0, no extension ray) adds 0 here.
                        const F3 le = light_le_escaped<EXT>(sc, sc.lights[ps.light], ps.d);
                        const F3 term = div(ps.f * le * ps.fw, ps.bsdf_pdf);
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

            // ---- shade the vertex: light sample -> shadow ray, BSDF sample -> next ray
            need_shadow = false;
            has_ext = false;
            contrib = f3(0, 0, 0);
            if (active && !finished) {
                const int b = ps.bounce;
                F3 wo = -ps.d;
                float u_light_c, u_light_1, u_light_2, u_pick, u_bsdf_c, u_bsdf_1, u_bsdf_2;
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
                const DevMaterial* mat = sc.materials + sc.instances[hit.inst].material;
                ResolvedMat rmat;   // EXT: the hit material with its textures evaluated
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
                        if (light_is_delta<EXT>(light)) {
                            contrib = div(f * ls.L * absdot(fr.n, ls.wi), ls.pdf);
                        } else {
                            float bp = EXT ? rmat_pdf(rmat, fr.n, wo, ls.wi) : mat_pdf(*mat, fr.n, wo, ls.wi);
                            float lw = power_heuristic(ls.pdf, bp);
                            contrib = div(f * ls.L * absdot(fr.n, ls.wi) * lw, ls.pdf);
                        }
                    }
                }
                F3 wi;
                float pdf;
                bool specular, null_sampled = false;
                F3 f = EXT ? rmat_sample(rmat, fr, wo, u_bsdf_c, u_bsdf_1, u_bsdf_2, &wi, &pdf, &specular, &null_sampled)
                           : mat_sample(*mat, fr, wo, u_bsdf_c, u_bsdf_1, u_bsdf_2, &wi, &pdf, &specular);
                // Russian roulette (build-side extension, off in every parity mode): a killed path still collects this
                // vertex's direct light
                bool rr_killed = false;
                float rr_inv = 1.0f;
                if (ra.russian_roulette && !REPLAY && !is_black(f) && pdf > 0.0f && ps.bounce >= 2) {
                    F3 tn = ps.throughput * div(f * absdot(wi, fr.n), pdf);
                    float q = fminf(0.95f, fmaxf(tn.x, fmaxf(tn.y, tn.z)));
                    float u = nat_u01(nat_mix(nat_mix(src.pixel_key, 0xBADC0DEu + ps.bounce), src.k));
                    if (!(u < q)) rr_killed = true;
                    else rr_inv = 1.0f / q;   // carried by cosw
                }
                if (!rr_killed && !is_black(f) && pdf > 0.0f) {
                    float fw = 1.0f;
                    if (!specular) fw = power_heuristic(pdf, light_pdf<EXT>(sc, sc.lights[ps.light], fr.p, wi));
                    ps.f = f;
                    ps.fw = fw;
                    ps.bsdf_pdf = pdf;
                    ps.cosw = absdot(wi, fr.n);
                    if (ra.russian_roulette && !REPLAY && rr_inv != 1.0f) ps.cosw = ps.cosw * rr_inv;
                    ps.o = fr.p;
                    ps.d = wi;
                    ps.mint = fr.eps;
                    has_ext = true;
                } else if (need_shadow) {
                    // the path ends here but its direct light is still pending on the shadow ray: one more round with the
                    // shadow ray alone; f = 0 makes the close do exactly Li += throughput * Ld / pickPdf, and the missing
                    // extension ray ends the path there   (:163-167)
                    ps.f = f3(0, 0, 0);
                    ps.fw = 0.0f;
                    ps.bsdf_pdf = 1.0f;
                    ps.cosw = 0.0f;
                    ps.o = fr.p;
                    ps.mint = fr.eps;
                } else {
                    // Li += throughput * Ld / pickLightPdf with Ld == 0 (no shadow ray pending); break
                    F3 add = div(ps.throughput * f3(0.0f, 0.0f, 0.0f), ps.pick_pdf);
                    ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
                    finished = true;
                }
            }
            // ---- path end: publish the sample's radiance (RenderTask::run: w * (tr * L + Lv), w = 1; the splat kernel filters it)
            if (active && finished) {
                reinterpret_cast<float4*>(ra.li_defer)[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                active = false;
                need_shadow = false;
                has_ext = false;
                paths_done += 1;
            }
        }
    }
    if (STATS) accumulate_stats(ra, cnt, paths_done);
}
