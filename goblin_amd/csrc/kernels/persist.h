// The lean path kernel with PERSISTENT TRAVERSAL: a lane's ray survives the shading of the lanes beside it.
//
// path_trace_kernel's wave runs an iteration as two wave-wide queries (extension rays, then shadow rays) with the shading between and
// behind them, and every query lasts as long as its longest ray: the counters show half of the closest-hit query's time serving the
// last <= 16 rays of each query -- 9 % of the ray steps.  Here a wave alternates between two phases instead:
//
//   traverse   every lane that holds a ray takes traversal steps (the lean dense loop of trace.h: leaf / instance step, then interior
//              step).  A lane whose SHADOW ray ends folds the result into Ld and starts its extension ray on the spot -- the BSDF
//              sample that made that ray does not depend on the shadow result, so both rays of a vertex are known when it has been
//              shaded (PathTracer::Li, GoblinPathtracer.cpp:96-167: the order of the additions into Ld and Li is kept).  A lane whose
//              EXTENSION ray ends waits.  The phase ends when `persist_wait` lanes wait (or nobody traverses).
//   shade      the waiting lanes close their bounce; lanes without a path fetch one (the primary pass has traced the camera rays:
//              kernels_quad.hip primary_kernel -- a miss costs a Black, a hit is a first vertex); every lane that stands at a vertex
//              samples its light and its BSDF; then back to traversing.  The lanes that were still traversing keep their ray, their
//              stack (LDS) and their place in the tree across this phase.
//
// So the traversal loop always runs with most of the wave's lanes, there is no tail per query and no quad phase; the price is that
// the shading code runs for `persist_wait` lanes at a time instead of all that have work, and that a traversal state lives across it.
// Work items are taken per WAVE (nothing here is shared by the workgroup but the read-only tree top in LDS), and a wave whose item
// runs out fetches the next one while its other lanes are still in flight: lanes only drain at the end of the launch.
// Same arithmetic per path as path_trace_kernel<GBL_SRC_NATIVE, false, false, true> -- bit-identical radiance (tests/test_gpu_primary.py).
#pragma once
#include "render_kernels.h"

#define GBL_PL_IDLE 0      // no path (fetch one in the next shade phase)
#define GBL_PL_SHADE 1     // the extension ray has ended: close the bounce
#define GBL_PL_SHADOW_DONE 2   // the shadow ray has ended (`occluded`): waits for the lanes beside it to get that far
#define GBL_PL_SHADOW 3    // traversing the vertex's shadow ray
#define GBL_PL_EXT 4       // traversing the extension ray

__global__ __launch_bounds__(GBL_BLOCK, GBL_PT_WAVES) void path_persist_kernel(DevScene sc, RenderArgs ra) {
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS as the quad kernels lay it out (gbl_api.hip sizes it once for both): quad records (unused here) | ctrl | stacks | tree top
    uint32_t* ctrl = reinterpret_cast<uint32_t*>(smem) + GBL_QUAD_LDS_WORDS;
    uint32_t* stack = ctrl + 4;
    HotLdsStack stk;
    stk.p = gbl_as_lds(stack + threadIdx.x);
    {
        uint4* hot = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(smem) + ra.hot_word);
        for (uint32_t i = threadIdx.x; i < 4u * ra.hot_count; i += GBL_BLOCK) hot[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
        stk.hot = (const gbl_lds_u4*)hot;
        stk.hot_count = ra.hot_count;
    }
    __syncthreads();
    LaneCounters cnt = {};
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_items = static_cast<uint32_t>(ra.local_tiles) * ra.chunks;
    const int sub_w = ra.window[1] - ra.window[0];
    const int full_w = sc.film.window[1] - sc.film.window[0];
    const uint32_t wait_for = ra.persist_wait;
    float4* const li = reinterpret_cast<float4*>(ra.li_defer);

    // the wave's work item and how many of its samples have been handed out
    ItemInfo it;
    it.px0 = it.py0 = it.k0 = 0;
    it.tw = it.th = 1;
    it.paths = 0;
    int cursor = 0;
    bool no_more_items = false;

    PathState ps;
    ps.bounce = 0;
    ps.light = 0;
    ps.path = 0;
    ps.punch = false;
    ps.first = true;
    ps.o = ps.d = ps.throughput = ps.Li = ps.Ld = ps.f = f3(0.0f, 0.0f, 0.0f);
    ps.mint = ps.cosw = ps.fw = ps.bsdf_pdf = ps.pick_pdf = 0.0f;
    SampleSource src;
    src.spp = ra.spp;
    src.root = ra.root;
    src.rec = nullptr;
    src.pixel_key = 0;
    src.k = 0;
    uint32_t out_index = 0;
    int state = GBL_PL_IDLE;
    bool exhausted = false;          // idle for good: the launch has no sample left for this lane
    bool occluded = false;           // GBL_PL_SHADOW_DONE: what the shadow ray found
    bool final_after_shadow = false; // the path ended at its vertex; its radiance is complete once the shadow ray has reported
    F3 sh_d = f3(0.0f, 0.0f, 1.0f), contrib = f3(0.0f, 0.0f, 0.0f);
    float sh_maxt = 0.0f;
    TravState st;
    trav_begin(sc, st, f3(0.0f, 0.0f, 0.0f), f3(0.0f, 0.0f, 1.0f), 0.0f, 0.0f, stk);
    st.cur = GBL_STACK_EXIT;

#ifdef GBL_PERSIST_CLOCK
    unsigned long long pk[12] = {};
    const unsigned long long pk_k0 = __builtin_amdgcn_s_memtime();
#define PK_NOW() __builtin_amdgcn_s_memtime()
#endif
    for (;;) {
        // ================= traverse
        __builtin_amdgcn_s_setprio(GBL_QUAD_PRIO_DENSE);
#ifdef GBL_PERSIST_CLOCK
        unsigned long long pk_t = PK_NOW();
#endif
        for (;;) {
            const bool trav = state >= GBL_PL_SHADOW;
            const unsigned long long tm = __ballot(trav);
            // the lanes whose shadow ray has ended go on together (a ray's set-up is three divisions: not something to run for
            // one lane at a time between two traversal steps): Ld takes the light sample, the extension ray starts
            const uint32_t reported = static_cast<uint32_t>(__popcll(__ballot(state == GBL_PL_SHADOW_DONE)));
            if (reported >= ra.persist_switch || (tm == 0ull && reported != 0u)) {
#ifdef GBL_PERSIST_CLOCK
                const unsigned long long s0 = PK_NOW();
                pk[0] += s0 - pk_t;
                pk[4] += 1;
                pk[5] += reported;
#endif
                if (state == GBL_PL_SHADOW_DONE) {
                    if (!occluded) ps.Ld = f3(ps.Ld.x + contrib.x, ps.Ld.y + contrib.y, ps.Ld.z + contrib.z);
                    if (final_after_shadow) {
                        // Li += throughput * Ld / pickLightPdf; break   (:163-167)
                        const F3 add = div(ps.throughput * ps.Ld, ps.pick_pdf);
                        ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
                        li[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                        state = GBL_PL_IDLE;
                    } else {
                        trav_begin(sc, st, ps.o, ps.d, ps.mint, INFINITY, stk);
                        state = GBL_PL_EXT;
                    }
                }
#ifdef GBL_PERSIST_CLOCK
                asm volatile("" ::"v"(st.cur), "v"(state));
                pk_t = PK_NOW();
                pk[3] += pk_t - s0;
#endif
                continue;
            }
            if (tm == 0ull) break;
            const uint32_t waiting = static_cast<uint32_t>(__popcll(__ballot(state == GBL_PL_SHADE || (state == GBL_PL_IDLE && !exhausted))));
            if (waiting >= wait_for) break;
#ifdef GBL_PERSIST_CLOCK
            pk[1] += 1;
            pk[2] += __popcll(tm);
#endif
            // one KIND of step per iteration (the wave's rays are at unrelated places in the tree: running both kinds every time makes
            // every branch of both run for a few lanes each -- wavefront.h wf_trace phases the same way): interior steps while
            // GBL_TRAV_TH lanes stand at interior nodes or nobody waits at a leaf, else the leaf / instance step
            const bool at_int = trav && trav_at_interior(st), at_oth = trav && !at_int;
            const unsigned long long mi = __ballot(at_int), mo = __ballot(at_oth);
            if (mo == 0ull || static_cast<uint32_t>(__popcll(mi)) >= ra.persist_th) {
                if (at_int) trav_interior<false, true>(sc, st, stk, cnt);
            } else if (at_oth) {
                bool occ = false;
                if (trav_other_kind<false, false, HotLdsStack, GBL_TIE_NONE, true>(sc, st, stk, cnt, state == GBL_PL_SHADOW, &occ, GBL_FILTER_NONE)) {
                    occluded = occ;
                    state = state == GBL_PL_SHADOW ? GBL_PL_SHADOW_DONE : GBL_PL_SHADE;
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef GBL_PERSIST_CLOCK
        {
            asm volatile("" ::"v"(st.cur), "v"(state));
            const unsigned long long s1 = PK_NOW();
            pk[0] += s1 - pk_t;
            pk_t = s1;
            pk[7] += 1;
            pk[9] += __popcll(__ballot(state == GBL_PL_SHADE));
        }
#endif

        // ================= shade
        bool at_vertex = false;   // the lane stands at a vertex to be shaded (fr, hit)
        Hit hit = st.hit;
        Frag fr;
        fr.p = fr.n = f3(0.0f, 0.0f, 1.0f);
        fr.eps = 0.0f;
        // ---- the lanes whose extension ray has ended: close the bounce (or, for a camera ray the primary pass handed back, open the path)
        if (state == GBL_PL_SHADE) {
            const bool got = hit.inst >= 0;
            bool finished = false;
            if (got) make_fragment<false>(sc, hit, ps.o, ps.d, fr);
            if (ps.bounce < 0) {
                if (!got) {
                    finished = true;
                } else {
                    const F3 le = hit_Le(sc, hit.inst, fr.n, -ps.d);
                    ps.Li = f3(ps.Li.x + le.x, ps.Li.y + le.y, ps.Li.z + le.z);
                    ps.bounce = 0;
                }
            } else {
                // MIS term for the sampled direction, then Li and throughput
                if (got && sc.instances[hit.inst].area_light == ps.light) {
                    const F3 le = hit_Le(sc, hit.inst, fr.n, -ps.d);
                    if (!is_black(le)) {
                        const F3 term = div(ps.f * le * ps.cosw * ps.fw, ps.bsdf_pdf);
                        ps.Ld = f3(ps.Ld.x + term.x, ps.Ld.y + term.y, ps.Ld.z + term.z);
                    }
                }
                const F3 add = div(ps.throughput * ps.Ld, ps.pick_pdf);
                ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
                const F3 scale = div(ps.f * ps.cosw, ps.bsdf_pdf);
                ps.throughput = ps.throughput * scale;
                ps.bounce += 1;
                if (!got) finished = true;
            }
            if (!finished && ps.bounce >= ra.max_depth - 1) finished = true;
            if (finished) {
                li[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                state = GBL_PL_IDLE;
            } else {
                at_vertex = true;
            }
        }
#ifdef GBL_PERSIST_CLOCK
        asm volatile("" ::"v"(state), "v"(ps.Li.x));
        const unsigned long long pk_f0 = PK_NOW();
#endif
        // ---- the lanes without a path: the next samples of the wave's item (of the next item, when this one is used up)
        for (;;) {
            const bool idle = state == GBL_PL_IDLE && !exhausted;
            const unsigned long long im = __ballot(idle);
            if (im == 0ull) break;
            if (cursor >= it.paths) {   // (wave-uniform)
                uint32_t item = n_items;
                if (!no_more_items) {
                    if (lane == 0) item = atomicAdd(ra.work_counter, 1u);
                    item = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(item)));
                }
                if (item >= n_items) {
                    no_more_items = true;
                    if (idle) exhausted = true;
                    break;
                }
                it = decode_item(ra, item);
                cursor = 0;
            }
            const int rank = static_cast<int>(__popcll(im & ((1ull << lane) - 1ull)));
            const int left = it.paths - cursor;
            const int f = cursor + rank;
            cursor += min(left, static_cast<int>(__popcll(im)));
            if (!idle || rank >= left) continue;
            const int pix = f / ra.chunk_spp;
            src.k = static_cast<uint32_t>(it.k0 + f % ra.chunk_spp);
            const int px = it.px0 + pix % it.tw, py = it.py0 + pix / it.tw;
            out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + src.k;
            hit.inst = ra.prim_inst[out_index];
            if (hit.inst == GBL_PRIM_MISS) {   // PathTracer::Li without a hit: Black (:58-66; no image based light in the lean build)
                li[out_index] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
                continue;
            }
            const uint32_t pixel = static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0]));
            src.pixel_key = nat_mix(ra.seed_key, pixel);
            float u, v;
            src.native_2d(0u, 1u, 0u, false, &u, &v);
            camera_ray<false>(sc.camera, px + u, py + v, 0.0f, 0.0f, &ps.o, &ps.d, &ps.mint);
            ps.throughput = f3(1.0f, 1.0f, 1.0f);
            ps.Li = f3(0.0f, 0.0f, 0.0f);
            ps.path = static_cast<uint32_t>(f);
            final_after_shadow = false;
            if (hit.inst == GBL_PRIM_TIED) {   // the packet could not answer for this camera ray: traced here like any other ray
                ps.bounce = -1;
                trav_begin(sc, st, ps.o, ps.d, ps.mint, INFINITY, stk);
                state = GBL_PL_EXT;
                continue;
            }
            const float4 h = reinterpret_cast<const float4*>(ra.prim_hit)[out_index];
            hit.t = h.x;
            hit.b1 = h.y;
            hit.b2 = h.z;
            hit.tri = __float_as_uint(h.w);
            make_fragment<false>(sc, hit, ps.o, ps.d, fr);
            const F3 le = hit_Le(sc, hit.inst, fr.n, -ps.d);
            ps.Li = f3(ps.Li.x + le.x, ps.Li.y + le.y, ps.Li.z + le.z);
            ps.bounce = 0;
            if (ps.bounce >= ra.max_depth - 1) {
                li[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                continue;   // (still idle: fetches again)
            }
            state = GBL_PL_SHADE;   // (any state but IDLE: the lane has its path)
            at_vertex = true;
        }
#ifdef GBL_PERSIST_CLOCK
        asm volatile("" ::"v"(state), "v"(ps.Li.x));
        pk[11] += PK_NOW() - pk_f0;
        pk[10] += __popcll(__ballot(at_vertex));
#endif
        // ---- every lane at a vertex: light sample (its shadow ray) and BSDF sample (its extension ray)
        if (at_vertex) {
            const F3 wo = -ps.d;
            const int b = ps.bounce;
            const float u_light_c = src.native_1d(3u * b + 0u);
            const float u_bsdf_c = src.native_1d(3u * b + 1u);
            const float u_pick = src.native_1d(3u * b + 2u);
            float u_light_1, u_light_2, u_bsdf_1, u_bsdf_2;
            src.native_2d(0x10000u + 2u * b, 1u, 0u, true, &u_light_1, &u_light_2);
            src.native_2d(0x10000u + 2u * b + 1u, 1u, 0u, true, &u_bsdf_1, &u_bsdf_2);
            // Scene::sampleLight: CDF1D::sampleDiscrete over the power distribution
            int lsel = 0;
            for (int i = 1; i <= sc.num_lights; ++i)
                if (sc.light_cdf[i] < u_pick) lsel = i;
            if (lsel >= sc.num_lights) lsel = sc.num_lights - 1;
            ps.light = lsel;
            ps.pick_pdf = sc.light_pick_pdf[lsel];
            ps.Ld = f3(0, 0, 0);
            const DevMaterial* mat = sc.materials + sc.instances[hit.inst].material;
            const DevLight& light = sc.lights[lsel];
            LightSampleOut ls;
            light_sample<false>(sc, light, fr.p, fr.eps, u_light_c, u_light_1, u_light_2, ls);
            bool need_shadow = false;
            if (!is_black(ls.L) && ls.pdf > 0.0f) {
                const F3 f = mat_bsdf(*mat, fr.n, wo, ls.wi);
                if (!is_black(f)) {
                    need_shadow = true;
                    sh_d = ls.wi;
                    sh_maxt = ls.maxt;
                    if (light_is_delta<false>(light)) {
                        contrib = div(f * ls.L * absdot(fr.n, ls.wi), ls.pdf);
                    } else {
                        const float bp = mat_pdf(*mat, fr.n, wo, ls.wi);
                        const float lw = power_heuristic(ls.pdf, bp);
                        contrib = div(f * ls.L * absdot(fr.n, ls.wi) * lw, ls.pdf);
                    }
                }
            }
            // BSDF sample: the next ray
            F3 wi;
            float pdf;
            bool specular;
            bool goes_on = false;
            const F3 f = mat_sample(*mat, fr, wo, u_bsdf_c, u_bsdf_1, u_bsdf_2, &wi, &pdf, &specular);
            if (!is_black(f) && pdf > 0.0f) {
                float fw = 1.0f;
                if (!specular) fw = power_heuristic(pdf, light_pdf<false>(sc, sc.lights[ps.light], fr.p, wi));
                ps.f = f;
                ps.fw = fw;
                ps.bsdf_pdf = pdf;
                ps.cosw = absdot(wi, fr.n);
                ps.o = fr.p;
                ps.d = wi;
                ps.mint = fr.eps;
                goes_on = true;
                if (ra.russian_roulette && ps.bounce >= 2) {   // build-side extension, off in every parity mode (render_kernels.h)
                    const F3 tn = ps.throughput * div(ps.f * ps.cosw, ps.bsdf_pdf);
                    const float q = fminf(0.95f, fmaxf(tn.x, fmaxf(tn.y, tn.z)));
                    const float u = nat_u01(nat_mix(nat_mix(src.pixel_key, 0xBADC0DEu + ps.bounce), src.k));
                    if (!(u < q)) goes_on = false;   // ends after accounting this vertex's direct light
                    else ps.f = ps.f * (1.0f / q);
                }
            }
            final_after_shadow = !goes_on;
            if (need_shadow) {
                trav_begin(sc, st, fr.p, sh_d, fr.eps, sh_maxt, stk);
                state = GBL_PL_SHADOW;
            } else if (goes_on) {
                trav_begin(sc, st, ps.o, ps.d, ps.mint, INFINITY, stk);
                state = GBL_PL_EXT;
            } else {
                // Li += throughput * Ld / pickLightPdf with nothing pending; break   (:163-167)
                const F3 add = div(ps.throughput * ps.Ld, ps.pick_pdf);
                ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
                li[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                state = GBL_PL_IDLE;
            }
        }
#ifdef GBL_PERSIST_CLOCK
        asm volatile("" ::"v"(state), "v"(st.cur));
        pk[6] += PK_NOW() - pk_t;
#endif
        // a lane that ended its path in this phase fetches in the next one; the wave is done when nobody holds a ray and nothing is left
        if (__ballot(state != GBL_PL_IDLE || !exhausted) == 0ull) break;
    }
#ifdef GBL_PERSIST_CLOCK
    if (lane == 0) {
        pk[8] = 0;
        for (int i = 0; i < 12; ++i) atomicAdd(ra.stats + i, pk[i]);
        atomicAdd(ra.stats + 30, PK_NOW() - pk_k0);
    }
#endif
}
