// Wave-pool schedule: the wavefront formulation at the scope of ONE WAVE, inside one persistent kernel.
//
// The megakernel keeps a path in a lane from birth to death, so a wave's traversal loop runs until its slowest ray is
// done (16 % of the lanes active in an interior step on the headline scene, profiles/r01_v4_probes.txt) and its shading
// code runs for whichever lanes happen to have finished.  The wavefront schedule fixes both by moving all path state
// through an 8 M-slot pool in HBM between three kernels per iteration.  Here every wave owns a small pool of
// WP_SLOTS path slots (cache-resident: 18 KB per wave) and is its own wavefront machine -- no barrier, no atomic,
// no launch between the stages, nothing shared with any other wave:
//
//   * TRACE   the lanes hold one ray each.  A lane that finishes its ray goes idle; when WP_REFILL lanes are idle the
//             wave leaves the traversal loop, the finished lanes write their hit into their slot and append the slot to
//             the wave's `ready` list, and every idle lane takes the next entry of the wave's job queue.  Lanes that
//             are still traversing keep their traversal state in registers across all of this.
//   * SHADE   as soon as 64 slots are ready, the wave shades them, one slot per lane, ALL lanes active: close the
//             bounce whose rays were traced (MIS term, Li, throughput), sample the light and the BSDF, store the
//             shadow and the extension ray in the slot and push it on the job queue -- or, if the path ended, write
//             its radiance and start the slot's next camera path from the wave's current work item.
//
// A job is a slot; its lane traces the shadow ray (any-hit) and then the extension ray (closest hit) of that slot, so a
// slot is shaded again exactly when both are known.  Queues live in LDS and are wave-private: positions come from
// __ballot + prefix popcount, heads and tails are wave-uniform registers.
//
// The arithmetic is the megakernel's, statement for statement (shade.h / trace.h, same operation order): per-sample
// radiance is bit-identical between the three schedules, and the parity suite runs every case under each.
//
// Replaces RenderTask::run + PathTracer::Li (GoblinRenderer.cpp:29-52, GoblinPathtracer.cpp:50-179), like
// path_trace_kernel.  Mask scenes stay on the megakernel (their filtered queries and attenuation walks are not here).
#pragma once
#include "../device_scene.h"
#include "render_kernels.h"
#include "sampler.h"
#include "shade.h"
#include "trace.h"
#include "vecmath.h"

#ifndef WP_SLOTS
#define WP_SLOTS 128   // path slots per wave (power of two, >= 128: 64 in the lanes + 64 gathering for the next shade)
#endif
#ifndef WP_REFILL
#define WP_REFILL 16   // leave the traversal loop when this many lanes are idle
#endif
#ifndef WP_TRAV_TH
#define WP_TRAV_TH 24  // interior step while at least this many lanes sit at interior nodes (see GBL_TRAV_TH)
#endif
#ifndef GBL_WP_WAVES
#define GBL_WP_WAVES 3
#endif
#define WP_FIELDS 9    // float4 fields per slot
#define WP_BOUNCE_EMPTY (-2)
#define WP_BOUNCE_DEAD (-3)
#define WP_JOB_SHADOW 0x100u
#define WP_JOB_EXT 0x200u

// pool fields of one wave: field f, slot s at pool[f * WP_SLOTS + s]
//   0 ray_o   o.xyz, mint                 (extension / camera ray; the shadow ray starts at the same point)
//   1 ray_d   d.xyz, -
//   2 sh_d    shadow d.xyz, maxt
//   3 sh_c    contrib.xyz (the light sample's term, added to Ld when unoccluded), .w = trace result: (inst + 1) | occluded << 31
//   4 s_thr   throughput.xyz, cosw
//   5 s_li    Li.xyz, fw
//   6 s_f     f.xyz, bsdf_pdf
//   7 hit     t, b1, b2, as_float(tri)
//   8 s_id    light | (bounce + 4) << 16, out_index, k, pixel_key

__device__ __forceinline__ void wp_sync() {
    // LDS and pool traffic between lanes of ONE wave: order it, nothing more (no other wave ever reads this memory)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Pool state: written once, read once a few microseconds later.  Plain accesses: it is re-read while still in L2
// (non-temporal accesses, WP_NT, were measured: 104 ms against 82 ms on config 2 -- they send every access to HBM).
typedef float wp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 wp_ld(const float4* p) {
#ifdef WP_NT
    const wp_f4 v = __builtin_nontemporal_load(reinterpret_cast<const wp_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void wp_st(float4* p, float4 v) {
#ifdef WP_NT
    const wp_f4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<wp_f4*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ uint4 wp_ldu(const uint4* p) {
    const float4 v = wp_ld(reinterpret_cast<const float4*>(p));
    return make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
}
__device__ __forceinline__ void wp_stu(uint4* p, uint4 v) {
    wp_st(reinterpret_cast<float4*>(p), make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
}

__device__ __forceinline__ uint32_t wp_bcast_first(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }

template <bool REPLAY, bool STATS, bool EXT>
__global__ __launch_bounds__(GBL_BLOCK, EXT ? GBL_EXT_WAVES : GBL_WP_WAVES) void wp_kernel(DevScene sc, RenderArgs ra) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem);
    const LdsStack stk = {gbl_as_lds(lds + threadIdx.x)};
    const int lane = threadIdx.x & 63;
    const uint32_t wave = wp_bcast_first(threadIdx.x >> 6);
    uint32_t* jobq = lds + sc.stack_entries * GBL_BLOCK + wave * 2u * WP_SLOTS;
    uint32_t* ready = jobq + WP_SLOTS;
    const uint32_t wave_gid = wp_bcast_first(blockIdx.x * (GBL_BLOCK / 64) + wave);
    float4* pool = reinterpret_cast<float4*>(ra.wp_pool) + static_cast<size_t>(wave_gid) * WP_FIELDS * WP_SLOTS;
    float4* const f_ray_o = pool;
    float4* const f_ray_d = pool + 1 * WP_SLOTS;
    float4* const f_sh_d = pool + 2 * WP_SLOTS;
    float4* const f_sh_c = pool + 3 * WP_SLOTS;
    float4* const f_thr = pool + 4 * WP_SLOTS;
    float4* const f_li = pool + 5 * WP_SLOTS;
    float4* const f_f = pool + 6 * WP_SLOTS;
    float4* const f_hit = pool + 7 * WP_SLOTS;
    uint4* const f_id = reinterpret_cast<uint4*>(pool + 8 * WP_SLOTS);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    constexpr uint32_t QM = WP_SLOTS - 1u;
    constexpr bool TIES = REPLAY || STATS;   // the lean native build leaves the tie rule out, like the lean megakernel (trace.h)

    LaneCounters cnt = {};
    uint32_t paths_done = 0;
    const uint32_t n_items = static_cast<uint32_t>(ra.local_tiles) * ra.chunks;
    const int sub_w = ra.window[1] - ra.window[0];
    const int full_w = sc.film.window[1] - sc.film.window[0];

    // every slot starts empty and ready to be shaded (= given its first camera path)
    for (uint32_t s = lane; s < WP_SLOTS; s += 64u) {
        wp_stu(&f_id[s], make_uint4(static_cast<uint32_t>(WP_BOUNCE_EMPTY + 4) << 16, 0u, 0u, 0u));
        ready[s] = s;
    }
    uint32_t jq_head = 0, jq_tail = 0, rd_head = 0, rd_tail = WP_SLOTS;   // wave-uniform, monotonic; index & QM
    uint32_t n_dead = 0;                                                    // slots that found no path left to start
    // the wave's current work item (tile x sample chunk) and how far it has been handed out
    ItemInfo it = {};
    uint32_t it_cursor = 0, it_paths = 0;
    bool items_left = true;
    wp_sync();

    // ---- per-lane trace state
    TravState st;
    st.sp = 0;
    st.cur = GBL_STACK_EXIT;
    st.inst = -1;
    st.mint = st.maxt = 0.0f;
    bool busy = false, have_job = false, any = false, has_ext = false, occl = false;
    uint32_t job_slot = 0;

    unsigned long long tick = STATS ? wall_clock64() : 0ull;   // probes: time per phase (publish + shade, refill, traverse)
    for (;;) {
        // =====================================================================================================
        // 1. lanes whose job is done publish its result; lanes between the two rays of a job move on to the second
        // =====================================================================================================
        {
            const bool fin = !busy && have_job && (!any || !has_ext);
            if (fin) {
                // (a job without an extension ray is a path that ended at this vertex with its light sample still to be
                //  tested: the close below sees a miss)
                const bool got = !any && st.hit.inst >= 0;
                if (got) wp_st(&f_hit[job_slot], make_float4(st.hit.t, st.hit.b1, st.hit.b2, __uint_as_float(st.hit.tri)));
                reinterpret_cast<uint32_t*>(f_sh_c + job_slot)[3] = (got ? static_cast<uint32_t>(st.hit.inst) + 1u : 0u) | (occl ? 0x80000000u : 0u);
            }
            const unsigned long long fm = __ballot(fin);
            if (fin) {
                ready[(rd_tail + static_cast<uint32_t>(__popcll(fm & lt_mask))) & QM] = job_slot;
                have_job = false;
            }
            rd_tail += static_cast<uint32_t>(__popcll(fm));
        }
        wp_sync();

        // =====================================================================================================
        // 2. shade a batch: 64 ready slots, or whatever is ready when the lanes have nothing else to do
        // =====================================================================================================
        for (;;) {
            const uint32_t n_ready = rd_tail - rd_head;
            const uint32_t n_jobs = jq_tail - jq_head;
            const uint32_t n_busy = static_cast<uint32_t>(__popcll(__ballot(busy || have_job)));
            if (n_ready == 0u) break;
            if (n_ready < 64u && (n_jobs > 0u || n_busy >= 64u - WP_REFILL + 1u)) break;   // keep gathering
            const uint32_t n_batch = min(n_ready, 64u);
            const bool on = static_cast<uint32_t>(lane) < n_batch;
            if (STATS && lane == 0) {   // probes: shade batches and the slots they shaded
                cnt.hist[1] += 1;
                cnt.hist[2] += n_batch;
            }
            const uint32_t slot = on ? ready[(rd_head + static_cast<uint32_t>(lane)) & QM] : 0u;
            rd_head += n_batch;

            PathState ps;
            ps.bounce = WP_BOUNCE_DEAD;
            ps.light = 0;
            ps.path = 0;
            ps.punch = false;
            uint32_t out_index = 0, k = 0, pixel_key = 0;
            float4 sh_c4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (on) {
                const uint4 id = wp_ldu(&f_id[slot]);
                ps.light = static_cast<int>(id.x & 0xffffu);
                ps.bounce = static_cast<int>(id.x >> 16) - 4;
                out_index = id.y;
                k = id.z;
                pixel_key = id.w;
            }
            const bool alive = ps.bounce >= -1;
            bool finished = false;
            Hit hit;
            hit.inst = -1;
            hit.t = INFINITY;
            hit.tri = 0;
            hit.b1 = hit.b2 = 0.0f;
            bool got = false, occluded = false;
            Frag fr;
            TexFrag tf;
            if (alive) {
                const float4 o = wp_ld(&f_ray_o[slot]), d = wp_ld(&f_ray_d[slot]);
                const float4 a = wp_ld(&f_thr[slot]), b = wp_ld(&f_li[slot]), e = wp_ld(&f_f[slot]);
                sh_c4 = wp_ld(&f_sh_c[slot]);
                ps.o = f3(o.x, o.y, o.z); ps.mint = o.w;
                ps.d = f3(d.x, d.y, d.z);
                ps.throughput = f3(a.x, a.y, a.z); ps.cosw = a.w;
                ps.Li = f3(b.x, b.y, b.z); ps.fw = b.w;
                ps.f = f3(e.x, e.y, e.z); ps.bsdf_pdf = e.w;
                const uint32_t res = __float_as_uint(sh_c4.w);
                occluded = (res & 0x80000000u) != 0u;
                hit.inst = static_cast<int>(res & 0x7fffffffu) - 1;
                got = hit.inst >= 0;
                if (got) {
                    const float4 h = wp_ld(&f_hit[slot]);
                    hit.t = h.x; hit.b1 = h.y; hit.b2 = h.z; hit.tri = __float_as_uint(h.w);
                }
                ps.pick_pdf = sc.num_lights > 0 ? sc.light_pick_pdf[ps.light] : 1.0f;
                // the light sample's term joins Ld when its shadow ray found nothing (Ld starts at 0, :88-113)
                ps.Ld = occluded ? f3(0.0f, 0.0f, 0.0f) : f3(0.0f + sh_c4.x, 0.0f + sh_c4.y, 0.0f + sh_c4.z);
            }
            SampleSource src;
            src.spp = ra.spp;
            src.root = ra.root;
            src.rec = nullptr;
            src.pixel_key = pixel_key;
            src.k = k;
            if (REPLAY && alive) src.rec = ra.replay + static_cast<size_t>(out_index) * ra.dims;

            // ---- close the bounce whose rays were just traced (the megakernel's statements)
            if (alive) {
                if (sc.num_lights == 0) {
                    finished = true;   // PathTracer::Li returns Black without lights (:53-56)
                } else {
                    if (got) {
                        make_fragment<EXT>(sc, hit, ps.o, ps.d, fr, &tf);
                        if (EXT && sc.materials[sc.instances[hit.inst].material].has_tex != 0u) {
                            float image_x = 0.0f, image_y = 0.0f;
                            if (ps.bounce < 0) {   // the camera sample this path started from
                                if (REPLAY) {
                                    image_x = src.rec[0];
                                    image_y = src.rec[1];
                                } else {
                                    const uint32_t pix = out_index / static_cast<uint32_t>(ra.spp);
                                    float u, v;
                                    src.native_2d(0u, 1u, 0u, false, &u, &v);
                                    image_x = (ra.window[0] + static_cast<int>(pix % sub_w)) + u;
                                    image_y = (ra.window[2] + static_cast<int>(pix / sub_w)) + v;
                                }
                            }
                            hit_differentials<REPLAY>(sc, src, ps.bounce < 0, image_x, image_y, fr, tf);
                        }
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
                            if (EXT && sc.has_bssrdf != 0) {   // Li += Lsubsurface (kernels/subsurface.h), GoblinPathtracer.cpp:69
                                const float4 ss = reinterpret_cast<const float4*>(ra.sss)[out_index];
                                ps.Li = f3(ps.Li.x + ss.x, ps.Li.y + ss.y, ps.Li.z + ss.z);
                            }
                            ps.bounce = 0;
                        }
                    } else {
                        if (got && sc.instances[hit.inst].area_light == ps.light) {
                            F3 le = hit_Le(sc, hit.inst, fr.n, -ps.d);
                            if (!is_black(le)) {
                                // Ld += f * tr * Li * absdot(wi, n) * fWeight / bsdfPdf   (tr == 1 without masks)
                                F3 term = EXT ? div(ps.f * f3(1.0f, 1.0f, 1.0f) * le * ps.cosw * ps.fw, ps.bsdf_pdf) : div(ps.f * le * ps.cosw * ps.fw, ps.bsdf_pdf);
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
            }

            // ---- shade the vertex: light sample -> shadow ray, BSDF sample -> next ray
            bool need_shadow = false, has_ray = false;
            F3 shadow_d = f3(0, 0, 1), contrib = f3(0, 0, 0);
            float shadow_maxt = 0.0f;
            if (alive && !finished) {
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
                // Russian roulette (build-side extension, off in every parity mode): as in the megakernel, a killed path
                // still collects this vertex's direct light
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
                    has_ray = true;
                } else if (need_shadow) {
                    // The path ends here but its direct light is still pending on the shadow ray: the slot stays one more
                    // round with a shadow-only job; f = 0 makes the next close do exactly Li += throughput * Ld / pickPdf
                    // and the missing extension ray ends the path there.
                    ps.f = f3(0, 0, 0);
                    ps.fw = 0.0f;
                    ps.bsdf_pdf = 1.0f;
                    ps.cosw = 0.0f;
                    ps.o = fr.p;
                    ps.mint = fr.eps;
                } else {
                    // Li += throughput * Ld / pickLightPdf with Ld == 0 (no shadow ray pending); break   (:163-167)
                    F3 add = div(ps.throughput * f3(0.0f, 0.0f, 0.0f), ps.pick_pdf);
                    ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
                    finished = true;
                }
            }

            // ---- path end: publish the sample's radiance (RenderTask::run: w * (tr * L + Lv), w = 1; the splat kernel
            //      filters it into the film), then start the slot's next camera path
            if (alive && finished) {
                reinterpret_cast<float4*>(ra.li_defer)[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
                paths_done += 1;
                need_shadow = false;
            }
            bool started = false;
            bool want_new = on && ((alive && finished) || ps.bounce == WP_BOUNCE_EMPTY);
            {
                unsigned long long want = __ballot(want_new);
                while (want != 0ull) {
                    if (it_cursor >= it_paths) {
                        if (!items_left) break;
                        uint32_t item = 0;
                        if (lane == 0) item = atomicAdd(ra.work_counter, 1u);
                        item = wp_bcast_first(item);
                        if (item >= n_items) {
                            items_left = false;
                            break;
                        }
                        it = decode_item(ra, item);
                        it_cursor = 0;
                        it_paths = static_cast<uint32_t>(it.paths);
                        continue;
                    }
                    const uint32_t rank = static_cast<uint32_t>(__popcll(want & lt_mask));
                    const uint32_t take = min(it_paths - it_cursor, static_cast<uint32_t>(__popcll(want)));
                    if (want_new && rank < take) {
                        const uint32_t fetched = it_cursor + rank;
                        const int pix = static_cast<int>(fetched) / ra.chunk_spp;
                        k = static_cast<uint32_t>(it.k0 + static_cast<int>(fetched) % ra.chunk_spp);
                        const int px = it.px0 + pix % it.tw, py = it.py0 + pix / it.tw;
                        out_index = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0])) * ra.spp + k;
                        src.k = k;
                        float image_x, image_y;
                        if (REPLAY) {
                            src.rec = ra.replay + static_cast<size_t>(out_index) * ra.dims;
                            image_x = src.rec[0];
                            image_y = src.rec[1];
                        } else {
                            const uint32_t pixel = static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0]));
                            pixel_key = nat_mix(ra.seed_key, pixel);
                            src.pixel_key = pixel_key;
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
                        ps.light = 0;
                        ps.bounce = -1;
                        started = true;
                        has_ray = true;
                        want_new = false;
                        if (STATS) cnt.dims += 2;
                    }
                    it_cursor += take;
                    want = __ballot(want_new);
                }
            }
            const bool dead = want_new;   // no path left for this slot: it leaves the pool
            n_dead += static_cast<uint32_t>(__popcll(__ballot(dead)));

            // ---- write back, enqueue
            const bool keep = (alive && !finished) || started;
            if (keep) {
                wp_st(&f_ray_o[slot], make_float4(ps.o.x, ps.o.y, ps.o.z, ps.mint));
                if (has_ray) wp_st(&f_ray_d[slot], make_float4(ps.d.x, ps.d.y, ps.d.z, 0.0f));
                if (need_shadow) wp_st(&f_sh_d[slot], make_float4(shadow_d.x, shadow_d.y, shadow_d.z, shadow_maxt));
                wp_st(&f_sh_c[slot], make_float4(contrib.x, contrib.y, contrib.z, 0.0f));
                wp_st(&f_thr[slot], make_float4(ps.throughput.x, ps.throughput.y, ps.throughput.z, ps.cosw));
                wp_st(&f_li[slot], make_float4(ps.Li.x, ps.Li.y, ps.Li.z, ps.fw));
                wp_st(&f_f[slot], make_float4(ps.f.x, ps.f.y, ps.f.z, ps.bsdf_pdf));
                wp_stu(&f_id[slot], make_uint4(static_cast<uint32_t>(ps.light) | (static_cast<uint32_t>(ps.bounce + 4) << 16), out_index, k, pixel_key));
            }
            {
                const unsigned long long km = __ballot(keep);
                if (keep)
                    jobq[(jq_tail + static_cast<uint32_t>(__popcll(km & lt_mask))) & QM] =
                        slot | (need_shadow ? WP_JOB_SHADOW : 0u) | (has_ray ? WP_JOB_EXT : 0u);
                jq_tail += static_cast<uint32_t>(__popcll(km));
            }
            wp_sync();
        }

        // =====================================================================================================
        // 3. idle lanes take the next jobs; every lane without a ray in flight starts the next ray of its job
        // =====================================================================================================
        if (STATS) {
            const unsigned long long t = wall_clock64();
            if (lane == 0) cnt.hist_steps[0] += static_cast<uint32_t>(t - tick);
            tick = t;
        }
        {
            if (STATS && lane == 0) cnt.hist[0] += 1;   // probe: refill events
            const bool idle = !busy && !have_job;
            const unsigned long long im = __ballot(idle);
            const uint32_t n_take = min(static_cast<uint32_t>(__popcll(im)), jq_tail - jq_head);
            const uint32_t rank = static_cast<uint32_t>(__popcll(im & lt_mask));
            if (idle && rank < n_take) {
                const uint32_t e = jobq[(jq_head + rank) & QM];
                job_slot = e & 0xffu;
                static_assert(WP_SLOTS <= 256, "job entries keep the slot in 8 bits");
                has_ext = (e & WP_JOB_EXT) != 0u;
                any = (e & WP_JOB_SHADOW) != 0u;   // shadow ray first when there is one
                occl = false;
                have_job = true;
            } else if (!busy && have_job) {
                any = false;   // the shadow ray is done (step 1 kept the job): now its extension ray
            }
            jq_head += n_take;
            if (!busy && have_job) {
                const float4 o4 = wp_ld(&f_ray_o[job_slot]);
                const float4 d4 = wp_ld(&(any ? f_sh_d : f_ray_d)[job_slot]);
                trav_begin(sc, st, f3(o4.x, o4.y, o4.z), f3(d4.x, d4.y, d4.z), o4.w, any ? d4.w : INFINITY, stk);
                busy = true;
                if (STATS) {
                    if (any) cnt.shadow += 1; else cnt.ext += 1;
                }
            }
        }
        if (STATS) {
            const unsigned long long t = wall_clock64();
            if (lane == 0) cnt.hist_steps[1] += static_cast<uint32_t>(t - tick);
            tick = t;
        }
        if (__ballot(busy) == 0ull) {
            if (rd_tail == rd_head && jq_tail == jq_head) break;   // nothing in flight, nothing ready, nothing queued
            continue;
        }

        // =====================================================================================================
        // 4. traverse until WP_REFILL lanes are idle and there is something for them to do
        // =====================================================================================================
        for (;;) {
            const unsigned long long bm = __ballot(busy);
            const uint32_t nb = static_cast<uint32_t>(__popcll(bm));
            if (nb == 0u) break;
            if (64u - nb >= WP_REFILL) {
                // idle lanes can be put to work if a job is queued, a lane waits for its second ray, or a batch can be shaded
                const uint32_t waiting = static_cast<uint32_t>(__popcll(__ballot(!busy && have_job)));
                if (jq_tail != jq_head || waiting != 0u) break;
            }
            const bool at_int = busy && trav_at_interior(st);
            const bool at_oth = busy && !at_int;
            const unsigned long long mi = __ballot(at_int), mo = __ballot(at_oth);
            if (STATS && lane == 0) {   // probes: traversal-loop iterations and the busy lanes they carried
                cnt.hist[3] += 1;
                cnt.hist[4] += nb;
            }
            if (mo == 0ull || static_cast<uint32_t>(__popcll(mi)) >= WP_TRAV_TH) {
                if (at_int) trav_interior<STATS, true>(sc, st, stk, cnt);
            } else if (at_oth) {
                bool o = false;
                if (trav_other_kind<STATS, EXT, LdsStack, TIES>(sc, st, stk, cnt, any, &o, GBL_FILTER_NONE)) {
                    if (any) occl = o;
                    busy = false;
                }
            }
        }
        if (STATS) {
            const unsigned long long t = wall_clock64();
            if (lane == 0) cnt.hist_steps[2] += static_cast<uint32_t>(t - tick);
            tick = t;
        }
    }
    (void)n_dead;
    if (STATS) accumulate_stats(ra, cnt, paths_done);
}
