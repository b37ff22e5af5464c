// Wavefront formulation of the same integrator.  Paths live in a pool of P slots in
// HBM; one iteration runs three kernels:
//
//   wf_trace<false>  extension rays : persistent waves; each wave owns a strided set of
//                                     64-entry queue regions and refills its idle lanes
//                                     from them while the busy lanes keep traversing
//   wf_shade         lane == slot   : closes the previous bounce (MIS, Li, throughput),
//                                     terminates + regenerates, samples light and BSDF,
//                                     emits the shadow ray and the next extension ray
//   wf_trace<true>   shadow rays    : any-hit; adds the deferred contribution to Ld
//
// and a final wf_splat per pass that filters the per-sample radiance into the film
// through the LDS tile.  The arithmetic is the megakernel's (same shade.h / trace.h
// device functions, same operation order): both schedules give the same per-sample
// radiance bit for bit, and tests run both.
//
// There is NOT ONE same-address global atomic on the timed path (a returning atomic on
// a single word saturates near 88 per microsecond on this chip, which throttled the
// first version of this file): queues are compacted per wave with __ballot + prefix
// popcount into that wave's own 64-entry region, consumers are assigned regions
// statically, and slot s traces the paths s, s+P, s+2P, ... of the pass.
#pragma once
#include "../device_scene.h"
#include "render_kernels.h"
#include "sampler.h"
#include "shade.h"
#include "trace.h"
#include "vecmath.h"
#include "wf_args.h"

#define WF_BOUNCE_EMPTY (-2)   // slot needs a new path
#define WF_BLOCK_NONE 0xffffffffu   // no block of path ids (the shared reserve is used up)
#define WF_BOUNCE_DEAD (-3)    // slot has traced all its paths of this pass

// Path state streams through the pool once per iteration: non-temporal accesses keep it from evicting the BVH nodes
// and triangles the trace kernels re-read from L2 (measured -3.5 % on config 2's size, neutral on the others).
typedef float wf_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 wf_ld_nt(const float4* p) {
#ifndef GBL_WF_NO_NT
    wf_f4 v = __builtin_nontemporal_load(reinterpret_cast<const wf_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void wf_st_nt(float4* p, float4 v) {
#ifndef GBL_WF_NO_NT
    wf_f4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<wf_f4*>(p));
#else
    *p = v;
#endif
}


__device__ __forceinline__ void wf_stats(unsigned long long* stats, const LaneCounters& c, uint32_t paths) {
    unsigned long long v[11] = {paths, c.ext, c.shadow, c.nodes, c.tris, c.splats, c.dims, c.int_lane, c.int_wave, c.oth_lane, c.oth_wave};
    for (int i = 0; i < 11; ++i) {
        unsigned long long x = v[i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(stats + i, x);
    }
}

// path id of a pass -> pixel / sample.  ids enumerate (owned tile, pixel in tile, sample in pass):
// consecutive ids are samples of one pixel, so a freshly generated wave traces coherent primaries.
struct PathId {
    int px, py;
    uint32_t k;          // sample index within the pixel (absolute)
    uint32_t li_index;   // index into li_buf
    bool valid;
};
__device__ __forceinline__ PathId decode_path(const RenderArgs& ra, const WfArgs& wa, uint32_t id) {
    PathId p;
    uint32_t per_tile = 64u * static_cast<uint32_t>(wa.pass_spp);
    uint32_t lt = id / per_tile, r = id % per_tile;
    uint32_t pix = r / wa.pass_spp, kk = r % wa.pass_spp;
    uint32_t tile = ra.shard_index + lt * ra.shard_count;
    int tx = tile % ra.tiles_x, ty = tile / ra.tiles_x;
    p.px = ra.window[0] + GBL_TILE * tx + static_cast<int>(pix % 8u);
    p.py = ra.window[2] + GBL_TILE * ty + static_cast<int>(pix / 8u);
    p.k = static_cast<uint32_t>(wa.pass_k0) + kk;
    // pixel-major over the render window (the C ABI's li_out order when the pass is the whole render)
    const int sub_w = ra.window[1] - ra.window[0];
    p.li_index = static_cast<uint32_t>((p.py - ra.window[2]) * sub_w + (p.px - ra.window[0])) * wa.pass_spp + kk;
    p.valid = p.px < ra.window[1] && p.py < ra.window[3];
    return p;
}

// ---------------------------------------------------------------------------
// wf_trace: persistent traversal with in-wave dynamic fetch.
// ---------------------------------------------------------------------------
#ifndef GBL_WF_FUSE
#define GBL_WF_FUSE 1   // a leaf whose pop uncovers the sentinel / exit marker takes those at once (trace.h FUSE): Cornell -7 %, grid -2.5 %
#endif
#ifndef WF_REFILL
#define WF_REFILL 16   // refill as soon as this many lanes of the wave are idle
#endif

#ifndef GBL_WF_TRACE_WAVES
#define GBL_WF_TRACE_WAVES 5   // 96 VGPRs (4 spilled): 5 waves per SIMD measured 3-5 % faster than 4; 6 spills too much
#endif
// MASKS: the scene has mask materials (implies EXT); only those builds carry the filtered queries and the
// attenuation walks, which would otherwise cost every EXT scene ~150 spilled registers.
// TIES: closest-hit ties at exactly equal t go the way the reference BVH's visiting order decides (trace.h); the lean
// native-sampler build leaves the rule out, like the lean megakernel.
template <bool ANY, bool STATS, bool EXT, bool MASKS = false, bool TIES = true>
__global__ __launch_bounds__(GBL_BLOCK, MASKS ? 3 : GBL_WF_TRACE_WAVES) void wf_trace(DevScene sc, RenderArgs ra, WfArgs wa) {
    extern __shared__ __align__(16) unsigned char smem[];
    HotSplitStack stk;
    stk.p = gbl_as_lds(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
    stk.g = gbl_as_global(wa.stack_spill + blockIdx.x * GBL_BLOCK + threadIdx.x);
    stk.gstride = gridDim.x * GBL_BLOCK;
    {   // the top of the tree, once per (persistent) workgroup: trace.h HotLdsStack
        uint4* hot = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(smem) + ra.hot_word);
        for (uint32_t i = threadIdx.x; i < 4u * ra.hot_count; i += GBL_BLOCK) hot[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
        stk.hot = (const gbl_lds_u4*)hot;
        stk.hot_count = ra.hot_count;
        if (ra.hot_count) __syncthreads();
    }
    LaneCounters cnt = {};
    const int lane = threadIdx.x & 63;
    const uint32_t n_regions = wa.pool_size / 64u;
    const uint32_t n_waves = gridDim.x * (GBL_BLOCK / 64);
    // this wave's regions: w, w + n_waves, w + 2 n_waves, ...
    uint32_t region = blockIdx.x * (GBL_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t* counts = ANY ? wa.sh_count : wa.ext_count;
    uint32_t r_off = 0, r_cnt = region < n_regions ? counts[region] : 0u;

    bool busy = false;
    TravState st;
    st.sp = 0;
    st.cur = GBL_STACK_EXIT;
    st.inst = -1;
    st.mint = st.maxt = 0.0f;
    uint32_t slot = 0, entry = 0;
    F3 contrib = f3(0, 0, 0);
    bool needs_mis = false;
    constexpr bool masks = EXT && MASKS;
    constexpr int filter = (ANY && masks) ? GBL_FILTER_OPAQUE : GBL_FILTER_NONE;   // shadow rays: scene->occluded(ray, isOpaque)

    for (;;) {
        // ---- refill idle lanes from this wave's regions (all scalar bookkeeping)
        unsigned long long idle = __ballot(!busy);
        int n_idle = __popcll(idle);
        if (region < n_regions && n_idle >= WF_REFILL) {
            int my_rank = __popcll(idle & ((1ull << lane) - 1ull));
            int assigned = 0;
            uint32_t my_entry = 0xffffffffu;
            while (assigned < n_idle && region < n_regions) {
                int avail = static_cast<int>(r_cnt - r_off);
                int take = min(avail, n_idle - assigned);
                if (!busy && my_rank >= assigned && my_rank < assigned + take)
                    my_entry = region * 64u + r_off + static_cast<uint32_t>(my_rank - assigned);
                assigned += take;
                r_off += take;
                if (r_off >= r_cnt) {
                    region += n_waves;
                    r_off = 0;
                    r_cnt = region < n_regions ? counts[region] : 0u;
                }
            }
            if (my_entry != 0xffffffffu) {
                float4 a, b;
                float maxt;
                if (ANY) {
                    slot = wa.sh_slot[my_entry];
                    a = wf_ld_nt(&wa.ray_o[slot]);   // the vertex the slot stands at: origin and mint of its shadow ray too
                    b = wf_ld_nt(&wa.sh_d[my_entry]);
                    if constexpr (masks) {
                        const float4 c = wf_ld_nt(&wa.sh_c[my_entry]);
                        contrib = f3(c.x, c.y, c.z);   // (lightPdf, isArea, -): the product is formed after the attenuation walk
                    }
                    maxt = b.w;
                    entry = my_entry;
                } else {
                    slot = wa.ext_q[my_entry];
                    a = wf_ld_nt(&wa.ray_o[slot]);
                    b = wf_ld_nt(&wa.ray_d[slot]);
                    maxt = INFINITY;
                    needs_mis = b.w != 0.0f;
                }
                trav_begin(sc, st, f3(a.x, a.y, a.z), f3(b.x, b.y, b.z), a.w, maxt, stk);
                busy = true;
                if (STATS) {
                    if (ANY) cnt.shadow += 1; else cnt.ext += 1;
                }
            }
        }
        const bool drained = region >= n_regions;
        if (__ballot(busy) == 0ull) {
            if (drained) break;
            continue;
        }
        // ---- traverse while enough lanes are busy (or nothing is left to fetch)
        const int keep = drained ? 1 : (64 - WF_REFILL + 1);
        while (__popcll(__ballot(busy)) >= keep) {
            const bool at_int = busy && trav_at_interior(st);
            const bool at_oth = busy && !at_int;
            const unsigned long long mi = __ballot(at_int), mo = __ballot(at_oth);
            if (mo == 0ull || __popcll(mi) >= GBL_TRAV_TH) {
                if (at_int) trav_interior<STATS, !ANY>(sc, st, stk, cnt);
            } else if (at_oth) {
                bool occluded = false;
                // (TIES: both rules inline at every accepted triangle.  The loop-plus-end-of-query-check form of trace() was measured here
                //  too: with the exact loop inlined behind it the 96-register kernels spill 70 - 140 registers, Cornell 126 against 79 ms.)
                if (trav_other<ANY, STATS, EXT, HotSplitStack, TIES ? GBL_TIE_EXACT : GBL_TIE_NONE, GBL_WF_FUSE != 0>(sc, st, stk, cnt, &occluded, filter)) {
                    if (ANY) {
                        if constexpr (masks) if (!occluded) {
                            // evalAttenuation along the unoccluded shadow segment, then f * tr * L * |n.wi| (* lWeight) / lightPdf
                            const F3 tr = eval_attenuation<STATS>(sc, st.world.o, st.world.d, st.mint, st.maxt, stk, cnt);
                            const float4 ff = wa.sh_f[entry], LL = wa.sh_L[entry];
                            const F3 lf = f3(ff.x, ff.y, ff.z), lL = f3(LL.x, LL.y, LL.z);
                            contrib = contrib.y != 0.0f ? div(lf * tr * lL * ff.w * LL.w, contrib.x) : div(lf * tr * lL * ff.w, contrib.x);
                        }
                        if (!occluded) {
                            // the light-sampled term waits in the slot (wf_shade put it there); it counts from now on.  (Until round 4
                            // this kernel added it into the slot's Ld: a 32-byte read-modify-write per unoccluded ray, and 28 more
                            // bytes of queue entry to carry the term and the origin here.)
                            if constexpr (masks) {
                                float* ld = reinterpret_cast<float*>(&wa.s_ld[slot]);
                                ld[0] = contrib.x;
                                ld[1] = contrib.y;
                                ld[2] = contrib.z;
                            }
                            wa.s_vis[slot] = 1;
                        }
                    } else {
                        wf_st_nt(&wa.hit[slot], make_float4(st.hit.t, st.hit.b1, st.hit.b2, __uint_as_float(st.hit.tri)));
                        wa.hit_inst[slot] = st.hit.inst;
                        if constexpr (masks) {
                            int h2 = -2;   // the MIS query would find the same surface
                            if (needs_mis && st.hit.inst >= 0 && sc.instances[st.hit.inst].is_mask != 0u) {
                                // the closest surface is a mask: the isOpaque-filtered query and the masks in front of its hit
                                const F3 ro = st.world.o, rd = st.world.d;
                                const float rmint = st.mint;
                                Hit ho;
                                const bool go = trace<false, STATS, EXT>(sc, ro, rd, rmint, INFINITY, stk, ho, cnt, GBL_FILTER_OPAQUE);
                                const F3 tr = eval_attenuation<STATS>(sc, ro, rd, rmint, go ? ho.t : INFINITY, stk, cnt);
                                h2 = go ? ho.inst : -1;
                                wa.hit2[slot] = make_float4(ho.t, ho.b1, ho.b2, __uint_as_float(ho.tri));
                                wa.mis_tr[slot] = make_float4(tr.x, tr.y, tr.z, 0.0f);
                            }
                            wa.hit2_inst[slot] = h2;
                        }
                    }
                    busy = false;
                }
                // (letting whoever stands at an interior node after this phase take that step in the same iteration, as the
                //  megakernel's loops do, was measured here: Cornell 65.3 against 59.1 ms, grid 25.5 against 23.6)
            }
        }
    }
    if (STATS) wf_stats(ra.stats, cnt, 0);
}

// ---------------------------------------------------------------------------
// wf_shade: lane == slot.
// ---------------------------------------------------------------------------
template <bool REPLAY, bool STATS, bool EXT>
__global__ __launch_bounds__(GBL_BLOCK) void wf_shade(DevScene sc, RenderArgs ra, WfArgs wa) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;   // grid covers the pool exactly
    const int lane = threadIdx.x & 63;
    const uint32_t wave_gid = slot >> 6;
    LaneCounters cnt = {};
    uint32_t paths_done = 0;
    PathState ps;
    ps.bounce = WF_BOUNCE_EMPTY;
    ps.light = 0;
    ps.path = 0;
    uint32_t out_index = 0, k = 0, pixel_key = 0;
    const int sub_w_id = ra.window[1] - ra.window[0];
    // the native sampler's key of the pixel a sample index belongs to (what the regeneration below derives from the path id)
    auto pixel_key_of = [&](uint32_t index) {
        const uint32_t pix = index / static_cast<uint32_t>(wa.pass_spp);
        const int px = ra.window[0] + static_cast<int>(pix % static_cast<uint32_t>(sub_w_id)), py = ra.window[2] + static_cast<int>(pix / static_cast<uint32_t>(sub_w_id));
        const int full_w = sc.film.window[1] - sc.film.window[0];
        return nat_mix(ra.seed_key, static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0])));
    };
    if (!wa.init) {
        const uint2 id = wa.s_id[slot];
        ps.light = static_cast<int>(id.x & 0xffffu);
        ps.bounce = static_cast<int>((id.x >> 16) & 0x3fffu) - 4;
        out_index = id.y;
        k = static_cast<uint32_t>(wa.pass_k0) + out_index % static_cast<uint32_t>(wa.pass_spp);
        ps.punch = (id.x & (1u << 30)) != 0u;
        ps.first = (id.x & (1u << 31)) != 0u;
    } else {
        ps.punch = false;
        ps.first = true;
    }
    const bool masks = EXT && sc.has_masks != 0;
    const bool alive = ps.bounce >= -1;
    bool finished = false;
    Hit hit;
    hit.inst = -1;
    bool got = false;
    Frag fr;
    TexFrag tf;
    if (alive) {
        float4 o = wf_ld_nt(&wa.ray_o[slot]), d = wf_ld_nt(&wa.ray_d[slot]);
        float4 h = wf_ld_nt(&wa.hit[slot]);
        float4 a = wf_ld_nt(&wa.s_thr[slot]), b = wf_ld_nt(&wa.s_li[slot]), c = wf_ld_nt(&wa.s_ld[slot]), e = wf_ld_nt(&wa.s_f[slot]);
        ps.o = f3(o.x, o.y, o.z); ps.mint = o.w;
        ps.d = f3(d.x, d.y, d.z);
        ps.throughput = f3(a.x, a.y, a.z); ps.cosw = a.w;
        ps.Li = f3(b.x, b.y, b.z); ps.fw = b.w;
        // Ld of the vertex the slot stands at: 0 + its light-sampled term if the shadow ray got through (the sum the shadow kernel
        // used to form in place)
        const bool lit = wa.s_vis[slot] != 0;
        ps.Ld = lit ? f3(0.0f + c.x, 0.0f + c.y, 0.0f + c.z) : f3(0.0f, 0.0f, 0.0f);
        ps.bsdf_pdf = c.w;
        ps.f = f3(e.x, e.y, e.z); ps.pick_pdf = e.w;
        if (!REPLAY) pixel_key = pixel_key_of(out_index);
        hit.t = h.x; hit.b1 = h.y; hit.b2 = h.z; hit.tri = __float_as_uint(h.w);
        hit.inst = wa.hit_inst[slot];
        got = hit.inst >= 0;
    }
    SampleSource src;
    src.spp = ra.spp;
    src.root = ra.root;
    src.rec = nullptr;
    src.pixel_key = pixel_key;
    src.k = k;
    if (REPLAY && alive)   // out_index = pixel * pass_spp + kk  ->  record pixel * spp + k
        src.rec = ra.replay + (static_cast<size_t>(out_index / wa.pass_spp) * ra.spp + k) * ra.dims;

    // ---- close the bounce whose extension ray was just traced (the megakernel's code)
    if (alive) {
        if (sc.num_lights == 0) {
            finished = true;
        } else {
            if (got) {
                make_fragment<EXT>(sc, hit, ps.o, ps.d, fr, &tf);
                if (EXT && sc.materials[sc.instances[hit.inst].material].has_tex != 0u) {
                    float image_x = 0.0f, image_y = 0.0f;
                    if (ps.bounce < 0) {   // the camera sample this path started from (wf regeneration below)
                        if (REPLAY) {
                            image_x = src.rec[0];
                            image_y = src.rec[1];
                        } else {
                            const int sub_w = ra.window[1] - ra.window[0];
                            const uint32_t pix = out_index / static_cast<uint32_t>(wa.pass_spp);
                            float u, v;
                            src.native_2d(0u, 1u, 0u, false, &u, &v);
                            image_x = (ra.window[0] + static_cast<int>(pix % sub_w)) + u;
                            image_y = (ra.window[2] + static_cast<int>(pix / sub_w)) + v;
                        }
                    }
                    hit_differentials<REPLAY>(sc, src, ps.bounce < 0, image_x, image_y, fr, tf);
                }
            }
            // mask scenes: the MIS query (isOpaque) and its attenuation, prepared by wf_trace when they differ
            int mis_inst = got ? hit.inst : -1;
            F3 mis_n = fr.n, mis_tr = f3(1.0f, 1.0f, 1.0f);
            if (masks && got && ps.bounce >= 0 && !ps.punch) {
                const int h2 = wa.hit2_inst[slot];
                if (h2 != -2) {
                    mis_inst = h2;
                    const float4 t4 = wa.mis_tr[slot];
                    mis_tr = f3(t4.x, t4.y, t4.z);
                    if (h2 >= 0) {
                        const float4 q = wa.hit2[slot];
                        Hit ho;
                        ho.t = q.x; ho.b1 = q.y; ho.b2 = q.z; ho.tri = __float_as_uint(q.w); ho.inst = h2;
                        Frag fo;
                        make_fragment<EXT>(sc, ho, ps.o, ps.d, fo);
                        mis_n = fo.n;
                    }
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
                        const float4 ss = reinterpret_cast<const float4*>(ra.sss)[static_cast<size_t>(out_index / wa.pass_spp) * ra.spp + k];
                        ps.Li = f3(ps.Li.x + ss.x, ps.Li.y + ss.y, ps.Li.z + ss.z);
                    }
                    ps.bounce = 0;
                }
            } else if (EXT && ps.punch) {
                ps.punch = false;   // the bounce that punched through a mask adds no direct light (GoblinPathtracer.cpp:122-136)
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
                if (mis_inst >= 0 && sc.instances[mis_inst].area_light == ps.light) {
                    F3 le = hit_Le(sc, mis_inst, mis_n, -ps.d);
                    if (!is_black(le)) {
                        F3 term = EXT ? div(ps.f * mis_tr * le * ps.cosw * ps.fw, ps.bsdf_pdf) : div(ps.f * le * ps.cosw * ps.fw, ps.bsdf_pdf);
                        ps.Ld = f3(ps.Ld.x + term.x, ps.Ld.y + term.y, ps.Ld.z + term.z);
                    }
                } else if (EXT && mis_inst < 0 && sc.has_ibl != 0) {
                    // the sampled direction left the scene: Ld += f * tr * light->Le(r) * fWeight / bsdfPdf (:157-161).  A slot
                    // that only waited for its shadow ray (f == 0, no extension ray) adds 0 here.
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
    }

    // ---- shade: light sample -> shadow entry, BSDF sample -> next ray
    bool need_shadow = false, has_ray = false, zombie = false;
    float4 sh_f4 = make_float4(0, 0, 0, 0), sh_L4 = make_float4(0, 0, 0, 0);
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
        int li = 0;
        for (int i = 1; i <= sc.num_lights; ++i)
            if (sc.light_cdf[i] < u_pick) li = i;
        if (li >= sc.num_lights) li = sc.num_lights - 1;
        ps.light = li;
        ps.pick_pdf = sc.light_pick_pdf[li];
        ps.Ld = f3(0, 0, 0);
        const DevMaterial* mat = sc.materials + sc.instances[hit.inst].material;
        ResolvedMat rmat;   // EXT: the hit material with its textures evaluated / its mask unwrapped
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
                if (masks) {   // the shadow kernel multiplies the attenuation in, in the reference's order
                    sh_f4 = make_float4(f.x, f.y, f.z, absdot(fr.n, ls.wi));
                    sh_L4 = make_float4(ls.L.x, ls.L.y, ls.L.z, lw);
                    contrib = f3(ls.pdf, light_is_delta<EXT>(light) ? 0.0f : 1.0f, 0.0f);
                }
            }
        }
        F3 wi;
        float pdf;
        bool specular, null_sampled = false;
        F3 f = EXT ? rmat_sample(rmat, fr, wo, u_bsdf_c, u_bsdf_1, u_bsdf_2, &wi, &pdf, &specular, &null_sampled)
                   : mat_sample(*mat, fr, wo, u_bsdf_c, u_bsdf_1, u_bsdf_2, &wi, &pdf, &specular);
        if (EXT && null_sampled && !is_black(f) && pdf > 0.0f) {
            // punch through the mask: throughput *= f / pdf, and this bounce's direct light is dropped with its shadow ray
            ps.throughput = ps.throughput * div(f, pdf);
            ps.o = fr.p;
            ps.d = wi;
            ps.mint = fr.eps;
            ps.punch = true;
            ps.f = f3(0, 0, 0);
            ps.fw = 0.0f;
            ps.bsdf_pdf = 1.0f;
            ps.cosw = 0.0f;
            need_shadow = false;
            has_ray = true;
        }
        // Russian roulette (build-side extension, off in every parity mode): as in the megakernel, a killed path still
        // collects this vertex's direct light -- here through the zombie / immediate-finish branches below
        bool rr_killed = false;
        float rr_inv = 1.0f;
        if (ra.russian_roulette && !REPLAY && !(EXT && null_sampled) && !is_black(f) && pdf > 0.0f && ps.bounce >= 2) {
            F3 tn = ps.throughput * div(f * absdot(wi, fr.n), pdf);
            float q = fminf(0.95f, fmaxf(tn.x, fmaxf(tn.y, tn.z)));
            float u = nat_u01(nat_mix(nat_mix(src.pixel_key, 0xBADC0DEu + ps.bounce), src.k));
            if (!(u < q)) rr_killed = true;
            else rr_inv = 1.0f / q;   // carried by f, see the megakernel
        }
        if (EXT && null_sampled && !is_black(f) && pdf > 0.0f) {
        } else if (!rr_killed && !is_black(f) && pdf > 0.0f) {
            float fw = 1.0f;
            if (!specular) fw = power_heuristic(pdf, light_pdf<EXT>(sc, sc.lights[ps.light], fr.p, wi));
            ps.f = f;
            ps.fw = fw;
            ps.bsdf_pdf = pdf;
            ps.cosw = absdot(wi, fr.n);
            if (ra.russian_roulette && !REPLAY && rr_inv != 1.0f) ps.f = ps.f * rr_inv;
            ps.o = fr.p;
            ps.d = wi;
            ps.mint = fr.eps;
            has_ray = true;
        } else if (need_shadow) {
            // The path ends here but its direct light is still pending on the shadow ray.  Keep the
            // slot one more iteration as a "zombie": no extension ray is traced, the hit is preset to
            // a miss, and f = 0 makes the next close do exactly Li += throughput * Ld / pickPdf.
            ps.f = f3(0, 0, 0);
            ps.fw = 0.0f;
            ps.bsdf_pdf = 1.0f;
            ps.cosw = 0.0f;
            zombie = true;
        } else {
            // Li += throughput * Ld / pickLightPdf with Ld == 0 (no shadow ray pending); break
            F3 add = div(ps.throughput * ps.Ld, ps.pick_pdf);
            ps.Li = f3(ps.Li.x + add.x, ps.Li.y + add.y, ps.Li.z + add.z);
            finished = true;
        }
    }
    // ---- shadow entries: compacted into this wave's region
    {
        unsigned long long m = __ballot(need_shadow);
        if (need_shadow) {
            uint32_t pos = wave_gid * 64u + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
            wa.sh_d[pos] = make_float4(shadow_d.x, shadow_d.y, shadow_d.z, shadow_maxt);
            wa.sh_slot[pos] = slot;
            if (masks) {
                wa.sh_c[pos] = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
                wa.sh_f[pos] = sh_f4;
                wa.sh_L[pos] = sh_L4;
            }
            wa.s_vis[slot] = 0;   // until the shadow ray says otherwise
        }
        if (lane == 0) wa.sh_count[wave_gid] = static_cast<uint32_t>(__popcll(m));
    }

    // ---- termination: publish the sample's radiance, then start this slot's next path
    if (alive && finished) {
        wa.li_buf[out_index] = make_float4(ps.Li.x, ps.Li.y, ps.Li.z, 1.0f);
        paths_done += 1;
    }
    bool started = false, dead = ps.bounce == WF_BOUNCE_DEAD;
    const bool want_new = (alive && finished) || ps.bounce == WF_BOUNCE_EMPTY;
    if (alive && finished) ps.bounce = WF_BOUNCE_EMPTY;
    {
        // Path ids come in blocks of 64 (= consecutive samples of one pixel).  Every shade-wave is dealt `static_blocks` of them up
        // front, round-robin, so its list is spread over the whole image; it hands their ids to its lanes in order -- ballot + prefix
        // popcount over a cursor only this wave touches (plain load / store).  A pixel's paths are as long as what the pixel sees,
        // so the waves' lists do NOT take equally long (on the Cornell box the slowest of 131 072 waves needed 18 % more iterations
        // than the mean, and every iteration costs the whole pool's launches): the last quarter of the blocks is a shared reserve, a
        // wave whose own list is used up takes its next block from there -- one atomic per 64 paths, by the waves that finish early.
        unsigned long long wm = __ballot(want_new);
        if (wm != 0ull) {
            const uint32_t waves = wa.pool_size >> 6;
            const uint32_t cursor = wa.wave_next[2u * wave_gid];
            const uint32_t held = wa.wave_next[2u * wave_gid + 1u];   // reserve block in use (index cursor >> 6, once that is past the static ones)
            const uint32_t n_new = static_cast<uint32_t>(__popcll(wm));
            const uint32_t local = cursor + static_cast<uint32_t>(__popcll(wm & ((1ull << lane) - 1ull)));
            // the ids of this round lie in block index j0, or j0 and j0 + 1; at most one of the two is begun in this round
            const uint32_t j0 = cursor >> 6, j1 = (cursor + n_new - 1u) >> 6;
            const bool begins = (cursor & 63u) == 0u || j1 != j0;
            const uint32_t jb = (cursor & 63u) == 0u ? j0 : j1;   // ... the one that is begun
            uint32_t begun = WF_BLOCK_NONE;
            if (begins) {
                if (jb < wa.static_blocks) {
                    begun = jb * waves + wave_gid;
                } else if (held != WF_BLOCK_NONE || jb == wa.static_blocks) {   // (NONE after the first reserve block: the reserve is empty)
                    uint32_t g = 0;
                    if (lane == 0) g = atomicAdd(wa.steal_next, 1u);
                    g = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(g)));
                    begun = wa.static_blocks * waves + g;
                    if (begun >= wa.total_blocks) begun = WF_BLOCK_NONE;
                }
            }
            if (lane == 0) {
                wa.wave_next[2u * wave_gid] = cursor + n_new;
                if (begins && jb >= wa.static_blocks) wa.wave_next[2u * wave_gid + 1u] = begun;
            }
            if (want_new) {
                const uint32_t j = local >> 6;
                uint32_t block = (begins && j == jb) ? begun : (j < wa.static_blocks ? j * waves + wave_gid : held);
                if (j >= wa.static_blocks && j != jb && held == WF_BLOCK_NONE) block = WF_BLOCK_NONE;
                const uint64_t id = static_cast<uint64_t>(block) * 64u + (local & 63u);
                if (block != WF_BLOCK_NONE && id < wa.total_paths) {
                    PathId pid = decode_path(ra, wa, static_cast<uint32_t>(id));
                    if (pid.valid) {   // (a pixel clipped off an edge tile leaves the slot EMPTY: it asks again next iteration)
                        const int full_w = sc.film.window[1] - sc.film.window[0];
                        float image_x, image_y;
                        out_index = pid.li_index;
                        k = pid.k;
                        src.k = k;
                        if (REPLAY) {
                            src.rec = ra.replay + (static_cast<size_t>(out_index / wa.pass_spp) * ra.spp + k) * ra.dims;
                            image_x = src.rec[0];
                            image_y = src.rec[1];
                        } else {
                            uint32_t pixel = static_cast<uint32_t>((pid.py - sc.film.window[2]) * full_w + (pid.px - sc.film.window[0]));
                            pixel_key = nat_mix(ra.seed_key, pixel);
                            src.pixel_key = pixel_key;
                            float u, v;
                            src.native_2d(0u, 1u, 0u, false, &u, &v);
                            image_x = pid.px + u;
                            image_y = pid.py + v;
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
                        ps.Ld = f3(0.0f, 0.0f, 0.0f);
                        ps.f = f3(0.0f, 0.0f, 0.0f);
                        ps.cosw = ps.fw = 0.0f;
                        ps.bsdf_pdf = ps.pick_pdf = 1.0f;
                        ps.light = 0;
                        ps.bounce = -1;
                        ps.punch = false;
                        ps.first = true;
                        started = true;
                        has_ray = true;
                        if (STATS) cnt.dims += 2;
                    }
                } else {
                    dead = true;
                    ps.bounce = WF_BOUNCE_DEAD;
                }
            }
        }
    }
    // ---- write back
    const bool keep = (alive && !finished) || started;
    if (keep) {
        if (has_ray) {
            wf_st_nt(&wa.ray_o[slot], make_float4(ps.o.x, ps.o.y, ps.o.z, ps.mint));
            // .w: this ray also serves the MIS estimate (not a camera ray, not a punch-through continuation)
            wf_st_nt(&wa.ray_d[slot], make_float4(ps.d.x, ps.d.y, ps.d.z, (ps.bounce >= 0 && !ps.punch) ? 1.0f : 0.0f));
        } else if (zombie) {
            // no extension ray, but the shadow ray still leaves from this vertex (the shadow kernel reads the origin here)
            wf_st_nt(&wa.ray_o[slot], make_float4(fr.p.x, fr.p.y, fr.p.z, fr.eps));
        }
        if (zombie) wa.hit_inst[slot] = -1;
        // the light-sampled term of the vertex just shaded waits for its shadow ray (mask scenes: the shadow kernel forms it)
        const F3 pending = (need_shadow && !masks) ? contrib : f3(0.0f, 0.0f, 0.0f);
        wf_st_nt(&wa.s_thr[slot], make_float4(ps.throughput.x, ps.throughput.y, ps.throughput.z, ps.cosw));
        wf_st_nt(&wa.s_li[slot], make_float4(ps.Li.x, ps.Li.y, ps.Li.z, ps.fw));
        wf_st_nt(&wa.s_ld[slot], make_float4(pending.x, pending.y, pending.z, ps.bsdf_pdf));
        wf_st_nt(&wa.s_f[slot], make_float4(ps.f.x, ps.f.y, ps.f.z, ps.pick_pdf));
    }
    if (keep || dead || want_new || wa.init)
        wa.s_id[slot] = make_uint2(static_cast<uint32_t>(ps.light) | (static_cast<uint32_t>(ps.bounce + 4) << 16) | (ps.punch ? 1u << 30 : 0u) | (ps.first ? 1u << 31 : 0u),
                                   out_index);
    // ---- extension queue: compacted into this wave's region
    {
        bool enq = keep && has_ray;
        unsigned long long m = __ballot(enq);
        if (enq) wa.ext_q[wave_gid * 64u + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)))] = slot;
        if (lane == 0) wa.ext_count[wave_gid] = static_cast<uint32_t>(__popcll(m));
        unsigned long long any_kept = __ballot(keep || ps.bounce == WF_BOUNCE_EMPTY);   // whole wave, not under `lane == 0`
        if (lane == 0 && any_kept != 0ull) wa.live_flags[wa.flag_index] = 1u;   // benign race: everyone stores 1
    }
    if (STATS) wf_stats(ra.stats, cnt, paths_done);
}

// ---------------------------------------------------------------------------
// wf_splat: one workgroup per owned tile, one lane per PIXEL of the tile (64 lanes of a
// wave splat 64 different footprints: no same-address LDS atomic storms); the four waves
// split the samples.  ImageTile::addSample (GoblinFilm.cpp:61-90).
// RenderTask::run: what the tile receives is w * (tr * L + Lv), w = 1 -- the medium's two terms come from vol_kernel
// (kernels/volume.h), per camera sample of the call
__device__ __forceinline__ float4 wf_apply_medium(const RenderArgs& ra, float4 L, size_t sample) {
    if (ra.vol == nullptr) return L;
    const float4 tr = reinterpret_cast<const float4*>(ra.vol)[2 * sample], lv = reinterpret_cast<const float4*>(ra.vol)[2 * sample + 1];
    return make_float4(1.0f * (tr.x * L.x + lv.x), 1.0f * (tr.y * L.y + lv.y), 1.0f * (tr.z * L.z + lv.z), L.w);
}

// ---------------------------------------------------------------------------
template <bool REPLAY, bool STATS>
__global__ __launch_bounds__(GBL_BLOCK) void wf_splat(DevScene sc, RenderArgs ra, WfArgs wa) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tp = GBL_TILE + 2 * sc.film.halo;
    float* tile = reinterpret_cast<float*>(smem);
    float* ftab = tile + 4 * tp * tp;
    for (int i = threadIdx.x; i < 256; i += GBL_BLOCK) ftab[i] = sc.filter_table[i];
    for (int i = threadIdx.x; i < 4 * tp * tp; i += GBL_BLOCK) tile[i] = 0.0f;
    __syncthreads();
    LaneCounters cnt = {};
    const uint32_t lt = blockIdx.x;
    const uint32_t tile_id = ra.shard_index + lt * ra.shard_count;
    const int tx = tile_id % ra.tiles_x, ty = tile_id / ra.tiles_x;
    const int px0 = ra.window[0] + GBL_TILE * tx, py0 = ra.window[2] + GBL_TILE * ty;
    const int tx0 = px0 - sc.film.halo, ty0 = py0 - sc.film.halo;
    const int full_w = sc.film.window[1] - sc.film.window[0];
    const int sub_w = ra.window[1] - ra.window[0];
    const int pix = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = px0 + (pix & 7), py = py0 + (pix >> 3);
    // Native samples lie inside their pixel, so with a filter half-width <= 2 the footprint of every
    // sample of pixel (px,py) is inside the 5x5 block around it: accumulate the lane's samples in
    // registers and touch LDS once per footprint pixel instead of once per sample.
    // (The stream sampler's records are the reference's own: also inside their pixel; their image positions were kept
    // in ra.image_xy.  Arbitrary replay records may belong anywhere and take the general path.)
    const bool fast5 = (!REPLAY || ra.image_xy != nullptr) && sc.film.wx <= 2.0f && sc.film.wy <= 2.0f;
    if (px < ra.window[1] && py < ra.window[3]) {
        SampleSource src;
        src.spp = ra.spp;
        src.root = ra.root;
        src.rec = nullptr;
        const uint32_t pixel = static_cast<uint32_t>((py - sc.film.window[2]) * full_w + (px - sc.film.window[0]));
        src.pixel_key = nat_mix(ra.seed_key, pixel);
        const uint32_t local_pixel = static_cast<uint32_t>((py - ra.window[2]) * sub_w + (px - ra.window[0]));
        if (fast5) {
            float acc[25][4];
#pragma unroll
            for (int i = 0; i < 25; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0f;
            const int xlo = sc.film.xstart, xhi = sc.film.xstart + sc.film.xcount - 1;
            const int ylo = sc.film.ystart, yhi = sc.film.ystart + sc.film.ycount - 1;
            for (int kk = wave; kk < wa.pass_spp; kk += GBL_BLOCK / 64) {
                src.k = static_cast<uint32_t>(wa.pass_k0 + kk);
                float image_x, image_y;
                if (REPLAY) {
                    const float* xy = ra.image_xy + 2 * (static_cast<size_t>(local_pixel) * ra.spp + src.k);
                    image_x = xy[0];
                    image_y = xy[1];
                } else {
                    float u, v;
                    src.native_2d(0u, 1u, 0u, false, &u, &v);
                    image_x = px + u;
                    image_y = py + v;
                }
                const float4 L = wf_apply_medium(ra, wa.li_buf[static_cast<size_t>(local_pixel) * wa.pass_spp + kk],
                                                 static_cast<size_t>(local_pixel) * ra.spp + src.k);
                if (L.x != L.x || L.y != L.y || L.z != L.z) continue;   // NaN sample: dropped
                const float dx = image_x - 0.5f, dy = image_y - 0.5f;
                const int x0 = max(static_cast<int>(ceilf(dx - sc.film.wx)), xlo), x1 = min(static_cast<int>(floorf(dx + sc.film.wx)), xhi);
                const int y0 = max(static_cast<int>(ceilf(dy - sc.film.wy)), ylo), y1 = min(static_cast<int>(floorf(dy + sc.film.wy)), yhi);
                int ix[5], iy[5];
#pragma unroll
                for (int o = 0; o < 5; ++o) {
                    const int x = px + o - 2, y = py + o - 2;
                    ix[o] = (x >= x0 && x <= x1) ? min(static_cast<int>(floorf(fabsf(16 * (x - dx) / sc.film.wx))), 15) : -1;
                    iy[o] = (y >= y0 && y <= y1) ? min(static_cast<int>(floorf(fabsf(16 * (y - dy) / sc.film.wy))), 15) : -1;
                }
#pragma unroll
                for (int oy = 0; oy < 5; ++oy) {
#pragma unroll
                    for (int ox = 0; ox < 5; ++ox) {
                        if (ix[ox] >= 0 && iy[oy] >= 0) {
                            const float w = ftab[iy[oy] * 16 + ix[ox]];
                            acc[oy * 5 + ox][0] += w * L.x;
                            acc[oy * 5 + ox][1] += w * L.y;
                            acc[oy * 5 + ox][2] += w * L.z;
                            acc[oy * 5 + ox][3] += w;
                            if (STATS) cnt.splats += 1;
                        }
                    }
                }
            }
#pragma unroll
            for (int oy = 0; oy < 5; ++oy) {
#pragma unroll
                for (int ox = 0; ox < 5; ++ox) {
                    const float* a = acc[oy * 5 + ox];
                    if (a[3] != 0.0f || a[0] != 0.0f || a[1] != 0.0f || a[2] != 0.0f) {
                        float* q = tile + 4 * ((py + oy - 2 - ty0) * tp + (px + ox - 2 - tx0));
                        atomicAdd(q + 0, a[0]);
                        atomicAdd(q + 1, a[1]);
                        atomicAdd(q + 2, a[2]);
                        atomicAdd(q + 3, a[3]);
                    }
                }
            }
        } else {
            for (int kk = wave; kk < wa.pass_spp; kk += GBL_BLOCK / 64) {
                const uint32_t k = static_cast<uint32_t>(wa.pass_k0 + kk);
                float image_x, image_y;
                if (REPLAY && ra.image_xy) {   // GBL_SAMPLES_STREAM: the records were transient, their image positions were kept
                    const float* xy = ra.image_xy + 2 * (static_cast<size_t>(local_pixel) * ra.spp + k);
                    image_x = xy[0];
                    image_y = xy[1];
                } else if (REPLAY) {
                    const float* rec = ra.replay + (static_cast<size_t>(local_pixel) * ra.spp + k) * ra.dims;
                    image_x = rec[0];
                    image_y = rec[1];
                } else {
                    src.k = k;
                    float u, v;
                    src.native_2d(0u, 1u, 0u, false, &u, &v);
                    image_x = px + u;
                    image_y = py + v;
                }
                float4 L = wf_apply_medium(ra, wa.li_buf[static_cast<size_t>(local_pixel) * wa.pass_spp + kk],
                                           static_cast<size_t>(local_pixel) * ra.spp + k);
                splat<STATS>(sc.film, ftab, tile, tx0, ty0, tp, image_x, image_y, f3(L.x, L.y, L.z), cnt);
            }
        }
    }
    __syncthreads();
    flush_tile(sc.film, tile, tx0, ty0, tp, ra.film);
    if (STATS) wf_stats(ra.stats, cnt, 0);
}
