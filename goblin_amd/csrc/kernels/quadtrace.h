// Quad-per-ray queries for the persistent megakernel and the AO kernel: once at most GBL_QUAD_MAX (16) of a wave's rays are
// unfinished, the wave's 64 lanes regroup as 16 quads, one quad per ray, and every lane of a quad tests ONE of a node's four
// children (the five compare-exchanges of trav_interior's sorting network run across the quad on DPP) or ONE of a leaf's
// triangles.
//
// Why: the megakernel is bound by VALU issue, and 60 % of its interior wave-steps run with <= 4 of 64 lanes,
// 83 % with <= 16 (the few rays of a wave that wander through the bunny's BLAS).  The round's earlier experiments made
// those sparse steps RARE by moving rays between lanes, waves or iterations, and lost to the idle waves that creates; this
// one leaves occupancy and the path state alone and makes the sparse step CHEAP: a quad step is one box test per lane (64
// VALU instructions with the sort) where the plain step is four (~150).  A ray's sequence of node visits and triangle tests
// is untouched, so hits, ties and radiance are bit-identical (Scene::intersect / occluded, GoblinBVH.cpp:189-280).
//
// A query starts as trace() does, one ray per lane.  Once at most 16 of the wave's rays are unfinished they MIGRATE: each
// publishes its traversal state to a record of the wave's LDS slab (rank order), quad q takes record q and owns that ray for
// the rest of the query -- interior steps with one child per lane, leaves with one triangle per lane (the tie-rule builds:
// trav_other's own loop, replicated in the four lanes), instance entry / exit from the world ray left in the record, pushing
// to and popping from the ray's own LDS stack column -- and hands the hit back through the record at the end.
// (A first form returned every ray to its lane after each run of interior steps: 27 % fewer VALU instructions and no
// faster -- two LDS round trips per run on a chain that is latency bound once the VALU work shrinks.)  LDS operations of
// one wave execute in order, so the slab and the foreign stack columns need no barrier, only compiler fences.
//
// TIES builds (`exact_ties` under the native sampler, the stream sampler) follow the reference's exact-t tie rule and its
// reachability test the way trace() does (trace.h, GBL_TIE_DETECT): the loops here only notice a triangle accepted at exactly the
// distance of the hit the ray holds -- the flag travels back in the record -- and trace_quad() checks every ray's final hit once,
// all lanes together, tracing the rare ray that tied or whose hit the reference would not have reached again in the exact loop.
#pragma once
#include "trace.h"

#ifndef GBL_QUAD_MAX
#define GBL_QUAD_MAX 16           // rays that migrate: 64 lanes / 4
#endif
// record: r.o r.d world.o world.d | mint maxt cur inst | hit.inst hit.tri hit.b1 hit.b2 | sp + (lane << 8)
// result (written by the quad's first lane when the ray is done): words 0-5 = hit.inst hit.tri hit.b1 hit.b2 hit.t occluded + 2 tied, 12 = steps
#define GBL_QUAD_REC_WORDS 21
#define GBL_QUAD_LDS_WORDS ((GBL_BLOCK / 64) * 16 * GBL_QUAD_REC_WORDS)

template <int CTRL>
__device__ __forceinline__ uint32_t quad_dpp(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float quad_dpp_f(float v) {
    return __uint_as_float(quad_dpp<CTRL>(__float_as_uint(v)));
}
#define GBL_QP_XOR1 0xB1   // quad_perm [1,0,3,2]
#define GBL_QP_XOR2 0x4E   // quad_perm [2,3,0,1]
#define GBL_QP_MID 0xD8    // quad_perm [0,2,1,3]
#define GBL_QP_BC0 0x00    // quad_perm [0,0,0,0]

// One compare-exchange of GBL_CSWAP across lanes.  `side` is -INFINITY in the lane that holds the pair's first element and
// +INFINITY in the one that holds its second: med3(t, partner, side) is then min(t, partner) / max(t, partner), and a lane
// takes the partner's pair exactly when that differs from its own t -- the strict `tb < ta` swap of GBL_CSWAP (equal
// entries stay where they are).  Entry distances are never NaN (child_entry).
template <int CTRL>
__device__ __forceinline__ void quad_cswap(float& t, uint32_t& r, float side) {
    const float tp = quad_dpp_f<CTRL>(t);
    const uint32_t rp = quad_dpp<CTRL>(r);
    const float nt = __builtin_amdgcn_fmed3f(t, tp, side);
    r = nt != t ? rp : r;
    t = nt;
}

__device__ __forceinline__ void quad_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The quad's lanes, set up once per migration: which child a lane tests and on which side of each compare-exchange it sits.
struct QuadLane {
    uint32_t c, bitc, selx, sely, selz;
    float side1, side2;
    gbl_lds_u32* col;   // the ray's LDS stack column (LdsStack layout)
};
// v_perm selectors: byte 0 <- this child's plane crossed first, byte 1 <- the one crossed last ({qhi, qlo} = bytes 4-7, 0-3)
__device__ __forceinline__ void quad_selectors(QuadLane& ql, F3 idir) {
    const uint32_t c = ql.c;
    ql.selx = idir.x < 0.0f ? (0x0c0c0004u + c + (c << 8)) : (0x0c0c0400u + c + (c << 8));
    ql.sely = idir.y < 0.0f ? (0x0c0c0004u + c + (c << 8)) : (0x0c0c0400u + c + (c << 8));
    ql.selz = idir.z < 0.0f ? (0x0c0c0004u + c + (c << 8)) : (0x0c0c0400u + c + (c << 8));
}

// trav_interior() for a ray held by a quad: st is replicated in the quad's four lanes, lane c tests child c.
template <bool SORTED, bool STATS>
__device__ __forceinline__ void quad_interior(TravState& st, const QuadLane& ql, LaneCounters& cnt, uint4 w0, uint4 w1, uint4 w2, uint32_t r,
                                              uint32_t popped) {
    // w0 = o.x o.y o.z scale.x | w1 = scale.y scale.z qlo.x qlo.y | w2 = qlo.z qhi.x qhi.y qhi.z | r = child[c]
    const int sp = st.sp;
    const RaySpace& rs = st.r;
    const F3 A = f3(__builtin_fmaf(__uint_as_float(w0.x), rs.idir.x, -rs.ood.x), __builtin_fmaf(__uint_as_float(w0.y), rs.idir.y, -rs.ood.y),
                    __builtin_fmaf(__uint_as_float(w0.z), rs.idir.z, -rs.ood.z));
    const F3 B = f3(__uint_as_float(w0.w) * rs.idir.x, __uint_as_float(w1.x) * rs.idir.y, __uint_as_float(w1.y) * rs.idir.z);
    const uint32_t px = __builtin_amdgcn_perm(w2.y, w1.z, ql.selx);
    const uint32_t py = __builtin_amdgcn_perm(w2.z, w1.w, ql.sely);
    const uint32_t pz = __builtin_amdgcn_perm(w2.w, w2.x, ql.selz);
    float t = child_entry(px, py, pz, px >> 8, py >> 8, pz >> 8, A, B, st.mint, st.maxt);
    if (STATS) {
        cnt.nodes += 1;   // four lanes per node visit: the same 4 per visit trav_interior counts
        if (ql.c == 0u) cnt.int_lane += 1;
        if ((threadIdx.x & 63u) == static_cast<uint32_t>(__ffsll(static_cast<long long>(__ballot(1)))) - 1u) cnt.int_wave += 1;
#ifdef GBL_PROBE_OCC
        // experiment build: quad wave-steps in hist[5], the rays inside them in hist[6] (trav_interior bins the one-ray-per-lane steps in hist[0..4])
        if ((threadIdx.x & 63u) == static_cast<uint32_t>(__ffsll(static_cast<long long>(__ballot(1)))) - 1u) cnt.hist[5] += 1;
        if (ql.c == 0u) cnt.hist[6] += 1;
#endif
    }
    if (SORTED) {   // trav_interior's network: (0,1) (2,3) | (0,2) (1,3) | (1,2)
        quad_cswap<GBL_QP_XOR1>(t, r, ql.side1);
        quad_cswap<GBL_QP_XOR2>(t, r, ql.side2);
        quad_cswap<GBL_QP_MID>(t, r, ql.side2);   // lanes 0 and 3 meet themselves: nothing changes
    }
    const bool h = t < INFINITY;
    uint32_t m = h ? ql.bitc : 0u;   // the quad's hit mask
    m |= quad_dpp<GBL_QP_XOR1>(m);
    m |= quad_dpp<GBL_QP_XOR2>(m);
    const int n = __popc(m);
    // position among the hit children in visiting order (sorted: the hits are the first n lanes)
    const int p = SORTED ? static_cast<int>(ql.c) : __popc(m & (ql.bitc - 1u));
    uint32_t first;
    if (SORTED) {
        first = quad_dpp<GBL_QP_BC0>(r);
    } else {
        first = (h && p == 0) ? r : 0u;
        first |= quad_dpp<GBL_QP_XOR1>(first);
        first |= quad_dpp<GBL_QP_XOR2>(first);
    }
    // the first is visited next, the others go on the stack farthest / last first
    if (h && p >= 1) ql.col[(sp + n - 1 - p) * GBL_BLOCK] = r;
    st.cur = static_cast<int>(n > 0 ? first : popped);
    st.sp = sp + n - 1;
}

// A triangle leaf for a ray held by a quad: lane c tests the leaf's triangle c, so a leaf is ONE memory round trip where the
// loop of trav_other makes up to four dependent ones.  What that loop leaves behind -- it accepts a triangle when
// mint <= t <= maxt and shrinks maxt to t, so the nearest accepted triangle stays and, among equal distances, the one
// tested last -- is the minimum over the quad with ties going to the higher lane.  (Used where the reference's tie rule is
// compiled out, GBL `TIES` = false, and for any-hit queries; a NaN distance, which the loop would accept, loses here.)
// TIES (the builds that follow the reference's tie rule): two accepted distances EXACTLY equal -- two lanes of the quad, or a lane
// and the hit the ray already holds in this instance -- or a NaN distance (which trav_other's loop accepts) set st.tied for the
// end-of-query check (trace.h GBL_TIE_DETECT); an any-hit query leaves its occluder in st.hit.
template <bool ANY, bool STATS, bool TIES = false>
__device__ __forceinline__ bool quad_leaf(TravState& st, const QuadLane& ql, LaneCounters& cnt, bool* occluded, uint4 w0, uint4 w1, uint4 w2,
                                          uint32_t popped) {
    const uint32_t ref = ~static_cast<uint32_t>(st.cur);
    const uint32_t first = ref >> 2, count = (ref & 3u) + 1u;
    float t = INFINITY, b1 = 0.0f, b2 = 0.0f;
    bool ok = false;
    if (STATS && ql.c == 0u) cnt.tris += count;   // (the other lanes' counters are put back after the quad phase)
    if (ql.c < count) {
        float4 q0 = make_float4(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), 0.0f);
        float4 q1 = make_float4(__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z), 0.0f);
        float4 q2 = make_float4(__uint_as_float(w2.x), __uint_as_float(w2.y), __uint_as_float(w2.z), 0.0f);
        tri_fetch_together(q0, q1, q2);   // one memory round trip (trace.h)
        ok = tri_test_regs(q0, q1, q2, st.r.o, st.r.d, st.mint, st.maxt, &t, &b1, &b2);
    }
    const float tq = ok ? t : INFINITY;
    float m = fminf(tq, quad_dpp_f<GBL_QP_XOR1>(tq));
    m = fminf(m, quad_dpp_f<GBL_QP_XOR2>(m));
    if (TIES && !ANY) {
        // a NaN distance passes tri_test_regs' range check as it passes the loop's: mark the lane (fminf drops the NaN)
        uint32_t odd = (ok && t != t) ? 1u : 0u;
        odd |= quad_dpp<GBL_QP_XOR1>(odd);
        odd |= quad_dpp<GBL_QP_XOR2>(odd);
        uint32_t wm = (ok && tq == m) ? ql.bitc : 0u;
        wm |= quad_dpp<GBL_QP_XOR1>(wm);
        wm |= quad_dpp<GBL_QP_XOR2>(wm);
        if (odd != 0u || (m < INFINITY && ((wm & (wm - 1u)) != 0u || (m == st.hit.t && st.hit.inst == st.inst)))) st.tied = true;
    }
    if (m < INFINITY) {   // (the same in the quad's four lanes)
        uint32_t wm = (ok && tq == m) ? ql.bitc : 0u;
        wm |= quad_dpp<GBL_QP_XOR1>(wm);
        wm |= quad_dpp<GBL_QP_XOR2>(wm);
        const uint32_t cw = 31u - static_cast<uint32_t>(__clz(static_cast<int>(wm)));
        if (ANY) {
            if (TIES) {   // the occluder, for the end-of-query check
                st.hit.inst = st.inst;
                st.hit.tri = first + cw;
            }
            *occluded = true;
            return true;
        }
        uint32_t u1 = ql.c == cw ? __float_as_uint(b1) : 0u, u2 = ql.c == cw ? __float_as_uint(b2) : 0u;
        u1 |= quad_dpp<GBL_QP_XOR1>(u1);
        u2 |= quad_dpp<GBL_QP_XOR1>(u2);
        u1 |= quad_dpp<GBL_QP_XOR2>(u1);
        u2 |= quad_dpp<GBL_QP_XOR2>(u2);
        st.maxt = m;
        st.hit.t = m;
        st.hit.inst = st.inst;
        st.hit.tri = first + cw;
        st.hit.b1 = __uint_as_float(u1);
        st.hit.b2 = __uint_as_float(u2);
    }
    st.cur = static_cast<int>(popped);
    st.sp -= 1;
    return false;
}

// The two steps of trav_other that need the WORLD ray, for a ray held by a quad: entering an instance (a TLAS leaf) and
// leaving one (the sentinel on the stack).  The quad keeps only the ray of the space it is in; the world ray's origin and
// direction wait in words 6-11 of the ray's record, and leaving an instance runs ray_space() on them again -- the same
// arithmetic on the same operands as when the ray started.
template <bool STATS, bool EXT>
__device__ __forceinline__ void quad_transition(const DevScene& sc, TravState& st, const LdsStack& stk, const gbl_lds_u32* rec, LaneCounters& cnt,
                                                int filter) {
    if (STATS) probe(cnt.oth_lane, cnt.oth_wave);
    const F3 wo = f3(__uint_as_float(rec[6]), __uint_as_float(rec[7]), __uint_as_float(rec[8]));
    const F3 wd = f3(__uint_as_float(rec[9]), __uint_as_float(rec[10]), __uint_as_float(rec[11]));
    if (st.cur == GBL_STACK_SENTINEL) {   // finished an instance: back to the world ray
        ray_space(st.r, wo, wd);
        st.inst = -1;
        st.cur = static_cast<int>(stk.load(--st.sp));
        return;
    }
    const uint32_t ref = ~static_cast<uint32_t>(st.cur);
    const DevInstance* ip = sc.instances + (ref >> 2);
    if (EXT && filter != GBL_FILTER_NONE && (ip->is_mask != 0u ? GBL_FILTER_MASK : GBL_FILTER_OPAQUE) != filter) {
        st.cur = static_cast<int>(stk.load(--st.sp));
        return;
    }
    st.inst = static_cast<int>(ref >> 2);
    ray_space(st.r, xf_point(ip->inv, wo), xf_vector(ip->inv, wd));
    stk.store(st.sp++, GBL_STACK_SENTINEL);
    st.cur = ip->root;
}

// trace() for a whole wave: every lane of the wave calls it (`want`: the lane has a ray); the exits are wave-uniform.
// `slab`: this wave's 16 records; `wave_stack`: the LDS stack column of the wave's lane 0 (LdsStack layout).
// ANY = true : Scene::occluded;  ANY = false: Scene::intersect -- as trace() (trace.h).
#ifndef GBL_QUAD_PRIO
#define GBL_QUAD_PRIO 3          // s_setprio of a wave in its quad phase ...
#endif
#ifndef GBL_QUAD_PRIO_DENSE
#define GBL_QUAD_PRIO_DENSE 2    // ... and in the one-ray-per-lane phase of a query (shading runs at 0)
#endif
template <bool ANY, bool STATS, bool EXT, bool TIES, class STK>
__device__ __forceinline__ bool trace_quad(const DevScene& sc, bool want, F3 o, F3 d, float mint, float maxt, const STK& stk, gbl_lds_u32* slab,
                                           gbl_lds_u32* wave_stack, Hit& hit, LaneCounters& cnt, int filter = GBL_FILTER_NONE) {
    constexpr int TM = TIES ? GBL_TIE_DETECT : GBL_TIE_NONE;
    TravState st;
    if (want) {
        trav_begin(sc, st, o, d, mint, maxt, stk);
    } else {
        st.tied = false;
        st.sp = 0;
        st.cur = GBL_STACK_EXIT;
        st.inst = -1;
        st.mint = st.maxt = st.maxt0 = 0.0f;
        st.hit.t = INFINITY;
        st.hit.inst = -1;
        st.hit.tri = 0;
        st.hit.b1 = st.hit.b2 = 0.0f;
        st.r.o = st.r.d = st.r.idir = st.r.ood = f3(0.0f, 0.0f, 0.0f);
        st.world = st.r;
    }
    bool done = !want, occluded = false;
    uint32_t steps = 0;
#ifdef GBL_PHASE_CLOCK
    const unsigned long long pc_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long pc_t1 = pc_t0, pc_t2 = pc_t0, pc_t3 = pc_t0;
#endif
    // Wave priority: a wave inside a query outranks the waves that shade (2 over 0), one in its quad phase -- a chain of
    // dependent node fetches with a few instructions between them -- outranks both (3): its instructions issue the moment
    // their operands arrive instead of queueing behind a shading wave's.  45.6 -> 44.5 ms on config [1], Cornell 78.7 ->
    // 77.4, grid 23.0 -> 22.6 (quad phase alone at 3: 44.8; at 1 or 2: 44.9).
    __builtin_amdgcn_s_setprio(GBL_QUAD_PRIO_DENSE);
    // ---- more than 16 rays in flight: one ray per lane, as trace()
    unsigned long long live = __ballot(!done);
    // (Holding the migration back for a query's first 4 / 8 / 12 steps, so that short rays never pay for it, was measured:
    //  slower on every scene -- config [1] 50.3 / 50.4 / 51.5 against 46.5 ms.)
    // (So was making it wait for a ray with 3 / 5 / 8 entries on its stack -- one with work ahead of it: 50.0 / 50.8 / 54.2 ms.
    //  The scenes the EXT builds run, small ones, are slower under this kernel whether or not their rays ever migrate.)
    while (__popcll(live) > GBL_QUAD_MAX) {
        if (!done) {
#ifdef GBL_PHASE_CLOCK
            {
                const bool at_int = trav_at_interior(st);
                if (__ballot(at_int) != 0ull && at_int) {
                    const unsigned long long a0 = __builtin_amdgcn_s_memtime();
                    trav_interior<STATS, !ANY>(sc, st, stk, cnt);
                    asm volatile("" ::"v"(st.cur), "v"(st.sp));
                    cnt.pc[11] += __builtin_amdgcn_s_memtime() - a0;
                    cnt.pc[13] += 1;
                    cnt.pc[15] += __popcll(__ballot(1));
                }
                if (!at_int) {
                    const unsigned long long b0 = __builtin_amdgcn_s_memtime();
                    done = trav_other<ANY, STATS, EXT, LdsStack, TM, false>(sc, st, stk, cnt, &occluded, filter);
                    asm volatile("" ::"v"(st.cur), "v"(st.sp));
                    cnt.pc[12] += __builtin_amdgcn_s_memtime() - b0;
                    cnt.pc[14] += 1;
                    cnt.pc[16] += __popcll(__ballot(1));
                }
            }
#else
            // Lean builds: the leaf / instance step first (a leaf whose pop uncovers the sentinel and the exit marker takes those
            // at once, trav_other<FUSE>), then the interior step of whoever stands at an interior node by then -- bunny -1 %,
            // Cornell -5 %, grid -3 % against one step of either kind per iteration, which the EXT builds keep (register pressure).
            if constexpr (EXT) {
                if (trav_at_interior(st)) {
                    trav_interior<STATS, !ANY>(sc, st, stk, cnt);
                    if (STATS) ++steps;
                } else {
                    done = trav_other<ANY, STATS, EXT, LdsStack, TM, false>(sc, st, stk, cnt, &occluded, filter);
                }
            } else {
                if (!trav_at_interior(st)) done = trav_other<ANY, STATS, EXT, LdsStack, TM, true>(sc, st, stk, cnt, &occluded, filter);
                if (!done && trav_at_interior(st)) {
                    trav_interior<STATS, !ANY>(sc, st, stk, cnt);
                    if (STATS) ++steps;
                }
                // (a second leaf / instance step behind the interior one -- three steps per iteration -- was measured: 44.7 against
                //  42.0 ms on configs[1], 74.6 against 69.0 on the Cornell box)
            }
#endif
        }
        live = __ballot(!done);
#ifdef GBL_PHASE_CLOCK
        cnt.pc[9] += 1;
#endif
    }
    Hit res = st.hit;   // (of the lanes that are done)
#ifdef GBL_PHASE_CLOCK
    asm volatile("" ::"v"(st.cur), "v"(st.sp));
    pc_t1 = pc_t2 = pc_t3 = __builtin_amdgcn_s_memtime();
#endif
    if (live != 0ull) {
        // ---- migration: ray of rank k -> record k -> quad k
        const uint32_t lane = threadIdx.x & 63u, q = lane >> 2;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(live >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(live), 0u));
        const uint32_t nl = static_cast<uint32_t>(__popcll(live));
        const bool owner = !done;
        if (owner) {
            gbl_lds_u32* rec = slab + rank * GBL_QUAD_REC_WORDS;
            rec[0] = __float_as_uint(st.r.o.x); rec[1] = __float_as_uint(st.r.o.y); rec[2] = __float_as_uint(st.r.o.z);
            rec[3] = __float_as_uint(st.r.d.x); rec[4] = __float_as_uint(st.r.d.y); rec[5] = __float_as_uint(st.r.d.z);
            rec[6] = __float_as_uint(st.world.o.x); rec[7] = __float_as_uint(st.world.o.y); rec[8] = __float_as_uint(st.world.o.z);
            rec[9] = __float_as_uint(st.world.d.x); rec[10] = __float_as_uint(st.world.d.y); rec[11] = __float_as_uint(st.world.d.z);
            rec[12] = __float_as_uint(st.mint); rec[13] = __float_as_uint(st.maxt);
            rec[14] = static_cast<uint32_t>(st.cur); rec[15] = static_cast<uint32_t>(st.inst);
            rec[16] = static_cast<uint32_t>(st.hit.inst); rec[17] = st.hit.tri;
            rec[18] = __float_as_uint(st.hit.b1); rec[19] = __float_as_uint(st.hit.b2);
            rec[20] = static_cast<uint32_t>(st.sp) | (lane << 8);
        }
        quad_fence();
        const bool tied_before = st.tied;   // (of the lane's own ray; the lane's state now becomes its quad's ray's)
        st.tied = false;
        const bool qlive = q < nl;
        st.world.o = st.world.d = st.world.idir = st.world.ood = f3(0.0f, 0.0f, 0.0f);   // (not kept in registers from here on)
        QuadLane ql;
        ql.c = lane & 3u;
        ql.bitc = 1u << ql.c;
        ql.side1 = (ql.c & 1u) == 0u ? -INFINITY : INFINITY;
        ql.side2 = (ql.c & 2u) == 0u ? -INFINITY : INFINITY;
        ql.col = wave_stack + lane;
        bool qdone = true, qocc = false;
        uint32_t qsteps = 0;
        gbl_lds_u32* const qrec = slab + q * GBL_QUAD_REC_WORDS;
        if (qlive) {
            // the same arithmetic on the same operands as the lane that started the ray: ray_space() of (o, d).  The world ray
            // stays in the record until the ray enters or leaves an instance (quad_transition).
            ray_space(st.r, f3(__uint_as_float(qrec[0]), __uint_as_float(qrec[1]), __uint_as_float(qrec[2])),
                      f3(__uint_as_float(qrec[3]), __uint_as_float(qrec[4]), __uint_as_float(qrec[5])));
            st.inst = static_cast<int>(qrec[15]);
            st.mint = __uint_as_float(qrec[12]);
            st.maxt = __uint_as_float(qrec[13]);
            st.cur = static_cast<int>(qrec[14]);
            st.hit.inst = static_cast<int>(qrec[16]);
            st.hit.tri = qrec[17];
            st.hit.b1 = __uint_as_float(qrec[18]);
            st.hit.b2 = __uint_as_float(qrec[19]);
            st.hit.t = st.hit.inst >= 0 ? st.maxt : INFINITY;   // the accepted distance is the ray's maxt (trav_other)
            const uint32_t w = qrec[20];
            st.sp = static_cast<int>(w & 0xffu);
            ql.col = wave_stack + (w >> 8);
            qdone = false;
        }
        quad_fence();
        // ---- a quad per ray until the ray is done
        __builtin_amdgcn_s_setprio(GBL_QUAD_PRIO);
        const LdsStack qstk = {ql.col};
#ifdef GBL_PHASE_CLOCK
        pc_t2 = __builtin_amdgcn_s_memtime();
#endif
        uint32_t keep_tris = cnt.tris, keep_ol = cnt.oth_lane, keep_ow = cnt.oth_wave;
        int sel_inst = -2;   // the instance space the v_perm selectors were made for
        // one triangle per lane at a leaf (quad_leaf); the instrumented builds' any-hit queries keep trav_other's loop, whose
        // early exit is what their triangle counter counts
        constexpr bool QUAD_LEAVES = !(ANY && STATS);
        while (!qdone) {
#ifdef GBL_PHASE_CLOCK
            cnt.pc[10] += 1;
            cnt.pc[23] += __popcll(__ballot(1)) >> 2;
            const unsigned long long q0t = __builtin_amdgcn_s_memtime();
            const int qkind = trav_at_interior(st) ? 0 : ((st.cur < 0 && st.inst >= 0 && st.cur != GBL_STACK_SENTINEL && st.cur != GBL_STACK_EXIT) ? 1 : 2);
#endif
            // (Fetching the step's record -- node or triangle -- ahead of the branch on the kind of step, so that the two kinds'
            //  loads travel together, was measured: 49.5 against 48.8 ms.  Letting a ray take up to three steps of different kinds
            //  per iteration -- sequential tests instead of this chain, in the orders transition / interior / leaf, interior / leaf /
            //  transition and leaf / transition / interior: 45.0 / 43.5 / 44.6 against 43.8 ms.  Following a popped sentinel at the
            //  head of the iteration -- the world ray's idir / ood kept in the record, the exit marker ending the ray there -- so that
            //  leaving an instance costs no iteration of its own: 44.5 against 42.2 ms, Cornell 74.2 against 69.1, 51 spilled
            //  registers against 26.)
            if (trav_at_interior(st)) {
                const uint4* np = node_ptr(sc, stk, st.cur);
                const uint4 w0 = np[0], w1 = np[1], w2 = np[2];
                const uint32_t w3 = reinterpret_cast<const uint32_t*>(np)[12 + ql.c];
                const uint32_t popped = ql.col[(st.sp - 1) * GBL_BLOCK];   // the stack's top, should every child be missed
                if (sel_inst != st.inst) {   // (making them at the migration and after every transition instead: 47.4 against 45.5 ms)
                    quad_selectors(ql, st.r.idir);
                    sel_inst = st.inst;
                }
                quad_interior<!ANY, STATS>(st, ql, cnt, w0, w1, w2, w3, popped);
                if (STATS) ++qsteps;
            } else if (QUAD_LEAVES && st.cur < 0 && st.inst >= 0 && (!EXT || (~static_cast<uint32_t>(st.cur) >> 2) < GBL_SHAPE_FIRST_DISK)) {
                const uint32_t lref = ~static_cast<uint32_t>(st.cur);
                const uint4* tp = reinterpret_cast<const uint4*>(sc.tris + (lref >> 2) + min(ql.c, lref & 3u));
                const uint4 w0 = tp[0], w1 = tp[1], w2 = tp[2];
                const uint32_t popped = ql.col[(st.sp - 1) * GBL_BLOCK];
                qdone = quad_leaf<ANY, STATS, TIES>(st, ql, cnt, &qocc, w0, w1, w2, popped);
            } else if (st.cur == GBL_STACK_SENTINEL || (st.cur < 0 && st.inst < 0)) {
                quad_transition<STATS, EXT>(sc, st, qstk, qrec, cnt, filter);
            } else if (!EXT && QUAD_LEAVES) {   // lean builds: all that is left is the exit marker
                qdone = true;
                (void)qstk;
            } else {   // the exit marker; analytic shapes; any-hit leaves of the instrumented builds
                qdone = trav_other<ANY, STATS, EXT, LdsStack, TM, false>(sc, st, qstk, cnt, &qocc, filter);
            }
#ifdef GBL_PHASE_CLOCK
            {   // wave-level: the iteration's time goes to the kind of the wave's first live quad
                asm volatile("" ::"v"(st.cur), "v"(st.sp));
                const int k0 = __builtin_amdgcn_readfirstlane(qkind);
                cnt.pc[17 + k0] += __builtin_amdgcn_s_memtime() - q0t;
                cnt.pc[20 + k0] += 1;
            }
#endif
        }
#ifdef GBL_PHASE_CLOCK
        asm volatile("" ::"v"(st.cur), "v"(st.sp));
        pc_t3 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_setprio(GBL_QUAD_PRIO_DENSE);
        if (STATS && ql.c != 0u) {   // leaf / instance steps ran in all four lanes: count them once
            cnt.tris = keep_tris;
            cnt.oth_lane = keep_ol;
            cnt.oth_wave = keep_ow;
        }
        if (qlive && ql.c == 0u) {
            qrec[0] = static_cast<uint32_t>(st.hit.inst);
            qrec[1] = st.hit.tri;
            qrec[2] = __float_as_uint(st.hit.b1);
            qrec[3] = __float_as_uint(st.hit.b2);
            qrec[4] = __float_as_uint(st.hit.t);
            qrec[5] = (qocc ? 1u : 0u) | (st.tied ? 2u : 0u);
            if (STATS) qrec[12] = qsteps;
        }
        quad_fence();
        if (owner) {
            const gbl_lds_u32* rec = slab + rank * GBL_QUAD_REC_WORDS;
            res.inst = static_cast<int>(rec[0]);
            res.tri = rec[1];
            res.b1 = __uint_as_float(rec[2]);
            res.b2 = __uint_as_float(rec[3]);
            res.t = __uint_as_float(rec[4]);
            occluded = (rec[5] & 1u) != 0u;
            st.tied = tied_before || (rec[5] & 2u) != 0u;
            if (STATS) steps += rec[12];
        } else {
            st.tied = tied_before;
        }
        quad_fence();   // the slab is free for the next query
    }
    if constexpr (TIES) {
        // every ray's final hit against the reference's box tests, all lanes together; the rare ray that fails, or tied, again
        // in the exact loop (trace.h trace_needs_redo)
        // (of the 6.9 ms this mode costs configs[1]: the flag in the loops 2.2, this check 2.6, the exact loop's presence 2.2)
        const bool redo = want && trace_needs_redo(sc, EXT, ANY ? occluded : res.inst >= 0, res, st.tied, o, d, mint, maxt);
        if (__ballot(redo) != 0ull && redo) {
            Hit h;
            LaneCounters again = {};   // (the first pass counted this ray's visits)
            const bool g = trace_loop<ANY, STATS, EXT, GBL_TIE_EXACT>(sc, o, d, mint, maxt, stk, h, again, filter, nullptr);
            if (ANY) occluded = g;
            else res = h;
        }
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef GBL_PHASE_CLOCK
    {
        asm volatile("" ::"v"(res.t), "v"(res.inst));
        const unsigned long long pc_t4 = __builtin_amdgcn_s_memtime();
        unsigned long long* q = cnt.pc + (ANY ? 4 : 0);
        q[0] += pc_t4 - pc_t0;
        q[1] += pc_t1 - pc_t0;
        q[2] += (pc_t2 - pc_t1) + (pc_t4 - pc_t3);
        q[3] += pc_t3 - pc_t2;
    }
#endif
#ifndef GBL_PROBE_OCC
    if (STATS && !ANY && want) {
        int b = steps <= 3 ? 0 : min(6, 30 - __clz(static_cast<int>(steps)));
        cnt.hist[b] += 1;
        cnt.hist_steps[b] += steps;
    }
#endif
    if (ANY) return occluded;
    hit = res;
    return res.inst >= 0;
}
