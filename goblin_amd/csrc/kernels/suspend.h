// Suspendable closest-hit queries for the persistent megakernel: a wave stops tracing when only a few stragglers are left
// and most of its lanes have a result to shade; the stragglers' traversal state is parked (16 words per lane in global
// memory, their LDS stack columns stay as they are) and they resume inside the wave's NEXT extension query, beside the
// fresh rays of the other lanes.
//
// Why: the probe build counts interior wave-steps by the number of lanes inside them -- on bunny.json 59.6 %
// of them run with <= 4 lanes, 72 % with <= 8 (the rays that wander through the bunny's BLAS while 60 lanes wait).  The
// kernel is bound by VALU issue, so those steps cost what full ones cost.  Moving rays between lanes or waves (wavepool.h,
// blocktrace.h, rayexchange.h) lost to its own overhead or to idle waves; here no ray moves: the long ray just keeps its
// lane for one more iteration, overlapping with the next batch's traversal, and the lane sits out one shading pass.
// A ray's sequence of node visits and triangle tests is unchanged, so hits, ties and radiance are bit-identical.
#pragma once
#include "trace.h"

#ifndef GBL_SUSP_T
#define GBL_SUSP_T 4          // stop when at most this many lanes are still traversing ...
#endif
#ifndef GBL_SUSP_READY
#define GBL_SUSP_READY 32     // ... and at least this many lanes finished a ray in this call (something to shade meanwhile)
#endif
#ifndef GBL_SUSP_MIN_STEPS
#define GBL_SUSP_MIN_STEPS 4  // ... and the call has made this many steps (a resumed ray always advances)
#endif
#define GBL_SUSP_WORDS 16

// `start`: the lane issues (o, d, mint) as a new ray.  `resume`: the lane's ray -- the same (o, d, mint), which its path
// state still holds -- was parked by the previous call.  Returns true with `hit` filled when the lane's ray finished in
// this call and hit something; *finished says whether it finished at all, *parked that it was left for the next call.
template <bool STATS, bool EXT, bool TIES, class STK>
__device__ __forceinline__ bool trace_suspendable(const DevScene& sc, bool start, bool resume, F3 o, F3 d, float mint, gbl_glb_u32* park,
                                                  uint32_t stride, const STK& stk, Hit& hit, LaneCounters& cnt, bool* finished, bool* parked) {
    TravState st;
    bool live = start || resume;
    if (resume) {
        const F3 ro = f3(__uint_as_float(park[0 * stride]), __uint_as_float(park[1 * stride]), __uint_as_float(park[2 * stride]));
        const F3 rd = f3(__uint_as_float(park[3 * stride]), __uint_as_float(park[4 * stride]), __uint_as_float(park[5 * stride]));
        st.maxt = __uint_as_float(park[6 * stride]);
        st.cur = static_cast<int>(park[7 * stride]);
        st.sp = static_cast<int>(park[8 * stride]);
        st.inst = static_cast<int>(park[9 * stride]);
        st.hit.inst = static_cast<int>(park[10 * stride]);
        st.hit.tri = park[11 * stride];
        st.hit.b1 = __uint_as_float(park[12 * stride]);
        st.hit.b2 = __uint_as_float(park[13 * stride]);
        st.hit.t = st.hit.inst >= 0 ? st.maxt : INFINITY;   // the accepted distance is the ray's maxt (trav_other)
        st.mint = mint;
        ray_space(st.world, o, d);
        if (st.inst < 0) st.r = st.world;
        else ray_space(st.r, ro, rd);
    } else if (start) {
        trav_begin(sc, st, o, d, mint, INFINITY, stk);
    } else {
        st.sp = 0;
        st.cur = GBL_STACK_EXIT;
        st.inst = -1;
        st.mint = st.maxt = 0.0f;
        st.hit.t = INFINITY;
        st.hit.inst = -1;
        st.hit.tri = 0;
        st.hit.b1 = st.hit.b2 = 0.0f;
        st.r.o = st.r.d = st.r.idir = st.r.ood = f3(0.0f, 0.0f, 0.0f);
        st.world = st.r;
    }
    // The loop is trace()'s own -- every lane takes the step its ray needs and leaves when the ray is done -- plus a look at
    // the wave every fourth step, all scalar: how many lanes entered, how many are still here.  (A wave-uniform form of the
    // loop -- ballots of live / done / at-interior lanes every iteration -- halved the interior steps as well but doubled
    // the kernel's scalar instructions and came out slower than not suspending at all.)
    bool done = false;
    const int entered = __popcll(__ballot(live));
    if (live) {
        uint32_t steps = 0;
        for (;;) {
            if (trav_at_interior(st)) {
                trav_interior<STATS, true>(sc, st, stk, cnt);
            } else {
                bool occluded = false;
                if (trav_other<false, STATS, EXT, STK, TIES>(sc, st, stk, cnt, &occluded, GBL_FILTER_NONE)) {
                    live = false;
                    done = true;
                    break;
                }
            }
            ++steps;
            if ((steps & 3u) == 0u && steps >= GBL_SUSP_MIN_STEPS) {
                const int here = __popcll(__ballot(1));   // the lanes still in this loop
                if (here <= GBL_SUSP_T && entered - here >= GBL_SUSP_READY) break;
            }
        }
    }
    if (live) {   // park: everything but the ray itself, which the caller keeps
        park[0 * stride] = __float_as_uint(st.r.o.x);
        park[1 * stride] = __float_as_uint(st.r.o.y);
        park[2 * stride] = __float_as_uint(st.r.o.z);
        park[3 * stride] = __float_as_uint(st.r.d.x);
        park[4 * stride] = __float_as_uint(st.r.d.y);
        park[5 * stride] = __float_as_uint(st.r.d.z);
        park[6 * stride] = __float_as_uint(st.maxt);
        park[7 * stride] = static_cast<uint32_t>(st.cur);
        park[8 * stride] = static_cast<uint32_t>(st.sp);
        park[9 * stride] = static_cast<uint32_t>(st.inst);
        park[10 * stride] = static_cast<uint32_t>(st.hit.inst);
        park[11 * stride] = st.hit.tri;
        park[12 * stride] = __float_as_uint(st.hit.b1);
        park[13 * stride] = __float_as_uint(st.hit.b2);
    }
    *parked = live;
    *finished = done;
    hit = st.hit;
    return done && st.hit.inst >= 0;
}
