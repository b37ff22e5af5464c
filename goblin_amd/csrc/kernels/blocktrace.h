// Workgroup-level ray tracing for the persistent megakernel: the 256 rays of a workgroup are traversed in rounds of a few
// node visits, and between rounds the rays still alive are packed into as few waves as hold them.
//
// Why: the rays a wave traces together differ wildly in length -- on bunny.json 64 % of the closest-hit rays finish within
// 3 interior steps (floor, sky) while the ones that enter the bunny's BLAS take 16-40 -- so a wave that traces its own 64
// rays to the end runs ~41 interior steps at 16 % lane utilisation (probe).  The kernel is bound by VALU
// issue, and a wave waiting at a barrier issues nothing: once the survivors of four waves fit into one, the other three
// step aside and the SIMDs go to the other resident workgroups.
//
// How: a ray's traversal state is 16 words (ray in its current space, maxt, node cursor, stack pointer, instance, hit,
// owner).  After each round every wave publishes its live count; if fewer waves could hold the survivors, each survivor is
// written to the slot of its rank (exclusive prefix over the workgroup) in an LDS exchange area and thread r adopts the
// ray in slot r.  The traversal stack is not copied: stacks are columns indexed by the ray's OWNER (the thread that
// issued it), and an adopted ray carries its owner along.  The owner's column of a second small area holds the world
// ray (re-entered when the ray leaves an instance) and, once the ray is done, its result, which the owner reads after
// the last round.  A ray's sequence of node visits and triangle tests is what it would be in its own lane, so hits,
// ties and radiance are unchanged bit for bit.
#pragma once
#include "trace.h"

#ifndef GBL_BT_ROUND0
#define GBL_BT_ROUND0 4   // steps of the first round (most rays end here)
#endif
#ifndef GBL_BT_ROUND
#define GBL_BT_ROUND 8    // steps of the later rounds
#endif
#define GBL_BT_STATE_WORDS 16
#define GBL_BT_OWNER_WORDS 7
#define GBL_BT_CTRL_WORDS 16
#define GBL_BT_LDS_WORDS ((GBL_BT_STATE_WORDS + GBL_BT_OWNER_WORDS) * GBL_BLOCK + GBL_BT_CTRL_WORDS)

struct BlockXch {
    gbl_lds_u32* state;   // GBL_BT_STATE_WORDS fields x GBL_BLOCK slots (field-major: a wave's accesses to one field are contiguous)
    gbl_lds_u32* owner;   // GBL_BT_OWNER_WORDS fields x GBL_BLOCK: world o, d, mint of thread tid's ray; later its result
    gbl_lds_u32* ctrl;    // [0..7] live lanes per wave, double-buffered by round parity; [8..11] per-wave flags of the caller
    gbl_lds_u32* stack;   // LDS part of the traversal stacks (SplitStack columns)
    gbl_glb_u32* spill;   // global backing of the deeper stack levels: this workgroup's GBL_BLOCK columns
    uint32_t spill_stride;
};

__device__ __forceinline__ SplitStack bt_stack(const BlockXch& x, int owner) {
    SplitStack s;
    s.p = x.stack + owner;
    s.g = x.spill + owner;
    s.gstride = x.spill_stride;
    return s;
}

// ANY / STATS / EXT / TIES / filter as trace() (trace.h).  Every thread of the workgroup calls this together; `valid`
// says whether this thread has a ray.  flag_in / flags_out: a per-wave boolean of the caller ORed over the workgroup on
// the way (the persistent loop's "any path still in flight"), so that the caller needs no barrier of its own for it.
// `phase` counts the rounds of all calls of this thread (uniform over the workgroup): consecutive rounds -- of one call or
// of two -- publish their live counts in alternating halves of ctrl[0..7], so a wave that runs ahead into the next round
// never overwrites counts a slower wave has yet to read (it cannot run two rounds ahead: there is a barrier per round).
template <bool ANY, bool STATS, bool EXT, bool TIES>
__device__ __forceinline__ bool trace_block(const DevScene& sc, bool valid, F3 o, F3 d, float mint, float maxt, const BlockXch& x, Hit& hit,
                                            LaneCounters& cnt, int filter, uint32_t& phase, bool flag_in, bool* flag_out) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    TravState st;
    int owner = tid;
    SplitStack stk = bt_stack(x, owner);
    bool live = valid;
    if (valid) {
        trav_begin(sc, st, o, d, mint, maxt, stk);
        x.owner[0 * GBL_BLOCK + tid] = __float_as_uint(o.x);
        x.owner[1 * GBL_BLOCK + tid] = __float_as_uint(o.y);
        x.owner[2 * GBL_BLOCK + tid] = __float_as_uint(o.z);
        x.owner[3 * GBL_BLOCK + tid] = __float_as_uint(d.x);
        x.owner[4 * GBL_BLOCK + tid] = __float_as_uint(d.y);
        x.owner[5 * GBL_BLOCK + tid] = __float_as_uint(d.z);
        x.owner[6 * GBL_BLOCK + tid] = __float_as_uint(mint);
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
    if (flag_out != nullptr) {
        const bool wave_flag = __ballot(flag_in) != 0ull;   // (the vote of the whole wave, taken before the one-lane store)
        if (lane == 0) x.ctrl[8 + wave] = wave_flag ? 1u : 0u;
    }
    for (uint32_t round = 0;; ++round) {
        // ---- a round: every live lane advances its ray by up to n node visits / leaf transitions
        int budget = round == 0 ? GBL_BT_ROUND0 : GBL_BT_ROUND;
        while (live && budget > 0) {
            --budget;
            bool done = false, occluded = false;
            if (trav_at_interior(st)) {
                trav_interior<STATS, !ANY>(sc, st, stk, cnt);
            } else {
                done = trav_other<ANY, STATS, EXT, SplitStack, TIES>(sc, st, stk, cnt, &occluded, filter);
            }
            if (done) {   // the result goes to the owner's column (its world ray is no longer needed)
                live = false;
                if (ANY) {
                    x.owner[0 * GBL_BLOCK + owner] = occluded ? 1u : 0u;
                } else {
                    x.owner[0 * GBL_BLOCK + owner] = __float_as_uint(st.hit.t);
                    x.owner[1 * GBL_BLOCK + owner] = static_cast<uint32_t>(st.hit.inst);
                    x.owner[2 * GBL_BLOCK + owner] = st.hit.tri;
                    x.owner[3 * GBL_BLOCK + owner] = __float_as_uint(st.hit.b1);
                    x.owner[4 * GBL_BLOCK + owner] = __float_as_uint(st.hit.b2);
                }
            }
        }
        // ---- who is left, workgroup-wide
        const unsigned long long m = __ballot(live);
        const uint32_t par = (phase++ & 1u) * 4u;
        if (lane == 0) x.ctrl[par + wave] = static_cast<uint32_t>(__popcll(m));
        __syncthreads();
        const uint32_t c0 = x.ctrl[par + 0], c1 = x.ctrl[par + 1], c2 = x.ctrl[par + 2], c3 = x.ctrl[par + 3];
        const uint32_t total = c0 + c1 + c2 + c3;
        if (round == 0 && flag_out != nullptr) *flag_out = (x.ctrl[8] | x.ctrl[9] | x.ctrl[10] | x.ctrl[11]) != 0u;
        if (total == 0u) break;
        const uint32_t waves_now = (c0 != 0u) + (c1 != 0u) + (c2 != 0u) + (c3 != 0u);
        const uint32_t waves_min = (total + 63u) >> 6;
        if (waves_min < waves_now) {
            // ---- pack the survivors: rank = exclusive prefix over the workgroup; thread r adopts slot r
            if (live) {
                const uint32_t base = wave == 0 ? 0u : (wave == 1 ? c0 : (wave == 2 ? c0 + c1 : c0 + c1 + c2));
                const uint32_t slot = base + static_cast<uint32_t>(__popcll(m & lt_mask));
                gbl_lds_u32* s = x.state + slot;
                s[0 * GBL_BLOCK] = __float_as_uint(st.r.o.x);
                s[1 * GBL_BLOCK] = __float_as_uint(st.r.o.y);
                s[2 * GBL_BLOCK] = __float_as_uint(st.r.o.z);
                s[3 * GBL_BLOCK] = __float_as_uint(st.r.d.x);
                s[4 * GBL_BLOCK] = __float_as_uint(st.r.d.y);
                s[5 * GBL_BLOCK] = __float_as_uint(st.r.d.z);
                s[6 * GBL_BLOCK] = __float_as_uint(st.maxt);
                s[7 * GBL_BLOCK] = static_cast<uint32_t>(st.cur);
                s[8 * GBL_BLOCK] = static_cast<uint32_t>(st.sp);
                s[9 * GBL_BLOCK] = static_cast<uint32_t>(st.inst);
                s[10 * GBL_BLOCK] = static_cast<uint32_t>(owner);
                if (!ANY) {
                    s[11 * GBL_BLOCK] = static_cast<uint32_t>(st.hit.inst);
                    s[12 * GBL_BLOCK] = st.hit.tri;
                    s[13 * GBL_BLOCK] = __float_as_uint(st.hit.b1);
                    s[14 * GBL_BLOCK] = __float_as_uint(st.hit.b2);
                }
            }
            __syncthreads();
            live = static_cast<uint32_t>(tid) < total;
            if (live) {
                const gbl_lds_u32* s = x.state + tid;
                const F3 ro = f3(__uint_as_float(s[0 * GBL_BLOCK]), __uint_as_float(s[1 * GBL_BLOCK]), __uint_as_float(s[2 * GBL_BLOCK]));
                const F3 rd = f3(__uint_as_float(s[3 * GBL_BLOCK]), __uint_as_float(s[4 * GBL_BLOCK]), __uint_as_float(s[5 * GBL_BLOCK]));
                st.maxt = __uint_as_float(s[6 * GBL_BLOCK]);
                st.cur = static_cast<int>(s[7 * GBL_BLOCK]);
                st.sp = static_cast<int>(s[8 * GBL_BLOCK]);
                st.inst = static_cast<int>(s[9 * GBL_BLOCK]);
                owner = static_cast<int>(s[10 * GBL_BLOCK]);
                if (!ANY) {
                    st.hit.inst = static_cast<int>(s[11 * GBL_BLOCK]);
                    st.hit.tri = s[12 * GBL_BLOCK];
                    st.hit.b1 = __uint_as_float(s[13 * GBL_BLOCK]);
                    st.hit.b2 = __uint_as_float(s[14 * GBL_BLOCK]);
                    st.hit.t = st.hit.inst >= 0 ? st.maxt : INFINITY;   // the accepted distance is the ray's maxt (trav_other)
                }
                stk = bt_stack(x, owner);
                const gbl_lds_u32* w = x.owner + owner;
                const F3 wo = f3(__uint_as_float(w[0 * GBL_BLOCK]), __uint_as_float(w[1 * GBL_BLOCK]), __uint_as_float(w[2 * GBL_BLOCK]));
                const F3 wd = f3(__uint_as_float(w[3 * GBL_BLOCK]), __uint_as_float(w[4 * GBL_BLOCK]), __uint_as_float(w[5 * GBL_BLOCK]));
                st.mint = __uint_as_float(w[6 * GBL_BLOCK]);
                ray_space(st.world, wo, wd);
                if (st.inst < 0) st.r = st.world;
                else ray_space(st.r, ro, rd);
            }
        }
    }
    // ---- every ray is done (the break above follows a barrier): the owner collects
    if (ANY) return valid && x.owner[0 * GBL_BLOCK + tid] != 0u;
    if (valid) {
        hit.t = __uint_as_float(x.owner[0 * GBL_BLOCK + tid]);
        hit.inst = static_cast<int>(x.owner[1 * GBL_BLOCK + tid]);
        hit.tri = x.owner[2 * GBL_BLOCK + tid];
        hit.b1 = __uint_as_float(x.owner[3 * GBL_BLOCK + tid]);
        hit.b2 = __uint_as_float(x.owner[4 * GBL_BLOCK + tid]);
    }
    return valid && hit.inst >= 0;
}
