// Ray exchange for the persistent megakernel: the waves of a workgroup hand the long rays to one another through LDS, without
// barriers, so that the lanes of a traversing wave stay busy and no wave ever waits idle while there is a ray to advance.
//
// Why: the rays a wave traces together differ wildly in length -- on bunny.json 64 % of the closest-hit rays
// finish within 3 interior steps while those that enter the bunny's BLAS take 16-40 -- so tracing its own 64 rays to the
// end a wave runs ~41 interior steps at 16 % lane utilisation.  Packing the survivors of the four waves at barriers
// (kernels/blocktrace.h) raises the utilisation to 40 % but is slower: the waves that stepped aside wait, and twelve waves
// per CU are too few to cover the ones that wait.  Here nobody waits:
//   * a wave advances its own rays for GBL_RX_ROUND0 steps (most end there, their results never leave the registers);
//   * if fewer than GBL_RX_KEEP survive it DONATES them: a ray's traversal state (16 words) goes to the LDS record of its
//     OWNER (the thread that issued it) and the owner's id into a ring; the traversal stack is not copied -- stacks are
//     columns indexed by owner, LDS levels and global backing alike;
//   * then, until the results of all its own rays are in, the wave HELPS: it tops its idle lanes up with rays from the ring
//     (any wave's, shadow and extension rays side by side, the kind travels with the ray) and advances them GBL_RX_ROUND steps
//     at a time; a finished ray's result goes to its owner's column with a flag the owner polls.
// A wave leaves when its own results are in and donates what it still holds; the owners of those rays are, by definition,
// still inside and pick them up.  Every ray is therefore always in the ring or in somebody's lane, and a ray's sequence
// of node visits and triangle tests is what it would be in its own lane: hits, ties and radiance are unchanged bit for bit.
#pragma once
#include "trace.h"

#ifndef GBL_RX_ROUND0
#define GBL_RX_ROUND0 4    // steps on the wave's own rays before the survivors are donated
#endif
#ifndef GBL_RX_ROUND
#define GBL_RX_ROUND 8     // steps between two top-ups of a helping wave
#endif
#ifndef GBL_RX_KEEP
#define GBL_RX_KEEP 48     // a wave with at least this many survivors keeps them
#endif
#ifndef GBL_RX_FILL
#define GBL_RX_FILL 40     // a wave holding nothing takes over the ring at once when at least this many rays wait there ...
#endif
#ifndef GBL_RX_PATIENCE
#define GBL_RX_PATIENCE 24 // ... or after this many polls, or when no other wave of the workgroup is traversing
#endif
#ifndef GBL_RX_POLL_LIMIT
#define GBL_RX_POLL_LIMIT (1u << 24)   // polls without progress after which a wave gives up loudly (s_trap: the launch fails) rather than spin forever
#endif
#define GBL_RX_REC_WORDS 16
#define GBL_RX_OWN_WORDS 7
#define GBL_RX_EMPTY 0xffffffffu
#define GBL_RX_LDS_WORDS ((GBL_RX_REC_WORDS + GBL_RX_OWN_WORDS + 2) * GBL_BLOCK + 4)

struct RayXch {
    gbl_lds_u32* rec;     // GBL_RX_REC_WORDS fields x GBL_BLOCK: the state of owner o's ray while nobody holds it (field-major)
    gbl_lds_u32* own;     // GBL_RX_OWN_WORDS fields x GBL_BLOCK: world o, d, mint of owner o's ray; later its result
    gbl_lds_u32* flag;    // GBL_BLOCK: 1 = owner o's result is in
    gbl_lds_u32* ring;    // GBL_BLOCK owner ids waiting for a lane, GBL_RX_EMPTY = free entry
    gbl_lds_u32* ctl;     // [0] head, [1] tail of the ring (free-running counters), [2] waves of the workgroup that hold rays in the helping loop
    gbl_lds_u32* stack;   // LDS part of the traversal stacks (SplitStack columns)
    gbl_glb_u32* spill;   // global backing of the deeper stack levels: this workgroup's GBL_BLOCK columns
    uint32_t spill_stride;
};

__device__ __forceinline__ SplitStack rx_stack(const RayXch& x, int owner) {
    SplitStack s;
    s.p = x.stack + owner;
    s.g = x.spill + owner;
    s.gstride = x.spill_stride;
    return s;
}

// once per kernel, before the first barrier
__device__ __forceinline__ void rx_init(const RayXch& x) {
    for (int i = threadIdx.x; i < GBL_BLOCK; i += GBL_BLOCK) {
        x.ring[i] = GBL_RX_EMPTY;
        x.flag[i] = 0u;
    }
    if (threadIdx.x < 4) x.ctl[threadIdx.x] = 0u;
}

__device__ __forceinline__ uint32_t rx_load(gbl_lds_u32* p) { return __hip_atomic_load((uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void rx_store(gbl_lds_u32* p, uint32_t v) { __hip_atomic_store((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// the lanes of `mask` (all live) hand their rays over
__device__ __forceinline__ void rx_donate(const RayXch& x, bool give, const TravState& st, int owner, uint32_t kind) {
    const unsigned long long m = __ballot(give);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    if (give) {
        gbl_lds_u32* s = x.rec + owner;
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
        s[10 * GBL_BLOCK] = kind;
        s[11 * GBL_BLOCK] = static_cast<uint32_t>(st.hit.inst);
        s[12 * GBL_BLOCK] = st.hit.tri;
        s[13 * GBL_BLOCK] = __float_as_uint(st.hit.b1);
        s[14 * GBL_BLOCK] = __float_as_uint(st.hit.b2);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // records (and the stacks) before the ring entries
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    uint32_t base = 0;
    if (lane == leader) base = __hip_atomic_fetch_add((uint32_t*)(x.ctl + 1), static_cast<uint32_t>(__popcll(m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = __shfl(base, leader);
    if (give) rx_store(x.ring + ((base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)))) & (GBL_BLOCK - 1)), static_cast<uint32_t>(owner));
}

// idle lanes (`!live`) take rays from the ring; returns nothing -- live / st / owner / kind / stk are updated in place
__device__ __forceinline__ void rx_grab(const DevScene& sc, const RayXch& x, bool& live, TravState& st, int& owner, uint32_t& kind, SplitStack& stk) {
    const unsigned long long idle = ~__ballot(live);
    if (idle == 0ull) return;
    const int lane = threadIdx.x & 63;
    const uint32_t want = static_cast<uint32_t>(__popcll(idle));
    uint32_t h = 0, n = 0;
    if (lane == 0) {
        for (;;) {
            h = rx_load(x.ctl + 0);
            const uint32_t t = rx_load(x.ctl + 1);
            n = t - h;
            if (n > want) n = want;
            if (n == 0u) break;
            uint32_t expect = h;
            if (__hip_atomic_compare_exchange_strong((uint32_t*)(x.ctl + 0), &expect, h + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        }
    }
    h = __shfl(h, 0);
    n = __shfl(n, 0);
    if (n == 0u) return;
    const uint32_t rank = static_cast<uint32_t>(__popcll(idle & ((1ull << lane) - 1ull)));
    const bool take = !live && rank < n;
    if (take) {
        gbl_lds_u32* e = x.ring + ((h + rank) & (GBL_BLOCK - 1));
        uint32_t id;
        while ((id = rx_load(e)) == GBL_RX_EMPTY) {}   // (claimed between the donor's tail bump and its store: a few cycles)
        rx_store(e, GBL_RX_EMPTY);
        owner = static_cast<int>(id);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (take) {
        const gbl_lds_u32* s = x.rec + owner;
        const F3 ro = f3(__uint_as_float(s[0 * GBL_BLOCK]), __uint_as_float(s[1 * GBL_BLOCK]), __uint_as_float(s[2 * GBL_BLOCK]));
        const F3 rd = f3(__uint_as_float(s[3 * GBL_BLOCK]), __uint_as_float(s[4 * GBL_BLOCK]), __uint_as_float(s[5 * GBL_BLOCK]));
        st.maxt = __uint_as_float(s[6 * GBL_BLOCK]);
        st.cur = static_cast<int>(s[7 * GBL_BLOCK]);
        st.sp = static_cast<int>(s[8 * GBL_BLOCK]);
        st.inst = static_cast<int>(s[9 * GBL_BLOCK]);
        kind = s[10 * GBL_BLOCK];
        st.hit.inst = static_cast<int>(s[11 * GBL_BLOCK]);
        st.hit.tri = s[12 * GBL_BLOCK];
        st.hit.b1 = __uint_as_float(s[13 * GBL_BLOCK]);
        st.hit.b2 = __uint_as_float(s[14 * GBL_BLOCK]);
        st.hit.t = st.hit.inst >= 0 ? st.maxt : INFINITY;   // the accepted distance is the ray's maxt (trav_other)
        stk = rx_stack(x, owner);
        const gbl_lds_u32* w = x.own + owner;
        const F3 wo = f3(__uint_as_float(w[0 * GBL_BLOCK]), __uint_as_float(w[1 * GBL_BLOCK]), __uint_as_float(w[2 * GBL_BLOCK]));
        const F3 wd = f3(__uint_as_float(w[3 * GBL_BLOCK]), __uint_as_float(w[4 * GBL_BLOCK]), __uint_as_float(w[5 * GBL_BLOCK]));
        st.mint = __uint_as_float(w[6 * GBL_BLOCK]);
        ray_space(st.world, wo, wd);
        if (st.inst < 0) st.r = st.world;
        else ray_space(st.r, ro, rd);
        live = true;
    }
}

// a ray held for somebody (or for this thread itself, after a round trip) is done: the result to the owner's column
__device__ __forceinline__ void rx_publish(const RayXch& x, int owner, bool any, bool occluded, const Hit& h) {
    if (any) {
        x.own[0 * GBL_BLOCK + owner] = occluded ? 1u : 0u;
    } else {
        x.own[0 * GBL_BLOCK + owner] = __float_as_uint(h.t);
        x.own[1 * GBL_BLOCK + owner] = static_cast<uint32_t>(h.inst);
        x.own[2 * GBL_BLOCK + owner] = h.tri;
        x.own[3 * GBL_BLOCK + owner] = __float_as_uint(h.b1);
        x.own[4 * GBL_BLOCK + owner] = __float_as_uint(h.b2);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    rx_store(x.flag + owner, 1u);
}

// ANY / STATS / EXT / TIES / filter as trace() (trace.h).  All lanes of the WAVE call this together (waves are not in step
// with one another); `valid` says whether this lane has a ray.  TIES is the closest-hit rays' (any-hit rays end before the
// tie rule), so one kernel's rays all agree on it.
template <bool ANY, bool STATS, bool EXT, bool TIES>
__device__ __forceinline__ bool trace_rx(const DevScene& sc, bool valid, F3 o, F3 d, float mint, float maxt, const RayXch& x, Hit& hit, LaneCounters& cnt,
                                         int filter) {
    const int tid = threadIdx.x;
    TravState st;
    int owner = tid;
    uint32_t kind = (ANY ? 1u : 0u) | (static_cast<uint32_t>(filter) << 1);
    SplitStack stk = rx_stack(x, owner);
    bool live = valid, pending = valid, occluded = false;
    if (valid) {
        trav_begin(sc, st, o, d, mint, maxt, stk);
        x.own[0 * GBL_BLOCK + tid] = __float_as_uint(o.x);
        x.own[1 * GBL_BLOCK + tid] = __float_as_uint(o.y);
        x.own[2 * GBL_BLOCK + tid] = __float_as_uint(o.z);
        x.own[3 * GBL_BLOCK + tid] = __float_as_uint(d.x);
        x.own[4 * GBL_BLOCK + tid] = __float_as_uint(d.y);
        x.own[5 * GBL_BLOCK + tid] = __float_as_uint(d.z);
        x.own[6 * GBL_BLOCK + tid] = __float_as_uint(mint);
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
    hit.t = INFINITY;
    hit.inst = -1;
    hit.tri = 0;
    hit.b1 = hit.b2 = 0.0f;
    // ---- the wave's own rays, one kind: results stay in registers
    {
        int budget = GBL_RX_ROUND0;
        while (live && budget > 0) {
            --budget;
            bool done = false, occ = false;
            if (trav_at_interior(st)) {
                trav_interior<STATS, !ANY>(sc, st, stk, cnt);
            } else {
                done = trav_other<ANY, STATS, EXT, SplitStack, TIES>(sc, st, stk, cnt, &occ, filter);
            }
            if (done) {
                live = false;
                pending = false;
                occluded = occ;
                hit = st.hit;
            }
        }
    }
    {
        const uint32_t c = static_cast<uint32_t>(__popcll(__ballot(live)));
        if (c != 0u && c < GBL_RX_KEEP) {
            rx_donate(x, live, st, owner, kind);
            live = false;
        }
    }
    // ---- until this wave's results are in: collect, top up, advance
    // A wave that holds rays tops its idle lanes up at every round boundary: that is where the donated rays of the others
    // usually go.  A wave that holds nothing does not take its own few survivors straight back (nothing would be gained): it
    // waits for a traversing wave to absorb them, and takes over the ring itself when enough rays have gathered there, when
    // nobody else is traversing, or when its patience runs out.
    const int lane = tid & 63;
    bool tracer = false;
    if (__ballot(live) != 0ull) {
        tracer = true;
        if (lane == 0) __hip_atomic_fetch_add((uint32_t*)(x.ctl + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    uint32_t waited = 0, polls = 0;
    for (;;) {
        if (pending && rx_load(x.flag + tid) != 0u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (ANY) {
                occluded = x.own[0 * GBL_BLOCK + tid] != 0u;
            } else {
                hit.t = __uint_as_float(x.own[0 * GBL_BLOCK + tid]);
                hit.inst = static_cast<int>(x.own[1 * GBL_BLOCK + tid]);
                hit.tri = x.own[2 * GBL_BLOCK + tid];
                hit.b1 = __uint_as_float(x.own[3 * GBL_BLOCK + tid]);
                hit.b2 = __uint_as_float(x.own[4 * GBL_BLOCK + tid]);
            }
            rx_store(x.flag + tid, 0u);
            pending = false;
        }
        const bool holding = __ballot(live) != 0ull;
        if (__ballot(pending) == 0ull) {
            rx_donate(x, live, st, owner, kind);   // what it still holds goes back: the owners are inside and take it
            if (tracer && lane == 0) __hip_atomic_fetch_sub((uint32_t*)(x.ctl + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
        if (!holding) {
            if (tracer) {
                tracer = false;
                if (lane == 0) __hip_atomic_fetch_sub((uint32_t*)(x.ctl + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const uint32_t avail = rx_load(x.ctl + 1) - rx_load(x.ctl + 0);
            const bool wait = avail == 0u || (avail < GBL_RX_FILL && waited < GBL_RX_PATIENCE && rx_load(x.ctl + 2) != 0u);
            if (wait) {
                if (avail != 0u) ++waited;
                if (++polls > GBL_RX_POLL_LIMIT) __builtin_trap();
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
        }
        rx_grab(sc, x, live, st, owner, kind, stk);
        if (__ballot(live) == 0ull) continue;
        waited = 0;
        polls = 0;
        if (!tracer) {
            tracer = true;
            if (lane == 0) __hip_atomic_fetch_add((uint32_t*)(x.ctl + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        int budget = GBL_RX_ROUND;
        while (live && budget > 0) {
            --budget;
            bool done = false, occ = false;
            if (trav_at_interior(st)) {
                trav_interior<STATS, true>(sc, st, stk, cnt);   // sorted for every kind: an any-hit query's answer does not depend on the order
            } else {
                done = trav_other_kind<STATS, EXT, SplitStack, TIES>(sc, st, stk, cnt, (kind & 1u) != 0u, &occ, static_cast<int>(kind >> 1));
            }
            if (done) {
                live = false;
                rx_publish(x, owner, (kind & 1u) != 0u, occ, st.hit);
            }
        }
    }
    if (ANY) return valid && occluded;
    return valid && hit.inst >= 0;
}
