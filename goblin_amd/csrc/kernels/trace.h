// Two-level BVH traversal + Moller-Trumbore for one ray per lane.
//
// Replaces Scene::intersect/occluded -> BVH::intersect/occluded ->
// InstancedPrimitive -> Model -> Triangle::intersect/occluded
// (GoblinScene.cpp:75-87, GoblinBVH.cpp:189-280, GoblinPrimitive.cpp:103-118,
// GoblinModel.cpp:28-55, GoblinTriangle.cpp:38-163).
//
// * 4-wide tree, 8-bit quantised child boxes: one 64-byte node fetch (four 16-byte
//   loads) tests four children.  The per-lane cost that bounds this kernel on gfx950 is
//   the L1 address pipe (about one lane-address per cycle for divergent 16-byte
//   gathers), so halving the steps per ray halves that cost and the number of dependent
//   memory round trips.  Boxes are rounded outwards: conservative, radiance unaffected.
// * Slab test in fused multiply-add form on the quantisation grid:
//       t = (o_a + q * s_a - ray.o_a) / d_a = fma(q, s_a * idir_a, fma(o_a, idir_a, -ray.o_a * idir_a))
//   with the near / far plane bytes of all four children selected per ray direction once per node
// * The world ray enters an instance by the same un-normalised inverse transform the
//   reference uses (Transform::invertRay), so object-space t is world t and hits from
//   different instances compare directly.
// * The triangle test is Moller-Trumbore in the reference's exact operation order
//   (-ffp-contract=off), +-1e-7 barycentric slack, inclusive [mint, maxt], no culling.
// * Per-lane stack lives in LDS, column-major over the workgroup
//   (stack[level * GBL_BLOCK + tid]): a lane only ever touches its own bank column.
// * trav_interior() / trav_other() advance a ray by ONE node visit or ONE leaf / instance
//   transition; callers phase them by majority state (see GBL_TRAV_TH) and the wavefront
//   trace kernel refills idle lanes between steps.
#pragma once
#include "../device_scene.h"
#include "vecmath.h"

#define GBL_STACK_EXIT 0x7ffffffe

struct Hit {
    float t;
    int inst;
    uint32_t tri;   // index into DevScene::tris
    float b1, b2;
};

struct LaneCounters {
    uint32_t ext, shadow, nodes, tris, splats, dims;
    // SIMD-efficiency probes (instrumented builds only): per phase, lane-steps executed and
    // wave-steps issued; utilisation = lane / (64 * wave)
    uint32_t int_lane, int_wave, oth_lane, oth_wave;
    // rays by interior steps: [0] <= 3, [1] 4-7, [2] 8-15, [3] 16-31, [4] 32-63, [5] 64-127, [6] >= 128;
    // hist_steps[b] sums the steps of bin b (closest-hit rays of the megakernel's trace())
    uint32_t hist[7], hist_steps[7];
#ifdef GBL_PHASE_CLOCK
    // measurement build (tools/phase_clock.py): shader-clock ticks of this wave per phase of the lean quad kernel --
    // [0..3] closest-hit query: whole call, one-ray-per-lane loop, migration + hand-back, quad loop; [4..7] the same for the
    // any-hit query; [8] the kernel; [9] / [10] dense / quad loop iterations; dense loop: [11] / [12] ticks inside interior / other
    // blocks, [13] / [14] their executions, [15] / [16] lanes inside them; quad loop: [17..19] ticks of interior / leaf / other
    // iterations, [20..22] their counts, [23] rays (quads) alive summed over iterations
    unsigned long long pc[24];
#endif
};

// first active lane of the current exec mask adds one wave-step
__device__ __forceinline__ void probe(uint32_t& lane_ctr, uint32_t& wave_ctr) {
    unsigned long long m = __ballot(1);
    lane_ctr += 1;
    if ((threadIdx.x & 63) == __ffsll(static_cast<long long>(m)) - 1) wave_ctr += 1;
}

struct RaySpace {
    F3 o, d, idir, ood;
};

__device__ __forceinline__ void ray_space(RaySpace& r, F3 o, F3 d) {
    r.o = o;
    r.d = d;
    // clamp tiny components so the fma-form slab test never produces inf - inf
    const float tiny = 1e-30f;
    float dx = fabsf(d.x) > tiny ? d.x : copysignf(tiny, d.x);
    float dy = fabsf(d.y) > tiny ? d.y : copysignf(tiny, d.y);
    float dz = fabsf(d.z) > tiny ? d.z : copysignf(tiny, d.z);
    r.idir = f3(1.0f / dx, 1.0f / dy, 1.0f / dz);
    r.ood = f3(o.x * r.idir.x, o.y * r.idir.y, o.z * r.idir.z);
}

// Triangle::intersect's acceptance test (GoblinTriangle.cpp:52-78) on a fetched DevTri (q0 = p0, q1 = e1, q2 = e2).
__device__ __forceinline__ bool tri_test_regs(float4 q0, float4 q1, float4 q2, F3 o, F3 d, float mint, float maxt, float* t_out, float* b1_out,
                                              float* b2_out) {
    F3 p0 = f3(q0.x, q0.y, q0.z), e1 = f3(q1.x, q1.y, q1.z), e2 = f3(q2.x, q2.y, q2.z);
    F3 s1 = cross(d, e2);
    float divisor = dot(s1, e1);
    if (divisor == 0.0f) return false;
    float inv = 1.0f / divisor;
    const float eps = 1e-7f;
    F3 s = o - p0;
    float b1 = dot(s, s1) * inv;
    if (b1 + eps < 0.0f || b1 - eps > 1.0f) return false;
    F3 s2 = cross(s, e1);
    float b2 = dot(d, s2) * inv;
    if (b2 + eps < 0.0f || b1 + b2 - eps > 1.0f) return false;
    float t = dot(e2, s2) * inv;
    if (t < mint || t > maxt) return false;
    *t_out = t;
    *b1_out = b1;
    *b2_out = b2;
    return true;
}
// The three loads of a triangle record travel together: left alone the compiler sinks the p0 load below the `divisor == 0`
// early-out, which makes every triangle test two dependent memory round trips instead of one (the traversal loops wait on
// memory 43 % of their time, profiles/r03_bunny.json).  The empty asm needs all nine floats, so they are fetched ahead of it.
__device__ __forceinline__ void tri_fetch_together(float4& q0, float4& q1, float4& q2) {
    asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q2.x), "+v"(q2.y), "+v"(q2.z));
}
__device__ __forceinline__ bool tri_test(const DevTri* tp, F3 o, F3 d, float mint, float maxt, float* t_out, float* b1_out,
                                         float* b2_out) {
    float4 q0 = reinterpret_cast<const float4*>(tp)[0];
    float4 q1 = reinterpret_cast<const float4*>(tp)[1];
    float4 q2 = reinterpret_cast<const float4*>(tp)[2];
    tri_fetch_together(q0, q1, q2);
    return tri_test_regs(q0, q1, q2, o, d, mint, maxt, t_out, b1_out, b2_out);
}

// Everything a lane carries for the ray it is traversing.
struct TravState {
    RaySpace world, r;   // world-space ray and the ray in the current space (world or instance)
    float mint, maxt;
    float maxt0;         // the query's own maxt (maxt shrinks with every accepted hit); read by the exact loop only (GBL_TIE_EXACT)
    bool tied;           // GBL_TIE_DETECT: a triangle was accepted at exactly the distance of the hit the ray held
    int sp, cur, inst;
    Hit hit;
};

// Traversal stacks.  LdsStack: the whole per-lane stack in LDS, column-major over the workgroup (megakernel, AO).
// SplitStack: the first GBL_WF_STACK_LDS levels in LDS and the (rarely reached) deeper ones in a global backing
// column, so the trace kernels of the wavefront schedule fit more workgroups per CU (44 KB -> 16 KB of LDS).
// The pointers carry their address space explicitly: through a plain `uint32_t*` member the compiler loses track of
// it on some paths and pops the stack with flat_load (slower than ds_read, and it ties vmcnt to lgkmcnt).
typedef __attribute__((address_space(3))) uint32_t gbl_lds_u32;
typedef __attribute__((address_space(1))) uint32_t gbl_glb_u32;
__device__ __forceinline__ gbl_lds_u32* gbl_as_lds(uint32_t* p) { return (gbl_lds_u32*)p; }
__device__ __forceinline__ gbl_glb_u32* gbl_as_global(uint32_t* p) { return (gbl_glb_u32*)p; }

struct LdsStack {
    gbl_lds_u32* p;
    __device__ __forceinline__ void store(int i, uint32_t v) const { p[i * GBL_BLOCK] = v; }
    __device__ __forceinline__ uint32_t load(int i) const { return p[i * GBL_BLOCK]; }
};
// LdsStack + the top of the two-level tree in LDS.  The first DevScene::hot_nodes nodes of the node array are the breadth-first
// prefix of the tree (scene_prep.cpp); the lean quad kernels copy as many of them as fit beside their stacks into LDS once per
// workgroup (RenderArgs::hot_count) and fetch a node whose reference lies below that count from there.  One instruction stream
// for both: the pointer is chosen per lane and the four 16-byte loads go through it (flat_load: LDS aperture or memory).
typedef __attribute__((address_space(3))) uint4 gbl_lds_u4;
struct HotLdsStack : LdsStack {
    const gbl_lds_u4* hot;
    uint32_t hot_count;
};
template <class STK> struct stk_is_hot { static constexpr bool value = false; };
template <> struct stk_is_hot<HotLdsStack> { static constexpr bool value = true; };
struct HotSplitStack;
template <> struct stk_is_hot<HotSplitStack> { static constexpr bool value = true; };
template <class STK>
__device__ __forceinline__ const uint4* node_ptr(const DevScene& sc, const STK& stk, int cur) {
    const uint4* g = reinterpret_cast<const uint4*>(sc.nodes + cur);
    if constexpr (stk_is_hot<STK>::value) {
        const uint4* l = (const uint4*)(stk.hot + 4 * cur);
        return static_cast<uint32_t>(cur) < stk.hot_count ? l : g;
    } else {
        return g;
    }
}
#ifndef GBL_WF_STACK_LDS
#define GBL_WF_STACK_LDS 16
#endif
struct SplitStack {
    gbl_lds_u32* p;     // LDS column of this lane
    gbl_glb_u32* g;     // global backing column of this thread
    uint32_t gstride;   // threads in the grid
    __device__ __forceinline__ void store(int i, uint32_t v) const {
        if (i < GBL_WF_STACK_LDS) p[i * GBL_BLOCK] = v;
        else g[static_cast<size_t>(i - GBL_WF_STACK_LDS) * gstride] = v;
    }
    __device__ __forceinline__ uint32_t load(int i) const {
        return i < GBL_WF_STACK_LDS ? p[i * GBL_BLOCK] : g[static_cast<size_t>(i - GBL_WF_STACK_LDS) * gstride];
    }
};

struct HotSplitStack : SplitStack {   // the wavefront trace kernels' stack + the top of the tree in LDS (see HotLdsStack)
    const gbl_lds_u4* hot;
    uint32_t hot_count;
};

template <class STK>
__device__ __forceinline__ void trav_begin(const DevScene& sc, TravState& st, F3 o, F3 d, float mint, float maxt, const STK& stk) {
    ray_space(st.world, o, d);
    st.r = st.world;
    st.mint = mint;
    st.maxt = maxt;
    st.maxt0 = maxt;
    st.tied = false;
    st.sp = 0;
    stk.store(st.sp++, GBL_STACK_EXIT);
    st.cur = sc.num_instances > 0 ? sc.tlas_root : GBL_STACK_EXIT;
    st.inst = -1;
    st.hit.t = INFINITY;
    st.hit.inst = -1;
    st.hit.tri = 0;
    st.hit.b1 = st.hit.b2 = 0.0f;
}

// Entry distance of one child, INFINITY if the ray misses it.  `n*` / `f*` hold the grid
// coordinates of the child's planes the ray crosses first / last on each axis (chosen per ray
// direction once per node), so no per-child min/max is needed, and an unused slot
// (qlo = 255 > qhi = 0) is an empty interval for every direction.  (Pairing the near and far plane of an
// axis in one v_pk_fma_f32 was measured: 52.7 ms against 51.5 ms for the scalar FMAs on config 2.)
__device__ __forceinline__ float child_entry(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz, F3 A, F3 B,
                                             float mint, float maxt) {
    float tn = fmaxf(fmaxf(__builtin_fmaf(static_cast<float>(nx & 0xffu), B.x, A.x), __builtin_fmaf(static_cast<float>(ny & 0xffu), B.y, A.y)),
                     fmaxf(__builtin_fmaf(static_cast<float>(nz & 0xffu), B.z, A.z), mint));
    float tf = fminf(fminf(__builtin_fmaf(static_cast<float>(fx & 0xffu), B.x, A.x), __builtin_fmaf(static_cast<float>(fy & 0xffu), B.y, A.y)),
                     fminf(__builtin_fmaf(static_cast<float>(fz & 0xffu), B.z, A.z), maxt));
    return tn <= tf ? tn : INFINITY;
}

#define GBL_CSWAP(ta, ra, tb, rb)            \
    do {                                     \
        bool sw_ = (tb) < (ta);              \
        float tt_ = sw_ ? (ta) : (tb);       \
        int rr_ = sw_ ? (ra) : (rb);         \
        (ta) = sw_ ? (tb) : (ta);            \
        (ra) = sw_ ? (rb) : (ra);            \
        (tb) = tt_;                          \
        (rb) = rr_;                          \
    } while (0)

// Interior step: st.cur must be an interior node reference.
// SORTED: visit the children front to back (closest-hit queries).  An any-hit query is done at the first accepted
// triangle wherever it lies, so it skips the 5-comparator sorting network and takes the hit children in slot order.
template <bool STATS, bool SORTED, class STK>
__device__ __forceinline__ void trav_interior(const DevScene& sc, TravState& st, const STK& stk, LaneCounters& cnt) {
    const uint4* np = node_ptr(sc, stk, st.cur);
    const uint4 w0 = np[0];   // o.x o.y o.z scale.x
    const uint4 w1 = np[1];   // scale.y scale.z qlo.x qlo.y
    const uint4 w2 = np[2];   // qlo.z qhi.x qhi.y qhi.z
    const uint4 w3 = np[3];   // child0..3
    const RaySpace& r = st.r;
    F3 A = f3(__builtin_fmaf(__uint_as_float(w0.x), r.idir.x, -r.ood.x), __builtin_fmaf(__uint_as_float(w0.y), r.idir.y, -r.ood.y),
              __builtin_fmaf(__uint_as_float(w0.z), r.idir.z, -r.ood.z));
    F3 B = f3(__uint_as_float(w0.w) * r.idir.x, __uint_as_float(w1.x) * r.idir.y, __uint_as_float(w1.y) * r.idir.z);
    // planes crossed first / last per axis, for all four children at once (4 bytes per word)
    const bool ngx = r.idir.x < 0.0f, ngy = r.idir.y < 0.0f, ngz = r.idir.z < 0.0f;
    const uint32_t nx = ngx ? w2.y : w1.z, fx = ngx ? w1.z : w2.y;
    const uint32_t ny = ngy ? w2.z : w1.w, fy = ngy ? w1.w : w2.z;
    const uint32_t nz = ngz ? w2.w : w2.x, fz = ngz ? w2.x : w2.w;
    float t0 = child_entry(nx, ny, nz, fx, fy, fz, A, B, st.mint, st.maxt);
    float t1 = child_entry(nx >> 8, ny >> 8, nz >> 8, fx >> 8, fy >> 8, fz >> 8, A, B, st.mint, st.maxt);
    float t2 = child_entry(nx >> 16, ny >> 16, nz >> 16, fx >> 16, fy >> 16, fz >> 16, A, B, st.mint, st.maxt);
    float t3 = child_entry(nx >> 24, ny >> 24, nz >> 24, fx >> 24, fy >> 24, fz >> 24, A, B, st.mint, st.maxt);
    int r0 = static_cast<int>(w3.x), r1 = static_cast<int>(w3.y), r2 = static_cast<int>(w3.z), r3 = static_cast<int>(w3.w);
    if (STATS) {
        cnt.nodes += 4;
        probe(cnt.int_lane, cnt.int_wave);
#ifdef GBL_PROBE_OCC
        {   // experiment build: interior wave-steps by the number of lanes inside them (bins <= 4, 8, 16, 32, 64 in hist[0..4])
            const unsigned long long m = __ballot(1);
            const int n = __popcll(m);
            if ((threadIdx.x & 63) == __ffsll(static_cast<long long>(m)) - 1) cnt.hist[n <= 4 ? 0 : (n <= 8 ? 1 : (n <= 16 ? 2 : (n <= 32 ? 3 : 4)))] += 1;
        }
#endif
    }
    int sp = st.sp;
    if (SORTED) {
        // sort the four (entry, ref) pairs by entry distance (5 compare-exchanges).  (Singling out only the nearest
        // child and pushing the rest unsorted was measured: -1.7 % on config 2, +2.4 % on the instanced grid.)
        GBL_CSWAP(t0, r0, t1, r1);
        GBL_CSWAP(t2, r2, t3, r3);
        GBL_CSWAP(t0, r0, t2, r2);
        GBL_CSWAP(t1, r1, t3, r3);
        GBL_CSWAP(t1, r1, t2, r2);
        // nearest child next; push the others farthest first (misses sort to the end: INFINITY)
        if (t3 < INFINITY) stk.store(sp++, static_cast<uint32_t>(r3));
        if (t2 < INFINITY) stk.store(sp++, static_cast<uint32_t>(r2));
        if (t1 < INFINITY) stk.store(sp++, static_cast<uint32_t>(r1));
        if (t0 < INFINITY) {
            st.cur = r0;
        } else {
            st.cur = static_cast<int>(stk.load(--sp));
        }
    } else {
        // any order: push every hit child, then pop one
        if (t3 < INFINITY) stk.store(sp++, static_cast<uint32_t>(r3));
        if (t2 < INFINITY) stk.store(sp++, static_cast<uint32_t>(r2));
        if (t1 < INFINITY) stk.store(sp++, static_cast<uint32_t>(r1));
        if (t0 < INFINITY) stk.store(sp++, static_cast<uint32_t>(r0));
        st.cur = static_cast<int>(stk.load(--sp));
    }
    st.sp = sp;
}

// Everything that is not an interior node: exit marker, instance sentinel, instance entry,
// triangle leaf.  Returns true when the ray is finished (for ANY: as soon as a triangle is
// accepted, with *occluded set).
// Sphere::intersect / occluded up to the accepted distance (GoblinSphere.cpp:12-31; quadratic, GoblinUtils.cpp:93-113)
__device__ __forceinline__ bool sphere_test(float radius, F3 o, F3 d, float mint, float maxt, float* t_out) {
    float A = sqlen(d);
    float B = 2.0f * dot(d, o);
    float C = sqlen(o) - radius * radius;
    float disc = B * B - 4.0f * A * C;
    if (disc < 0.0f) return false;
    float root = sqrtf(disc);
    float q = B < 0 ? -0.5f * (B - root) : -0.5f * (B + root);
    float t1 = q / A, t2 = C / q;
    if (t1 > t2) {
        float tmp = t1;
        t1 = t2;
        t2 = tmp;
    }
    if (t1 > maxt || t2 < mint) return false;
    float t_hit = t1;
    if (t_hit < mint) {
        t_hit = t2;
        if (t_hit > maxt) return false;
    }
    *t_out = t_hit;
    return true;
}
// Disk::intersect / occluded (GoblinDisk.cpp:12-31, 63-74)
__device__ __forceinline__ bool disk_test(float radius, F3 o, F3 d, float mint, float maxt, float* t_out) {
    if (fabsf(d.z) < 1e-7f) return false;
    float t = -o.z / d.z;
    F3 p = o + t * d;
    if (t < mint || t > maxt) return false;
    float square_r = p.x * p.x + p.y * p.y;
    if (square_r > radius * radius) return false;
    *t_out = t;
    return true;
}

// Two triangles of one instance accepted at EXACTLY the same t (a ray through their shared edge or vertex): the
// reference keeps whichever its own BLAS tests later (DevTriOrder) -- if it gets to test it at all: by then maxt is t,
// and the box test in front of the later leaf is strict (tMin < maxt, GoblinBVH.cpp:156-187), so a leaf whose box the
// ray enters exactly AT the hit point (the hit is a corner or an edge of the triangle's bound) is skipped and the
// earlier triangle stays.
// static intersect(bbox, ray, invDir, dirIsNeg), GoblinBVH.cpp:156-187: the reference's node test
__device__ __forceinline__ bool ref_box_reached(F3 lo, F3 hi, F3 o, F3 d, float mint, float maxt) {
    const F3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float tmin = ((d.x < 0.0f ? hi.x : lo.x) - o.x) * inv.x;
    float tmax = ((d.x < 0.0f ? lo.x : hi.x) - o.x) * inv.x;
    const float tymin = ((d.y < 0.0f ? hi.y : lo.y) - o.y) * inv.y;
    const float tymax = ((d.y < 0.0f ? lo.y : hi.y) - o.y) * inv.y;
    if (tymax < tmin || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    const float tzmin = ((d.z < 0.0f ? hi.z : lo.z) - o.z) * inv.z;
    const float tzmax = ((d.z < 0.0f ? lo.z : hi.z) - o.z) * inv.z;
    if (tzmax < tmin || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return tmin < maxt && tmax > mint;
}
__device__ __forceinline__ bool ref_leaf_reached(const DevScene& sc, uint32_t tri, F3 o, F3 d, float mint, float maxt) {
    const float4* bp = reinterpret_cast<const float4*>(sc.tri_bounds + tri);
    const float4 b0 = bp[0], b1 = bp[1];
    return ref_box_reached(f3(b0.x, b0.y, b0.z), f3(b0.w, b1.x, b1.y), o, d, mint, maxt);
}
// true: `cand` replaces the current hit `cur` (both accepted at t).
// A function of its own (plain pointers and floats: no DevScene to materialise for the call): a handful of ties per 10^7 paths,
// and out of line its two dozen registers stay out of the loops that call it.  That is the only reason left -- for a round this
// attribute was also what hid a miscompile (inlined into the quad queries' leaf loop the rule "decided" 114 of 6.8e7 Cornell
// samples the other way at -O3, not with a printf beside it, not out of line).  The cause is a StructurizeCFG bug of this compiler,
// found from the ISA and reduced to sixty lines of IR in round 4 (tools/compiler_bugs/); the code that triggered it was the caller's
// accept block, not this function, and trav_other_kind no longer has such a block.  -DGBL_TIE_INLINE builds the inlined form
// (tests/test_gpu_ties.py runs both against the oracle on a scene where every hit is a tie).
#ifdef GBL_TIE_INLINE
#define GBL_TIE_ATTR __forceinline__
#else
#define GBL_TIE_ATTR __attribute__((noinline))
#endif
static __device__ GBL_TIE_ATTR bool tie_goes_to_impl(const DevTri* tris, const DevTriOrder* tri_order, const DevTriBound* tri_bounds, uint32_t cur,
                                                           uint32_t cand, float ox, float oy, float oz, float dx, float dy, float dz, float mint, float t) {
    const uint32_t sa = tris[cur].shade, sb = tris[cand].shade;
    const DevTriOrder a = tri_order[sa], b = tri_order[sb];
    const uint32_t da = a.depth_rank & 0xffu, db = b.depth_rank & 0xffu;
    const uint32_t common = min(da, db);
    const uint32_t x = (a.path ^ b.path) & (common >= 32u ? 0xffffffffu : ((1u << common) - 1u));
    if (x == 0u) return (b.depth_rank >> 8) > (a.depth_rank >> 8);   // one leaf: its triangles are tested in order
    const uint32_t l = static_cast<uint32_t>(__ffs(static_cast<int>(x))) - 1u;
    const uint32_t axis = l < 16u ? (a.axes_lo >> (2u * l)) & 3u : (a.axes_hi >> (2u * (l - 16u))) & 3u;
    const float dc = axis == 0u ? dx : (axis == 1u ? dy : dz);
    const bool near_is_second = dc < 0.0f;                 // dirIsNeg[axis]: the second child is entered first
    const bool cand_in_second = ((b.path >> l) & 1u) != 0u;
    const bool cand_is_later = cand_in_second != near_is_second;   // the far child is visited later
    // the later one wins if its leaf is reached with maxt == t; its single-triangle leaf's box is the triangle's bound
    const float4* bp = reinterpret_cast<const float4*>(tri_bounds + (cand_is_later ? cand : cur));
    const float4 b0 = bp[0], b1 = bp[1];
    const bool reached = ref_box_reached(f3(b0.x, b0.y, b0.z), f3(b0.w, b1.x, b1.y), f3(ox, oy, oz), f3(dx, dy, dz), mint, t);
    return cand_is_later ? reached : !reached;
}
__device__ __forceinline__ bool tie_goes_to(const DevScene& sc, uint32_t cur, uint32_t cand, F3 o, F3 d, float mint, float t) {
    return tie_goes_to_impl(sc.tris, sc.tri_order, sc.tri_bounds, cur, cand, o.x, o.y, o.z, d.x, d.y, d.z, mint, t);
}

// Would the reference's own traversal get to test this triangle at all?  Its box tests are not watertight (static
// intersect(bbox, ...), GoblinBVH.cpp:156-187: no padding, strict comparisons), while Triangle::intersect accepts with a
// 1e-7 barycentric slack -- so a ray that grazes the silhouette of an axis-aligned object (the mirror block of the Cornell
// scene: one sample in 65 536 at 1024 spp) can be accepted by the triangle test yet never reach it in the reference, whose
// leaf box (the triangle's own bound, one triangle per leaf) or TLAS leaf box (the instance's world bound) the same ray
// misses by a rounding.  Every ancestor's box contains those two, so they decide.  The TIES builds (replay, stream,
// exact_ties, instrumented) therefore accept a triangle only if both pass the reference's test, with the query's own maxt.
__device__ __forceinline__ bool ref_reached(const DevScene& sc, int inst, uint32_t tri, F3 wo, F3 wd, F3 oo, F3 od, float mint, float maxt0) {
    const DevInstanceBound wb = sc.instance_bounds[inst];
    if (!ref_box_reached(f3(wb.lo[0], wb.lo[1], wb.lo[2]), f3(wb.hi[0], wb.hi[1], wb.hi[2]), wo, wd, mint, maxt0)) return false;
    const float4* bp = reinterpret_cast<const float4*>(sc.tri_bounds + tri);
    const float4 b0 = bp[0], b1 = bp[1];
    const F3 lo = f3(b0.x, b0.y, b0.z), hi = f3(b0.w, b1.x, b1.y);
    return ref_box_reached(lo, hi, oo, od, mint, maxt0);
}

// EXT: the scene may hold analytic shapes (DevScene::extended); plain scenes compile the branch out.
// `filter` (EXT builds): GBL_FILTER_* -- instances whose material is / is not a mask are skipped whole, which is
// what Model::intersect does with the isOpaque / notOpaque IntersectFilter (GoblinModel.cpp:30-32, 44-46).
// TM: what the loop does about the reference's exact-t tie rule (tie_goes_to) and its reachability test (ref_reached).
//   GBL_TIE_NONE    nothing: the last triangle tested at a distance keeps it, every accepted triangle counts
//   GBL_TIE_EXACT   both, inline, at every triangle the test accepts (needs st.world and st.maxt0)
//   GBL_TIE_DETECT  the loop of GBL_TIE_NONE plus one compare: st.tied is set when a triangle is accepted at exactly the distance
//                   of the hit the ray holds; an any-hit query also leaves its occluder in st.hit.  The callers (trace(),
//                   trace_quad(), wf_trace) then check the FINAL hit with ref_reached once per query, all lanes together,
//                   and run the rare ray that tied or whose hit the reference would not have reached again under
//                   GBL_TIE_EXACT (trace_needs_redo below says why that is the same answer).  Inline, the two rules cost the
//                   headline kernel 32 %: a triangle is accepted 1.5 - 3 times per query, by a few lanes at a time.
// `any`: the query kind as a value -- a compile-time constant through trav_other<ANY, ...> below.
// FUSE: a leaf whose pop uncovers the instance's sentinel (and then, possibly, the exit marker) takes those steps at once instead
// of spending an iteration of the caller's loop on each (needs st.world: the one-ray-per-lane loops only).
#define GBL_TIE_NONE 0
#define GBL_TIE_EXACT 1
#define GBL_TIE_DETECT 2
template <bool STATS, bool EXT, class STK, int TM, bool FUSE = false>
__device__ __forceinline__ bool trav_other_kind(const DevScene& sc, TravState& st, const STK& stk, LaneCounters& cnt, const bool ANY,
                                                bool* occluded, int filter) {
    constexpr bool TIES = TM == GBL_TIE_EXACT;
    const int cur = st.cur;
    if (STATS) probe(cnt.oth_lane, cnt.oth_wave);
    if (cur == GBL_STACK_EXIT) return true;
    if (cur == GBL_STACK_SENTINEL) {   // finished an instance: back to the world ray
        st.r = st.world;
        st.inst = -1;
        st.cur = static_cast<int>(stk.load(--st.sp));
        return false;
    }
    const uint32_t ref = ~static_cast<uint32_t>(cur);
    if (st.inst < 0) {
        const DevInstance* ip = sc.instances + (ref >> 2);
        if (EXT && filter != GBL_FILTER_NONE && (ip->is_mask != 0u ? GBL_FILTER_MASK : GBL_FILTER_OPAQUE) != filter) {
            st.cur = static_cast<int>(stk.load(--st.sp));
            return false;
        }
        // (A copy of every mesh instance's BLAS root node at an address that follows from the TLAS leaf alone, one word of it
        //  fetched here next to the instance record so that the interior step behind this one finds the line on its way: no
        //  gain -- configs[1] 42.4 against 42.1 ms, grid 20.8 / 20.8, AO 27.5 against 27.0.)
        st.inst = static_cast<int>(ref >> 2);
        ray_space(st.r, xf_point(ip->inv, st.world.o), xf_vector(ip->inv, st.world.d));
        stk.store(st.sp++, GBL_STACK_SENTINEL);
        st.cur = ip->root;
        return false;
    }
    const uint32_t first = ref >> 2, count = (ref & 3u) + 1u;
    if (EXT && first >= GBL_SHAPE_FIRST_DISK) {   // Model::intersect of an intersectable geometry (GoblinModel.cpp:46-54)
        const float radius = sc.instances[st.inst].radius;
        float t;
        if (STATS) cnt.tris += 1;
        bool got = first == GBL_SHAPE_FIRST_SPHERE ? sphere_test(radius, st.r.o, st.r.d, st.mint, st.maxt, &t)
                                                   : disk_test(radius, st.r.o, st.r.d, st.mint, st.maxt, &t);
        if (TIES && got) {   // the TLAS leaf box in front of the shape (ref_reached)
            const DevInstanceBound wb = sc.instance_bounds[st.inst];
            got = ref_box_reached(f3(wb.lo[0], wb.lo[1], wb.lo[2]), f3(wb.hi[0], wb.hi[1], wb.hi[2]), st.world.o, st.world.d, st.mint, st.maxt0);
        }
        if (got) {
            if (ANY) {
                if (TM == GBL_TIE_DETECT) {
                    st.hit.inst = st.inst;
                    st.hit.tri = 0;
                }
                *occluded = true;
                return true;
            }
            st.maxt = t;
            st.hit.t = t;
            st.hit.inst = st.inst;
            st.hit.tri = 0;
            st.hit.b1 = st.hit.b2 = 0.0f;
        }
        st.cur = static_cast<int>(stk.load(--st.sp));
        return false;
    }
    for (uint32_t i = 0; i < count; ++i) {
        float t, b1, b2;
        if (STATS) cnt.tris += 1;
        if (tri_test(sc.tris + first + i, st.r.o, st.r.d, st.mint, st.maxt, &t, &b1, &b2)) {
            if (TIES && !ref_reached(sc, st.inst, first + i, st.world.o, st.world.d, st.r.o, st.r.d, st.mint, st.maxt0)) continue;
            if (ANY) {
                if (TM == GBL_TIE_DETECT) {
                    st.hit.inst = st.inst;
                    st.hit.tri = first + i;
                }
                *occluded = true;
                return true;
            }
            if (TM == GBL_TIE_DETECT && t == st.hit.t && st.hit.inst == st.inst) st.tied = true;
            if constexpr (TIES) {
                // The accepted hit is written with SELECTS on a value the optimiser cannot see through, never from a block of its
                // own.  Written the plain way -- `if (tie && !tie_goes_to(..)) continue; st.hit.tri = first + i; ...` -- the accept
                // block is entered both straight from the tie check and from the end of the tie rule, and this compiler
                // (AMD clang 22.0.0git roc-7.2.0) miscompiles exactly that shape once the rule is inlined: StructurizeCFG hoists
                // the block's zero-cost `insertelement (inst, first + i)` above the tie check and then replaces the join's phi by
                // the Flow phi [rejected pair, hoisted pair], so a lane that WINS a tie keeps the old triangle id with the new
                // barycentrics (114 of 6.8e7 Cornell samples, round 3).  Reduced case and the ISA evidence:
                // tools/compiler_bugs/structurizecfg_hoisted_phi.ll, DESIGN.md 6.  (A bool would do as well until JumpThreading
                // unfolds a select on a phi of constants back into that block; the empty asm keeps it from knowing `take`.)
                uint32_t take = 1u;
#ifndef GBL_NO_TIE_RULE
                if (t == st.hit.t && st.hit.inst == st.inst && sc.tri_order != nullptr)
                    take = tie_goes_to(sc, st.hit.tri, first + i, st.r.o, st.r.d, st.mint, t) ? 1u : 0u;
#endif
                asm volatile("" : "+v"(take));
                const bool acc = take != 0u;
                st.maxt = acc ? t : st.maxt;
                st.hit.t = acc ? t : st.hit.t;
                st.hit.inst = acc ? st.inst : st.hit.inst;
                st.hit.tri = acc ? first + i : st.hit.tri;
                st.hit.b1 = acc ? b1 : st.hit.b1;
                st.hit.b2 = acc ? b2 : st.hit.b2;
            } else {
                st.maxt = t;
                st.hit.t = t;
                st.hit.inst = st.inst;
                st.hit.tri = first + i;
                st.hit.b1 = b1;
                st.hit.b2 = b2;
            }
        }
    }
    st.cur = static_cast<int>(stk.load(--st.sp));
    if (FUSE) {
        if (st.cur == GBL_STACK_SENTINEL) {
            st.r = st.world;
            st.inst = -1;
            st.cur = static_cast<int>(stk.load(--st.sp));
        }
        if (st.cur == GBL_STACK_EXIT) return true;
    }
    return false;
}
template <bool ANY, bool STATS, bool EXT, class STK, int TM, bool FUSE = false>
__device__ __forceinline__ bool trav_other(const DevScene& sc, TravState& st, const STK& stk, LaneCounters& cnt, bool* occluded,
                                           int filter = GBL_FILTER_NONE) {
    return trav_other_kind<STATS, EXT, STK, TM, FUSE>(sc, st, stk, cnt, ANY, occluded, filter);
}

__device__ __forceinline__ bool trav_at_interior(const TravState& st) {
    return static_cast<uint32_t>(st.cur) < static_cast<uint32_t>(GBL_REF_NONE);
}

// A wave runs ONE kind of step per iteration: interior steps while at least GBL_TRAV_TH of its
// lanes sit at interior nodes (or nobody waits at a leaf), otherwise the leaf / instance phase.
// Running whatever each lane needs in the same iteration would execute every branch each time
// (~20 % lane utilisation on wave64); phasing by majority state keeps the lanes together.
#ifndef GBL_TRAV_TH
#define GBL_TRAV_TH 24
#endif

// After a GBL_TIE_DETECT query: does this ray have to be traced again under GBL_TIE_EXACT?  `got`: it hit (or, any-hit, was
// occluded by) triangle h.tri of instance h.inst.
// Why checking the end of the query is enough.  Let C be the triangles whose test the ray passes within [mint, maxt]; the loop of
// GBL_TIE_NONE returns the nearest of C (it accepts whatever is nearer than what it holds, and never culls a node in front of
// that).  The reference returns the nearest REACHED member of C.  If the nearest of C is reached, and no second member sits at
// exactly its distance (the loop would have tested it against the hit it held: st.tied), the two are the same triangle.  An
// any-hit query is occluded exactly when some member of C is reached: if the one the loop stopped at is, it is.  Everything else
// -- a tie, a hit the reference's box tests would have passed by -- goes through the exact loop, a handful of rays per 10^6.
// One case this (and ref_reached) does NOT restate: the reference tests a box against the ray's SHRINKING maxt (GoblinBVH.cpp:156-187
// reads ray.maxt, which Triangle::intersect lowers), these against the query's own.  A triangle C that is nearer than the hit H the
// reference holds when it gets to C's leaf, but whose flat leaf box has a tMin that rounds up to >= t(H) -- C and H within an ulp or
// two of each other along the ray, not exactly tied -- is skipped by the reference and accepted here.  st.tied only catches exact
// equality.  No fixture, none of the 2 300 fuzz scenes and none of the full-size blocks has produced one; it is stated as a limit
//  rather than paid for with a "within k ulp" retrace whose k nobody could justify.
__device__ __forceinline__ bool trace_needs_redo(const DevScene& sc, bool EXT, bool got, const Hit& h, bool tied, F3 o, F3 d, float mint, float maxt) {
    if (tied) return true;
    if (!got) return false;
    const DevInstanceBound wb = sc.instance_bounds[h.inst];
    if (!ref_box_reached(f3(wb.lo[0], wb.lo[1], wb.lo[2]), f3(wb.hi[0], wb.hi[1], wb.hi[2]), o, d, mint, maxt)) return true;
    const DevInstance* ip = sc.instances + h.inst;
    if (EXT && ip->shape != 0u) return false;   // an analytic shape: the TLAS leaf box is all that stands in front of it
    const float4* bp = reinterpret_cast<const float4*>(sc.tri_bounds + h.tri);
    const float4 b0 = bp[0], b1 = bp[1];
    return !ref_box_reached(f3(b0.x, b0.y, b0.z), f3(b0.w, b1.x, b1.y), xf_point(ip->inv, o), xf_vector(ip->inv, d), mint, maxt);
}

// ANY = true : Scene::occluded (first accepted triangle ends the query)
// ANY = false: Scene::intersect (closest hit; hit.t shrinks like ray.maxt)
template <bool ANY, bool STATS, bool EXT, int TM, class STK>
__device__ __forceinline__ bool trace_loop(const DevScene& sc, F3 o, F3 d, float mint, float maxt, const STK& stk, Hit& hit,
                                           LaneCounters& cnt, int filter, bool* tied) {
    TravState st;
    trav_begin(sc, st, o, d, mint, maxt, stk);
    bool occluded = false;
    // (Every query starts at the TLAS root: taking that node from the kernel arguments -- through the scalar cache -- for a first
    //  step ahead of the loop saved nothing, configs[1] 42.4 against 42.1 ms, and cost the EXT kernels 30 %.)
    // The megakernel's waves are mostly coherent (lanes are samples of one pixel), so every lane
    // simply takes the step it needs; phasing by majority state (GBL_TRAV_TH) only pays in the
    // wavefront trace kernel, whose waves mix rays of many pixels and depths (measured: 64.6 ms vs
    // 76.3 ms per 68 M-path frame here, 82.4 ms vs 79.6 ms there).
    // Lean builds, one iteration: the leaf / instance step of the lanes that stand at one, then the interior step of every lane
    // that stands at an interior node by then -- a lane that leaves a leaf for a node (or enters an instance) takes both in one
    // iteration (measured against one step of either kind per iteration, the quad kernels' dense phase: bunny -1 %, Cornell
    // -4 %, grid -3 %).  The EXT builds keep one step per iteration: the longer body cost them 100+ spilled registers.
    uint32_t steps = 0;
    for (;;) {
        if constexpr (EXT) {
            if (trav_at_interior(st)) {
                trav_interior<STATS, !ANY>(sc, st, stk, cnt);
                if (STATS) ++steps;
            } else if (trav_other<ANY, STATS, EXT, STK, TM>(sc, st, stk, cnt, &occluded, filter)) {
                break;
            }
        } else {
            if (!trav_at_interior(st) && trav_other<ANY, STATS, EXT, STK, TM, true>(sc, st, stk, cnt, &occluded, filter)) break;
            if (trav_at_interior(st)) {
                trav_interior<STATS, !ANY>(sc, st, stk, cnt);
                if (STATS) ++steps;
            }
        }
    }
#ifndef GBL_PROBE_OCC
    if (STATS && !ANY) {
        int b = steps <= 3 ? 0 : min(6, 30 - __clz(static_cast<int>(steps)));
        cnt.hist[b] += 1;
        cnt.hist_steps[b] += steps;
    }
#endif
    if (TM == GBL_TIE_DETECT) *tied = st.tied;
    if (ANY) {
        if (TM == GBL_TIE_DETECT) hit = st.hit;   // (the occluder)
        return occluded;
    }
    hit = st.hit;
    return st.hit.inst >= 0;
}
// TIES: follow the reference's exact-t tie rule and reachability test (GBL_TIE_DETECT loop, end-of-query check, exact loop for
// the rare ray that needs it); false: the bare loop.
template <bool ANY, bool STATS, bool EXT, bool TIES = true, class STK>
__device__ __forceinline__ bool trace(const DevScene& sc, F3 o, F3 d, float mint, float maxt, const STK& stk, Hit& hit,
                                      LaneCounters& cnt, int filter = GBL_FILTER_NONE) {
    if constexpr (!TIES) {
        return trace_loop<ANY, STATS, EXT, GBL_TIE_NONE>(sc, o, d, mint, maxt, stk, hit, cnt, filter, nullptr);
    } else {
        bool tied = false;
        Hit h;
        bool got = trace_loop<ANY, STATS, EXT, GBL_TIE_DETECT>(sc, o, d, mint, maxt, stk, h, cnt, filter, &tied);
        if (trace_needs_redo(sc, EXT, got, h, tied, o, d, mint, maxt)) {
            LaneCounters again = {};   // (instrumented builds: the first pass counted this ray's visits already)
            got = trace_loop<ANY, STATS, EXT, GBL_TIE_EXACT>(sc, o, d, mint, maxt, stk, h, again, filter, nullptr);
        }
        if (!ANY) hit = h;
        return got;
    }
}
