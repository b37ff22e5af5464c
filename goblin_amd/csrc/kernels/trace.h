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
};

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

// Triangle::intersect's acceptance test (GoblinTriangle.cpp:52-78).
__device__ __forceinline__ bool tri_test(const DevTri* tp, F3 o, F3 d, float mint, float maxt, float* t_out, float* b1_out,
                                         float* b2_out) {
    const float4 q0 = reinterpret_cast<const float4*>(tp)[0];
    const float4 q1 = reinterpret_cast<const float4*>(tp)[1];
    const float4 q2 = reinterpret_cast<const float4*>(tp)[2];
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

// Everything a lane carries for the ray it is traversing.
struct TravState {
    RaySpace world, r;   // world-space ray and the ray in the current space (world or instance)
    float mint, maxt;
    int sp, cur, inst;
    Hit hit;
};

__device__ __forceinline__ void trav_begin(const DevScene& sc, TravState& st, F3 o, F3 d, float mint, float maxt, uint32_t* stk) {
    ray_space(st.world, o, d);
    st.r = st.world;
    st.mint = mint;
    st.maxt = maxt;
    st.sp = 0;
    stk[(st.sp++) * GBL_BLOCK] = GBL_STACK_EXIT;
    st.cur = sc.num_instances > 0 ? sc.tlas_root : GBL_STACK_EXIT;
    st.inst = -1;
    st.hit.t = INFINITY;
    st.hit.inst = -1;
    st.hit.tri = 0;
    st.hit.b1 = st.hit.b2 = 0.0f;
}

// entry distance of child c on the node's quantisation grid, INFINITY if missed
__device__ __forceinline__ float child_entry(uint32_t lx, uint32_t ly, uint32_t lz, uint32_t hx, uint32_t hy, uint32_t hz, F3 A, F3 B,
                                             float mint, float maxt) {
    // v_cvt_f32_ubyte0: the low byte of each word is this child's grid coordinate
    float x0 = __builtin_fmaf(static_cast<float>(lx & 0xffu), B.x, A.x), x1 = __builtin_fmaf(static_cast<float>(hx & 0xffu), B.x, A.x);
    float y0 = __builtin_fmaf(static_cast<float>(ly & 0xffu), B.y, A.y), y1 = __builtin_fmaf(static_cast<float>(hy & 0xffu), B.y, A.y);
    float z0 = __builtin_fmaf(static_cast<float>(lz & 0xffu), B.z, A.z), z1 = __builtin_fmaf(static_cast<float>(hz & 0xffu), B.z, A.z);
    float lo = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), mint));
    float hi = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), maxt));
    return lo <= hi ? lo : INFINITY;
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
template <bool STATS>
__device__ __forceinline__ void trav_interior(const DevScene& sc, TravState& st, uint32_t* stk, LaneCounters& cnt) {
    const uint4* np = reinterpret_cast<const uint4*>(sc.nodes + st.cur);
    const uint4 w0 = np[0];   // o.x o.y o.z exps
    const uint4 w1 = np[1];   // qlo.x qlo.y qlo.z qhi.x
    const uint4 w2 = np[2];   // qhi.y qhi.z child0 child1
    const uint4 w3 = np[3];   // child2 child3 - -
    const RaySpace& r = st.r;
    F3 A = f3(__builtin_fmaf(__uint_as_float(w0.x), r.idir.x, -r.ood.x), __builtin_fmaf(__uint_as_float(w0.y), r.idir.y, -r.ood.y),
              __builtin_fmaf(__uint_as_float(w0.z), r.idir.z, -r.ood.z));
    F3 B = f3(__uint_as_float((w0.w & 0xffu) << 23) * r.idir.x, __uint_as_float(((w0.w >> 8) & 0xffu) << 23) * r.idir.y,
              __uint_as_float(((w0.w >> 16) & 0xffu) << 23) * r.idir.z);
    float t0 = child_entry(w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, A, B, st.mint, st.maxt);
    float t1 = child_entry(w1.x >> 8, w1.y >> 8, w1.z >> 8, w1.w >> 8, w2.x >> 8, w2.y >> 8, A, B, st.mint, st.maxt);
    float t2 = child_entry(w1.x >> 16, w1.y >> 16, w1.z >> 16, w1.w >> 16, w2.x >> 16, w2.y >> 16, A, B, st.mint, st.maxt);
    float t3 = child_entry(w1.x >> 24, w1.y >> 24, w1.z >> 24, w1.w >> 24, w2.x >> 24, w2.y >> 24, A, B, st.mint, st.maxt);
    int r0 = static_cast<int>(w2.z), r1 = static_cast<int>(w2.w), r2 = static_cast<int>(w3.x), r3 = static_cast<int>(w3.y);
    // unused child slots (a min/max slab test cannot see an inverted box)
    t0 = r0 == static_cast<int>(GBL_REF_NONE) ? INFINITY : t0;
    t1 = r1 == static_cast<int>(GBL_REF_NONE) ? INFINITY : t1;
    t2 = r2 == static_cast<int>(GBL_REF_NONE) ? INFINITY : t2;
    t3 = r3 == static_cast<int>(GBL_REF_NONE) ? INFINITY : t3;
    if (STATS) cnt.nodes += 4;
    // sort the four (entry, ref) pairs by entry distance (5 compare-exchanges)
    GBL_CSWAP(t0, r0, t1, r1);
    GBL_CSWAP(t2, r2, t3, r3);
    GBL_CSWAP(t0, r0, t2, r2);
    GBL_CSWAP(t1, r1, t3, r3);
    GBL_CSWAP(t1, r1, t2, r2);
    // nearest child next; push the others farthest first
    int sp = st.sp;
    if (t3 < INFINITY) stk[(sp++) * GBL_BLOCK] = static_cast<uint32_t>(r3);
    if (t2 < INFINITY) stk[(sp++) * GBL_BLOCK] = static_cast<uint32_t>(r2);
    if (t1 < INFINITY) stk[(sp++) * GBL_BLOCK] = static_cast<uint32_t>(r1);
    if (t0 < INFINITY) {
        st.cur = r0;
    } else {
        st.cur = static_cast<int>(stk[(--sp) * GBL_BLOCK]);
    }
    st.sp = sp;
}

// Everything that is not an interior node: exit marker, instance sentinel, instance entry,
// triangle leaf.  Returns true when the ray is finished (for ANY: as soon as a triangle is
// accepted, with *occluded set).
template <bool ANY, bool STATS>
__device__ __forceinline__ bool trav_other(const DevScene& sc, TravState& st, uint32_t* stk, LaneCounters& cnt, bool* occluded) {
    const int cur = st.cur;
    if (cur == GBL_STACK_EXIT) return true;
    if (cur == GBL_STACK_SENTINEL) {   // finished an instance: back to the world ray
        st.r = st.world;
        st.inst = -1;
        st.cur = static_cast<int>(stk[(--st.sp) * GBL_BLOCK]);
        return false;
    }
    const uint32_t ref = ~static_cast<uint32_t>(cur);
    if (st.inst < 0) {
        st.inst = static_cast<int>(ref >> 2);
        const DevInstance* ip = sc.instances + st.inst;
        ray_space(st.r, xf_point(ip->inv, st.world.o), xf_vector(ip->inv, st.world.d));
        stk[(st.sp++) * GBL_BLOCK] = GBL_STACK_SENTINEL;
        st.cur = ip->root;
        return false;
    }
    const uint32_t first = ref >> 2, count = (ref & 3u) + 1u;
    for (uint32_t i = 0; i < count; ++i) {
        float t, b1, b2;
        if (STATS) cnt.tris += 1;
        if (tri_test(sc.tris + first + i, st.r.o, st.r.d, st.mint, st.maxt, &t, &b1, &b2)) {
            if (ANY) {
                *occluded = true;
                return true;
            }
            st.maxt = t;
            st.hit.t = t;
            st.hit.inst = st.inst;
            st.hit.tri = first + i;
            st.hit.b1 = b1;
            st.hit.b2 = b2;
        }
    }
    st.cur = static_cast<int>(stk[(--st.sp) * GBL_BLOCK]);
    return false;
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

// ANY = true : Scene::occluded (first accepted triangle ends the query)
// ANY = false: Scene::intersect (closest hit; hit.t shrinks like ray.maxt)
template <bool ANY, bool STATS>
__device__ __forceinline__ bool trace(const DevScene& sc, F3 o, F3 d, float mint, float maxt, uint32_t* stk, Hit& hit,
                                      LaneCounters& cnt) {
    TravState st;
    trav_begin(sc, st, o, d, mint, maxt, stk);
    bool occluded = false;
    // The megakernel's waves are mostly coherent (lanes are samples of one pixel), so every lane
    // simply takes the step it needs; phasing by majority state (GBL_TRAV_TH) only pays in the
    // wavefront trace kernel, whose waves mix rays of many pixels and depths (measured: 64.6 ms vs
    // 76.3 ms per 68 M-path frame here, 82.4 ms vs 79.6 ms there).
    for (;;) {
        if (trav_at_interior(st)) {
            trav_interior<STATS>(sc, st, stk, cnt);
        } else if (trav_other<ANY, STATS>(sc, st, stk, cnt, &occluded)) {
            break;
        }
    }
    if (ANY) return occluded;
    hit = st.hit;
    return st.hit.inst >= 0;
}
