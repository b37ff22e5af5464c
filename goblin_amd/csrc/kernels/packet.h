// Packet traversal: the 64 rays of a wave walk the two-level tree TOGETHER -- one node reference, one stack, wave-uniform control
// flow -- each lane testing the shared node's four children (or the shared leaf's triangles) against its OWN ray and keeping its
// own closest hit.  For rays that are coherent by construction: the camera rays of one pixel (64 samples of a pixel start from one
// point through one pixel's footprint), which is what the primary pass traces (kernels_quad.hip primary_kernel).
//
// Why it pays: the path tracer is bound by VALU issue with a third of the lanes on.  One ray per lane, a wave's
// rays drift apart -- each interior step serves the lanes that stand at an interior node, about half of them, the tail of a query a
// handful.  Here every step serves every lane that is in the subtree, there is no per-lane stack traffic and no per-lane child sort
// (the order is decided once per node, on the scalar unit, from the first lane's entry distances).
//
// A lane's result is what its ray would have found alone (Scene::intersect, GoblinBVH.cpp:234-280, as trace.h's lean loop restates
// it).  Every stack entry carries the mask of the lanes whose OWN ray entered that child's box, and a lane tests a leaf's triangles
// only if it is in the leaf's mask: testing a ray against triangles of boxes it misses is not harmless, because tri_test widens every
// edge by 1e-7 in barycentric units -- a ray just outside a wall's last triangle would hit it here and miss it alone (seen: a handful
// of samples in 10^9 on the Cornell box).  What remains order dependent is which of two triangles at EXACTLY the same distance is
// kept (tri_test accepts t <= maxt, the later one wins; the packet's order is the lead lane's): such a lane reports `tied` and the
// caller has the ray traced on its own.
#pragma once
#include "trace.h"

#define GBL_PACKET_STACK 64   // entries of the wave's shared stack (the per-lane stacks' need is DevScene::stack_entries <= this)
#define GBL_PACKET_STACK_WORDS (3 * GBL_PACKET_STACK)   // {reference, lane mask lo, lane mask hi}

// Wave-uniform addresses go through the scalar cache (s_load): the node, the instance record and the triangles of a packet step are
// the same for all 64 lanes; their words then sit in SGPRs and feed the VALU as scalar operands -- no vector-memory instruction, no
// 64 copies of the same 64 bytes in VGPRs, a shorter round trip.  (Read-only data: the scalar cache is not coherent with stores.)
typedef uint32_t gbl_u32x4 __attribute__((ext_vector_type(4)));
struct gbl_k_u4 {   // sixteen-byte words behind a wave-uniform address, read with s_load
    const __attribute__((address_space(4))) gbl_u32x4* p;
    __device__ __forceinline__ uint4 operator[](int i) const {
        const gbl_u32x4 v = p[i];
        return make_uint4(v.x, v.y, v.z, v.w);
    }
};
__device__ __forceinline__ gbl_k_u4 pk_k(const void* p) { return gbl_k_u4{(const __attribute__((address_space(4))) gbl_u32x4*)p}; }
__device__ __forceinline__ float4 pk_f4(uint4 w) { return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)); }
// the instance record's inverse transform (floats 12..23) and BLAS root (word 24)
__device__ __forceinline__ int pk_enter_instance(const DevScene& sc, uint32_t index, const RaySpace& world, RaySpace& r) {
    const gbl_k_u4 ip = pk_k(sc.instances + index);
    const uint4 a = ip[3], b = ip[4], c = ip[5], e = ip[6];
    const float inv[12] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                           __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w),
                           __uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w)};
    ray_space(r, xf_point(inv, world.o), xf_vector(inv, world.d));
    return static_cast<int>(e.x);
}
__device__ __forceinline__ bool pk_tri_test(const DevScene& sc, uint32_t tri, F3 o, F3 d, float mint, float maxt, float* t, float* b1, float* b2) {
    const gbl_k_u4 tp = pk_k(sc.tris + tri);
    return tri_test_regs(pk_f4(tp[0]), pk_f4(tp[1]), pk_f4(tp[2]), o, d, mint, maxt, t, b1, b2);
}

__device__ __forceinline__ unsigned long long pk_uniform64(unsigned long long v) {
    return static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)))) |
           (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32)))) << 32);
}
__device__ __forceinline__ void pk_push(gbl_lds_u32* wstack, int& sp, int ref, unsigned long long mask) {
    wstack[3 * sp] = static_cast<uint32_t>(ref);
    wstack[3 * sp + 1] = static_cast<uint32_t>(mask);
    wstack[3 * sp + 2] = static_cast<uint32_t>(mask >> 32);
    ++sp;
}
__device__ __forceinline__ void pk_pop(gbl_lds_u32* wstack, int& sp, int& ref, unsigned long long& mask) {
    --sp;
    ref = __builtin_amdgcn_readfirstlane(static_cast<int>(wstack[3 * sp]));
    mask = pk_uniform64(static_cast<unsigned long long>(wstack[3 * sp + 1]) | (static_cast<unsigned long long>(wstack[3 * sp + 2]) << 32));
}

// `live`: the lane has a ray.  `wstack`: GBL_PACKET_STACK_WORDS words of LDS private to this wave.  Returns the lane's closest hit;
// `tied`: a triangle was accepted at exactly the distance of the hit the lane held (see above).
__device__ __forceinline__ bool packet_closest(const DevScene& sc, bool live, F3 o, F3 d, float mint, gbl_lds_u32* wstack, Hit& hit, bool& tied) {
    RaySpace world, r;
    ray_space(world, o, d);
    r = world;
    float maxt = INFINITY;
    hit.t = INFINITY;
    hit.inst = -1;
    hit.tri = 0;
    hit.b1 = hit.b2 = 0.0f;
    tied = false;
    int sp = 0;
    int inst = -1;
    pk_push(wstack, sp, GBL_STACK_EXIT, 0ull);
    int cur = sc.num_instances > 0 ? sc.tlas_root : GBL_STACK_EXIT;
    unsigned long long mask = __ballot(live);   // the lanes whose own ray is in the subtree of `cur`
    for (;;) {
        const bool here = __builtin_amdgcn_inverse_ballot_w64(mask);
        if (static_cast<uint32_t>(cur) < static_cast<uint32_t>(GBL_REF_NONE)) {
            // ---- interior node: every lane of the mask tests the four children against its own ray
            const gbl_k_u4 np = pk_k(sc.nodes + cur);
            const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
            const F3 A = f3(__builtin_fmaf(__uint_as_float(w0.x), r.idir.x, -r.ood.x), __builtin_fmaf(__uint_as_float(w0.y), r.idir.y, -r.ood.y),
                            __builtin_fmaf(__uint_as_float(w0.z), r.idir.z, -r.ood.z));
            const F3 B = f3(__uint_as_float(w0.w) * r.idir.x, __uint_as_float(w1.x) * r.idir.y, __uint_as_float(w1.y) * r.idir.z);
            const bool ngx = r.idir.x < 0.0f, ngy = r.idir.y < 0.0f, ngz = r.idir.z < 0.0f;
            const uint32_t nx = ngx ? w2.y : w1.z, fx = ngx ? w1.z : w2.y;
            const uint32_t ny = ngy ? w2.z : w1.w, fy = ngy ? w1.w : w2.z;
            const uint32_t nz = ngz ? w2.w : w2.x, fz = ngz ? w2.x : w2.w;
            const float t0 = child_entry(nx, ny, nz, fx, fy, fz, A, B, mint, maxt);
            const float t1 = child_entry(nx >> 8, ny >> 8, nz >> 8, fx >> 8, fy >> 8, fz >> 8, A, B, mint, maxt);
            const float t2 = child_entry(nx >> 16, ny >> 16, nz >> 16, fx >> 16, fy >> 16, fz >> 16, A, B, mint, maxt);
            const float t3 = child_entry(nx >> 24, ny >> 24, nz >> 24, fx >> 24, fy >> 24, fz >> 24, A, B, mint, maxt);
            // who enters which child; the order: by the entry distances of the first lane that enters anything
            // (scalar from here on; positive floats order like their bit patterns, INFINITY last)
            unsigned long long m0 = __ballot(here && t0 < INFINITY), m1 = __ballot(here && t1 < INFINITY);
            unsigned long long m2 = __ballot(here && t2 < INFINITY), m3 = __ballot(here && t3 < INFINITY);
            const unsigned long long any = m0 | m1 | m2 | m3;
            if (any == 0ull) {
                pk_pop(wstack, sp, cur, mask);
                continue;
            }
            const int lead = __ffsll(static_cast<long long>(any)) - 1;
            uint32_t k0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(t0)), lead));
            uint32_t k1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(t1)), lead));
            uint32_t k2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(t2)), lead));
            uint32_t k3 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(t3)), lead));
            const uint32_t inf = 0x7f800000u;
            // a child the lead lane misses but another lane enters goes behind the lead's (key just under INFINITY); nobody's: INFINITY
            if (k0 >= inf) k0 = m0 != 0ull ? inf - 1u : inf;
            if (k1 >= inf) k1 = m1 != 0ull ? inf - 1u : inf;
            if (k2 >= inf) k2 = m2 != 0ull ? inf - 1u : inf;
            if (k3 >= inf) k3 = m3 != 0ull ? inf - 1u : inf;
            int r0 = __builtin_amdgcn_readfirstlane(static_cast<int>(w3.x)), r1 = __builtin_amdgcn_readfirstlane(static_cast<int>(w3.y));
            int r2 = __builtin_amdgcn_readfirstlane(static_cast<int>(w3.z)), r3 = __builtin_amdgcn_readfirstlane(static_cast<int>(w3.w));
#define GBL_PK_CSWAP(ka, ra, ma, kb, rb, mb)        \
    do {                                            \
        if ((kb) < (ka)) {                          \
            const uint32_t tk_ = (ka);              \
            const int tr_ = (ra);                   \
            const unsigned long long tm_ = (ma);    \
            (ka) = (kb); (ra) = (rb); (ma) = (mb);  \
            (kb) = tk_; (rb) = tr_; (mb) = tm_;     \
        }                                           \
    } while (0)
            GBL_PK_CSWAP(k0, r0, m0, k1, r1, m1);
            GBL_PK_CSWAP(k2, r2, m2, k3, r3, m3);
            GBL_PK_CSWAP(k0, r0, m0, k2, r2, m2);
            GBL_PK_CSWAP(k1, r1, m1, k3, r3, m3);
            GBL_PK_CSWAP(k1, r1, m1, k2, r2, m2);
#undef GBL_PK_CSWAP
            if (k3 < inf) pk_push(wstack, sp, r3, m3);
            if (k2 < inf) pk_push(wstack, sp, r2, m2);
            if (k1 < inf) pk_push(wstack, sp, r1, m1);
            cur = r0;   // (k0 < inf: somebody enters something)
            mask = m0;
            continue;
        }
        if (cur == GBL_STACK_EXIT) break;
        if (cur == GBL_STACK_SENTINEL) {   // the instance is done: back to the world rays
            r = world;
            inst = -1;
            pk_pop(wstack, sp, cur, mask);
            continue;
        }
        const uint32_t ref = ~static_cast<uint32_t>(cur);
        if (inst < 0) {   // a TLAS leaf: every lane's ray into the instance's space (Transform::invertRay, un-normalised: t is shared)
            inst = static_cast<int>(ref >> 2);
            pk_push(wstack, sp, GBL_STACK_SENTINEL, 0ull);
            cur = pk_enter_instance(sc, ref >> 2, world, r);   // (same lanes: `mask` stays)
            continue;
        }
        // ---- a triangle leaf: the lanes of the mask test every triangle of it
        const uint32_t first = ref >> 2, count = (ref & 3u) + 1u;
        for (uint32_t i = 0; i < count; ++i) {
            float t, b1, b2;
            if (pk_tri_test(sc, first + i, r.o, r.d, mint, maxt, &t, &b1, &b2) && here) {
                tied = tied || t == hit.t;
                maxt = t;
                hit.t = t;
                hit.inst = inst;
                hit.tri = first + i;
                hit.b1 = b1;
                hit.b2 = b2;
            }
        }
        pk_pop(wstack, sp, cur, mask);
    }
    return hit.inst >= 0;
}

