// Two-level BVH traversal + Moller-Trumbore for one ray per lane.
//
// Replaces Scene::intersect/occluded -> BVH::intersect/occluded ->
// InstancedPrimitive -> Model -> Triangle::intersect/occluded
// (GoblinScene.cpp:75-87, GoblinBVH.cpp:189-280, GoblinPrimitive.cpp:103-118,
// GoblinModel.cpp:28-55, GoblinTriangle.cpp:38-163).
//
// * One 64-byte node fetch tests both children (min/max slab test, fused
//   multiply-add form with a clamped reciprocal direction; conservative: boxes
//   were nudged outwards by the packer).
// * The world ray enters an instance by the same un-normalised inverse
//   transform the reference uses (Transform::invertRay), so object-space t is
//   world t and hits from different instances compare directly.
// * The triangle test is Moller-Trumbore in the reference's exact operation
//   order, +-1e-7 barycentric slack, inclusive [mint, maxt], no culling.
// * Per-lane stack lives in LDS, column-major over the workgroup
//   (stack[level * GBL_BLOCK + tid]): conflict-free for ds_read/write_b32 since a
//   lane only ever touches its own column.
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

// ANY = true : Scene::occluded (first accepted triangle ends the query)
// ANY = false: Scene::intersect (closest hit; hit.t shrinks like ray.maxt)
template <bool ANY, bool STATS>
__device__ __forceinline__ bool trace(const DevScene& sc, F3 o, F3 d, float mint, float maxt, uint32_t* stk, Hit& hit,
                                      LaneCounters& cnt) {
    if (sc.num_instances == 0) return false;
    RaySpace world, r;
    ray_space(world, o, d);
    r = world;
    int sp = 0;
    stk[(sp++) * GBL_BLOCK] = GBL_STACK_EXIT;
    int cur = sc.tlas_root;
    int inst = -1;
    bool found = false;
    const DevNode* __restrict__ nodes = sc.nodes;
    for (;;) {
        // ---- interior nodes: descend while the reference is a plain node index
        while (static_cast<uint32_t>(cur) < static_cast<uint32_t>(GBL_STACK_EXIT)) {
            const float4* np = reinterpret_cast<const float4*>(nodes + cur);
            const float4 n0 = np[0];   // c0 lo.x hi.x lo.y hi.y
            const float4 n1 = np[1];   // c1 lo.x hi.x lo.y hi.y
            const float4 nz = np[2];   // c0 lo.z hi.z  c1 lo.z hi.z
            const int2 ch = *reinterpret_cast<const int2*>(np + 3);
            float ax0 = __builtin_fmaf(n0.x, r.idir.x, -r.ood.x), ax1 = __builtin_fmaf(n0.y, r.idir.x, -r.ood.x);
            float ay0 = __builtin_fmaf(n0.z, r.idir.y, -r.ood.y), ay1 = __builtin_fmaf(n0.w, r.idir.y, -r.ood.y);
            float az0 = __builtin_fmaf(nz.x, r.idir.z, -r.ood.z), az1 = __builtin_fmaf(nz.y, r.idir.z, -r.ood.z);
            float bx0 = __builtin_fmaf(n1.x, r.idir.x, -r.ood.x), bx1 = __builtin_fmaf(n1.y, r.idir.x, -r.ood.x);
            float by0 = __builtin_fmaf(n1.z, r.idir.y, -r.ood.y), by1 = __builtin_fmaf(n1.w, r.idir.y, -r.ood.y);
            float bz0 = __builtin_fmaf(nz.z, r.idir.z, -r.ood.z), bz1 = __builtin_fmaf(nz.w, r.idir.z, -r.ood.z);
            float a_lo = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), mint));
            float a_hi = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fminf(fmaxf(az0, az1), maxt));
            float b_lo = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fmaxf(fminf(bz0, bz1), mint));
            float b_hi = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fminf(fmaxf(bz0, bz1), maxt));
            bool ha = a_lo <= a_hi, hb = b_lo <= b_hi;
            if (STATS) cnt.nodes += 2;
            if (ha && hb) {
                bool a_first = a_lo <= b_lo;
                stk[(sp++) * GBL_BLOCK] = static_cast<uint32_t>(a_first ? ch.y : ch.x);
                cur = a_first ? ch.x : ch.y;
            } else if (ha) {
                cur = ch.x;
            } else if (hb) {
                cur = ch.y;
            } else {
                cur = static_cast<int>(stk[(--sp) * GBL_BLOCK]);
            }
        }
        if (cur == GBL_STACK_EXIT) break;
        if (cur == GBL_STACK_SENTINEL) {   // finished an instance: back to the world ray
            r = world;
            inst = -1;
            cur = static_cast<int>(stk[(--sp) * GBL_BLOCK]);
            continue;
        }
        // ---- leaf
        uint32_t ref = ~static_cast<uint32_t>(cur);
        if (inst < 0) {
            inst = static_cast<int>(ref >> 2);
            const DevInstance* ip = sc.instances + inst;
            ray_space(r, xf_point(ip->inv, world.o), xf_vector(ip->inv, world.d));
            stk[(sp++) * GBL_BLOCK] = GBL_STACK_SENTINEL;
            cur = ip->root;
        } else {
            uint32_t first = ref >> 2, count = (ref & 3u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                float t, b1, b2;
                if (STATS) cnt.tris += 1;
                if (tri_test(sc.tris + first + i, r.o, r.d, mint, maxt, &t, &b1, &b2)) {
                    if (ANY) return true;
                    maxt = t;
                    hit.t = t;
                    hit.inst = inst;
                    hit.tri = first + i;
                    hit.b1 = b1;
                    hit.b2 = b2;
                    found = true;
                }
            }
            cur = static_cast<int>(stk[(--sp) * GBL_BLOCK]);
        }
    }
    return found;
}
