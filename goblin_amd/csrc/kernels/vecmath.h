// float3 / colour helpers for the device integrator.
//
// Arithmetic that feeds a discrete decision or a radiance value is written in
// the reference's operation order (GoblinVector.h, GoblinColor.h) and the whole
// translation unit is compiled with -ffp-contract=off, so add/mul/div/sqrt round
// exactly like the CPU reference.  The only deliberate fused ops are the BVH
// slab tests (trace.h), which use __builtin_fmaf and can only add candidates.
#pragma once
#include <hip/hip_runtime.h>

struct F3 {
    float x, y, z;
};

__device__ __forceinline__ F3 f3(float x, float y, float z) {
    F3 r;
    r.x = x; r.y = y; r.z = z;
    return r;
}
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }   // Color * Color
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float absdot(F3 a, F3 b) { return fabsf(dot(a, b)); }
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float sqlen(F3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float length(F3 a) { return sqrtf(sqlen(a)); }
// Vector3 / float and Color / float multiply by the reciprocal (GoblinVector.h:166-169, GoblinColor.h:76-79)
__device__ __forceinline__ F3 div(F3 a, float s) {
    float inv = 1.0f / s;
    return f3(a.x * inv, a.y * inv, a.z * inv);
}
__device__ __forceinline__ F3 normalize(F3 a) { return div(a, length(a)); }
// Color == Color::Black with alpha fixed at 1 on this path (GoblinColor.h:104-106)
__device__ __forceinline__ bool is_black(F3 c) { return c.x == 0.0f && c.y == 0.0f && c.z == 0.0f; }
__device__ __forceinline__ F3 load3(const float* p) { return f3(p[0], p[1], p[2]); }

// 3x4 row-major affine helpers (Transform::onPoint/onVector/onNormal/invert*, GoblinTransform.cpp:97-160)
__device__ __forceinline__ F3 xf_point(const float* m, F3 p) {
    return f3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
__device__ __forceinline__ F3 xf_vector(const float* m, F3 v) {
    return f3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// (inv)^T * n : columns of the inverse
__device__ __forceinline__ F3 xf_normal(const float* inv, F3 n) {
    return f3(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z,
              inv[2] * n.x + inv[6] * n.y + inv[10] * n.z);
}

#define GBL_PI 3.14159265358979323f
#define GBL_TWO_PI 6.28318530718f
#define GBL_INV_PI 0.31830988618379067154f
#define GBL_INV_TWOPI 0.15915494309189533577f
#define GBL_INV_PI 0.31830988618379067154f
#define GBL_INV_TWOPI 0.15915494309189533577f

#include "refmath.h"
