// Surface interaction at a hit: fragment reconstruction, BSDFs, lights.
//
// Restates, in the reference's float operation order:
//   Triangle::intersect outputs      GoblinTriangle.cpp:79-123
//   Fragment::transform / frame      GoblinGeometry.cpp:17-37
//   Lambert / Blinn / Transparent / Mirror   GoblinMaterial.cpp:285-726
//   Point / Spot / Area lights       GoblinLight.cpp:87-99, 225-237, 277-287, 313-343, 368-394, 457-461
//   Geometry::pdf                    GoblinGeometry.cpp:44-62
//   warps, power heuristic           GoblinSampler.cpp:420-424, 517-557; GoblinSampler.h:286-290
#pragma once
#include "../device_scene.h"
#include "trace.h"
#include "vecmath.h"
#include "refmath.h"
#include "image.h"

#define GBL_MAT_LAMBERT 0u
#define GBL_MAT_BLINN 1u
#define GBL_MAT_TRANSPARENT 2u
#define GBL_MAT_MIRROR 3u
#define GBL_MAT_MASK 4u
#define GBL_MAT_SUBSURFACE 5u
#define GBL_LIGHT_POINT 0u
#define GBL_LIGHT_DIRECTIONAL 1u
#define GBL_LIGHT_SPOT 2u
#define GBL_LIGHT_AREA 3u
#define GBL_LIGHT_IBL 4u

struct Frag {
    F3 p, n;      // world position / shading normal
    F3 t, b;      // tangent frame rows (Fragment::getWorldToShade)
    float eps;    // 1e-3 * t
};

// What texture lookups read from the Fragment (EXT builds, filled only when the hit material has textures)
struct TexFrag {
    float u, v;
    F3 dpdu, dpdv;                    // world space (Fragment::transform)
    F3 dpdx, dpdy;                    // Intersection::computeUVDifferential; zero without ray differentials
    float dudx, dvdx, dudy, dvdy;
};

// What a texture lookup reads of the scene, by value: an out-of-line function taking `const DevScene&` would force the
// 1 KB kernel argument block into scratch (image.h).
struct TexCtx {
    const DevTexture* textures;
    ImgCtx img;
};
__device__ __forceinline__ TexCtx tex_ctx(const DevScene& sc) {
    TexCtx c = {sc.textures, img_ctx(sc)};
    return c;
}
template <int DEPTH>
__device__ __forceinline__ F3 tex_eval_ctx(const TexCtx& tc, int id, const Frag& fr, const TexFrag& tf);
// Material::perturb -> BumpShaders::evaluate (GoblinMaterial.cpp:221-283), which Scene::intersect runs on the closest hit's
// fragment (GoblinScene.cpp:75-83) before any ray differential is attached to it: the lookups see a zero footprint.
// `bump` displaces the surface along its normal (forward differences over du = dv = 0.002 in uv, the position moved along
// dpdu / dpdv), `normal` holds 2 n - 1 in the shading frame.  Out of line: three texture lookups nobody else needs inlined.
// Arguments and result travel by value (registers): handing the callee pointers to the caller's Frag / TexFrag put both
// into scratch for every EXT kernel (149 spilled registers, bump maps in the scene or not).
struct BumpOut {
    F3 n, dpdu, dpdv;
};
__device__ __noinline__ BumpOut perturb_fragment(TexCtx sc, int tex_bump, int tex_normal, F3 p, F3 n_in, float u, float v, F3 dpdu_in, F3 dpdv_in) {
    Frag fr;
    TexFrag tf;
    fr.p = p;
    fr.n = n_in;
    fr.t = fr.b = f3(0.0f, 0.0f, 0.0f);   // (texture lookups read p, uv and the differentials only)
    fr.eps = 0.0f;
    tf.u = u;
    tf.v = v;
    tf.dpdu = dpdu_in;
    tf.dpdv = dpdv_in;
    tf.dudx = tf.dvdx = tf.dudy = tf.dvdy = 0.0f;
    tf.dpdx = tf.dpdy = f3(0.0f, 0.0f, 0.0f);
    if (tex_bump >= 0) {
        const F3 n = fr.n;
        const float bump_d = tex_eval_ctx<GBL_TEX_MAX_DEPTH>(sc, tex_bump, fr, tf).x;
        const float du = 0.002f;
        Frag fdu = fr;
        TexFrag tdu = tf;
        fdu.p = p + du * tf.dpdu;
        tdu.u = u + du;
        tdu.v = v + 0.0f;
        const float bump_ddu = tex_eval_ctx<GBL_TEX_MAX_DEPTH>(sc, tex_bump, fdu, tdu).x;
        const F3 bump_dpdu = tf.dpdu + (bump_ddu - bump_d) / du * n;
        const float dv = 0.002f;
        Frag fdv = fr;
        TexFrag tdv = tf;
        fdv.p = p + dv * tf.dpdv;
        tdv.u = u + 0.0f;
        tdv.v = v + dv;
        const float bump_ddv = tex_eval_ctx<GBL_TEX_MAX_DEPTH>(sc, tex_bump, fdv, tdv).x;
        const F3 bump_dpdv = tf.dpdv + (bump_ddv - bump_d) / dv * n;
        F3 bump_n = normalize(cross(bump_dpdu, bump_dpdv));
        if (dot(bump_n, n) < 0.0f) bump_n = bump_n * -1.0f;
        fr.n = bump_n;
        tf.dpdu = bump_dpdu;
        tf.dpdv = bump_dpdv;
    }
    if (tex_normal >= 0) {
        // (the lookup reads the fragment as the bump map left it; Fragment::getWorldToShade: rows t, b, n)
        const F3 c = tex_eval_ctx<GBL_TEX_MAX_DEPTH>(sc, tex_normal, fr, tf);
        const F3 ns = 2.0f * c - f3(1.0f, 1.0f, 1.0f);
        const F3 n = fr.n;
        const F3 t = normalize(tf.dpdu - n * dot(tf.dpdu, n));
        const F3 b = cross(n, t);
        F3 nw = f3(t.x * ns.x + b.x * ns.y + n.x * ns.z, t.y * ns.x + b.y * ns.y + n.y * ns.z, t.z * ns.x + b.z * ns.y + n.z * ns.z);
        nw = normalize(nw);
        if (dot(nw, fr.n) < 0.0f) nw = nw * -1.0f;
        fr.n = nw;
    }
    BumpOut o;
    o.n = fr.n;
    o.dpdu = tf.dpdu;
    o.dpdv = tf.dpdv;
    return o;
}
// make_fragment's tail for a material with bump shaders
__device__ __forceinline__ void apply_perturb(const DevScene& sc, const DevMaterial& m, Frag& fr, TexFrag& tf) {
    const BumpOut o = perturb_fragment(tex_ctx(sc), m.tex_bump, m.tex_normal, fr.p, fr.n, tf.u, tf.v, tf.dpdu, tf.dpdv);
    fr.n = o.n;
    tf.dpdu = o.dpdu;
    tf.dpdv = o.dpdv;
    fr.t = normalize(tf.dpdu - fr.n * dot(tf.dpdu, fr.n));
    fr.b = cross(fr.n, fr.t);
}

// Rebuild the reference's Fragment for the closest hit and move it to world space.
template <bool EXT>
__device__ __forceinline__ void make_fragment(const DevScene& sc, const Hit& h, F3 wo_origin, F3 w_dir, Frag& fr, TexFrag* tf_out = nullptr) {
    const DevInstance* ip = sc.instances + h.inst;
    // the object-space ray the triangle test saw (Transform::invertRay)
    F3 ro = xf_point(ip->inv, wo_origin), rd = xf_vector(ip->inv, w_dir);
    const bool bumped = EXT && (sc.materials[ip->material].tex_bump >= 0 || sc.materials[ip->material].tex_normal >= 0);
    // (computed into a local and copied out at the end: selecting between the caller's TexFrag and a local one by pointer
    //  put both through scratch at every call site -- +10 % on every EXT scene)
    TexFrag tf_local;
    TexFrag* const tf = &tf_local;
    const bool want_tex = EXT && (tf_out != nullptr || bumped) && sc.materials[ip->material].has_tex != 0u;
    if (EXT && ip->shape != 0u) {
        // Sphere::intersect (GoblinSphere.cpp:32-86) / Disk::intersect (GoblinDisk.cpp:33-60): position, normal and
        // dpdu are algebraic in the hit point; uv and dpdv (atan2 / acos) only feed textures and bump maps
        F3 pos = ro + h.t * rd;
        F3 nrm = ip->shape == 1u ? normalize(pos) : f3(0.0f, 0.0f, 1.0f);
        F3 dpdu = f3(-GBL_TWO_PI * pos.y, GBL_TWO_PI * pos.x, 0.0f);
        if (want_tex) {
            float phi = gbl_atan2f(pos.y, pos.x);
            if (phi < 0.0f) phi += GBL_TWO_PI;
            tf->u = phi * GBL_INV_TWOPI;
            F3 dpdv;
            if (ip->shape == 1u) {
                float theta = gbl_acosf(pos.z / ip->radius);
                tf->v = theta * GBL_INV_PI;
                float inv_r = 1.0f / sqrtf(pos.x * pos.x + pos.y * pos.y);
                float cos_phi = pos.x * inv_r, sin_phi = pos.y * inv_r;
                dpdv = GBL_PI * f3(pos.z * cos_phi, pos.z * sin_phi, -ip->radius * gbl_sinf(theta));
            } else {
                float r = sqrtf(pos.x * pos.x + pos.y * pos.y);
                tf->v = r / ip->radius;
                dpdv = f3(ip->radius * pos.x / r, ip->radius * pos.y / r, 0.0f);
            }
            tf->dpdu = xf_vector(ip->m, dpdu);
            tf->dpdv = xf_vector(ip->m, dpdv);
        }
        fr.p = xf_point(ip->m, pos);
        fr.n = normalize(xf_normal(ip->inv, nrm));
        F3 dpdu_w = xf_vector(ip->m, dpdu);
        fr.t = normalize(dpdu_w - fr.n * dot(dpdu_w, fr.n));
        fr.b = cross(fr.n, fr.t);
        fr.eps = 1e-3f * h.t;
        if (bumped) apply_perturb(sc, sc.materials[ip->material], fr, *tf);
        if (tf_out != nullptr && want_tex) *tf_out = tf_local;
        return;
    }
    const DevTri* tp = sc.tris + h.tri;
    const float4 q0 = reinterpret_cast<const float4*>(tp)[0];
    const float4 q1 = reinterpret_cast<const float4*>(tp)[1];
    const float4 q2 = reinterpret_cast<const float4*>(tp)[2];
    F3 e1 = f3(q1.x, q1.y, q1.z), e2 = f3(q2.x, q2.y, q2.z);
    // (the record's own copy of the mesh's flags: a mesh without vertex normals and uvs -- face normal, default uvs -- needs
    //  nothing from tri_shade, one dependent fetch less at every hit)
    const uint32_t tri_flags = __float_as_uint(q1.w);
    DevTriShade sh = {};
    if (tri_flags != 0u) sh = sc.tri_shade[__float_as_uint(q0.w)];
    float b1 = h.b1, b2 = h.b2;
    float b0 = 1.0f - b1 - b2;
    F3 pos = ro + h.t * rd;
    F3 nrm;
    if (tri_flags & 1u) {
        F3 n0 = load3(sc.normals + 3 * sh.v[0]), n1 = load3(sc.normals + 3 * sh.v[1]), n2 = load3(sc.normals + 3 * sh.v[2]);
        nrm = normalize(b0 * n0 + b1 * n1 + b2 * n2);
    } else {
        nrm = normalize(cross(e1, e2));
    }
    F3 dpdu;
    if (tri_flags & 2u) {
        const float* uv = sc.uvs;
        float u0 = uv[2 * sh.v[0]], v0 = uv[2 * sh.v[0] + 1];
        float u1 = uv[2 * sh.v[1]], v1 = uv[2 * sh.v[1] + 1];
        float u2 = uv[2 * sh.v[2]], v2 = uv[2 * sh.v[2] + 1];
        float du1 = u1 - u0, dv1 = v1 - v0, du2 = u2 - u0, dv2 = v2 - v0;
        float det = du1 * dv2 - dv1 * du2;   // never 0: the packer rejects degenerate-uv meshes
        float inv_det = 1.0f / det;
        dpdu = inv_det * (dv2 * e1 - dv1 * e2);
        if (want_tex) {
            tf->u = b0 * u0 + b1 * u1 + b2 * u2;
            tf->v = b0 * v0 + b1 * v1 + b2 * v2;
            tf->dpdv = xf_vector(ip->m, inv_det * (-du2 * e1 + du1 * e2));
        }
    } else {
        dpdu = e1;   // default uvs (0,0),(1,0),(0,1): invDet * (1*e1 - 0*e2)
        if (want_tex) {
            tf->u = b0 * 0.0f + b1 * 1.0f + b2 * 0.0f;
            tf->v = b0 * 0.0f + b1 * 0.0f + b2 * 1.0f;
            tf->dpdv = xf_vector(ip->m, e2);
        }
    }
    // Fragment::transform
    fr.p = xf_point(ip->m, pos);
    fr.n = normalize(xf_normal(ip->inv, nrm));
    F3 dpdu_w = xf_vector(ip->m, dpdu);
    if (want_tex) tf->dpdu = dpdu_w;
    // Fragment::getWorldToShade
    fr.t = normalize(dpdu_w - fr.n * dot(dpdu_w, fr.n));
    fr.b = cross(fr.n, fr.t);
    fr.eps = 1e-3f * h.t;
    if (EXT && bumped) apply_perturb(sc, sc.materials[ip->material], fr, *tf);
    if (EXT && tf_out != nullptr && want_tex) *tf_out = tf_local;
}

// ------------------------------------------------------------------- textures
// Intersection::computeUVDifferential (GoblinPrimitive.cpp:32-97) for the camera ray's auxiliary rays.
__device__ __forceinline__ void uv_differential(const Frag& fr, TexFrag& tf, bool has, F3 dxo, F3 dxd, F3 dyo, F3 dyd) {
    tf.dudx = tf.dvdx = tf.dudy = tf.dvdy = 0.0f;
    tf.dpdx = tf.dpdy = f3(0, 0, 0);
    if (!has) return;
    const F3 p = fr.p, n = fr.n;
    float minus_d = dot(p, n);
    float tdx = (minus_d - dot(dxo, n)) / dot(dxd, n);
    float tdy = (minus_d - dot(dyo, n)) / dot(dyd, n);
    if (isnan(tdx) || isnan(tdy)) return;
    F3 pdx = dxo + tdx * dxd, pdy = dyo + tdy * dyd;
    tf.dpdx = pdx - p;
    tf.dpdy = pdy - p;
    float a00, a01, a10, a11, bx0, bx1, by0, by1;
    if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) {
        a00 = tf.dpdu.y; a01 = tf.dpdv.y; a10 = tf.dpdu.z; a11 = tf.dpdv.z;
        bx0 = tf.dpdx.y; bx1 = tf.dpdx.z; by0 = tf.dpdy.y; by1 = tf.dpdy.z;
    } else if (fabsf(n.y) > fabsf(n.z)) {
        a00 = tf.dpdu.x; a01 = tf.dpdv.x; a10 = tf.dpdu.z; a11 = tf.dpdv.z;
        bx0 = tf.dpdx.x; bx1 = tf.dpdx.z; by0 = tf.dpdy.x; by1 = tf.dpdy.z;
    } else {
        a00 = tf.dpdu.x; a01 = tf.dpdv.x; a10 = tf.dpdu.y; a11 = tf.dpdv.y;
        bx0 = tf.dpdx.x; bx1 = tf.dpdx.y; by0 = tf.dpdy.x; by1 = tf.dpdy.y;
    }
    float det = a00 * a11 - a01 * a10;   // solve2x2LinearSystem, GoblinUtils.h:151-163
    if (fabsf(det) < 1e-10f) return;
    float x = (+a11 * bx0 - a01 * bx1) / det, y = (-a10 * bx0 + a00 * bx1) / det;
    if (!(isnan(x) || isnan(y))) {
        tf.dudx = x;
        tf.dvdx = y;
    }
    x = (+a11 * by0 - a01 * by1) / det;
    y = (-a10 * by0 + a00 * by1) / det;
    if (!(isnan(x) || isnan(y))) {
        tf.dudy = x;
        tf.dvdy = y;
    }
}

// SphericalMapping::pointToST, GoblinTexture.cpp:339-347
__device__ __forceinline__ void point_to_st(const float* to_tex, F3 p, float* s, float* t) {
    F3 v = normalize(xf_point(to_tex, p) - f3(0.0f, 0.0f, 0.0f));
    float theta = gbl_acosf(fminf(fmaxf(v.z, -1.0f), 1.0f));
    float phi = gbl_atan2f(v.y, v.x);
    phi = phi < 0.0f ? phi + GBL_TWO_PI : phi;
    *s = phi * GBL_INV_TWOPI;
    *t = theta * GBL_INV_PI;
}
__device__ __forceinline__ float integrate_checker(float x) {
    float xh = 0.5f * x;
    return floorf(xh) + 2.0f * fmaxf(xh - floorf(xh) - 0.5f, 0.0f);
}
// TextureMapping::map (UVMapping / SphericalMapping, GoblinTexture.cpp:296-347)
__device__ __forceinline__ TexCoord tex_map(const DevTexture& g, const Frag& fr, const TexFrag& tf) {
    TexCoord tc;
    if (g.mapping == 1u) {
        point_to_st(g.to_tex, fr.p, &tc.s, &tc.t);
        float sdx, tdx, sdy, tdy;
        point_to_st(g.to_tex, fr.p + tf.dpdx, &sdx, &tdx);
        point_to_st(g.to_tex, fr.p + tf.dpdy, &sdy, &tdy);
        float dsdx = sdx - tc.s;
        if (dsdx > 0.5f) dsdx -= 1.0f;
        else if (dsdx < -0.5f) dsdx += 1.0f;
        float dsdy = sdy - tc.s;
        if (dsdy > 0.5f) dsdy -= 1.0f;
        else if (dsdy < -0.5f) dsdy += 1.0f;
        tc.dsdx = dsdx;
        tc.dsdy = dsdy;
        tc.dtdx = tdx - tc.t;
        tc.dtdy = tdy - tc.t;
    } else {
        tc.s = g.uv_scale[0] * tf.u + g.uv_offset[0];
        tc.t = g.uv_scale[1] * tf.v + g.uv_offset[1];
        tc.dsdx = g.uv_scale[0] * tf.dudx;
        tc.dtdx = g.uv_scale[1] * tf.dvdx;
        tc.dsdy = g.uv_scale[0] * tf.dudy;
        tc.dtdy = g.uv_scale[1] * tf.dvdy;
    }
    return tc;
}
// Texture<T>::lookup (float textures carry their value in every channel)
template <int DEPTH>
__device__ __forceinline__ F3 tex_eval_ctx(const TexCtx& sc, int id, const Frag& fr, const TexFrag& tf) {
    const DevTexture& g = sc.textures[id];
    const F3 value = f3(g.value[0], g.value[1], g.value[2]);
    if constexpr (DEPTH == 0) {
        return value;   // the packer rejects graphs deeper than GBL_TEX_MAX_DEPTH
    } else {
        if (g.type == 0u) return value;
        if (g.type == 3u) {   // ImageTexture<T>::lookup: map, then the MIPMap
            const TexCoord tc = tex_map(g, fr, tf);
            return mip_lookup(sc.img, sc.img.images + g.image, g.is_float != 0u, tc, g.filter, g.address, g.max_aniso);
        }
        const F3 a = tex_eval_ctx<DEPTH - 1>(sc, g.child[0], fr, tf);
        const F3 b = tex_eval_ctx<DEPTH - 1>(sc, g.child[1], fr, tf);
        if (g.type == 2u) return a * b.x;   // ScaleTexture: mScale->lookup * mTexture->lookup
        const TexCoord tc = tex_map(g, fr, tf);
        const float s = tc.s, t = tc.t, dsdx = tc.dsdx, dtdx = tc.dtdx, dsdy = tc.dsdy, dtdy = tc.dtdy;
        const bool first = (static_cast<int>(floorf(s)) + static_cast<int>(floorf(t))) % 2 == 0;
        if (!g.filter) return first ? a : b;
        float ds = fmaxf(fabsf(dsdx), fabsf(dsdy)), dt = fmaxf(fabsf(dtdx), fabsf(dtdy));
        float s0 = s - ds, s1 = s + ds, t0 = t - dt, t1 = t + dt;
        if (static_cast<int>(floorf(s0)) == static_cast<int>(floorf(s1)) && static_cast<int>(floorf(t0)) == static_cast<int>(floorf(t1)))
            return first ? a : b;
        float sr = (integrate_checker(s1) - integrate_checker(s0)) / (2.0f * ds);
        float tr = (integrate_checker(t1) - integrate_checker(t0)) / (2.0f * dt);
        float area2 = sr + tr - 2.0f * sr * tr;
        if (ds > 1.0f || dt > 1.0f) area2 = 0.5f;
        return (1.0f - area2) * a + area2 * b;
    }
}
template <int DEPTH>
__device__ __forceinline__ F3 tex_eval(const DevScene& sc, int id, const Frag& fr, const TexFrag& tf) {
    return tex_eval_ctx<DEPTH>(tex_ctx(sc), id, fr, tf);
}
// The hit's material with its texture slots evaluated at this fragment (every lookup of a bounce sees the same
// Fragment, so resolving once is what the reference's repeated lookups return).
__device__ __forceinline__ void resolve_material(const DevScene& sc, const DevMaterial& m, const Frag& fr, const TexFrag& tf, DevMaterial& out) {
    out = m;
    if (m.tex_color >= 0) {
        F3 c = tex_eval<GBL_TEX_MAX_DEPTH>(sc, m.tex_color, fr, tf);
        out.color[0] = c.x; out.color[1] = c.y; out.color[2] = c.z;
    }
    if (m.tex_color2 >= 0) {
        F3 c = tex_eval<GBL_TEX_MAX_DEPTH>(sc, m.tex_color2, fr, tf);
        out.color2[0] = c.x; out.color2[1] = c.y; out.color2[2] = c.z;
    }
    if (m.tex_exponent >= 0) out.exponent = tex_eval<GBL_TEX_MAX_DEPTH>(sc, m.tex_exponent, fr, tf).x;
    if (m.tex_color3 >= 0) {
        F3 c = tex_eval<GBL_TEX_MAX_DEPTH>(sc, m.tex_color3, fr, tf);
        out.color3[0] = c.x; out.color3[1] = c.y; out.color3[2] = c.z;
    }
}

// shadeToWorld * v, shadeToWorld = transpose(rows t, b, n)
__device__ __forceinline__ F3 shade_to_world(const Frag& fr, F3 v) {
    return f3(fr.t.x * v.x + fr.b.x * v.y + fr.n.x * v.z, fr.t.y * v.x + fr.b.y * v.y + fr.n.y * v.z,
              fr.t.z * v.x + fr.b.z * v.y + fr.n.z * v.z);
}

__device__ __forceinline__ F3 cosine_sample_hemisphere(float u1, float u2) {
    float sin_t = sqrtf(u1);
    float cos_t = sqrtf(fmaxf(0.0f, 1.0f - u1));
    float phi = GBL_TWO_PI * u2;
    return f3(sin_t * gbl_cosf(phi), sin_t * gbl_sinf(phi), cos_t);
}
__device__ __forceinline__ F3 uniform_sample_hemisphere(float u1, float u2) {
    float sin_t = sqrtf(fmaxf(0.0f, 1.0f - u1 * u1));
    float phi = GBL_TWO_PI * u2;
    return f3(sin_t * gbl_cosf(phi), sin_t * gbl_sinf(phi), u1);
}
__device__ __forceinline__ float power_heuristic(float pa, float pb) {
    float A = 1.0f * pa, B = 1.0f * pb;
    return A * A / (A * A + B * B);
}

// ------------------------------------------------------------------ materials
__device__ __forceinline__ float clamp_pm1(float f) { return f < -1.0f ? -1.0f : (f > 1.0f ? 1.0f : f); }

__device__ __forceinline__ float fresnel_dielectric(float cosi, float etai, float etat) {
    cosi = clamp_pm1(cosi);
    float sint = (etai / etat) * sqrtf(fmaxf(0.0f, 1.0f - cosi * cosi));
    if (sint >= 1.0f) return 1.0f;
    float cost = sqrtf(fmaxf(0.0f, 1 - sint * sint));
    cosi = fabsf(cosi);
    float r_parl = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    float r_perp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
__device__ __forceinline__ float fresnel_conductor(float cosi, float eta, float k) {
    float tmp = (eta * eta + k * k);
    float cosi2 = cosi * cosi;
    float r_parl2 = (tmp * cosi2 - 2.0f * eta * cosi + 1.0f) / (tmp * cosi2 + 2.0f * eta * cosi + 1.0f);
    float r_perp2 = (tmp - 2.0f * eta * cosi + cosi2) / (tmp + 2.0f * eta * cosi + cosi2);
    return (r_parl2 + r_perp2) * 0.5f;
}

__device__ __forceinline__ bool same_hemisphere(F3 n, F3 wo, F3 wi) { return dot(wo, n) * dot(wi, n) > 0.0f; }

__device__ __forceinline__ F3 blinn_bsdf(const DevMaterial& m, F3 n, F3 wo, F3 wi) {
    // getSampleType strips Reflection when wo, wi are on opposite sides -> no match
    if (!(dot(n, wo) * dot(n, wi) > 0.0f)) return f3(0, 0, 0);
    float cosi = absdot(n, wi), coso = absdot(n, wo);
    if (cosi == 0.0f || coso == 0.0f) return f3(0, 0, 0);
    F3 wh = normalize(wo + wi);
    float cosh = absdot(n, wh);
    float e = m.exponent;
    float D = (e + 2.0f) * GBL_INV_TWOPI * gbl_powf_inline(cosh, e);
    float wo_wh = absdot(wo, wh);
    float G = fminf(1.0f, fminf(2.0f * cosh * coso / wo_wh, 2.0f * cosh * cosi / wo_wh));
    float F = m.k > 0.0f ? fresnel_conductor(wo_wh, m.index, m.k) : fresnel_dielectric(wo_wh, 1.0f, m.index);
    F3 kg = f3(m.color[0], m.color[1], m.color[2]);
    return div(kg * D * G * F, 4.0f * cosi * coso);
}
__device__ __forceinline__ float blinn_pdf(const DevMaterial& m, F3 n, F3 wo, F3 wi) {
    if (!same_hemisphere(n, wo, wi)) return 0.0f;
    F3 wh = normalize(wo + wi);
    float cos_h = absdot(wh, n);
    float e = m.exponent;
    return (e + 1.0f) * gbl_powf_inline(cos_h, e) / (GBL_TWO_PI * 4.0f * dot(wo, wh));
}

// material->bsdf(fragment, wo, wi)
__device__ __forceinline__ F3 mat_bsdf(const DevMaterial& m, F3 n, F3 wo, F3 wi) {
    if (m.type == GBL_MAT_LAMBERT) {
        if (dot(n, wo) * dot(n, wi) > 0.0f) {
            F3 kd = f3(m.color[0], m.color[1], m.color[2]);
            F3 v = kd * GBL_INV_PI;
            return f3(0.0f + v.x, 0.0f + v.y, 0.0f + v.z);   // f(Black) += Kd * INV_PI
        }
        return f3(0, 0, 0);
    }
    if (m.type == GBL_MAT_BLINN) return blinn_bsdf(m, n, wo, wi);
    return f3(0, 0, 0);
}
__device__ __forceinline__ float mat_pdf(const DevMaterial& m, F3 n, F3 wo, F3 wi) {
    if (m.type == GBL_MAT_LAMBERT) return same_hemisphere(n, wo, wi) ? absdot(n, wi) * GBL_INV_PI : 0.0f;
    if (m.type == GBL_MAT_BLINN) return blinn_pdf(m, n, wo, wi);
    return 0.0f;
}

// material->sampleBSDF(fragment, wo, bs, &wi, &pdf, BSDFAll, &sampledType); *specular = sampledType & BSDFSpecular
__device__ __forceinline__ F3 mat_sample(const DevMaterial& m, const Frag& fr, F3 wo, float u_comp, float u1, float u2, F3* wi,
                                         float* pdf, bool* specular) {
    F3 n = fr.n;
    if (m.type == GBL_MAT_LAMBERT) {
        F3 local = cosine_sample_hemisphere(u1, u2);
        if (dot(wo, n) < 0.0f) local = local * -1.0f;
        *wi = shade_to_world(fr, local);
        *pdf = mat_pdf(m, n, wo, *wi);
        *specular = false;
        return f3(m.color[0], m.color[1], m.color[2]) * GBL_INV_PI;
    }
    if (m.type == GBL_MAT_BLINN) {
        float e = m.exponent;
        float cos_t = gbl_powf_inline(u1, 1.0f / (e + 1.0f));
        float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
        float phi = u2 * GBL_TWO_PI;
        F3 wh_local = f3(sin_t * gbl_cosf(phi), sin_t * gbl_sinf(phi), cos_t);
        if (dot(wo, n) < 0.0f) wh_local = wh_local * -1.0f;
        F3 wh = shade_to_world(fr, wh_local);
        *wi = -wo + 2.0f * dot(wo, wh) * wh;
        *pdf = blinn_pdf(m, n, wo, *wi);
        *specular = false;
        return blinn_bsdf(m, n, wo, *wi);
    }
    *specular = true;
    if (m.type == GBL_MAT_TRANSPARENT) {
        // specularReflectDieletric / specularRefract with etai = 1, etat = index
        float cosi = dot(n, wo);
        bool entering = cosi > 0.0f;
        F3 nn = entering ? n : -n;
        float ci = entering ? cosi : -cosi;
        float ei = entering ? 1.0f : m.index, et = entering ? m.index : 1.0f;
        float fr_refl = fresnel_dielectric(ci, ei, et);
        F3 w_refl = 2 * ci * nn - wo;
        float reflect = fr_refl / ci;
        // refraction: (et, ei) roles as in specularRefract (etao = 1, etai = index)
        float ro_et = entering ? 1.0f : m.index, ro_ei = entering ? m.index : 1.0f;
        float f2 = fresnel_dielectric(ci, ro_et, ro_ei);
        float refract = 0.0f;
        F3 w_refr = f3(0, 0, 0);
        if (f2 != 1.0f) {
            float eta = ro_et / ro_ei;
            w_refr = normalize(nn * (eta * ci - sqrtf(fmaxf(0.0f, 1.0f - eta * eta * (1.0f - ci * ci)))) - eta * wo);
            refract = eta * eta * (1.0f - f2) / absdot(w_refr, nn);
        }
        float chance = reflect * absdot(w_refl, n);
        if (u_comp < chance) {
            *wi = w_refl;
            *pdf = chance;
            return f3(m.color[0], m.color[1], m.color[2]) * reflect;
        }
        *wi = w_refr;
        *pdf = 1.0f - chance;
        return f3(m.color2[0], m.color2[1], m.color2[2]) * refract;
    }
    if (m.type == GBL_MAT_SUBSURFACE) {
        // SubsurfaceMaterial::sampleBSDF (GoblinMaterial.cpp:728-745): Kr * specularReflectDieletric(1, eta), pdf 1
        float cosi = dot(n, wo);
        bool entering = cosi > 0.0f;
        F3 nn = entering ? n : -n;
        float ci = entering ? cosi : -cosi;
        float ei = entering ? 1.0f : m.index, et = entering ? m.index : 1.0f;
        float f = fresnel_dielectric(ci, ei, et);
        *wi = 2 * ci * nn - wo;
        *pdf = 1.0f;
        return f3(m.color3[0], m.color3[1], m.color3[2]) * (f / ci);
    }
    // mirror
    float cosi = dot(n, wo);
    *pdf = 1.0f;
    if (cosi <= 0.0f) {
        *wi = f3(0, 0, 0);
        return f3(m.color[0], m.color[1], m.color[2]) * 0.0f;
    }
    float f = fresnel_conductor(cosi, m.index, m.k);
    *wi = 2 * cosi * n - wo;
    return f3(m.color[0], m.color[1], m.color[2]) * (f / cosi);
}

// MaskMaterial (GoblinMaterial.cpp:747-811): `m` is the wrapped material with its textures resolved, alpha and
// tcolor the mask's own two lookups; for any other material is_mask is false and `m` the material itself.
struct ResolvedMat {
    DevMaterial m;
    float alpha;
    F3 tcolor;
    bool is_mask;
};
__device__ __forceinline__ void resolve_hit_material(const DevScene& sc, int material, const Frag& fr, const TexFrag& tf, ResolvedMat& r) {
    const DevMaterial& outer = sc.materials[material];
    r.is_mask = false;
    r.alpha = 1.0f;
    r.tcolor = f3(1.0f, 1.0f, 1.0f);
    if (outer.type != GBL_MAT_MASK) {
        if (outer.has_tex != 0u) resolve_material(sc, outer, fr, tf, r.m);
        else r.m = outer;
        return;
    }
    r.is_mask = true;
    r.alpha = outer.tex_exponent >= 0 ? tex_eval<GBL_TEX_MAX_DEPTH>(sc, outer.tex_exponent, fr, tf).x : outer.exponent;
    r.tcolor = outer.tex_color >= 0 ? tex_eval<GBL_TEX_MAX_DEPTH>(sc, outer.tex_color, fr, tf) : f3(outer.color[0], outer.color[1], outer.color[2]);
    const DevMaterial& inner = sc.materials[outer.masked];
    if (inner.has_tex != 0u) resolve_material(sc, inner, fr, tf, r.m);
    else r.m = inner;
}
// bsdf / pdf / sampleBSDF with type == BSDFAll; *null_sampled: the alpha branch was taken (sampledType == BSDFnullptr)
__device__ __forceinline__ F3 rmat_bsdf(const ResolvedMat& r, F3 n, F3 wo, F3 wi) {
    F3 f = mat_bsdf(r.m, n, wo, wi);
    return r.is_mask ? r.alpha * f : f;
}
__device__ __forceinline__ float rmat_pdf(const ResolvedMat& r, F3 n, F3 wo, F3 wi) {
    float p = mat_pdf(r.m, n, wo, wi);
    return r.is_mask ? r.alpha * p : p;
}
__device__ __forceinline__ F3 rmat_sample(const ResolvedMat& r, const Frag& fr, F3 wo, float u_comp, float u1, float u2, F3* wi, float* pdf,
                                          bool* specular, bool* null_sampled) {
    *null_sampled = false;
    if (r.is_mask && !(u_comp < r.alpha)) {
        *wi = -normalize(wo);
        *pdf = 1.0f - r.alpha;
        *specular = false;
        *null_sampled = true;
        return (1.0f - r.alpha) * r.tcolor;
    }
    F3 f = mat_sample(r.m, fr, wo, u_comp, u1, u2, wi, pdf, specular);
    if (r.is_mask) {
        f = r.alpha * f;
        *pdf *= r.alpha;
    }
    return f;
}

// --------------------------------------------------------------------- lights
// Geometry::pdf for one emitting triangle, light-local space
__device__ __forceinline__ float light_tri_pdf(const DevLightTri& lt, F3 p, F3 wi) {
    F3 p0 = f3(lt.p0[0], lt.p0[1], lt.p0[2]), p1 = f3(lt.p1[0], lt.p1[1], lt.p1[2]), p2 = f3(lt.p2[0], lt.p2[1], lt.p2[2]);
    F3 e1 = p1 - p0, e2 = p2 - p0;
    F3 s1 = cross(wi, e2);
    float divisor = dot(s1, e1);
    if (divisor == 0.0f) return 0.0f;
    float inv = 1.0f / divisor;
    const float eps = 1e-7f;
    F3 s = p - p0;
    float b1 = dot(s, s1) * inv;
    if (b1 + eps < 0.0f || b1 - eps > 1.0f) return 0.0f;
    F3 s2 = cross(s, e1);
    float b2 = dot(wi, s2) * inv;
    if (b2 + eps < 0.0f || b1 + b2 - eps > 1.0f) return 0.0f;
    float t = dot(e2, s2) * inv;
    if (t < 1e-3f || t > INFINITY) return 0.0f;
    float b0 = 1.0f - b1 - b2;
    F3 pos = p + t * wi;
    F3 nrm;
    if (lt.has_normal != 0.0f) {
        F3 n0 = f3(lt.n0[0], lt.n0[1], lt.n0[2]), n1 = f3(lt.n1[0], lt.n1[1], lt.n1[2]), n2 = f3(lt.n2[0], lt.n2[1], lt.n2[2]);
        nrm = normalize(b0 * n0 + b1 * n1 + b2 * n2);
    } else {
        nrm = normalize(cross(e1, e2));
    }
    float pdf = sqlen(p - pos) / (lt.area * absdot(-wi, nrm));
    if (isinf(pdf)) pdf = 0.0f;
    return pdf;
}
// GeometrySet::pdf
__device__ __forceinline__ float light_geoset_pdf(const DevScene& sc, const DevLight& l, F3 p, F3 wi) {
    float pdf = 0.0f;
    for (uint32_t k = 0; k < l.tri_count; ++k) {
        const DevLightTri& lt = sc.light_tris[l.tri_first + k];
        pdf += lt.area * light_tri_pdf(lt, p, wi);
    }
    pdf /= l.sum_area;
    return pdf;
}

struct LightSampleOut {
    F3 L, wi;
    float pdf, maxt;
};

// coordinateAxises, GoblinUtils.cpp:58-69
__device__ __forceinline__ void coordinate_axes(F3 a1, F3* a2, F3* a3) {
    if (fabsf(a1.x) > fabsf(a1.y)) {
        float inv = 1.0f / sqrtf(a1.x * a1.x + a1.z * a1.z);
        *a2 = f3(-a1.z * inv, 0.0f, a1.x * inv);
    } else {
        float inv = 1.0f / sqrtf(a1.y * a1.y + a1.z * a1.z);
        *a2 = f3(0.0f, -a1.z * inv, a1.y * inv);
    }
    *a3 = cross(a1, *a2);
}
// uniformSampleDisk, GoblinSampler.cpp:565-602
__device__ __forceinline__ void uniform_sample_disk(float u1, float u2, float* ox, float* oy) {
    float r, theta;
    float x = 2.0f * u1 - 1.0f;
    float y = 2.0f * u2 - 1.0f;
    if (x + y > 0) {
        if (x > y) {
            r = x;
            theta = 0.25f * GBL_PI * (y / x);
        } else {
            r = y;
            theta = 0.25f * GBL_PI * (2.0f - x / y);
        }
    } else {
        if (x < y) {
            r = -x;
            theta = 0.25f * GBL_PI * (4.0f + y / x);
        } else {
            r = -y;
            theta = y != 0.0f ? 0.25f * GBL_PI * (6.0f - x / y) : 0.0f;
        }
    }
    *ox = r * gbl_cosf(theta);
    *oy = r * gbl_sinf(theta);
}
// Geometry::pdf for an analytic emitter, light-local space (GoblinGeometry.cpp:44-62)
__device__ __forceinline__ float light_shape_area_pdf(const DevLight& l, F3 p, F3 wi) {
    float t;
    const bool got = l.shape == 1u ? sphere_test(l.radius, p, wi, 1e-3f, INFINITY, &t) : disk_test(l.radius, p, wi, 1e-3f, INFINITY, &t);
    if (!got) return 0.0f;
    F3 pos = p + t * wi;
    F3 nrm = l.shape == 1u ? normalize(pos) : f3(0.0f, 0.0f, 1.0f);
    float pdf = sqlen(p - pos) / (l.sum_area * absdot(-wi, nrm));
    if (isinf(pdf)) pdf = 0.0f;
    return pdf;
}
// GeometrySet::pdf over one Sphere (Sphere::pdf, GoblinSphere.cpp:126-138) or Disk (Geometry::pdf)
__device__ __forceinline__ float light_shape_pdf(const DevLight& l, F3 p, F3 wi) {
    float g;
    const float r2 = l.radius * l.radius, d2 = sqlen(p);
    if (l.shape == 1u && !(d2 - r2 < 1e-4f)) {
        float sin_max2 = r2 / d2;
        float cos_max = sqrtf(fmaxf(0.0f, 1.0f - sin_max2));
        g = 1.0f / (GBL_TWO_PI * (1.0f - cos_max));   // uniformConePdf
    } else {
        g = light_shape_area_pdf(l, p, wi);
    }
    float pdf = 0.0f;
    pdf += l.sum_area * g;
    pdf /= l.sum_area;
    return pdf;
}
// Sphere::sample(p, u1, u2, &n) (GoblinSphere.cpp:109-124) / Disk::sample (GoblinDisk.cpp:76-80)
__device__ __forceinline__ F3 light_shape_sample(const DevLight& l, F3 p, float u1, float u2, F3* normal) {
    if (l.shape != 1u) {
        *normal = f3(0.0f, 0.0f, 1.0f);
        float x, y;
        uniform_sample_disk(u1, u2, &x, &y);
        return f3(l.radius * x, l.radius * y, 0.0f);
    }
    const float r2 = l.radius * l.radius, d2 = sqlen(p);
    if (d2 - r2 < 1e-4f) {   // uniformSampleSphere
        float z = 1.0f - 2.0f * u1;
        float sin_t = sqrtf(fmaxf(0.0f, 1.0f - z * z));
        float phi = GBL_TWO_PI * u2;
        *normal = f3(sin_t * gbl_cosf(phi), sin_t * gbl_sinf(phi), z);
        return l.radius * (*normal);
    }
    F3 z_axis = normalize(-p), x_axis, y_axis;
    coordinate_axes(z_axis, &x_axis, &y_axis);
    float sin_max2 = r2 / d2;
    float cos_max = sqrtf(fmaxf(0.0f, 1.0f - sin_max2));
    float cos_t = 1.0f - u1 + u1 * cos_max;
    float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
    float phi = GBL_TWO_PI * u2;
    F3 d = x_axis * sin_t * gbl_cosf(phi) + y_axis * sin_t * gbl_sinf(phi) + z_axis * cos_t;
    float t;
    F3 p_hit;
    if (sphere_test(l.radius, p, d, 1e-3f, INFINITY, &t)) p_hit = p + t * d;
    else p_hit = p + (sqrtf(d2) * cos_max) * d;   // the ray scratches over the sphere's surface
    *normal = normalize(p_hit);
    return p_hit;
}

// light->sampleL(p, epsilon, ls, ...)
template <bool EXT>
__device__ __forceinline__ void light_sample(const DevScene& sc, const DevLight& l, F3 p, float epsilon, float u_comp, float u1,
                                             float u2, LightSampleOut& o) {
    F3 color = f3(l.color[0], l.color[1], l.color[2]);
    if (EXT && l.type == GBL_LIGHT_IBL) {   // ImageBasedLight::sampleL, GoblinLight.cpp:529-555: the shadow ray has no far end
        const float4 r = ibl_sample(img_ctx(sc), &l, u1, u2, &o.wi);
        o.L = f3(r.x, r.y, r.z);
        o.pdf = r.w;
        o.maxt = INFINITY;
        return;
    }
    if (EXT && l.type == GBL_LIGHT_DIRECTIONAL) {   // DirectionalLight::sampleL, GoblinLight.cpp:145-154
        o.wi = -f3(l.axis[0], l.axis[1], l.axis[2]);
        o.pdf = 1.0f;
        o.maxt = INFINITY;
        o.L = color;
        return;
    }
    if (EXT && l.type == GBL_LIGHT_AREA && l.shape != 0u) {   // AreaLight::sampleL over one analytic geometry
        F3 p_local = xf_point(l.inv, p);
        F3 ns_local;
        F3 ps_local = light_shape_sample(l, p_local, u1, u2, &ns_local);
        F3 wi_local = normalize(ps_local - p_local);
        o.pdf = light_shape_pdf(l, p_local, wi_local);
        F3 ps = xf_point(l.m, ps_local);
        F3 ns = normalize(xf_normal(l.inv, ns_local));
        o.wi = normalize(ps - p);
        o.maxt = length(ps - p) - epsilon;
        o.L = dot(ns, -o.wi) > 0.0f ? color : f3(0, 0, 0);
        return;
    }
    if (l.type == GBL_LIGHT_AREA) {
        F3 p_local = xf_point(l.inv, p);
        // CDF1D::sampleDiscrete: lower_bound(cdf, u) - 1, clamped at 0
        uint32_t tri = 0;
        for (uint32_t k = 0; k < l.tri_count; ++k)
            if (sc.light_tris[l.tri_first + k].cdf_hi < u_comp) tri = k + 1;
        if (tri >= l.tri_count) tri = l.tri_count - 1;
        const DevLightTri& lt = sc.light_tris[l.tri_first + tri];
        float root = sqrtf(u1);
        float b0 = 1.0f - root, b1 = root * u2;
        F3 p0 = f3(lt.p0[0], lt.p0[1], lt.p0[2]), p1 = f3(lt.p1[0], lt.p1[1], lt.p1[2]), p2 = f3(lt.p2[0], lt.p2[1], lt.p2[2]);
        F3 ns_local = normalize(cross(p1 - p0, p2 - p0));
        F3 ps_local = b0 * p0 + b1 * p1 + (1.0f - b0 - b1) * p2;
        F3 wi_local = normalize(ps_local - p_local);
        o.pdf = light_geoset_pdf(sc, l, p_local, wi_local);
        F3 ps = xf_point(l.m, ps_local);
        F3 ns = normalize(xf_normal(l.inv, ns_local));
        o.wi = normalize(ps - p);
        o.maxt = length(ps - p) - epsilon;
        o.L = dot(ns, -o.wi) > 0.0f ? color : f3(0, 0, 0);
        return;
    }
    F3 dir = f3(l.pos[0], l.pos[1], l.pos[2]) - p;
    o.wi = normalize(dir);
    o.pdf = 1.0f;
    float d2 = sqlen(dir);
    o.maxt = sqrtf(d2) - epsilon;
    if (l.type == GBL_LIGHT_SPOT) {
        float cos_t = dot(-o.wi, f3(l.axis[0], l.axis[1], l.axis[2]));
        float fall;
        if (cos_t < l.cos_max) {
            fall = 0.0f;
        } else if (cos_t > l.cos_falloff) {
            fall = 1.0f;
        } else {
            float dl = (cos_t - l.cos_max) / (l.cos_falloff - l.cos_max);
            fall = dl * dl * dl * dl;
        }
        o.L = div(fall * color, d2);
    } else {
        o.L = div(color, d2);
    }
}

// light->pdf(p, wi): 0 for delta lights, AreaLight::pdf otherwise (wi is NOT renormalised in light space)
template <bool EXT>
__device__ __forceinline__ float light_pdf(const DevScene& sc, const DevLight& l, F3 p, F3 wi) {
    if (EXT && l.type == GBL_LIGHT_IBL) return ibl_pdf(img_ctx(sc), &l, wi);
    if (l.type != GBL_LIGHT_AREA) return 0.0f;
    if (EXT && l.shape != 0u) return light_shape_pdf(l, xf_point(l.inv, p), xf_vector(l.inv, wi));
    return light_geoset_pdf(sc, l, xf_point(l.inv, p), xf_vector(l.inv, wi));
}

// Light::isDelta (GoblinLight.h:116, :296, :346): every light but the area and the image based ones
template <bool EXT>
__device__ __forceinline__ bool light_is_delta(const DevLight& l) {
    return l.type != GBL_LIGHT_AREA && !(EXT && l.type == GBL_LIGHT_IBL);
}
// light->Le(ray) of a ray that left the scene: Black but for an image based light (GoblinLight.h:76, GoblinLight.cpp:520-527)
template <bool EXT>
__device__ __forceinline__ F3 light_le_escaped(const DevScene& sc, const DevLight& l, F3 dir) {
    if (EXT && l.type == GBL_LIGHT_IBL) return ibl_le(img_ctx(sc), &l, dir);
    return f3(0.0f, 0.0f, 0.0f);
}
// Scene::evalEnvironmentLight (GoblinScene.cpp:89-95): the sum over every light
template <bool EXT>
__device__ __forceinline__ F3 environment_le(const DevScene& sc, F3 dir) {
    F3 L = f3(0.0f, 0.0f, 0.0f);
    if (EXT && sc.has_ibl != 0)
        for (int i = 0; i < sc.num_lights; ++i) {
            const F3 le = light_le_escaped<EXT>(sc, sc.lights[i], dir);
            L = f3(L.x + le.x, L.y + le.y, L.z + le.z);
        }
    return L;
}

// Intersection::Le(out): the hit instance's area light, one-sided
__device__ __forceinline__ F3 hit_Le(const DevScene& sc, int inst, F3 n, F3 out_dir) {
    int al = sc.instances[inst].area_light;
    if (al < 0) return f3(0, 0, 0);
    const DevLight& l = sc.lights[al];
    return dot(n, out_dir) > 0.0f ? f3(l.color[0], l.color[1], l.color[2]) : f3(0, 0, 0);
}
