// MIPMap<T> lookups (image textures) and the image based light.
//
//   ImageBuffer<T>::texel            GoblinTexture.cpp:10-38     address modes, incl. the clamp mode's `t = clamp(s, ...)`
//   MIPMap<T>::lookup(level, s, t)   :276-291                     bilinear inside one level
//   lookupNearest / Bilinear / Trilinear / EWA   :99-258          level selection from the footprint, EWA over the ellipse
//   ImageBasedLight::Le / sampleL / pdf          GoblinLight.cpp:520-555, 615-628
//   CDF1D / CDF2D::sampleContinuous / pdf        GoblinSampler.cpp:334-405
//
// The pyramids are built on the host exactly as MIPMap's constructor builds them (csrc/host/mipmap.cpp) and uploaded as
// one float pool; a float texture's texel is one float, a colour texture's four.  T = float and T = Color run the same
// float operations per channel except where Color's operators differ from float's (Color /= s multiplies by 1 / s,
// GoblinColor.h:93-99): `is_float` selects.  One deviation: MIPMap::lookup clamps the level to [0, levels] and then
// indexes mPyramid[levels] out of bounds when the footprint is wider than the image (:278, :106-109); the device clamps
// to the last level.
#pragma once
#include "../device_scene.h"
#include "vecmath.h"

#define GBL_EWA_LUT_SIZE 128

// What the lookups read of the scene, BY VALUE: the out-of-line functions below must not take `const DevScene&` -- that
// would force the kernel's argument block into per-lane scratch and turn every scene pointer load of the whole kernel
// into a scratch access (measured: the EXT kernels ran 1.5x slower on scenes without a single image).
struct ImgCtx {
    const float* texels;
    const float* ewa_lut;
    const DevImage* images;
    const float* ibl_dist;
};
__device__ __forceinline__ ImgCtx img_ctx(const DevScene& sc) {
    ImgCtx c = {sc.texels, sc.ewa_lut, sc.images, sc.ibl_dist};
    return c;
}

__device__ __forceinline__ int img_floor(float f) { return static_cast<int>(floorf(f)); }
__device__ __forceinline__ int img_ceil(float f) { return static_cast<int>(ceilf(f)); }

// ImageBuffer<T>::texel of level `level`
__device__ __forceinline__ F3 image_texel(const ImgCtx sc, const DevImage& im, int level, int s, int t, uint32_t mode) {
    const int w = max(1, static_cast<int>(im.width >> level)), h = max(1, static_cast<int>(im.height >> level));
    if (mode == 1u) {   // AddressClamp, as written: t is clamped from s
        s = min(max(s, 0), w - 1);
        t = min(max(s, 0), h - 1);
    } else if (mode == 2u) {   // AddressBorder
        if (s < 0 || t < 0 || s >= w || t >= h) return f3(0.0f, 0.0f, 0.0f);
    } else {   // AddressRepeat
        s = s % w;
        t = t % h;
        if (s < 0) s += w;
        if (t < 0) t += h;
    }
    const float* p = sc.texels + im.offset + im.level_offset[level] + (static_cast<size_t>(t) * w + s) * im.channels;
    if (im.channels == 1u) {
        const float v = p[0];
        return f3(v, v, v);
    }
    const float4 q = *reinterpret_cast<const float4*>(p);
    return f3(q.x, q.y, q.z);
}

// MIPMap<T>::lookup(level, s, t, m)
__device__ __forceinline__ F3 mip_level(const ImgCtx sc, const DevImage& im, int level, float s, float t, uint32_t mode) {
    level = min(max(level, 0), static_cast<int>(im.levels) - 1);
    const int w = max(1, static_cast<int>(im.width >> level)), h = max(1, static_cast<int>(im.height >> level));
    const float s_res = s * w - 0.5f, t_res = t * h - 0.5f;
    const int s0 = img_floor(s_res), t0 = img_floor(t_res);
    const float ds = s_res - static_cast<float>(s0), dt = t_res - static_cast<float>(t0);
    return (1.0f - ds) * (1.0f - dt) * image_texel(sc, im, level, s0, t0, mode) + (ds) * (1.0f - dt) * image_texel(sc, im, level, s0 + 1, t0, mode) +
           (1.0f - ds) * (dt)*image_texel(sc, im, level, s0, t0 + 1, mode) + (ds) * (dt)*image_texel(sc, im, level, s0 + 1, t0 + 1, mode);
}

__device__ __forceinline__ F3 mip_trilinear(const ImgCtx sc, const DevImage& im, float s, float t, float width, uint32_t mode) {
    const int levels = static_cast<int>(im.levels);
    const float level = levels - 1 + gbl_ref_log2(fmaxf(width, 1e-8f));
    const int il = img_floor(level);
    if (il < 0) return mip_level(sc, im, 0, s, t, mode);
    if (il >= levels - 1) return mip_level(sc, im, levels - 1, s, t, mode);
    const float delta = level - static_cast<float>(il);
    return (1.0f - delta) * mip_level(sc, im, il, s, t, mode) + (delta)*mip_level(sc, im, il + 1, s, t, mode);
}

// MIPMap<T>::EWA
__device__ __forceinline__ F3 mip_ewa_level(const ImgCtx sc, const DevImage& im, bool is_float, int level, float s, float t, float A, float B, float C,
                                   uint32_t mode) {
    const int w = max(1, static_cast<int>(im.width >> level)), h = max(1, static_cast<int>(im.height >> level));
    const float s_res = static_cast<float>(w), t_res = static_cast<float>(h);
    s = s * w - 0.5f;
    t = t * h - 0.5f;
    A = A / (s_res * s_res);
    B = B / (s_res * t_res);
    C = C / (t_res * t_res);
    const float inv_det = 1.0f / (-B * B + 4.0f * A * C);
    const float off_s = 2.0f * sqrtf(C * inv_det), off_t = 2.0f * sqrtf(A * inv_det);
    const int s0 = img_ceil(s - off_s), s1 = img_floor(s + off_s), t0 = img_ceil(t - off_t), t1 = img_floor(t + off_t);
    float weight_sum = 0.0f;
    F3 result = f3(0.0f, 0.0f, 0.0f);
    for (int is = s0; is <= s1; ++is) {
        for (int it = t0; it <= t1; ++it) {
            const float ss = is - s, tt = it - t;
            const float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 <= 1.0f) {
                const int li = min(img_floor(r2 * GBL_EWA_LUT_SIZE), GBL_EWA_LUT_SIZE - 1);
                // EWALut[i] = expf(-2 r2_i) - expf(-2), r2_i = i / (EWA_LUT_SIZE - 1)   (initEWALut, :262-271; uploaded table)
                const float weight = sc.ewa_lut[li];
                const F3 tx = image_texel(sc, im, level, is, it, mode);
                result = f3(result.x + weight * tx.x, result.y + weight * tx.y, result.z + weight * tx.z);
                weight_sum += weight;
            }
        }
    }
    if (weight_sum > 0.0f) {
        if (is_float) return f3(result.x / weight_sum, result.y / weight_sum, result.z / weight_sum);   // float /= float
        const float inv = 1.0f / weight_sum;                                                             // Color /= float
        return f3(result.x * inv, result.y * inv, result.z * inv);
    }
    return image_texel(sc, im, level, static_cast<int>(s), static_cast<int>(t), mode);
}

struct TexCoord {
    float s, t, dsdx, dtdx, dsdy, dtdy;
};

// MIPMap<T>::lookup(tc, filter, address).  Out of line on purpose: a material evaluates up to four texture slots, each a
// graph two levels deep, in every kernel -- inlined, the EWA loops multiplied the kernels' code (and the build time) many
// times over for a path only image-textured scenes take.
__device__ __attribute__((noinline)) F3 mip_lookup(const ImgCtx sc, const DevImage* imp, bool is_float, const TexCoord tc, uint32_t filter, uint32_t mode, float max_aniso) {
    const DevImage& im = *imp;
    if (filter == 1u) {   // bilinear: one level, rounded
        const float width = fmaxf(fmaxf(fabsf(tc.dsdx), fabsf(tc.dtdx)), fmaxf(fabsf(tc.dsdy), fabsf(tc.dtdy)));
        const float level = static_cast<int>(im.levels) - 1 + gbl_ref_log2(fmaxf(width, 1e-8f));
        return mip_level(sc, im, img_floor(level + 0.5f), tc.s, tc.t, mode);
    }
    if (filter == 2u) {
        const float width = fmaxf(fmaxf(fabsf(tc.dsdx), fabsf(tc.dtdx)), fmaxf(fabsf(tc.dsdy), fabsf(tc.dtdy)));
        return mip_trilinear(sc, im, tc.s, tc.t, width, mode);
    }
    if (filter == 3u) {   // lookupEWA
        float ds0 = tc.dsdx, dt0 = tc.dtdx, ds1 = tc.dsdy, dt1 = tc.dtdy;
        float major = sqrtf(ds0 * ds0 + dt0 * dt0), minor = sqrtf(ds1 * ds1 + dt1 * dt1);
        if (major < minor) {
            float x = ds0; ds0 = ds1; ds1 = x;
            x = dt0; dt0 = dt1; dt1 = x;
            x = major; major = minor; minor = x;
        }
        if (minor * max_aniso < major && minor > 0.0f) {
            const float scale = major / (minor * max_aniso);
            minor *= scale;
            ds1 *= scale;
            dt1 *= scale;
        }
        float A = dt0 * dt0 + dt1 * dt1;
        float B = -2.0f * (ds0 * dt0 + ds1 * dt1);
        float C = ds0 * ds0 + ds1 * ds1;
        const float F = A * C - 0.25f * B * B;
        if (minor == 0.0f || F <= 0.0f) return mip_trilinear(sc, im, tc.s, tc.t, minor, mode);
        const float inv_f = 1.0f / F;
        A *= inv_f;
        B *= inv_f;
        C *= inv_f;
        const int levels = static_cast<int>(im.levels);
        const float level = levels - 1 + gbl_ref_log2(minor);
        const int il = img_floor(level);
        if (il < 0) return mip_level(sc, im, 0, tc.s, tc.t, mode);
        if (il >= levels - 1) return mip_level(sc, im, levels - 1, tc.s, tc.t, mode);
        const float delta = level - static_cast<float>(il);
        return (1.0f - delta) * mip_ewa_level(sc, im, is_float, il, tc.s, tc.t, A, B, C, mode) +
               (delta)*mip_ewa_level(sc, im, is_float, il + 1, tc.s, tc.t, A, B, C, mode);
    }
    return mip_level(sc, im, 0, tc.s, tc.t, mode);   // FilterNone: lookupNearest is the bilinear lookup of level 0
}

// ---------------------------------------------------------------------------------------------------------- the IBL
// CDF1D::sampleContinuous over cdf[0..n] (normalised) and func[0..n): std::lower_bound, then linear inside the cell
__device__ __forceinline__ float cdf1d_sample(const float* cdf, const float* func, float integral, int n, float u, float* pdf, int* index) {
    int lo = 0, len = n + 1;   // first element not less than u
    while (len > 0) {
        const int half = len >> 1;
        if (cdf[lo + half] < u) {
            lo += half + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    const int offset = max(0, lo - 1);
    const float d = (u - cdf[offset]) / (cdf[offset + 1] - cdf[offset]);
    *pdf = func[offset] / integral;
    *index = offset;
    return (static_cast<float>(offset) + d) / n;
}

// layout of an IBL's distribution in DevScene::ibl_dist (floats), W x H = the level the reference builds it from:
//   marginal: func[H], cdf[H + 1], integral        rows r: func[W], cdf[W + 1], integral
__device__ __forceinline__ const float* ibl_marginal(const ImgCtx sc, const DevLight& l) { return sc.ibl_dist + l.dist_offset; }
__device__ __forceinline__ const float* ibl_row(const ImgCtx sc, const DevLight& l, int row) {
    return sc.ibl_dist + l.dist_offset + (2 * l.dist_h + 2) + static_cast<size_t>(row) * (2 * l.dist_w + 2);
}

__device__ __forceinline__ float spherical_theta(F3 v) { return gbl_acosf(fminf(fmaxf(v.z, -1.0f), 1.0f)); }
__device__ __forceinline__ float spherical_phi(F3 v) {
    const float phi = gbl_atan2f(v.y, v.x);
    return phi < 0.0f ? phi + GBL_TWO_PI : phi;
}

// ImageBasedLight::Le(ray): mRadiance->lookup(0, phi / 2pi, theta / pi) of the direction in the light's frame
__device__ __attribute__((noinline)) F3 ibl_le(const ImgCtx sc, const DevLight* lp, F3 dir) {
    const DevLight& l = *lp;
    const F3 w = xf_vector(l.inv, dir);
    const float s = spherical_phi(w) * GBL_INV_TWOPI, t = spherical_theta(w) * GBL_INV_PI;
    return mip_level(sc, sc.images[l.image], 0, s, t, 0u);
}

// ImageBasedLight::sampleL
__device__ __attribute__((noinline)) float4 ibl_sample(const ImgCtx sc, const DevLight* lp, float u1, float u2, F3* wi_out) {
    const DevLight& l = *lp;
    F3 wi_v;
    float pdf_v;
    F3* wi = &wi_v;
    float* pdf = &pdf_v;
    const float* mg = ibl_marginal(sc, l);
    float pdf_row, pdf_col;
    int row, col;
    const float v = cdf1d_sample(mg + l.dist_h, mg, mg[2 * l.dist_h + 1], static_cast<int>(l.dist_h), u2, &pdf_row, &row);
    const float* rw = ibl_row(sc, l, row);
    const float u = cdf1d_sample(rw + l.dist_w, rw, rw[2 * l.dist_w + 1], static_cast<int>(l.dist_w), u1, &pdf_col, &col);
    const float pdf_st = pdf_row * pdf_col;
    const float theta = v * GBL_PI, phi = u * GBL_TWO_PI;
    const float cos_theta = gbl_cosf(theta), sin_theta = gbl_sinf(theta), cos_phi = gbl_cosf(phi), sin_phi = gbl_sinf(phi);
    *wi = xf_vector(l.m, f3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
    *pdf = pdf_st / (GBL_TWO_PI * GBL_PI * sin_theta);   // (the sinTheta == 0 guard above it is overwritten, :540-543)
    const F3 L = mip_level(sc, sc.images[l.image], 0, u, v, 0u);
    *wi_out = wi_v;
    return make_float4(L.x, L.y, L.z, pdf_v);   // radiance, pdf
}

// ImageBasedLight::pdf
__device__ __attribute__((noinline)) float ibl_pdf(const ImgCtx sc, const DevLight* lp, F3 wi) {
    const DevLight& l = *lp;
    const F3 w = xf_vector(l.inv, wi);
    const float theta = spherical_theta(w);
    const float sin_theta = gbl_sinf(theta);
    if (sin_theta == 0.0f) return 0.0f;
    const float phi = spherical_phi(w);
    const float u = phi * GBL_INV_TWOPI, v = theta * GBL_INV_PI;
    const float* mg = ibl_marginal(sc, l);
    const int h = static_cast<int>(l.dist_h), wd = static_cast<int>(l.dist_w);
    const int row = min(max(img_floor(h * v), 0), h - 1);
    const float* rw = ibl_row(sc, l, row);
    const int col = min(max(img_floor(wd * u), 0), wd - 1);
    const float integral = mg[2 * h + 1] * rw[2 * wd + 1];
    if (integral == 0.0f) return 0.0f;
    const float p = mg[row] * rw[col] / integral;
    return p / (GBL_TWO_PI * GBL_PI * sin_theta);
}
