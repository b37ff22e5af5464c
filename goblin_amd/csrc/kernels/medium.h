// The participating medium around the camera ray: RenderTask::run adds  w * (tr * L + Lv)  to the tile
// (GoblinRenderer.cpp:40-47) with tr = Renderer::transmittance and Lv = Renderer::Lv (:298-455, homogeneous branch)
// over a HomogeneousVolumeRegion (GoblinVolume.cpp:12-36, GoblinVolume.h:72-112), both evaluated on the camera ray
// clipped at the first surface (Ray::maxt is mutable: Li's first scene query shrinks it).
//
// Neither term feeds the integrator, so they run as a pass of their own (vol_kernel: one lane per camera sample,
// 2 float4 per sample); the splat folds them into every sample it filters (wf_apply_medium) -- for every integrator
// and both schedules -- and vol_combine_kernel does the same to the caller's li_out afterwards.
//
// Random numbers.  The reference draws them straight from the tile's generator AFTER the sample's Li (9 per light
// sample: pick, LightSample, equi-angular u, distance u, LightSample).  The native and replay samplers take them from
// a hash of the sample's image position instead (VolRand below; oracle/goblin_oracle.cpp restates the same rule);
// the stream sampler's tile walk feeds the generator's own outputs (path_trace_kernel<.., STREAM>, which includes this
// file; vol_kernel itself is in volume.h).
#pragma once
#include "bssrdf.h"
#include "shade.h"
#include "stream.h"
#include "trace.h"

struct VolRand {
    const uint32_t* raw;   // stream sampler: this sample's raw generator outputs, consumed in order; null: hashed
    uint32_t key, i;
    __device__ __forceinline__ float f() {
        const uint32_t n = i++;
        return raw ? stream_u01(raw[n]) : nat_u01(nat_mix(key, 0x766f6c00u + n));
    }
};
__device__ __forceinline__ VolRand vol_rand_hashed(float image_x, float image_y) {
    VolRand r;
    r.raw = nullptr;
    r.key = nat_mix(__float_as_uint(image_x), __float_as_uint(image_y));
    r.i = 0u;
    return r;
}

// BBox::intersect(ray, &tMin, &tMax), GoblinBBox.cpp:57-77, on the region's own box with the ray moved into its space
__device__ __forceinline__ bool vol_intersect(const DevVolume& v, F3 o, F3 d, float mint, float maxt, float* tmin, float* tmax) {
    const F3 lo_ = xf_point(v.inv, o), ld = xf_vector(v.inv, d);
    const float oo[3] = {lo_.x, lo_.y, lo_.z}, dd[3] = {ld.x, ld.y, ld.z};
    float t0 = mint, t1 = maxt;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float inv = 1.0f / dd[i];
        float tn = (v.lo[i] - oo[i]) * inv, tf = (v.hi[i] - oo[i]) * inv;
        if (tn > tf) {
            const float tmp = tn;
            tn = tf;
            tf = tmp;
        }
        t0 = (tn > t0) ? tn : t0;
        t1 = (tf < t1) ? tf : t1;
        if (t0 > t1) return false;
    }
    *tmin = t0;
    *tmax = t1;
    return true;
}
__device__ __forceinline__ bool vol_contains(const DevVolume& v, F3 p) {
    const F3 q = xf_point(v.inv, p);
    return v.lo[0] <= q.x && q.x <= v.hi[0] && v.lo[1] <= q.y && q.y <= v.hi[1] && v.lo[2] <= q.z && q.z <= v.hi[2];
}
// VolumeGrid::getVoxel / eval (GoblinVolume.cpp:148-196): trilinear interpolation of the density grid at a point of the
// region's own space.  (As written there the index scale is  normalize * n - 0.5: the half-cell shift multiplies.)
__device__ __forceinline__ F3 grid_voxel(const DevVolume& v, const float* density, int x, int y, int z) {
    if (x < 0 || x >= v.nx || y < 0 || y >= v.ny || z < 0 || z >= v.nz) return f3(0.0f, 0.0f, 0.0f);
    if (v.nch == 1) {
        const float c = density[z * v.nx * v.ny + y * v.nx + x];
        return f3(c, c, c);
    }
    const int off = 3 * (z * v.nx * v.ny + y * v.nx + x);
    return f3(density[off], density[off + 1], density[off + 2]);
}
__device__ __forceinline__ F3 f3_lerp(float t, F3 a, F3 b) { return (1.0f - t) * a + t * b; }   // lerp<T>, GoblinUtils.h:109-112
__device__ __noinline__ F3 grid_eval(const DevVolume& v, const float* density, F3 p_local) {
    F3 f = f3(p_local.x - v.lo[0], p_local.y - v.lo[1], p_local.z - v.lo[2]);
    f.x *= v.normalize[0] * v.nx - 0.5f;
    f.y *= v.normalize[1] * v.ny - 0.5f;
    f.z *= v.normalize[2] * v.nz - 0.5f;
    const int ix = static_cast<int>(floorf(f.x)), iy = static_cast<int>(floorf(f.y)), iz = static_cast<int>(floorf(f.z));
    const float dx = f.x - ix, dy = f.y - iy, dz = f.z - iz;
    const F3 d00 = f3_lerp(dx, grid_voxel(v, density, ix, iy, iz), grid_voxel(v, density, ix + 1, iy, iz));
    const F3 d10 = f3_lerp(dx, grid_voxel(v, density, ix, iy + 1, iz), grid_voxel(v, density, ix + 1, iy + 1, iz));
    const F3 d01 = f3_lerp(dx, grid_voxel(v, density, ix, iy, iz + 1), grid_voxel(v, density, ix + 1, iy, iz + 1));
    const F3 d11 = f3_lerp(dx, grid_voxel(v, density, ix, iy + 1, iz + 1), grid_voxel(v, density, ix + 1, iy + 1, iz + 1));
    return f3_lerp(dz, f3_lerp(dy, d00, d10), f3_lerp(dy, d01, d11));
}
// HeterogeneousVolumeRegion::getAttenuation (:313-321)
__device__ __forceinline__ F3 hetero_attenuation(const DevScene& sc, F3 p) {
    const DevVolume& v = sc.volume;
    const F3 q = xf_point(v.inv, p);
    const bool inside = v.lo[0] <= q.x && q.x <= v.hi[0] && v.lo[1] <= q.y && q.y <= v.hi[1] && v.lo[2] <= q.z && q.z <= v.hi[2];
    return inside ? grid_eval(v, sc.vol_density, q) : f3(0.0f, 0.0f, 0.0f);
}
// HomogeneousVolumeRegion::transmittance (GoblinVolume.cpp:25-36): Beer's law over the segment inside the box, no random
// number.  HeterogeneousVolumeRegion::transmittance (:323-341): a jittered ray march -- one random number -- and BLACK,
// not white, for a ray that misses the region.
__device__ __forceinline__ F3 vol_transmittance(const DevScene& sc, F3 o, F3 d, float mint, float maxt, VolRand& rnd) {
    const DevVolume& v = sc.volume;
    float tmin, tmax;
    if (v.hetero != 0u) {
        if (!vol_intersect(v, o, d, mint, maxt, &tmin, &tmax)) return f3(0.0f, 0.0f, 0.0f);
        const float step = v.step;
        float t = tmin;
        const float jitter = rnd.f() * step;
        F3 tau = jitter * hetero_attenuation(sc, o + t * d);
        t += jitter;
        while (t + step < tmax) {
            const F3 a = step * hetero_attenuation(sc, o + t * d);
            tau = f3(tau.x + a.x, tau.y + a.y, tau.z + a.z);
            t += step;
        }
        const F3 a = (tmax - t) * hetero_attenuation(sc, o + t * d);
        tau = f3(tau.x + a.x, tau.y + a.y, tau.z + a.z);
        return f3(gbl_expf(-tau.x), gbl_expf(-tau.y), gbl_expf(-tau.z));
    }
    if (!vol_intersect(v, o, d, mint, maxt, &tmin, &tmax)) return f3(1.0f, 1.0f, 1.0f);
    const float len = length((o + tmax * d) - (o + tmin * d));
    const F3 tau = len * f3(v.attenuation[0], v.attenuation[1], v.attenuation[2]);
    return f3(gbl_expf(-tau.x), gbl_expf(-tau.y), gbl_expf(-tau.z));
}

// Light::samplePosition (GoblinLight.cpp:101-108, 161-175, 239-244, 396-409)
__device__ __forceinline__ F3 light_sample_position(const DevScene& sc, const DevLight& l, float u_comp, float u1, float u2) {
    if (l.type == GBL_LIGHT_AREA) {
        F3 p_local;
        if (l.shape == 1u) {          // Sphere::sample(u1, u2, &n): uniformSampleSphere * radius
            const float z = 1.0f - 2.0f * u1;
            const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - z * z));
            const float phi = GBL_TWO_PI * u2;
            p_local = l.radius * f3(sin_t * gbl_cosf(phi), sin_t * gbl_sinf(phi), z);
        } else if (l.shape == 2u) {   // Disk::sample
            float x, y;
            uniform_sample_disk(u1, u2, &x, &y);
            p_local = f3(l.radius * x, l.radius * y, 0.0f);
        } else {                      // GeometrySet::sample(ls, &n): a triangle by area, a uniform point on it
            uint32_t tri = 0;
            for (uint32_t k = 0; k < l.tri_count; ++k)
                if (sc.light_tris[l.tri_first + k].cdf_hi < u_comp) tri = k + 1;
            if (tri >= l.tri_count) tri = l.tri_count - 1;
            const DevLightTri& lt = sc.light_tris[l.tri_first + tri];
            const float root = sqrtf(u1);
            const float b0 = 1.0f - root, b1 = root * u2;
            const F3 p0 = f3(lt.p0[0], lt.p0[1], lt.p0[2]), p1 = f3(lt.p1[0], lt.p1[1], lt.p1[2]), p2 = f3(lt.p2[0], lt.p2[1], lt.p2[2]);
            p_local = b0 * p0 + b1 * p1 + (1.0f - b0 - b1) * p2;
        }
        return xf_point(l.m, p_local);
    }
    if (l.type == GBL_LIGHT_DIRECTIONAL) {   // a disc of the scene's bounding sphere, pushed back along the light's direction
        const DevVolume& v = sc.volume;
        const F3 z = f3(l.axis[0], l.axis[1], l.axis[2]);
        F3 x, y;
        coordinate_axes(z, &x, &y);
        float dx, dy;
        uniform_sample_disk(u1, u2, &dx, &dy);
        const F3 disk = f3(v.bound_center[0], v.bound_center[1], v.bound_center[2]) + v.bound_radius * (dx * x + dy * y);
        return disk - z * v.bound_radius;
    }
    if (l.type == GBL_LIGHT_IBL) {   // ImageBasedLight::samplePosition (GoblinLight.cpp:556-568): a point of the scene's bounding sphere
        const DevVolume& v = sc.volume;
        const float z = 1.0f - 2.0f * u1;
        const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - z * z));
        const float phi = GBL_TWO_PI * u2;
        return f3(v.bound_center[0], v.bound_center[1], v.bound_center[2]) + v.bound_radius * f3(sin_t * gbl_cosf(phi), sin_t * gbl_sinf(phi), z);
    }
    return f3(l.pos[0], l.pos[1], l.pos[2]);
}

// Scene::occluded for a light sample taken from a point INSIDE the medium: sampleL is called with epsilon 0, so the
// shadow segment ends exactly ON the emitter and whether the emitter's own triangle counts as an occluder is decided by
// the last bit -- in the reference, first of all by its BVH's strict box test in front of that triangle's leaf
// (tMin < maxt, GoblinBVH.cpp:156-187; an emitter quad's bound is flat, so tMin IS the hit distance).  The closest hit
// tells the two cases apart: anything nearer than the emitter occludes; the emitter itself only if the reference would
// reach its leaf (ref_leaf_reached, trace.h).
template <bool STATS, class STK>
__device__ __forceinline__ bool vol_shadow_occluded(const DevScene& sc, int li, F3 p, F3 wi, float maxt, const STK& stk, LaneCounters& cnt) {
    Hit h;
    if (!trace<false, STATS, true>(sc, p, wi, 0.0f, maxt, stk, h, cnt)) return false;
    const DevInstance& in = sc.instances[h.inst];
    if (in.area_light != li) return true;
    // the hit is on the sampled emitter itself: the reference's box tests on the way to it -- the instance's world bound in
    // the scene BVH (a disk's is flat: a segment that ends on it enters the box AT maxt and the strict test culls it, found
    // by the round-2 fuzz scenes), then, for a mesh, the triangle's bound in the instance's space
    const DevInstanceBound wb = sc.instance_bounds[h.inst];
    if (!ref_box_reached(f3(wb.lo[0], wb.lo[1], wb.lo[2]), f3(wb.hi[0], wb.hi[1], wb.hi[2]), p, wi, 0.0f, maxt)) return false;
    if (in.shape != 0u || sc.tri_order == nullptr) return true;
    return ref_leaf_reached(sc, h.tri, xf_point(in.inv, p), xf_vector(in.inv, wi), 0.0f, maxt);
}

// Renderer::Lv, heterogeneous branch (GoblinRenderer.cpp:397-445): march the camera ray in steps of step_size from a jittered
// start; at every sample point one light sample (pick + LightSample: 4 random numbers, and one more for the shadow ray's
// own jittered transmittance when it is unoccluded).
template <bool STATS, class STK>
__device__ __forceinline__ F3 volume_lv_hetero(const DevScene& sc, F3 o, F3 d, float tmin, float tmax, VolRand& rnd, const STK& stk, LaneCounters& cnt) {
    const DevVolume& vol = sc.volume;
    const F3 albedo = f3(vol.albedo[0], vol.albedo[1], vol.albedo[2]);
    F3 Lv = f3(0, 0, 0);
    const float step = vol.step;
    F3 p_prev = o + tmin * d;
    float t = tmin + step * rnd.f();
    F3 p = o + t * d;
    F3 transmittance = f3(1.0f, 1.0f, 1.0f);
    while (t <= tmax) {
        const F3 sigma_t = hetero_attenuation(sc, p);   // HeterogeneousVolumeRegion::eval (:297-311)
        const F3 sigma_s = sigma_t * albedo;
        const F3 tau = sigma_t * length(p - p_prev);
        transmittance = transmittance * f3(gbl_expf(-tau.x), gbl_expf(-tau.y), gbl_expf(-tau.z));
        {   // Lv += transmittance * emission, emission = Black
            const F3 e = transmittance * f3(0.0f, 0.0f, 0.0f);
            Lv = f3(Lv.x + e.x, Lv.y + e.y, Lv.z + e.z);
        }
        const float pick = rnd.f();
        int li = -1;
        float pick_pdf = 0.0f;
        if (sc.num_lights != 0) {
            li = 0;
            for (int k = 1; k <= sc.num_lights; ++k)
                if (sc.light_cdf[k] < pick) li = k;
            if (li >= sc.num_lights) li = sc.num_lights - 1;
            pick_pdf = sc.light_pick_pdf[li];
        }
        if (li >= 0 && pick_pdf != 0.0f) {
            const DevLight& light = sc.lights[li];
            const float u_comp = rnd.f(), u1 = rnd.f(), u2 = rnd.f();   // LightSample ls(rng)
            LightSampleOut ls;
            light_sample<true>(sc, light, p, 0.0f, u_comp, u1, u2, ls);
            if (!is_black(ls.L) && ls.pdf > 0.0f) {
                if (STATS) cnt.shadow += 1;
                if (!vol_shadow_occluded<STATS>(sc, li, p, ls.wi, ls.maxt, stk, cnt)) {
                    const F3 tr_light = vol_transmittance(sc, p, ls.wi, 0.0f, ls.maxt, rnd);
                    const F3 Ld = div(tr_light * ls.L, pick_pdf * ls.pdf);
                    const float phase = vol_contains(vol, p) ? phase_hg(d, ls.wi, vol.g) : 0.0f;   // VolumeRegion::phase
                    const F3 term = transmittance * sigma_s * phase * Ld;
                    Lv = f3(Lv.x + term.x, Lv.y + term.y, Lv.z + term.z);
                }
            }
        }
        t += step;
        p_prev = p;
        p = o + t * d;
    }
    return step * Lv;
}

// Renderer::Lv, homogeneous branch (GoblinRenderer.cpp:298-391).  (o, d, mint, maxt): the camera ray after Li.
template <bool STATS, class STK>
__device__ __forceinline__ F3 volume_lv(const DevScene& sc, F3 o, F3 d, float mint, float maxt, VolRand& rnd, const STK& stk, LaneCounters& cnt) {
    const DevVolume& vol = sc.volume;
    float tmin, tmax;
    if (!vol_intersect(vol, o, d, mint, maxt, &tmin, &tmax)) return f3(0, 0, 0);
    if ((tmax - tmin) < 1e-5f) return f3(0, 0, 0);
    if (vol.hetero != 0u) return volume_lv_hetero<STATS>(sc, o, d, tmin, tmax, rnd, stk, cnt);
    const F3 att = f3(vol.attenuation[0], vol.attenuation[1], vol.attenuation[2]), sca = f3(vol.scatter[0], vol.scatter[1], vol.scatter[2]);
    const F3 zero = f3(0, 0, 0);
    F3 Lv = f3(0, 0, 0);
    for (int i = 0; i < vol.sample_num; ++i) {
        const float pick = rnd.f();
        if (sc.num_lights == 0) continue;
        int li = 0;
        for (int k = 1; k <= sc.num_lights; ++k)
            if (sc.light_cdf[k] < pick) li = k;
        if (li >= sc.num_lights) li = sc.num_lights - 1;
        const float pick_pdf = sc.light_pick_pdf[li];
        if (pick_pdf == 0.0f) continue;
        const DevLight& light = sc.lights[li];
        const float e_comp = rnd.f(), e_u1 = rnd.f(), e_u2 = rnd.f();   // LightSample lsEqui(rng)
        const F3 p_light = light_sample_position(sc, light, e_comp, e_u1, e_u2);
        const float delta = dot(p_light - o, d);
        const float a = tmin - delta, b = tmax - delta;
        const float D = length(p_light - (o + delta * d));
        const float theta_a = gbl_atan2f(a, D), theta_b = gbl_atan2f(b, D);
        const float ue = rnd.f();
        const float te = D * gbl_tanf((1 - ue) * theta_a + ue * theta_b);           // equiAngularSample
        const float pdf_te = D / ((theta_b - theta_a) * (D * D + te * te));    // equiAngularPdf
        const F3 p_e = o + (delta + te) * d;
        const bool in_e = vol_contains(vol, p_e);
        const F3 sigma_te = in_e ? att : zero, scatter_e = in_e ? sca : zero;
        const F3 tr_e = f3(gbl_expf(-sigma_te.x * (te - a)), gbl_expf(-sigma_te.y * (te - a)), gbl_expf(-sigma_te.z * (te - a)));
        {
            LightSampleOut ls;
            light_sample<true>(sc, light, p_e, 0.0f, e_comp, e_u1, e_u2, ls);
            if (!is_black(ls.L) && ls.pdf > 0.0f) {
                if (STATS) cnt.shadow += 1;
                if (!vol_shadow_occluded<STATS>(sc, li, p_e, ls.wi, ls.maxt, stk, cnt)) {
                    const F3 tr_light = vol_transmittance(sc, p_e, ls.wi, 0.0f, ls.maxt, rnd);
                    const F3 Ld = div(tr_light * ls.L, pick_pdf * ls.pdf);
                    const float phase = in_e ? phase_hg(d, ls.wi, vol.g) : 0.0f;   // VolumeRegion::phase
                    const float sig = sss_luminance(sigma_te);
                    const float pdf_td = sig / (gbl_expf(sig * (te - a)) - gbl_expf(sig * (te - b)));   // exponentialPdf(t, sigma, a, b)
                    const float mis = power_heuristic(pdf_te, pdf_td);
                    const F3 t = div(mis * tr_e * scatter_e * phase * Ld, pdf_te);
                    Lv = f3(Lv.x + t.x, Lv.y + t.y, Lv.z + t.z);
                }
            }
        }
        // distance sampling
        const F3 sigma_td = vol_contains(vol, o + (0.5f * (tmin + tmax)) * d) ? att : zero;
        const float ud = rnd.f();
        const float sig_d = sss_luminance(sigma_td);
        const float td = a - gbl_logf(1.0f - ud * (1.0f - gbl_expf(sig_d * (a - b)))) / sig_d;   // exponentialSample(u, sigma, a, b)
        const float pdf_td = sig_d / (gbl_expf(sig_d * (td - a)) - gbl_expf(sig_d * (td - b)));
        const F3 p_d = o + (delta + td) * d;
        const F3 tr_d = f3(gbl_expf(-sigma_td.x * (td - a)), gbl_expf(-sigma_td.y * (td - a)), gbl_expf(-sigma_td.z * (td - a)));
        const bool in_d = vol_contains(vol, p_d);
        const F3 scatter_d = in_d ? sca : zero;
        const float d_comp = rnd.f(), d_u1 = rnd.f(), d_u2 = rnd.f();   // LightSample lsDistance(rng)
        {
            LightSampleOut ls;
            light_sample<true>(sc, light, p_d, 0.0f, d_comp, d_u1, d_u2, ls);
            if (!is_black(ls.L) && ls.pdf > 0.0f) {
                if (STATS) cnt.shadow += 1;
                if (!vol_shadow_occluded<STATS>(sc, li, p_d, ls.wi, ls.maxt, stk, cnt)) {
                    const F3 tr_light = vol_transmittance(sc, p_d, ls.wi, 0.0f, ls.maxt, rnd);
                    const F3 Ld = div(tr_light * ls.L, pick_pdf * ls.pdf);
                    const float phase = in_d ? phase_hg(d, ls.wi, vol.g) : 0.0f;
                    const float pdf_te2 = D / ((theta_b - theta_a) * (D * D + td * td));
                    const float mis = power_heuristic(pdf_td, pdf_te2);
                    const F3 t = div(mis * tr_d * scatter_d * phase * Ld, pdf_td);
                    Lv = f3(Lv.x + t.x, Lv.y + t.y, Lv.z + t.z);
                }
            }
        }
    }
    return div(Lv, static_cast<float>(vol.sample_num));
}

