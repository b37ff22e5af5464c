// Subsurface scattering at the camera hit: Renderer::Lsubsurface = LbssrdfSingle +
// LbssrdfDiffusion (GoblinRenderer.cpp:128-296) over the BSSRDF of a
// SubsurfaceMaterial (GoblinMaterial.cpp:32-220, GoblinMaterial.h:61-116).
//
// PathTracer::Li adds the term once, right after the emission of the first hit and
// with throughput 1 (GoblinPathtracer.cpp:66-70).  It depends on nothing the bounce
// loop computes, so it runs as its own pass AHEAD of the path kernels: sss_kernel
// writes one float4 per camera sample (zero where the first hit carries no BSSRDF)
// and the path kernels add it at bounce 0, in the reference's order
// (Li = Le; Li += Lsubsurface; then the bounce terms).  Scenes without a subsurface
// material never launch it.  (sss_kernel itself: subsurface.h.  The stream sampler's
// tile walk, whose records only exist pixel by pixel, evaluates the same functions
// inline at the first hit instead.)
//
// Every random number comes from the camera sample's BSSRDF block
// (BSSRDFSampleIndex / BSSRDFSample, GoblinLight.cpp:35-61), n = RenderArgs::bssrdf_n
// slots per pattern.
#pragma once
#include "../device_scene.h"
#include "sampler.h"
#include "shade.h"
#include "trace.h"
#include "vecmath.h"

__device__ __forceinline__ F3 sss_div(F3 a, F3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }   // Color / Color
__device__ __forceinline__ F3 sss_sqrt(F3 c) { return f3(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z)); }
__device__ __forceinline__ F3 sss_exp(F3 c) { return f3(gbl_expf(c.x), gbl_expf(c.y), gbl_expf(c.z)); }
__device__ __forceinline__ float sss_clamp0(float f) { return f < 0.0f ? 0.0f : (f > INFINITY ? INFINITY : f); }
__device__ __forceinline__ float sss_luminance(F3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }

// `m` is the subsurface material resolved at the fragment in question: color = sigma_a, color2 = sigma_s', k = g
__device__ __forceinline__ F3 bssrdf_scatter(const DevMaterial& m) { return div(f3(m.color2[0], m.color2[1], m.color2[2]), 1.0f - m.k); }
__device__ __forceinline__ F3 bssrdf_attenuation(const DevMaterial& m) { return bssrdf_scatter(m) + f3(m.color[0], m.color[1], m.color[2]); }
__device__ __forceinline__ F3 bssrdf_sigma_tr(const DevMaterial& m) {
    F3 sigma_a = f3(m.color[0], m.color[1], m.color[2]), sigma_sp = f3(m.color2[0], m.color2[1], m.color2[2]);
    F3 sigma_tp = sigma_a + sigma_sp;
    return sss_sqrt(3.0f * sigma_a * sigma_tp);
}
// BSSRDF::Rd, the dipole diffusion profile (GoblinMaterial.cpp:60-81)
__device__ __forceinline__ F3 bssrdf_rd(const DevMaterial& m, float d2) {
    const float A = m.exponent;
    F3 sigma_a = f3(m.color[0], m.color[1], m.color[2]), sigma_sp = f3(m.color2[0], m.color2[1], m.color2[2]);
    F3 sigma_tp = sigma_a + sigma_sp;
    F3 sigma_tr = sss_sqrt(3.0f * sigma_a * sigma_tp);
    F3 one = f3(1.0f, 1.0f, 1.0f);
    F3 zr = sss_div(one, sigma_tp);
    F3 zv = zr * (1.0f + 4.0f / 3.0f * A);
    F3 dd = f3(d2, d2, d2);
    F3 dr = sss_sqrt(zr * zr + dd);
    F3 dv = sss_sqrt(zv * zv + dd);
    F3 alpha_p = sss_div(sigma_sp, sigma_tp);
    F3 s_dr = sigma_tr * dr, s_dv = sigma_tr * dv;
    F3 rd = (0.25f * GBL_INV_PI) * alpha_p *
            (sss_div(zr * (one + s_dr) * sss_exp(-s_dr), dr * dr * dr) + sss_div(zv * (one + s_dv) * sss_exp(-s_dv), dv * dv * dv));
    return f3(sss_clamp0(rd.x), sss_clamp0(rd.y), sss_clamp0(rd.z));
}
// Henyey-Greenstein (GoblinVolume.h:126-134); `g < 1e-3` in double there selects the same floats as `g < 1e-3f`
__device__ __forceinline__ float phase_hg(F3 wi, F3 wo, float g) {
    if (g < 1e-3f) return 0.25f * GBL_INV_PI;
    float cos_theta = dot(wi, wo);
    return 0.25f * GBL_INV_PI * (1.0f - g * g) / gbl_powf(1.0f + g * g - 2.0f * g * cos_theta, 1.5f);
}
// Goblin::specularRefract(wo, n, etai, etat), GoblinMaterial.cpp:418-434
__device__ __forceinline__ F3 refract_dir(F3 wo, F3 n, float etai, float etat) {
    float eta = etai / etat;
    float cosi = absdot(n, wo);
    return normalize(n * (eta * cosi - sqrtf(fmaxf(0.0f, 1.0f - eta * eta * (1.0f - cosi * cosi)))) - eta * wo);
}
// gaussianSample2DPdf on the disc of radius rmax (GoblinSampler.h:194-204) and of a point projected on the plane
// through `center` with normal N (GoblinSampler.cpp:645-657)
__device__ __forceinline__ float gaussian_pdf_2d(float x, float y, float falloff, float rmax) {
    return (GBL_INV_PI * falloff * gbl_expf(-falloff * (x * x + y * y))) / (1.0f - gbl_expf(-falloff * rmax * rmax));
}
__device__ __forceinline__ float gaussian_pdf_proj(F3 center, F3 sample, F3 N, float falloff, float rmax) {
    F3 d = sample - center;
    F3 projected = d - N * dot(d, N);
    return (GBL_INV_PI * falloff * gbl_expf(-falloff * sqlen(projected))) / (1.0f - gbl_expf(-falloff * rmax * rmax));
}

#define SSS_U_AXIS 0
#define SSS_V_AXIS 1
#define SSS_N_AXIS 2
// BSSRDF::MISWeight (GoblinMaterial.cpp:83-127): the three probe axes are picked 1 : 1 : 2 (U : V : N)
__device__ __forceinline__ float bssrdf_mis_weight(const Frag& fo, const TexFrag& to, F3 pwi, F3 ni, int axis, float pdf, float sigma_tr,
                                                   float rmax) {
    const F3 pwo = fo.p;
    if (axis == SSS_N_AXIS) {
        F3 u = normalize(to.dpdu), v = normalize(to.dpdv);
        float u_pdf = 0.25f * gaussian_pdf_proj(pwo, pwi, u, sigma_tr, rmax) * absdot(u, ni);
        float v_pdf = 0.25f * gaussian_pdf_proj(pwo, pwi, v, sigma_tr, rmax) * absdot(v, ni);
        float num = 4 * pdf * pdf;
        return num / (num + u_pdf * u_pdf + v_pdf * v_pdf);
    }
    if (axis == SSS_U_AXIS) {
        F3 n = fo.n, v = normalize(to.dpdv);
        float n_pdf = 0.5f * gaussian_pdf_proj(pwo, pwi, n, sigma_tr, rmax) * absdot(n, ni);
        float v_pdf = 0.25f * gaussian_pdf_proj(pwo, pwi, v, sigma_tr, rmax) * absdot(v, ni);
        float num = pdf * pdf;
        return num / (4 * n_pdf * n_pdf + num + v_pdf * v_pdf);
    }
    F3 n = fo.n, u = normalize(to.dpdu);
    float n_pdf = 0.5f * gaussian_pdf_proj(pwo, pwi, n, sigma_tr, rmax) * absdot(n, ni);
    float u_pdf = 0.25f * gaussian_pdf_proj(pwo, pwi, u, sigma_tr, rmax) * absdot(u, ni);
    float num = pdf * pdf;
    return num / (4 * n_pdf * n_pdf + u_pdf * u_pdf + num);
}

struct SssSample {   // BSSRDFSample(sample, index, n), GoblinLight.cpp:53-61
    float ls_comp, ls_geo0, ls_geo1, pick_light, pick_axis, disc0, disc1, single;
};
// Slot i of the camera sample's BSSRDF block.  Record layout (SampleQuota, GoblinSampler.cpp:23-58): the 1D patterns
// ls / pickLight / pickAxis / singleScatter and the 2D patterns ls / disc follow the integrator's own -- 3 D and 2 D
// per-bounce ones under the path tracer, 2 per light slot + pickLight and 2 per light slot under Whitted
// (RenderArgs::sss_off1 / sss_off2, sss_pat1 / sss_pat2).
template <bool REPLAY>
__device__ __forceinline__ SssSample sss_sample(const RenderArgs& ra, const SampleSource& src, uint32_t i) {
    SssSample s;
    const uint32_t n = static_cast<uint32_t>(ra.bssrdf_n), n2 = static_cast<uint32_t>(ra.bssrdf_n2);
    if (REPLAY) {
        const float* r1 = src.rec + ra.sss_off1;
        const float* r2 = src.rec + ra.sss_off2;
        s.ls_comp = r1[i];
        s.pick_light = r1[n + i];
        s.pick_axis = r1[2 * n + i];
        s.single = r1[3 * n + i];
        s.ls_geo0 = r2[2 * i];
        s.ls_geo1 = r2[2 * i + 1];
        s.disc0 = r2[2 * n2 + 2 * i];
        s.disc1 = r2[2 * n2 + 2 * i + 1];
    } else {
        s.ls_comp = src.native_1d_n(ra.sss_pat1 + 0u, n, i);
        s.pick_light = src.native_1d_n(ra.sss_pat1 + 1u, n, i);
        s.pick_axis = src.native_1d_n(ra.sss_pat1 + 2u, n, i);
        s.single = src.native_1d_n(ra.sss_pat1 + 3u, n, i);
        src.native_2d_slot(0x10000u + ra.sss_pat2, n2, i, &s.ls_geo0, &s.ls_geo1);
        src.native_2d_slot(0x10000u + ra.sss_pat2 + 1u, n2, i, &s.disc0, &s.disc1);
    }
    return s;
}

// Scene::sampleLight: CDF1D::sampleDiscrete over the power distribution
__device__ __forceinline__ int sss_pick_light(const DevScene& sc, float u, float* pdf) {
    int li = 0;
    for (int i = 1; i <= sc.num_lights; ++i)
        if (sc.light_cdf[i] < u) li = i;
    if (li >= sc.num_lights) li = sc.num_lights - 1;
    *pdf = sc.light_pick_pdf[li];
    return li;
}

// the subsurface material `material` resolved at a fragment that has seen no ray differentials (PathTracer::Li runs
// Lsubsurface before computeUVDifferential) or, KEEP, with the ones the caller computed (WhittedRenderer::Li: after)
template <bool KEEP = false>
__device__ __forceinline__ void sss_resolve(const DevScene& sc, int material, const Frag& fr, TexFrag& tf, DevMaterial& out) {
    if (!KEEP) uv_differential(fr, tf, false, fr.p, fr.p, fr.p, fr.p);
    resolve_material(sc, sc.materials[material], fr, tf, out);
}

// Renderer::LbssrdfSingle, GoblinRenderer.cpp:128-204
template <bool REPLAY, bool STATS, class STK>
__device__ __forceinline__ F3 l_bssrdf_single(const DevScene& sc, const RenderArgs& ra, const SampleSource& src, const Frag& fr,
                                               const DevMaterial& mo, int material, F3 wo, const STK& stk, LaneCounters& cnt) {
    const F3 pwo = fr.p, no = fr.n;
    const float coso = absdot(wo, fr.n);
    const float eta = mo.index;
    const float Ft = 1.0f - fresnel_dielectric(coso, 1.0f, eta);
    const F3 scatter = bssrdf_scatter(mo);
    const F3 sigma_t = bssrdf_attenuation(mo);
    const float falloff = sss_luminance(sigma_t);
    const F3 wo_refract = refract_dir(wo, no, 1.0f, eta);
    F3 Ls = f3(0, 0, 0);
    for (int i = 0; i < ra.bssrdf_n; ++i) {
        const SssSample bs = sss_sample<REPLAY>(ra, src, static_cast<uint32_t>(i));
        if (STATS) cnt.dims += 8;
        const float d = -gbl_logf(bs.single) / falloff;              // exponentialSample
        const F3 p_sample = pwo + d * wo_refract;
        const float sample_pdf = falloff * gbl_expf(-falloff * d);   // exponentialPdf
        float pick_pdf;
        const int li = sss_pick_light(sc, bs.pick_light, &pick_pdf);
        LightSampleOut ls;
        light_sample<true>(sc, sc.lights[li], p_sample, 1e-5f, bs.ls_comp, bs.ls_geo0, bs.ls_geo1, ls);
        if (is_black(ls.L) || ls.pdf == 0.0f) continue;
        Hit wh;
        if (STATS) cnt.ext += 1;
        if (!trace<false, STATS, true>(sc, p_sample, ls.wi, 1e-5f, ls.maxt, stk, wh, cnt)) continue;
        if (sc.instances[wh.inst].material != material) continue;   // getBSSRDF() == bssrdf: the same material object
        Frag fwi;
        TexFrag twi;
        make_fragment<true>(sc, wh, p_sample, ls.wi, fwi, &twi);
        if (STATS) cnt.shadow += 1;
        if (trace<true, STATS, true>(sc, p_sample, ls.wi, wh.t + fwi.eps, ls.maxt, stk, wh, cnt)) continue;
        DevMaterial mi;
        sss_resolve(sc, material, fwi, twi, mi);
        const F3 ni = fwi.n;
        const float ph = phase_hg(ls.wi, wo_refract, mo.k);
        const float cosi = absdot(ni, ls.wi);
        const float Fti = 1.0f - fresnel_dielectric(cosi, 1.0f, eta);
        const F3 sigma_ti = bssrdf_attenuation(mi);
        const float G = absdot(ni, wo_refract) / cosi;
        const F3 sigma_tc = sigma_t + G * sigma_ti;
        const float di = length(fwi.p - p_sample);
        const float et = 1.0f / eta;
        const float di_prime = di * absdot(ls.wi, ni) / sqrtf(1.0f - et * et * (1.0f - cosi * cosi));
        const F3 term = div(sss_div((Ft * Fti * ph) * scatter, sigma_tc) * sss_exp(-di_prime * sigma_ti) * sss_exp(-d * sigma_t) * ls.L,
                            ls.pdf * pick_pdf * sample_pdf);
        Ls = Ls + term;
    }
    return div(Ls, static_cast<float>(ra.bssrdf_n));
}

// Renderer::LbssrdfDiffusion, GoblinRenderer.cpp:206-274
template <bool REPLAY, bool STATS, class STK>
__device__ __forceinline__ F3 l_bssrdf_diffusion(const DevScene& sc, const RenderArgs& ra, const SampleSource& src, const Frag& fr,
                                                  const TexFrag& tf, const DevMaterial& mo, int material, F3 wo, const STK& stk,
                                                  LaneCounters& cnt) {
    const F3 pwo = fr.p;
    const float coso = absdot(wo, fr.n);
    const float eta = mo.index;
    const float Ft = 1.0f - fresnel_dielectric(coso, 1.0f, eta);
    const float sigma_tr = sss_luminance(bssrdf_sigma_tr(mo));
    const float skip_ratio = 0.01f;
    const float rmax = sqrtf(gbl_logf(skip_ratio) / -sigma_tr);
    F3 Lm = f3(0, 0, 0);
    for (int i = 0; i < ra.bssrdf_n; ++i) {
        const SssSample bs = sss_sample<REPLAY>(ra, src, static_cast<uint32_t>(i));
        // BSSRDF::sampleProbeRay (GoblinMaterial.cpp:129-164): a gaussian disc sample, probed along N, U or V (2 : 1 : 1)
        const float r = sqrtf(gbl_logf(1.0f - bs.disc0 * (1.0f - gbl_expf(-sigma_tr * rmax * rmax))) / -sigma_tr);
        const float theta = GBL_TWO_PI * bs.disc1;
        const float sx = r * gbl_cosf(theta), sy = r * gbl_sinf(theta);
        const float half_len = sqrtf(rmax * rmax - (sx * sx + sy * sy));
        F3 po, pd;
        int axis;
        float disc_pdf;
        if (bs.pick_axis <= 0.5f) {
            po = pwo + shade_to_world(fr, f3(sx, sy, -half_len));
            pd = fr.n;
            axis = SSS_N_AXIS;
            disc_pdf = 0.5f;
        } else if (bs.pick_axis <= 0.75f) {
            po = pwo + shade_to_world(fr, f3(-half_len, sx, sy));
            pd = normalize(tf.dpdu);
            axis = SSS_U_AXIS;
            disc_pdf = 0.25f;
        } else {
            po = pwo + shade_to_world(fr, f3(sy, -half_len, sx));
            pd = normalize(tf.dpdv);
            axis = SSS_V_AXIS;
            disc_pdf = 0.25f;
        }
        disc_pdf *= gaussian_pdf_2d(sx, sy, sigma_tr, rmax);
        Hit ph;
        if (STATS) cnt.ext += 1;
        if (!trace<false, STATS, true>(sc, po, pd, 0.0f, 2.0f * half_len, stk, ph, cnt)) continue;
        if (sc.instances[ph.inst].material != material) continue;
        Frag pf;
        TexFrag pt;
        make_fragment<true>(sc, ph, po, pd, pf, &pt);
        DevMaterial mp;
        sss_resolve(sc, material, pf, pt, mp);
        const F3 p_probe = pf.p;
        const F3 Rd = bssrdf_rd(mp, sqlen(p_probe - pwo));
        float pick_pdf;
        const int li = sss_pick_light(sc, bs.pick_light, &pick_pdf);
        const F3 ni = pf.n;
        LightSampleOut ls;
        light_sample<true>(sc, sc.lights[li], p_probe, pf.eps, bs.ls_comp, bs.ls_geo0, bs.ls_geo1, ls);
        if (is_black(ls.L) || ls.pdf == 0.0f) continue;
        if (STATS) cnt.shadow += 1;
        Hit dummy;
        if (trace<true, STATS, true>(sc, p_probe, ls.wi, pf.eps, ls.maxt, stk, dummy, cnt)) continue;
        const float cosi = absdot(ni, ls.wi);
        const F3 irradiance = div(ls.L * cosi, ls.pdf * pick_pdf);
        const float Fti = 1.0f - fresnel_dielectric(cosi, 1.0f, eta);
        const float pdf = disc_pdf * absdot(pd, ni);
        const float w = bssrdf_mis_weight(fr, tf, p_probe, ni, axis, pdf, sigma_tr, rmax);
        Lm = Lm + div((w * GBL_INV_PI * Ft * Fti) * Rd * irradiance, pdf);
    }
    return div(Lm, static_cast<float>(ra.bssrdf_n));
}

