// Host + device restatements of libm functions whose last bit matters for parity (see below).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef __HIPCC__
#define GBL_HD __host__ __device__ __forceinline__
// the table-driven double precision routines are called, not inlined: one copy per code object instead of one per call site
// (inlined they multiplied the build time of the kernels by seven)
#define GBL_HD_CALL __host__ __device__ __attribute__((noinline)) inline
#else
#define GBL_HD inline
#define GBL_HD_CALL inline
#endif

// sinf / cosf as the reference binary computes them: glibc's implementation (sysdeps/ieee754/flt-32/s_sinf.c,
// s_cosf.c, sincosf.h -- the ARM optimized-routines algorithm: the argument reduced by pi/2 in double, then one of two
// degree-7 / degree-8 double polynomials, the quadrant folded into sign and table choice), restated so that every
// sampled direction is the reference's bit for bit.  A device libm that is merely accurate differs from glibc in the
// last bit on a few per cent of the arguments; that is harmless per sample but flips a discrete outcome (a shadow test,
// a side of a surface) once in ~10^6 paths, and in the stream sampler a flipped path desynchronises the rest of its
// tile.  tests/test_refmath.py compiles this header for the host and checks it against libm itself on 2 x 10^7 arguments;
// the GPU suite does the same through gbl_selftest_sincos.
// Valid for |y| < 120 (the callers pass angles in [0, 2 pi]); larger arguments fall back to the device sinf / cosf.
GBL_HD float gbl_sincos_poly(double x, double x2, bool neg_cos, int n) {
    const double c0 = neg_cos ? -0x1p0 : 0x1p0, c1 = neg_cos ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2,
                 c2 = neg_cos ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5, c3 = neg_cos ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10,
                 c4 = neg_cos ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    // the multiply-adds are fused, as in the FMA build of these routines that glibc's ifunc selects on any x86-64
    // with FMA3 (sysdeps/x86_64/fpu/multiarch/s_sinf.c); the plain build differs from it on ~4e-7 of the arguments >= 6
    if ((n & 1) == 0) {
        double x3 = x * x2, t1 = fma(x2, s3, s2);
        double x7 = x3 * x2, s = fma(x3, s1, x);
        return static_cast<float>(fma(x7, t1, s));
    }
    double x4 = x2 * x2, k2 = fma(x2, c4, c3), k1 = fma(x2, c1, c0);
    double x6 = x4 * x2, c = fma(x4, c2, k1);
    return static_cast<float>(fma(x6, k2, c));
}
GBL_HD uint32_t gbl_abstop12(float x) { uint32_t u;
    memcpy(&u, &x, sizeof(u));
    return (u >> 20) & 0x7ffu; }
template <bool COS>
GBL_HD float gbl_ref_sincosf(float y) {
    double x = y;
    if (gbl_abstop12(y) < gbl_abstop12(0x1.921FB6p-1f)) {   // |y| < pi/4
        if (gbl_abstop12(y) < gbl_abstop12(0x1p-12f)) return COS ? 1.0f : y;
        return gbl_sincos_poly(x, x * x, false, COS ? 1 : 0);
    }
    if (!(gbl_abstop12(y) < gbl_abstop12(120.0f))) return COS ? cosf(y) : sinf(y);
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double r = x * hpi_inv;
    int n = (static_cast<int32_t>(r) + 0x800000) >> 24;
    x = fma(-static_cast<double>(n), hpi, x);
    const int q = COS ? n + 1 : n;
    const double sign = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;   // sign[4] = {1, -1, -1, 1}
    return gbl_sincos_poly(x * sign, x * x, (q & 2) != 0, COS ? n ^ 1 : n);
}
GBL_HD float gbl_sinf(float y) { return gbl_ref_sincosf<false>(y); }
GBL_HD float gbl_cosf(float y) { return gbl_ref_sincosf<true>(y); }

// expf / logf as the reference binary computes them: glibc's implementation (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c -- the
// ARM optimized-routines algorithms: a 32-entry 2^(i/32) table and a cubic in double for expf; a 16-entry (1/c, log c) table and
// a cubic in double for logf), restated with the library's own tables.  The medium's distance samples (GoblinRenderer.cpp:298-391),
// the BSSRDF's and the MIPMap's EWA weights pass through them; a device libm that is merely accurate moves a sample point by an
// ulp, and a shadow segment that ends exactly ON an emitter (epsilon 0) then flips its self-occlusion test.
// tests/test_refmath.py checks both against libm itself on the host; the GPU suite does through gbl_selftest_explog.
GBL_HD double gbl_asdouble(uint64_t u) {
    double d;
    memcpy(&d, &u, sizeof(d));
    return d;
}
GBL_HD uint64_t gbl_asuint64(double d) {
    uint64_t u;
    memcpy(&u, &d, sizeof(u));
    return u;
}
GBL_HD uint32_t gbl_asuint(float f) {
    uint32_t u;
    memcpy(&u, &f, sizeof(u));
    return u;
}
GBL_HD float gbl_asfloat(uint32_t u) {
    float f;
    memcpy(&f, &u, sizeof(f));
    return f;
}
#ifndef GBL_LIBM_PLAIN
#define GBL_FMA(a, b, c) fma((a), (b), (c))   // the FMA build glibc's ifunc selects on x86-64 with FMA3
#else
#define GBL_FMA(a, b, c) ((a) * (b) + (c))
#endif
// __exp2f_data: tab[i] = asuint64(2^(i/32)) - (i << 47)
GBL_HD uint64_t gbl_exp2f_tab(uint32_t i) {
    const uint64_t T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    return T[i];
}
GBL_HD_CALL float gbl_expf(float x) {
    const double shift = 0x1.8p+52, inv_ln2_n = 0x1.71547652b82fep+5;
    const double c0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, c1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, c2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double xd = static_cast<double>(x);
    const uint32_t abstop = (gbl_asuint(x) >> 20) & 0x7ffu;
    if (abstop >= ((gbl_asuint(88.0f) >> 20) & 0x7ffu)) {   // |x| >= 88 or nan
        if (gbl_asuint(x) == gbl_asuint(-INFINITY)) return 0.0f;
        if (abstop >= ((gbl_asuint(INFINITY) >> 20) & 0x7ffu)) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;      // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;         // underflow
    }
    double z = inv_ln2_n * xd;
    double kd = GBL_FMA(inv_ln2_n, xd, shift);      // z + shift, likewise
    const uint64_t ki = gbl_asuint64(kd);
    kd -= shift;
    const double r = GBL_FMA(inv_ln2_n, xd, -kd);   // z - kd, contracted in the FMA build
    uint64_t t = gbl_exp2f_tab(static_cast<uint32_t>(ki % 32));
    t += ki << (52 - 5);
    const double s = gbl_asdouble(t);
    z = GBL_FMA(c0, r, c1);
    const double r2 = r * r;
    double y = GBL_FMA(c2, r, 1.0);
    y = GBL_FMA(z, r2, y);
    y = y * s;
    return static_cast<float>(y);
}
GBL_HD_CALL float gbl_logf(float x) {
    const double T[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2},
    {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2},
    {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3},
    {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4},
    {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},
    {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},
    {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
    {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double ln2 = 0x1.62e42fefa39efp-1, a0 = -0x1.00ea348b88334p-2, a1 = 0x1.5575b0be00b6ap-2, a2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = gbl_asuint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {   // x < 0x1p-126 or inf or nan
        if (ix * 2u == 0u) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (x - x) / 0.0f;
        ix = gbl_asuint(x * 0x1p23f);   // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = static_cast<int>((tmp >> (23 - 4)) % 16u);
    const int k = static_cast<int32_t>(tmp) >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = T[i][0], logc = T[i][1];
    const double z = static_cast<double>(gbl_asfloat(iz));
    const double r = GBL_FMA(z, invc, -1.0);
    const double y0 = GBL_FMA(static_cast<double>(k), ln2, logc);
    const double r2 = r * r;
    double y = GBL_FMA(a1, r, a2);
    y = GBL_FMA(a0, r2, y);
    y = GBL_FMA(y, r2, y0 + r);
    return static_cast<float>(y);
}

// atanf / atan2f / tanf as the reference binary computes them.  In glibc 2.35 these are still the fdlibm float routines
// (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c, k_tanf.c): argument reduction to one of four intervals and a degree-11 odd /
// even split polynomial for atanf; the quadrant logic around it for atan2f; for tanf the kernel on [-pi/4, pi/4] with the
// double-precision range reduction of sincosf (e_rem_pio2f.c since 2.28: reduce_fast, split into a float pair).  Restated
// operation for operation in float; the equi-angular distance samples of the medium (Renderer::Lv) and the spherical texture
// mapping pass through them.  tanf is restated for |x| < 120 (the medium's angles lie in (-pi/2, pi/2)).
GBL_HD float gbl_atanf(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const float one = 1.0f, huge = 1.0e30f;
    const int32_t hx = static_cast<int32_t>(gbl_asuint(x));
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {   // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {   // |x| < 0.4375
        if (ix < 0x31000000 && huge + x > one) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) {
                id = 0;
                x = (2.0f * x - one) / (2.0f + x);
            } else {
                id = 1;
                x = (x - one) / (x + one);
            }
        } else if (ix < 0x401c0000) {
            id = 2;
            x = (x - 1.5f) / (one + 1.5f * x);
        } else {
            id = 3;
            x = -1.0f / x;
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -z : z;
}
GBL_HD_CALL float gbl_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = static_cast<int32_t>(gbl_asuint(x)), hy = static_cast<int32_t>(gbl_asuint(y));
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return gbl_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);   // 2 * sign(x) + sign(y)
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = gbl_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return gbl_asfloat(gbl_asuint(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}
GBL_HD float gbl_kernel_tanf(float x, float y, int iy) {
    const float T[13] = {3.3333334327e-01f, 1.3333334029e-01f, 5.3968254477e-02f, 2.1869488060e-02f, 8.8632395491e-03f, 3.5920790397e-03f, 1.4562094584e-03f,
                         5.8804126456e-04f, 2.4646313977e-04f, 7.8179444245e-05f, 7.1407252108e-05f, -1.8558637748e-05f, 2.5907305826e-05f};
    const float one = 1.0f, pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const int32_t hx = static_cast<int32_t>(gbl_asuint(x));
    const int32_t ix = hx & 0x7fffffff;
    if (ix < 0x39000000 && static_cast<int>(x) == 0) {   // |x| < 2^-13
        if ((ix | (iy + 1)) == 0) return one / fabsf(x);
        return iy == 1 ? x : -one / x;
    }
    if (ix >= 0x3f2ca140) {   // |x| >= 0.6744
        if (hx < 0) {
            x = -x;
            y = -y;
        }
        const float z0 = pio4 - x, w0 = pio4lo - y;
        x = z0 + w0;
        y = 0.0f;
        if (fabsf(x) < 0x1p-13f) return (1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
    }
    float z = x * x;
    float w = z * z;
    float r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
    float v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T[0] * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = static_cast<float>(iy);
        return static_cast<float>(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    // -1 / (x + r), accurately
    z = gbl_asfloat(gbl_asuint(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w;
    const float t = gbl_asfloat(gbl_asuint(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}
GBL_HD_CALL float gbl_tanf(float x) {
    const int32_t ix = static_cast<int32_t>(gbl_asuint(x)) & 0x7fffffff;
    if (ix <= 0x3f490fda) return gbl_kernel_tanf(x, 0.0f, 1);   // |x| <= pi / 4
    if (!(gbl_abstop12(x) < gbl_abstop12(120.0f))) return tanf(x);
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double dx = x;
    const double r = dx * hpi_inv;
    const int n = (static_cast<int32_t>(r) + 0x800000) >> 24;
    dx = dx - n * hpi;
    const float y0 = static_cast<float>(dx), y1 = static_cast<float>(dx - y0);
    return gbl_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}

// acosf: fdlibm's e_acosf.c (glibc 2.35 still ships it): a rational approximation of (asin(x) - x) / x^3 on |x| < 0.5 and the
// half-angle identities outside.  The spherical texture mapping and the image based light's direction -> (u, v) use it.
GBL_HD_CALL float gbl_acosf(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int32_t hx = static_cast<int32_t>(gbl_asuint(x));
    const int32_t ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {   // |x| < 0.5
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {   // x < -0.5
        const float z = (one + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrtf(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (one - x) * 0.5f;   // x > 0.5
    const float s = sqrtf(z);
    const float df = gbl_asfloat(gbl_asuint(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}

// log2f / powf: glibc's e_log2f.c and e_powf.c (the optimized-routines algorithms again): log2 through the 16-entry
// (1/c, log2 c) table and a polynomial in double, powf = exp2(y * log2 x) through the 2^(i/32) table of expf.  The MIPMap's
// level selection takes log2f of the filter width; the Blinn / Phong lobes and the Henyey-Greenstein phase function take powf.
GBL_HD double gbl_log2f_tab(uint32_t i, int col) {
    const double T[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
    return T[i][col];
}
GBL_HD_CALL float gbl_log2f(float x) {
    const double a0 = -0x1.712b6f70a7e4dp-2, a1 = 0x1.ecabf496832e0p-2, a2 = -0x1.715479ffae3dep-1, a3 = 0x1.715475f35c8b8p+0;
    uint32_t ix = gbl_asuint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {   // x < 0x1p-126 or inf or nan
        if (ix * 2u == 0u) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (x - x) / 0.0f;
        ix = gbl_asuint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> (23 - 4)) % 16u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = static_cast<int32_t>(tmp) >> 23;
    const double invc = gbl_log2f_tab(i, 0), logc = gbl_log2f_tab(i, 1);
    const double z = static_cast<double>(gbl_asfloat(iz));
    const double r = GBL_FMA(z, invc, -1.0);
    const double y0 = logc + static_cast<double>(k);
    const double r2 = r * r;
    double y = GBL_FMA(a1, r, a2);
    y = GBL_FMA(a0, r2, y);
    const double p = GBL_FMA(a3, r, y0);
    y = GBL_FMA(y, r2, p);
    return static_cast<float>(y);
}
GBL_HD int gbl_powf_checkint(uint32_t iy) {   // 0: not an integer, 1: odd, 2: even
    const int e = static_cast<int>(iy >> 23 & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
GBL_HD bool gbl_powf_zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }
// (inlined: the Blinn lobe of the base kernels calls this one -- a call, even one never taken, costs the register allocation of
// the persistent kernel 1 % of its run time)
GBL_HD float gbl_powf_inline(float x, float y) {
    const double b0 = 0x1.27616c9496e0bp-2, b1 = -0x1.71969a075c67ap-2, b2 = 0x1.ec70a6ca7baddp-2, b3 = -0x1.7154748bef6c8p-1, b4 = 0x1.71547652ab82bp+0;
    const double c0 = 0x1.c6af84b912394p-5, c1 = 0x1.ebfce50fac4f3p-3, c2 = 0x1.62e42ff0c52d6p-1, shift = 0x1.8p+52 / 32;
    uint32_t sign_bias = 0;
    uint32_t ix = gbl_asuint(x);
    const uint32_t iy = gbl_asuint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || gbl_powf_zeroinfnan(iy)) {
        if (gbl_powf_zeroinfnan(iy)) {
            if (2u * iy == 0u) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (gbl_powf_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && gbl_powf_checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {   // finite x < 0
            const int yint = gbl_powf_checkint(iy);
            if (yint == 0) return (x - x) / 0.0f;
            if (yint == 1) sign_bias = 1u << (5 + 11);
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {   // subnormal
            ix = gbl_asuint(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> (23 - 4)) % 16u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = static_cast<int32_t>(top) >> 23;
    const double invc = gbl_log2f_tab(i, 0), logc = gbl_log2f_tab(i, 1);
    const double z = static_cast<double>(gbl_asfloat(iz));
    const double r = GBL_FMA(z, invc, -1.0);
    const double y0 = logc + static_cast<double>(k);
    const double r2 = r * r;
    double yy = GBL_FMA(b0, r, b1);
    const double p = GBL_FMA(b2, r, b3);
    const double r4 = r2 * r2;
    double q = GBL_FMA(b4, r, y0);
    q = GBL_FMA(p, r2, q);
    yy = GBL_FMA(yy, r4, q);
    const double ylogx = static_cast<double>(y) * yy;
    if ((gbl_asuint64(ylogx) >> 47 & 0xffffu) >= gbl_asuint64(126.0) >> 47) {   // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -INFINITY : INFINITY;
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
    }
    // exp2_inline
    double kd = ylogx + shift;
    const uint64_t ki = gbl_asuint64(kd);
    kd -= shift;
    const double rr = ylogx - kd;
    uint64_t t = gbl_exp2f_tab(static_cast<uint32_t>(ki % 32));
    const uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    const double s = gbl_asdouble(t);
    const double zz = GBL_FMA(c0, rr, c1);
    const double rr2 = rr * rr;
    double e = GBL_FMA(c2, rr, 1.0);
    e = GBL_FMA(zz, rr2, e);
    e = e * s;
    return static_cast<float>(e);
}
GBL_HD_CALL float gbl_powf(float x, float y) { return gbl_powf_inline(x, y); }
// Goblin::log2 (GoblinUtils.h:84-87) -- the reference's namespace shadows libm's log2 with logf(n) * (1.0f / logf(2.0f)); the
// MIPMap's level selection goes through this one.  logf(2.0f) = 0x1.62e43p-1 in glibc.
GBL_HD float gbl_ref_log2(float n) {
    const float inv_log2 = 1.0f / 0x1.62e43p-1f;
    return gbl_logf(n) * inv_log2;
}
