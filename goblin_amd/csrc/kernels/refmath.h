// Host + device restatements of libm functions whose last bit matters for parity (see below).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef __HIPCC__
#define GBL_HD __host__ __device__ __forceinline__
#else
#define GBL_HD inline
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
