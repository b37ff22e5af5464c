"""kernels/refmath.h restates glibc's float libm (sinf / cosf; expf / logf / log2f / powf; atanf / atan2f / tanf / acosf) so that
sampled directions, distances and lobes are the reference binary's bit for bit.  The header is host + device: here it is
compiled with g++ and checked against libm itself; the GPU suite runs the device build of the same code through
gbl_selftest_sincos / gbl_selftest_libm."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "%s/goblin_amd/csrc/kernels/refmath.h"
extern "C" {
// mismatches of the restatement against libm on n arguments spread over [lo, hi)
long refmath_mismatches(float lo, float hi, long n, int cosine) {
    long bad = 0;
    unsigned long long s = 88172645463325252ull;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        float x = lo + (hi - lo) * (static_cast<float>(s >> 40) / 16777216.0f);
        float a = cosine ? cosf(x) : sinf(x), b = cosine ? gbl_cosf(x) : gbl_sinf(x);
        if (!(a == b)) ++bad;
    }
    return bad;
}
void libm_sincos(const float* in, float* s, float* c, long n) {
    for (long i = 0; i < n; ++i) { s[i] = sinf(in[i]); c[i] = cosf(in[i]); }
}
static float libm_fn(int fn, float x, float y) {
    switch (fn) {
        case 0: return expf(x);
        case 1: return logf(x);
        case 2: return log2f(x);
        case 3: return powf(x, y);
        case 4: return atanf(x);
        case 5: return atan2f(x, y);
        case 6: return tanf(x);
        default: return acosf(x);
    }
}
static float gbl_fn(int fn, float x, float y) {
    switch (fn) {
        case 0: return gbl_expf(x);
        case 1: return gbl_logf(x);
        case 2: return gbl_log2f(x);
        case 3: return gbl_powf(x, y);
        case 4: return gbl_atanf(x);
        case 5: return gbl_atan2f(x, y);
        case 6: return gbl_tanf(x);
        default: return gbl_acosf(x);
    }
}
static bool same(float a, float b) { return memcmp(&a, &b, 4) == 0 || (a != a && b != b); }
// mismatches (bit patterns; any NaN equals any NaN) of restated fn against libm on the n argument pairs
long libm_mismatches(int fn, const float* a, const float* b, long n) {
    long bad = 0;
    for (long i = 0; i < n; ++i) bad += !same(libm_fn(fn, a[i], b[i]), gbl_fn(fn, a[i], b[i]));
    return bad;
}
void libm_eval(int fn, const float* a, const float* b, float* out, long n) {
    for (long i = 0; i < n; ++i) out[i] = libm_fn(fn, a[i], b[i]);
}
}
""" % REPO


@pytest.fixture(scope="session")
def refmath_lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("refmath")
    src = d / "refmath_check.cpp"
    src.write_text(SRC)
    so = d / "librefmath_check.so"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-o", str(so), str(src), "-lm"])
    lib = C.CDLL(str(so))
    lib.refmath_mismatches.argtypes = [C.c_float, C.c_float, C.c_long, C.c_int]
    lib.refmath_mismatches.restype = C.c_long
    lib.libm_sincos.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    lib.libm_sincos.restype = None
    lib.libm_mismatches.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    lib.libm_mismatches.restype = C.c_long
    lib.libm_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    lib.libm_eval.restype = None
    return lib


LIBM_FNS = {"expf": 0, "logf": 1, "log2f": 2, "powf": 3, "atanf": 4, "atan2f": 5, "tanf": 6, "acosf": 7}


def libm_arguments(name, n, seed=11):
    """Argument pairs over the ranges the renderer reaches, plus the special values."""
    rng = np.random.default_rng(seed)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 1e-38, 3.0, 88.5, -104.0, 1e30], np.float32)
    if name == "expf":
        a = np.concatenate([rng.uniform(-110, 90, n), rng.uniform(-1e-3, 1e-3, n // 8)])
    elif name in ("logf", "log2f"):
        a = np.concatenate([np.ldexp(rng.uniform(0.5, 1.5, n), rng.integers(-140, 128, n)), rng.uniform(0.5, 2.0, n // 4)])
    elif name == "powf":
        a = np.concatenate([rng.uniform(0, 1, n // 2), rng.uniform(1, 4, n // 4), np.ldexp(rng.uniform(0.5, 1, n // 4), rng.integers(-120, 120, n // 4))])
    elif name == "atanf":
        a = np.concatenate([np.ldexp(rng.uniform(-1, 1, n), rng.integers(-30, 30, n))])
    elif name == "atan2f":
        a = np.concatenate([rng.uniform(-8, 8, n // 2), np.ldexp(rng.uniform(-1, 1, n // 2), rng.integers(-70, 70, n // 2))])
    elif name == "tanf":
        a = np.concatenate([rng.uniform(-np.pi / 2, np.pi / 2, n // 2), rng.uniform(-119, 119, n // 4), np.ldexp(rng.uniform(-1, 1, n // 4), rng.integers(-30, 0, n // 4))])
    else:
        a = np.concatenate([rng.uniform(-1, 1, n // 2), np.ldexp(rng.uniform(-1, 1, n // 2), rng.integers(-40, 1, n // 2))])
    a = a.astype(np.float32)
    if name == "powf":
        b = np.concatenate([rng.uniform(0, 2000, a.size // 2), 1.0 / (1.0 + rng.uniform(0, 500, a.size - a.size // 2))]).astype(np.float32)
        rng.shuffle(b)
        b[::7] = 1.5
    elif name == "atan2f":
        b = np.concatenate([rng.uniform(-8, 8, a.size // 2), np.ldexp(rng.uniform(-1, 1, a.size - a.size // 2), rng.integers(-70, 70, a.size - a.size // 2))]).astype(np.float32)
        rng.shuffle(b)
    else:
        b = np.zeros_like(a)
    sa, sb = np.meshgrid(special, special)
    return np.ascontiguousarray(np.concatenate([a, sa.ravel()])), np.ascontiguousarray(np.concatenate([b, sb.ravel()]))


@pytest.mark.parametrize("name", sorted(LIBM_FNS))
def test_restated_libm_equals_libm(refmath_lib, name):
    a, b = libm_arguments(name, 8_000_000)
    assert refmath_lib.libm_mismatches(LIBM_FNS[name], a.ctypes.data, b.ctypes.data, a.size) == 0


@pytest.mark.parametrize("lo,hi", [(0.0, 6.2831855), (-6.2831855, 0.0), (0.0, 0.8), (6.0, 119.0), (0.0, 1e-3)])
def test_restated_sinf_cosf_equal_libm(refmath_lib, lo, hi):
    assert refmath_lib.refmath_mismatches(lo, hi, 10_000_000, 0) == 0
    assert refmath_lib.refmath_mismatches(lo, hi, 10_000_000, 1) == 0


@pytest.mark.gpu
def test_device_sinf_cosf_equal_libm(refmath_lib):
    import torch
    from goblin_amd import scene as gs
    from goblin_amd.renderer import HipPathTracer
    tr = HipPathTracer(gs.load_scene("bunny", gs.config_overrides(resolution=(16, 16), spp=1, depth=2)), 0)
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 4_000_000), rng.uniform(-7, 7, 1_000_000), rng.uniform(0, 119, 500_000),
                        rng.uniform(0, 1e-3, 100_000), [0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi]]).astype(np.float32)
    xd = torch.from_numpy(x).to(tr.device)
    sd, cd = torch.empty_like(xd), torch.empty_like(xd)
    st = tr.lib.gbl_selftest_sincos(tr.handle, xd.data_ptr(), sd.data_ptr(), cd.data_ptr(), x.size)
    assert st == 0
    s_ref, c_ref = np.empty_like(x), np.empty_like(x)
    refmath_lib.libm_sincos(x.ctypes.data, s_ref.ctypes.data, c_ref.ctypes.data, x.size)
    np.testing.assert_array_equal(sd.cpu().numpy(), s_ref)
    np.testing.assert_array_equal(cd.cpu().numpy(), c_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(LIBM_FNS))
def test_device_libm_equals_libm(refmath_lib, name):
    import torch
    from goblin_amd import scene as gs
    from goblin_amd.renderer import HipPathTracer
    tr = HipPathTracer(gs.load_scene("bunny", gs.config_overrides(resolution=(16, 16), spp=1, depth=2)), 0)
    a, b = libm_arguments(name, 2_000_000, seed=12)
    ad, bd = torch.from_numpy(a).to(tr.device), torch.from_numpy(b).to(tr.device)
    od = torch.empty_like(ad)
    assert tr.lib.gbl_selftest_libm(tr.handle, LIBM_FNS[name], ad.data_ptr(), bd.data_ptr(), od.data_ptr(), a.size) == 0
    ref = np.empty_like(a)
    refmath_lib.libm_eval(LIBM_FNS[name], a.ctypes.data, b.ctypes.data, ref.ctypes.data, a.size)
    np.testing.assert_array_equal(od.cpu().numpy().view(np.uint32) & np.where(np.isnan(ref), 0, 0xFFFFFFFF).astype(np.uint32),
                                  ref.view(np.uint32) & np.where(np.isnan(ref), 0, 0xFFFFFFFF).astype(np.uint32))
    assert np.array_equal(np.isnan(od.cpu().numpy()), np.isnan(ref))
