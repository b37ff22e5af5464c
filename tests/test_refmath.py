"""kernels/refmath.h restates glibc's sinf / cosf so that sampled directions are the reference binary's bit for bit.
The header is host + device: here it is compiled with g++ and checked against libm itself; the GPU suite runs the
device build of the same code through gbl_selftest_sincos."""
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
    return lib


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
