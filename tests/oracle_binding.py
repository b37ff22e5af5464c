"""TEST INFRASTRUCTURE: ctypes face of oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Nothing under goblin_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_HARNESS = os.path.join(ORACLE_DIR, "_ref", "ref_harness")


class orc_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "closest_queries", "anyhit_queries", "filtered_queries", "nodes",
                                          "tris", "splats", "dims")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def build_oracle():
    """Compile liboracle.so if it is missing or stale (g++, a few seconds)."""
    src = os.path.join(ORACLE_DIR, "goblin_oracle.cpp")
    hdr = os.path.join(os.path.dirname(ORACLE_DIR), "include", "goblin_hip.h")   # the ABI structs (and version) it is compiled against
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        from goblin_amd import _abi
        L = C.CDLL(ORACLE_SO)
        L.orc_create.argtypes = [C.POINTER(_abi.gbl_scene_desc)]
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_destroy.restype = None
        L.orc_sample_window.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_sample_window.restype = None
        L.orc_sample_dimension.argtypes = [C.POINTER(_abi.gbl_render_setting)]
        L.orc_sample_dimension_scene.argtypes = [C.c_void_p, C.POINTER(_abi.gbl_render_setting)]
        L.orc_pt_offsets.argtypes = [C.POINTER(_abi.gbl_render_setting), C.POINTER(C.c_int32)]
        L.orc_pt_offsets.restype = None
        L.orc_glibc_rand.argtypes = [C.POINTER(C.c_int32), C.c_int32]
        L.orc_glibc_rand.restype = None
        L.orc_filter_table.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_filter_table.restype = None
        L.orc_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_camera_ray.restype = None
        L.orc_light_power.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_occluded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        L.orc_li_replay.argtypes = [C.c_void_p, C.POINTER(_abi.gbl_render_setting), C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int32, C.POINTER(orc_counters)]
        L.orc_splat.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_native_samples.argtypes = [C.c_void_p, C.POINTER(_abi.gbl_render_setting), C.c_uint64,
                                         C.POINTER(C.c_int32), C.c_void_p]
        L.orc_li_native.argtypes = [C.c_void_p, C.POINTER(_abi.gbl_render_setting), C.c_uint64, C.POINTER(C.c_int32), C.c_int32,
                                    C.c_void_p, C.c_int32]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(_abi.gbl_render_setting), C.c_int32, C.c_int32, C.c_int32,
                                 C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                                 C.POINTER(orc_counters)]
        L.orc_hardware_threads.restype = C.c_int32
        L.orc_mip_lookup.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        L.orc_mip_lookup.restype = C.c_int32
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """The CPU restatement bound to one loaded scene (goblin_amd.scene.Scene)."""

    def __init__(self, scene):
        self.scene = scene  # keeps the host arrays alive
        self.h = lib().orc_create(scene.desc_ptr)
        if not self.h:
            raise RuntimeError("orc_create failed")

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            lib().orc_destroy(h)

    # -- facts ---------------------------------------------------------------
    def window(self):
        w = (C.c_int32 * 4)()
        lib().orc_sample_window(self.h, w)
        return tuple(w)

    def dims(self, setting=None):
        return lib().orc_sample_dimension_scene(self.h, C.byref(setting or self.scene.desc.setting))

    def pt_offsets(self, setting=None):
        s = setting or self.scene.desc.setting
        out = (C.c_int32 * (5 * max(1, s.max_ray_depth)))()
        lib().orc_pt_offsets(C.byref(s), out)
        return np.array(out, dtype=np.int32).reshape(-1, 5)

    # -- known-answer probes -------------------------------------------------
    def filter_table(self):
        out = np.zeros(256, np.float32)
        lib().orc_filter_table(self.h, _ptr(out))
        return out

    def camera_ray(self, x, y):
        out = np.zeros(8, np.float32)
        lib().orc_camera_ray(self.h, x, y, _ptr(out))
        return out

    def mip_lookup(self, image, queries, filter, mode, max_aniso=10.0, is_float=False):
        """MIPMap<T>::lookup of the scene's image `image` for n x {s, t, dsdx, dtdx, dsdy, dtdy} -> n x rgba."""
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, 6)
        out = np.zeros((q.shape[0], 4), np.float32)
        if lib().orc_mip_lookup(self.h, image, int(is_float), _ptr(q), q.shape[0], filter, mode, max_aniso, _ptr(out)) != 0:
            raise RuntimeError("orc_mip_lookup failed")
        return out

    def light_power(self):
        out = np.zeros(4 * max(1, self.scene.desc.num_lights), np.float32)
        n = lib().orc_light_power(self.h, _ptr(out))
        return out[:4 * n].reshape(n, 4)

    def intersect(self, o, d, mint=1e-3, maxt=-1.0):
        o = np.asarray(o, np.float32)
        d = np.asarray(d, np.float32)
        out = np.zeros(12, np.float32)
        hit = lib().orc_intersect(self.h, _ptr(o), _ptr(d), mint, maxt, _ptr(out))
        return out if hit else None

    def occluded(self, o, d, mint=1e-3, maxt=-1.0):
        o = np.asarray(o, np.float32)
        d = np.asarray(d, np.float32)
        return bool(lib().orc_occluded(self.h, _ptr(o), _ptr(d), mint, maxt))

    # -- integrator ----------------------------------------------------------
    def li_replay(self, samples, setting=None, threads=1):
        s = setting or self.scene.desc.setting
        samples = np.ascontiguousarray(samples, np.float32)
        n = samples.shape[0]
        assert samples.shape[1] == self.dims(s), (samples.shape, self.dims(s))
        li = np.zeros((n, 4), np.float32)
        cnt = orc_counters()
        lib().orc_li_replay(self.h, C.byref(s), _ptr(samples), n, _ptr(li), threads, C.byref(cnt))
        return li, cnt.as_dict()

    def splat(self, samples, li, film=None):
        d = self.scene.desc
        if film is None:
            film = np.zeros((d.film.yres, d.film.xres, 4), np.float32)
        samples = np.ascontiguousarray(samples, np.float32)
        li = np.ascontiguousarray(li, np.float32)
        lib().orc_splat(self.h, _ptr(samples), samples.shape[1], _ptr(li), samples.shape[0], _ptr(film))
        return film

    def native_samples(self, seed, window=None, setting=None):
        s = setting or self.scene.desc.setting
        w = window or self.window()
        spp = int(np.ceil(np.sqrt(np.float32(s.sample_per_pixel)))) ** 2
        n = (w[1] - w[0]) * (w[3] - w[2]) * spp
        out = np.zeros((n, self.dims(s)), np.float32)
        lib().orc_native_samples(self.h, C.byref(s), seed, (C.c_int32 * 4)(*w), _ptr(out))
        return out

    def li_native(self, seed, rr=False, window=None, setting=None, threads=4):
        """Li of the native sampler's records for a window, pixel-major (the device's li order); rr: with the device's Russian
        roulette extension restated (same counter-based kill draw, same 1 / q placement)."""
        s = setting or self.scene.desc.setting
        w = window or self.window()
        spp = int(np.ceil(np.sqrt(np.float32(s.sample_per_pixel)))) ** 2
        li = np.zeros(((w[1] - w[0]) * (w[3] - w[2]) * spp, 4), np.float32)
        lib().orc_li_native(self.h, C.byref(s), seed, (C.c_int32 * 4)(*w), 1 if rr else 0, _ptr(li), threads)
        return li

    def render(self, setting=None, threads=1, ref_faithful=0, sampler=0, seed=0, want_samples=False):
        """The reference's whole render loop.  Returns dict(film, seconds, counters[, samples, li])."""
        s = setting or self.scene.desc.setting
        d = self.scene.desc
        film = np.zeros((d.film.yres, d.film.xres, 4), np.float32)
        w = self.window()
        spp = int(np.ceil(np.sqrt(np.float32(s.sample_per_pixel)))) ** 2
        n = (w[1] - w[0]) * (w[3] - w[2]) * spp
        samples = np.zeros((n, self.dims(s)), np.float32) if want_samples else None
        li = np.zeros((n, 4), np.float32) if want_samples else None
        sec = C.c_double()
        cnt = orc_counters()
        lib().orc_render(self.h, C.byref(s), threads, ref_faithful, sampler, seed, _ptr(film), _ptr(samples), _ptr(li),
                         C.byref(sec), C.byref(cnt))
        out = {"film": film, "seconds": sec.value, "counters": cnt.as_dict()}
        if want_samples:
            out["samples"] = samples
            out["li"] = li
        return out


def hardware_threads():
    return lib().orc_hardware_threads()


def normalize_film(film):
    """Film::writeImage's rgb/weight (GoblinFilm.cpp:164-172); weight-0 pixels -> 0."""
    w = film[..., 3:4]
    with np.errstate(divide="ignore", invalid="ignore"):
        rgb = film[..., :3] * (np.float32(1.0) / w)
    return np.where(w > 0, rgb, 0).astype(np.float32)
