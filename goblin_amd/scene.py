"""Scene loading: Goblin JSON (+OBJ) -> flattened ``gbl_scene_desc``.

Thin Python face of libgoblin_host.so, the host-side mirror of the reference's
``ContextLoader::load`` (/root/reference/src/GoblinContextLoader.cpp:447-504).
"""
import copy
import ctypes as C
import json
import os

from . import _abi

SCENE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")


def scene_path(name):
    """Path of a bundled scene (``bunny``, ``cornell``, ``grid`` ...)."""
    if os.path.exists(name):
        return name
    p = os.path.join(SCENE_DIR, name if name.endswith(".json") else name + ".json")
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


class Scene:
    """A loaded scene: owns the host arrays the description points into."""

    def __init__(self, handle, source):
        self._handle = handle
        self.source = source
        self.desc_ptr = _abi.host_lib().gbl_host_desc(handle)
        self.desc = self.desc_ptr.contents

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _abi.host_lib().gbl_host_free(h)
            except Exception:
                pass

    @property
    def setting(self):
        return self.desc.setting

    @property
    def film(self):
        return self.desc.film

    def sample_window(self):
        """Film::getSampleRange: (x0, x1, y0, y1), the film padded by the filter radius."""
        out = (C.c_int32 * 4)()
        _abi.host_lib().gbl_host_sample_window(C.byref(self.desc.film), out)
        return tuple(out)

    def spp(self):
        """Samples per pixel the sampler really takes: roundToSquare(sample_per_pixel)."""
        return _abi.host_lib().gbl_host_round_to_square(self.desc.setting.sample_per_pixel)

    def sample_dimension(self):
        return _abi.host_lib().gbl_host_sample_dimension_scene(C.byref(self.desc), C.byref(self.desc.setting))

    def num_paths(self):
        x0, x1, y0, y1 = self.sample_window()
        return (x1 - x0) * (y1 - y0) * self.spp()


def load_scene_text(text, scene_dir):
    lib = _abi.host_lib()
    handle = C.c_void_p()
    st = lib.gbl_host_load_string(text.encode(), os.fsencode(scene_dir), C.byref(handle))
    if st != _abi.GBL_OK:
        raise _abi.GoblinError(st, lib.gbl_host_last_error().decode())
    return Scene(handle, text)


def load_scene(name_or_path, overrides=None):
    """Load a scene file; ``overrides`` is a dict deep-merged over the JSON
    (e.g. ``{"render_setting": {"sample_per_pixel": 16}, "camera": {"film": {"resolution": [256, 256]}}}``)."""
    path = scene_path(name_or_path)
    if not overrides:   # ContextLoader::load on the file itself (also yields the reference's default output path)
        lib = _abi.host_lib()
        handle = C.c_void_p()
        st = lib.gbl_host_load_file(os.fsencode(path), C.byref(handle))
        if st != _abi.GBL_OK:
            raise _abi.GoblinError(st, lib.gbl_host_last_error().decode())
        return Scene(handle, path)
    with open(path) as f:
        doc = json.load(f)
    _merge(doc, overrides)
    return load_scene_text(json.dumps(doc), os.path.dirname(os.path.abspath(path)))


def config_overrides(resolution=None, spp=None, depth=None, method=None, ao_samples=None, filter=None):
    """Convenience builder for the usual BASELINE config overrides."""
    o = {}
    rs = {}
    if spp is not None:
        rs["sample_per_pixel"] = int(spp)
    if depth is not None:
        rs["max_ray_depth"] = int(depth)
    if method is not None:
        rs["render_method"] = method
    if ao_samples is not None:
        rs["ao_sample_num"] = int(ao_samples)
    if rs:
        o["render_setting"] = rs
    cam = {}
    if resolution is not None:
        cam["film"] = {"resolution": [int(resolution[0]), int(resolution[1])]}
    if filter is not None:
        cam["filter"] = filter
    if cam:
        o["camera"] = cam
    return o
