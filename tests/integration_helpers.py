"""Shared by tests/test_integration_binding.py (CPU) and tests/test_gpu_integration.py: the test-only g_ray with the HIP renderer
bound in (oracle/_ref/g_ray_hipbind, built by `make -C oracle hipbind` where /root/reference exists) and scene files for it."""
import ctypes as C
import json
import os
import struct

import numpy as np

from goblin_amd import _abi
from goblin_amd import scene as gs

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPBIND = os.path.join(REPO, "oracle", "_ref", "g_ray_hipbind")


def write_scene(scene, overrides, path, film_file=None):
    """The bundled scene with `overrides` merged in, mesh paths made absolute, written to `path` (what tests/golden/make_golden.py
    hands the reference).  Returns the document."""
    src = gs.scene_path(scene)
    with open(src) as f:
        doc = json.load(f)
    gs._merge(doc, overrides)
    doc.setdefault("render_setting", {})["thread_num"] = 1
    for section in ("geometries", "textures", "lights"):
        for g in doc.get(section, []):
            if "file" in g:
                g["file"] = os.path.join(os.path.dirname(src), g["file"])
    if film_file:
        doc["camera"].setdefault("film", {})["file"] = film_file
    with open(path, "w") as f:
        json.dump(doc, f)
    return doc


def read_desc_dump(path):
    """g_ray_hipbind --dump-desc: {tag: bytes}."""
    out = {}
    with open(path, "rb") as f:
        while True:
            head = f.read(24)
            if len(head) < 24:
                break
            tag = head[:16].split(b"\0")[0].decode()
            n, = struct.unpack("<Q", head[16:])
            out[tag] = f.read(n)
    return out


def structs(raw, ctype):
    n = len(raw) // C.sizeof(ctype)
    return [ctype.from_buffer_copy(raw, i * C.sizeof(ctype)) for i in range(n)]
