"""Film::writeImage's tail on the host side: bloom, tone mapping, PPM and HALF EXR output.

The reference's GoblinImageIO.cpp does not compile here (MSVC fopen_s), so these are checked against numpy
restatements written from GoblinImageIO.cpp:101-127 (PPM), :169-218 (bloom), :220-236 (toneMapping) and
tinyexr.h:7164-7199 (float -> half), and the EXR container is read back with an independent minimal reader."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from goblin_amd import _abi
from goblin_amd import scene as gs


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _image(w=24, h=16, seed=3):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 3), dtype=np.float32) ** 3 * 4.0
    img[2, 3] = [60.0, 50.0, 40.0]     # a highlight for bloom / tone mapping to act on
    return np.ascontiguousarray(img)


def _luminance(c):
    return np.float32(0.212671) * c[..., 0] + np.float32(0.715160) * c[..., 1] + np.float32(0.072169) * c[..., 2]


def test_bloom_matches_restatement():
    img = _image()
    h, w, _ = img.shape
    radius, weight = np.float32(0.2), np.float32(0.35)
    fw = int(np.ceil(radius * max(w, h))) // 2
    assert fw >= 2
    filt = np.zeros((fw, fw), np.float32)
    for y in range(fw):
        for x in range(fw):
            d = np.float32(np.sqrt(np.float32(x * x + y * y))) / np.float32(fw)
            filt[y, x] = max(np.float32(0.0), np.float32(1.0) - d) ** np.float32(4.0)
    want = np.zeros_like(img)
    for y in range(h):
        for x in range(w):
            acc, ws = np.zeros(3, np.float32), np.float32(0.0)
            for py in range(max(0, y - fw + 1), min(y + fw - 1, h - 1) + 1):
                for px in range(max(0, x - fw + 1), min(x + fw - 1, w - 1) + 1):
                    fx, fy = abs(px - x), abs(py - y)
                    if fx == 0 and fy == 0:
                        continue
                    acc += filt[fy, fx] * img[py, px]
                    ws += filt[fy, fx]
            want[y, x] = (np.float32(1.0) - weight) * img[y, x] + weight * (acc * (np.float32(1.0) / ws))
    got = img.copy()
    _abi.host_lib().gbl_host_bloom(_ptr(got), w, h, float(radius), float(weight))
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)
    same = img.copy()
    _abi.host_lib().gbl_host_bloom(_ptr(same), w, h, 0.0, 0.5)       # radius or weight <= 0: untouched
    _abi.host_lib().gbl_host_bloom(_ptr(same), w, h, 0.5, 0.0)
    np.testing.assert_array_equal(same, img)


def test_tone_mapping_matches_restatement():
    img = _image()
    h, w, _ = img.shape
    y = _luminance(img)
    ywa = np.float32(0.0)
    for v in y.reshape(-1):
        ywa += np.log(np.float32(1e4) + v, dtype=np.float32)
    ywa = np.exp(ywa / np.float32(w * h), dtype=np.float32)
    s = (np.float32(1.0) + y * (np.float32(1.0) / (ywa * ywa))) / (np.float32(1.0) + y)
    want = img * s[..., None]
    got = img.copy()
    _abi.host_lib().gbl_host_tone_map(_ptr(got), w, h)
    np.testing.assert_allclose(got, want, rtol=1e-5)


def _half_bits_tinyexr(f):
    """tinyexr.h:7164-7199 float_to_half_full: round half UP on the dropped mantissa bits."""
    u = struct.unpack("<I", struct.pack("<f", f))[0]
    sign, exp, man = u >> 31, (u >> 23) & 0xff, u & 0x7fffff
    o = 0
    if exp == 0:
        o = 0
    elif exp == 255:
        o = (31 << 10) | (0x200 if man else 0)
    else:
        ne = exp - 127 + 15
        if ne >= 31:
            o = 31 << 10
        elif ne <= 0:
            if 14 - ne <= 24:
                m = man | 0x800000
                o = (m >> (14 - ne)) & 0x3ff
                if (m >> (13 - ne)) & 1:
                    o += 1
        else:
            o = (ne << 10) | (man >> 13)
            if man & 0x1000:
                o += 1
    return (o | (sign << 15)) & 0xffff


def _read_exr(path):
    """Minimal OpenEXR 2.0 reader: single-part, scanline, NO_COMPRESSION."""
    data = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", data, 0)
    assert magic == 20000630 and version == 2
    pos, attrs = 8, {}
    while data[pos] != 0:
        end = data.index(b"\0", pos)
        name = data[pos:end].decode()
        pos = end + 1
        end = data.index(b"\0", pos)
        typ = data[pos:end].decode()
        pos = end + 1
        size = struct.unpack_from("<i", data, pos)[0]
        pos += 4
        attrs[name] = (typ, data[pos:pos + size])
        pos += size
    pos += 1
    chans, c = [], attrs["channels"][1]
    i = 0
    while c[i] != 0:
        end = c.index(b"\0", i)
        ptype, _plin, xs, ys = struct.unpack_from("<iIii", c, end + 1)
        chans.append((c[i:end].decode(), ptype, xs, ys))
        i = end + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    offsets = struct.unpack_from("<%dQ" % h, data, pos)
    planes = np.zeros((len(chans), h, w), np.uint16)
    for row, off in enumerate(offsets):
        yy, nbytes = struct.unpack_from("<ii", data, off)
        assert yy == y0 + row and nbytes == w * 2 * len(chans)
        line = np.frombuffer(data, np.uint16, w * len(chans), off + 8).reshape(len(chans), w)
        planes[:, row] = line
    return attrs, chans, planes


def test_exr_is_half_bgr_with_tinyexr_rounding(tmp_path):
    img = _image(20, 11)
    img[0, 0] = [0.0, -0.0, 1e-9]            # zero / underflow
    img[0, 1] = [70000.0, 65504.0, 6.1e-5]   # overflow -> inf, max half, smallest normal region
    img[0, 2] = [1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 2.0 ** -20]   # exact ties: tinyexr rounds them UP
    h, w, _ = img.shape
    path = str(tmp_path / "out.exr")
    assert _abi.host_lib().gbl_host_write_exr(path.encode(), _ptr(img), w, h) == 0
    attrs, chans, planes = _read_exr(path)
    assert [(n, t, xs, ys) for n, t, xs, ys in chans] == [("B", 1, 1, 1), ("G", 1, 1, 1), ("R", 1, 1, 1)]   # HALF, B G R
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    assert struct.unpack("<4i", attrs["displayWindow"][1]) == (0, 0, w - 1, h - 1)
    want = np.vectorize(_half_bits_tinyexr, otypes=[np.uint16])(img)
    np.testing.assert_array_equal(planes[2], want[..., 0])   # R
    np.testing.assert_array_equal(planes[1], want[..., 1])   # G
    np.testing.assert_array_equal(planes[0], want[..., 2])   # B
    # and the stored halves are the image to half precision
    back = planes.view(np.float16).astype(np.float32)
    ok = np.isfinite(back[2])
    np.testing.assert_allclose(back[2][ok], img[..., 0][ok], rtol=1e-3, atol=1e-7)
    assert _half_bits_tinyexr(1.0 + 2.0 ** -11) == 0x3c01    # a tie goes up (numpy's float16 would give 0x3c00)
    assert np.float32(1.0 + 2.0 ** -11).astype(np.float16).view(np.uint16) == 0x3c00


def test_ppm_and_extension_dispatch(tmp_path):
    img = _image(7, 5)
    h, w, _ = img.shape
    lib = _abi.host_lib()
    p = str(tmp_path / "a.ppm")
    assert lib.gbl_host_write_image(p.encode(), _ptr(img.copy()), w, h, 0) == 0
    toks = open(p).read().split()
    assert toks[:4] == ["P3", str(w), str(h), "255"]
    vals = np.array(toks[4:], np.int32).reshape(h, w, 3)
    want = (np.clip(img ** np.float32(1.0 / 2.2), 0.0, 1.0) * np.float32(255.0)).astype(np.int32)
    assert np.abs(vals - want).max() <= 1     # powf vs numpy's pow at the integer boundary
    # tone mapping only applies to .ppm (GoblinImageIO.cpp:155-159)
    t = img.copy()
    assert lib.gbl_host_write_image(str(tmp_path / "b.ppm").encode(), _ptr(t), w, h, 1) == 0
    assert not np.array_equal(t, img)
    t = img.copy()
    assert lib.gbl_host_write_image(str(tmp_path / "b.exr").encode(), _ptr(t), w, h, 1) == 0
    np.testing.assert_array_equal(t, img)
    # no extension / unknown extension -> "<name>.ppm" (:148-150, :162-166)
    assert lib.gbl_host_write_image(str(tmp_path / "noext").encode(), _ptr(img.copy()), w, h, 0) == 0
    assert lib.gbl_host_write_image(str(tmp_path / "x.tiff").encode(), _ptr(img.copy()), w, h, 0) == 0
    assert os.path.exists(tmp_path / "noext.ppm") and os.path.exists(tmp_path / "x.tiff.ppm")


def test_film_output_parameters(tmp_path):
    doc = {"camera": {"film": {"resolution": [8, 8], "tone_mapping": True, "bloom_radius": 0.1, "bloom_weight": 0.25}}}
    s = gs.load_scene_text(json.dumps(doc), ".")
    f = s.desc.film
    assert (f.tone_mapping, f.bloom_radius, f.bloom_weight) == (1, np.float32(0.1), 0.25)
    assert _abi.host_lib().gbl_host_output_path(s._handle) == b"goblin.exr"      # createImageFilm's default
    p = tmp_path / "my.scene.json"
    p.write_text(json.dumps({"camera": {"film": {"resolution": [8, 8]}}}))
    s = gs.load_scene(str(p))
    assert (s.desc.film.tone_mapping, s.desc.film.bloom_radius) == (0, 0.0)
    assert _abi.host_lib().gbl_host_output_path(s._handle).decode() == str(tmp_path / "my.scene.exr")   # <scene>.exr
    p.write_text(json.dumps({"camera": {"film": {"file": "out/pic.ppm"}}}))
    s = gs.load_scene(str(p))
    assert _abi.host_lib().gbl_host_output_path(s._handle) == b"out/pic.ppm"
