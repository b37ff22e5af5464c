"""Film::writeImage's tail and Goblin::loadImage on the host side: bloom, tone mapping, PPM and HALF
EXR output, EXR input.

Two layers of checks: numpy restatements written from GoblinImageIO.cpp:101-127 (PPM), :169-218 (bloom), :220-236
(toneMapping) and tinyexr.h:7164-7199 (float -> half) with an independent minimal EXR container reader (the first half of
this file, kept from round 1), and -- the pin -- fixtures captured from the reference's own GoblinImageIO.cpp compiled
into oracle/_ref (tests/golden/make_image_golden.py; second half)."""
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


# ---------------------------------------------------------------------------------------------------------------------
# Pinned against the reference itself: fixtures captured from GoblinImageIO.cpp compiled into oracle/_ref
# (tests/golden/make_image_golden.py): bloom, toneMapping, the .exr / .ppm files Goblin::writeImage wrote and
# Goblin::loadImage of that .exr.
# ---------------------------------------------------------------------------------------------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AUTHORING = os.path.exists("/root/reference/src/GoblinImageIO.cpp")
HARNESS = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "ref_harness")


def _fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def _equal(got, want, what):
    """Bit-equal where the fixtures were made (same libm); one decade above float epsilon elsewhere."""
    if AUTHORING:
        np.testing.assert_array_equal(got, want, err_msg=what)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7, err_msg=what)


@pytest.mark.parametrize("name", ["image_a", "image_b"])
def test_bloom_and_tone_map_equal_the_references(name):
    fx = _fixture(name)
    h, w, _ = fx["input"].shape
    rgb = np.ascontiguousarray(fx["input"][..., :3])
    got = rgb.copy()
    _abi.host_lib().gbl_host_bloom(_ptr(got), w, h, float(fx["bloom_radius"]), float(fx["bloom_weight"]))
    _equal(got, fx["bloom"][..., :3], "bloom")
    got = rgb.copy()
    _abi.host_lib().gbl_host_tone_map(_ptr(got), w, h)
    _equal(got, fx["tone"][..., :3], "toneMapping")


@pytest.mark.parametrize("name", ["image_a", "image_b"])
def test_exr_reader_decodes_the_references_file(name, tmp_path):
    """The .exr Goblin::writeImage produced (HALF B, G, R, ZIP-compressed by tinyexr) read by this repository's reader
    equals what Goblin::loadImage returned for the same file: every pixel, alpha defaulting to 1."""
    fx = _fixture(name)
    path = tmp_path / (name + ".exr")
    fx["exr_bytes"].tofile(path)
    got = _abi.read_image(str(path))
    assert got.shape == fx["load"].shape
    np.testing.assert_array_equal(got.view(np.uint32), fx["load"].view(np.uint32))
    assert (got[..., 3] == 1.0).all()


@pytest.mark.parametrize("name", ["image_a", "image_b"])
def test_exr_and_ppm_writers_equal_the_references(name, tmp_path):
    """This repository's .exr holds the reference's pixels (same float -> half rule; stored uncompressed, so the bytes
    differ): read back, it equals Goblin::loadImage of the reference's own file.  The tone-mapped .ppm is the same text."""
    fx = _fixture(name)
    h, w, _ = fx["input"].shape
    rgb = np.ascontiguousarray(fx["input"][..., :3])
    out = tmp_path / "mine.exr"
    assert _abi.host_lib().gbl_host_write_exr(os.fsencode(str(out)), _ptr(rgb), w, h) == _abi.GBL_OK
    mine = _abi.read_image(str(out))
    np.testing.assert_array_equal(mine.view(np.uint32), fx["load"].view(np.uint32))
    ppm = tmp_path / "mine.ppm"
    img = rgb.copy()
    assert _abi.host_lib().gbl_host_write_image(os.fsencode(str(ppm)), _ptr(img), w, h, 1) == _abi.GBL_OK
    want = bytes(fx["ppm_bytes"])
    got = ppm.read_bytes()
    if AUTHORING:
        assert got == want
    # elsewhere a pow() rounding may move a value across an integer boundary: at most a handful of digits differ
    a, b = np.array(got.split()[4:], np.int32), np.array(want.split()[4:], np.int32)
    assert got.split()[:4] == want.split()[:4] and a.shape == b.shape and np.abs(a - b).max() <= 1


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="needs oracle/_ref/ref_harness (the authoring container)")
def test_the_reference_reads_this_repositorys_exr(tmp_path):
    """The other direction: Goblin::loadImage (tinyexr) opens the uncompressed HALF file this repository writes."""
    import subprocess
    fx = _fixture("image_b")
    h, w, _ = fx["input"].shape
    rgb = np.ascontiguousarray(fx["input"][..., :3])
    out = tmp_path / "mine.exr"
    assert _abi.host_lib().gbl_host_write_exr(os.fsencode(str(out)), _ptr(rgb), w, h) == _abi.GBL_OK
    dump = tmp_path / "back.f32"
    meta = json.loads(subprocess.check_output([HARNESS, "imageload", str(out), str(dump)]).decode())
    assert (meta["width"], meta["height"]) == (w, h)
    back = np.fromfile(dump, np.float32).reshape(h, w, 4)
    np.testing.assert_array_equal(back.view(np.uint32), fx["load"].view(np.uint32))


def test_exr_reader_formats_and_errors(tmp_path):
    """FLOAT channels, a single (luminance) channel replicated, RLE and per-line ZIP blocks, an offset data window; files
    outside the reader (tiled, PIZ) and non-.exr names fail the way the reference's nullptr does."""
    import zlib

    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value

    def exr(channels, compression, planes, x0=0, y0=0, version=2):
        h, w = planes[0].shape
        chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", t, 0, 0, 0, 0, 1, 1) for n, t in channels) + b"\0"
        hdr = struct.pack("<II", 20000630, version) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
        box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
        hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
        hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
        hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
        per = {0: 1, 1: 1, 2: 1, 3: 16}[compression]
        blocks = []
        for b0 in range(0, h, per):
            raw = b"".join(planes[ci][y].astype(np.float16 if t == 1 else np.float32).tobytes() for y in range(b0, min(h, b0 + per)) for ci, (_, t) in enumerate(channels))
            if compression in (2, 3):
                a = np.frombuffer(raw, np.uint8)
                half = (a.size + 1) // 2
                re = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
                assert re[:half].size == half
                d = re.copy()
                d[1:] = (re[1:] - re[:-1] + 128 + 256) % 256
                packed = zlib.compress(d.astype(np.uint8).tobytes())
                data = packed if len(packed) < len(raw) else raw
            elif compression == 1:
                a = np.frombuffer(raw, np.uint8)
                re = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
                d = re.copy()
                d[1:] = (re[1:] - re[:-1] + 128 + 256) % 256
                data = b"".join(bytes([0xff, int(v)]) for v in d)     # every byte as a literal run of one
            else:
                data = raw
            blocks.append(struct.pack("<ii", y0 + b0, len(data)) + data)
        table_at = len(hdr)
        offs, pos = [], table_at + 8 * len(blocks)
        for b in blocks:
            offs.append(pos)
            pos += len(b)
        return hdr + b"".join(struct.pack("<Q", o) for o in offs) + b"".join(blocks)

    rng = np.random.default_rng(5)
    r, g, b, a = (rng.random((19, 13), dtype=np.float32) for _ in range(4))
    for comp in (0, 1, 2, 3):
        p = tmp_path / ("c%d.exr" % comp)
        p.write_bytes(exr([("A", 2), ("B", 2), ("G", 1), ("R", 2)], comp, [a, b, g, r], x0=-3, y0=7))
        got = _abi.read_image(str(p))
        np.testing.assert_array_equal(got[..., 0], r)
        np.testing.assert_array_equal(got[..., 1], g.astype(np.float16).astype(np.float32))
        np.testing.assert_array_equal(got[..., 2], b)
        np.testing.assert_array_equal(got[..., 3], a)
    p = tmp_path / "y.exr"
    p.write_bytes(exr([("Y", 1)], 3, [r]))
    got = _abi.read_image(str(p))
    for k in range(4):
        np.testing.assert_array_equal(got[..., k], r.astype(np.float16).astype(np.float32))
    for bad, status in ((exr([("G", 1), ("R", 1)], 0, [g, r]), _abi.GBL_ERR_IO),                     # B channel not found
                        (exr([("B", 1), ("G", 1), ("R", 1)], 0, [b, g, r], version=2 | 0x200), _abi.GBL_ERR_UNSUPPORTED),   # tiled
                        (b"not an exr file at all", _abi.GBL_ERR_IO)):
        p = tmp_path / "bad.exr"
        p.write_bytes(bad)
        with pytest.raises(_abi.GoblinError) as e:
            _abi.read_image(str(p))
        assert e.value.status == status
    piz = bytearray(exr([("B", 1), ("G", 1), ("R", 1)], 0, [b, g, r]))
    i = bytes(piz).index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    piz[i] = 4
    p = tmp_path / "piz.exr"
    p.write_bytes(bytes(piz))
    with pytest.raises(_abi.GoblinError) as e:
        _abi.read_image(str(p))
    assert e.value.status == _abi.GBL_ERR_UNSUPPORTED
    with pytest.raises(_abi.GoblinError) as e:
        _abi.read_image(str(tmp_path / "picture.png"))
    assert e.value.status == _abi.GBL_ERR_IO and "unsupported format" in str(e.value)
