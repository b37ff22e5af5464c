#!/usr/bin/env python3
"""Author the image files of the image-textured and image-lit scenes (analytic, no RNG: bit-reproducible).

    python goblin_amd/scenes/make_images.py

  images/tiles.exr   48 x 32 (not powers of two: MIPMap resizes it), colour: warm tiles with grout lines and a gradient
  images/env.exr     64 x 32 latitude-longitude sky: blue gradient above, dim ground below, a small hot sun
Written through libgoblin_host.so's own writer (three HALF channels, uncompressed).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from goblin_amd import _abi


def write(name, img):
    img = np.ascontiguousarray(img, np.float32)
    h, w, _ = img.shape
    path = os.path.join(HERE, "images", name)
    st = _abi.host_lib().gbl_host_write_exr(os.fsencode(path), img.ctypes.data_as(C.c_void_p), w, h)
    assert st == _abi.GBL_OK
    print(path, w, "x", h)


def tiles(w=48, h=32):
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    tx, ty = (x % 12) / 12.0, (y % 8) / 8.0
    grout = (tx < 0.12) | (ty < 0.16)
    base = np.stack([0.75 - 0.3 * (x / w), 0.35 + 0.4 * (y / h), 0.2 + 0.15 * np.sin(0.5 * x) * np.cos(0.4 * y)], axis=-1)
    alt = ((x // 12 + y // 8) % 2)[..., None] * np.array([0.1, -0.05, 0.25], np.float32)
    img = np.clip(base + alt, 0.02, 1.0)
    img[grout] = [0.08, 0.08, 0.09]
    return img


def env(w=64, h=32):
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    theta = (y + 0.5) / h * np.pi            # 0 at the pole the light's frame maps to "up"
    phi = (x + 0.5) / w * 2 * np.pi
    up = np.cos(theta)
    sky = np.stack([0.25 + 0.15 * up, 0.35 + 0.25 * up, 0.55 + 0.45 * up], axis=-1)
    ground = np.stack([0.12 + 0 * up, 0.10 + 0 * up, 0.08 + 0 * up], axis=-1)
    img = np.where((up > 0)[..., None], sky, ground).astype(np.float32)
    sun = np.exp(-((theta - 0.9) ** 2 + (phi - 2.2) ** 2) / 0.02)
    img += sun[..., None] * np.array([60.0, 52.0, 40.0], np.float32)
    return img


if __name__ == "__main__":
    write("tiles.exr", tiles())
    write("env.exr", env())
