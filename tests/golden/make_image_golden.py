#!/usr/bin/env python3
"""Golden fixtures of the reference's image path (GoblinImageIO.cpp), from the REAL reference.

Runs only in the authoring container (needs oracle/_ref/ref_harness with GoblinImageIO.cpp compiled in, `make -C oracle
ref`).  Inputs are seeded synthetic HDR images; outputs are what the reference computes from them: bloom, toneMapping,
the .exr file Goblin::writeImage produces (three HALF channels, ZIP, through the tinyexr copy the reference links), the
.ppm it writes with tone mapping, and Goblin::loadImage of that .exr.  Data only.

    python tests/golden/make_image_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(REPO, "oracle", "_ref", "ref_harness")

# name -> (width, height, seed, bloom radius, bloom weight)
CASES = {"image_a": (24, 16, 11, 0.2, 0.15), "image_b": (37, 21, 12, 0.12, 0.4)}


def synthetic(w, h, seed):
    """A smooth gradient with a few very bright pixels (what bloom and the tone map exist for), subnormal-half and
    beyond-half-range values included so the float -> half rule is exercised at its edges."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([0.2 + 0.8 * x / w, 0.1 + 0.5 * y / h, 0.3 + 0.3 * np.sin(0.4 * x + 0.3 * y), np.ones_like(x)], axis=-1).astype(np.float32)
    img[..., :3] *= rng.uniform(0.5, 1.5, size=(h, w, 3)).astype(np.float32)
    for _ in range(6):
        img[rng.integers(h), rng.integers(w), :3] = rng.uniform(20.0, 400.0, size=3).astype(np.float32)
    img[0, 0, :3] = [1e-7, 3e-6, 7e4]        # subnormal halves, and a value past the largest half
    img[1, 1, :3] = [0.0, 65504.0, 1.0 / 3.0]
    return img


def main():
    if not os.path.exists(HARNESS):
        sys.exit("oracle/_ref/ref_harness is missing: run `make -C oracle ref` (needs /root/reference)")
    for name, (w, h, seed, radius, weight) in CASES.items():
        img = synthetic(w, h, seed)
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "in.f32")
            img.tofile(src)
            prefix = os.path.join(tmp, name)
            meta = json.loads(subprocess.check_output([HARNESS, "image", src, str(w), str(h), prefix, repr(radius), repr(weight)]).decode())
            assert (meta["width"], meta["height"]) == (w, h)
            arrays = {"input": img, "bloom_radius": np.float32(radius), "bloom_weight": np.float32(weight)}
            for k in ("bloom", "tone", "load"):
                arrays[k] = np.fromfile(prefix + "." + k + ".f32", np.float32).reshape(h, w, 4)
            arrays["exr_bytes"] = np.fromfile(prefix + ".exr", np.uint8)
            arrays["ppm_bytes"] = np.fromfile(prefix + ".ppm", np.uint8)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
            print(name, w, "x", h, "exr", arrays["exr_bytes"].size, "bytes, ppm", arrays["ppm_bytes"].size, "bytes")


if __name__ == "__main__":
    main()
