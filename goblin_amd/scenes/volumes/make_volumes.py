"""Writes the density grids of the heterogeneous-medium scenes (Mitsuba .vol, float32 encoding -- the layout
HeterogeneousVolumeRegion's loader reads, GoblinVolume.cpp:43-66): puff.vol, one channel, and tint.vol, three."""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def write_vol(path, data, lo, hi):
    nz, ny, nx, nch = data.shape
    with open(path, "wb") as f:
        f.write(b"VOL\x03")
        f.write(struct.pack("<5i", 1, nx, ny, nz, nch))
        f.write(struct.pack("<6f", *lo, *hi))
        f.write(np.ascontiguousarray(data, np.float32).tobytes())


def puff(nx, ny, nz, nch, seed):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.linspace(-1, 1, nz), np.linspace(-1, 1, ny), np.linspace(-1, 1, nx), indexing="ij")
    r2 = x * x + 1.4 * y * y + 0.8 * z * z
    base = np.clip(1.6 * np.exp(-2.2 * r2) + 0.25 * np.sin(5 * x + 1) * np.cos(4 * z) * np.exp(-r2), 0, None)
    base = base * (0.85 + 0.3 * rng.random(base.shape))
    if nch == 1:
        return base[..., None].astype(np.float32)
    tint = np.stack([base * 0.9, base * (0.7 + 0.3 * (y + 1) / 2), base * (1.1 - 0.4 * (x + 1) / 2)], axis=-1)
    return tint.astype(np.float32)


if __name__ == "__main__":
    write_vol(os.path.join(HERE, "puff.vol"), puff(12, 10, 8, 1, 7), (-1.0, -0.8, -0.9), (1.0, 0.8, 0.9))
    write_vol(os.path.join(HERE, "tint.vol"), puff(5, 6, 7, 3, 8), (-0.5, -1.0, -1.0), (1.5, 1.0, 1.0))
