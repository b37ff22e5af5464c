"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def tile_order_index(window, spp, tile=8):
    """Index array mapping the oracle's tile-then-pixel record order to the
    pixel-major order the C ABI's replay mode expects.

    out[pixel_major_record] = tile_order_record
    """
    x0, x1, y0, y1 = window
    w, h = x1 - x0, y1 - y0
    order = np.zeros((h, w), np.int64)
    n = 0
    for ty in range(y0, y1, tile):
        for tx in range(x0, x1, tile):
            tw, th = min(tile, x1 - tx), min(tile, y1 - ty)
            blk = n + np.arange(tw * th).reshape(th, tw)
            order[ty - y0:ty - y0 + th, tx - x0:tx - x0 + tw] = blk
            n += tw * th
    pix = order.reshape(-1)
    return (pix[:, None] * spp + np.arange(spp)[None, :]).reshape(-1)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def li_mismatch_fraction(a, b, rtol=1e-3, atol=1e-5):
    """Fraction of samples whose radiance differs by more than rtol (a 'flipped' sample:
    a discrete decision -- reflect/refract pick, hit/miss at an edge -- went the other way)."""
    a = np.asarray(a, np.float64)[:, :3]
    b = np.asarray(b, np.float64)[:, :3]
    bad = np.any(np.abs(a - b) > atol + rtol * np.abs(b), axis=1)
    return float(bad.mean())
