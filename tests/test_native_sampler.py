"""The counter-based ("native") sampler: same stratification law as
Sampler::requestSamples (GoblinSampler.cpp:108-197), checked structurally and
statistically against the reference-stream sampler."""
import numpy as np
import pytest

import helpers
import oracle_binding as ob
from goblin_amd import scene as gs


def test_every_pattern_is_stratified_per_pixel():
    scene = gs.load_scene("bunny", gs.config_overrides(resolution=(8, 8), spp=16, depth=3))
    o = ob.Oracle(scene)
    w = o.window()
    spp, root = 16, 4
    s = o.native_samples(seed=11).reshape((w[3] - w[2]) * (w[1] - w[0]), spp, -1)
    dims = s.shape[2]
    assert dims == o.dims() == 4 + 7 * 3 + 32
    off = o.pt_offsets()
    for pix in range(0, s.shape[0], 7):
        rec = s[pix]
        px, py = w[0] + pix % (w[1] - w[0]), w[2] + pix // (w[1] - w[0])
        # image samples: sample k sits in sub-cell k of the pixel, not permuted
        cx = np.floor((rec[:, 0] - px) * root).astype(int)
        cy = np.floor((rec[:, 1] - py) * root).astype(int)
        np.testing.assert_array_equal(cy * root + cx, np.arange(spp))
        for b in range(3):
            for o1 in (off[b, 0], off[b, 2], off[b, 4]):          # 1D slots: one sample per 1/spp stratum
                strata = np.sort(np.floor(rec[:, o1] * spp).astype(int))
                np.testing.assert_array_equal(strata, np.arange(spp))
            for o2 in (off[b, 1], off[b, 3]):                      # 2D slots: one sample per sub-cell
                cell = np.floor(rec[:, o2 + 1] * root).astype(int) * root + np.floor(rec[:, o2] * root).astype(int)
                np.testing.assert_array_equal(np.sort(cell), np.arange(spp))
    assert (s[..., 4:] >= 0).all() and (s[..., 4:] < 1).all()


def test_stream_sampler_has_the_same_structure():
    """Same structural test on the reference-law stream, i.e. the two samplers obey one law."""
    scene = gs.load_scene("bunny", gs.config_overrides(resolution=(8, 8), spp=16, depth=3))
    o = ob.Oracle(scene)
    res = o.render(threads=1, want_samples=True)
    s = res["samples"].reshape(-1, 16, o.dims())
    off = o.pt_offsets()
    for pix in range(0, s.shape[0], 5):
        rec = s[pix]
        for o1 in (off[0, 0], off[1, 2], off[2, 4]):
            np.testing.assert_array_equal(np.sort(np.floor(rec[:, o1] * 16).astype(int)), np.arange(16))
        for o2 in (off[0, 1], off[2, 3]):
            cell = np.floor(rec[:, o2 + 1] * 4).astype(int) * 4 + np.floor(rec[:, o2] * 4).astype(int)
            np.testing.assert_array_equal(np.sort(cell), np.arange(16))


def test_native_and_stream_renders_agree_within_monte_carlo_noise():
    scene = gs.load_scene("bunny", gs.config_overrides(resolution=(40, 40), spp=64, depth=4))
    o = ob.Oracle(scene)
    stream = ob.normalize_film(o.render(threads=4)["film"])
    nat_a = ob.normalize_film(o.render(threads=4, sampler=1, seed=1)["film"])
    nat_b = ob.normalize_film(o.render(threads=4, sampler=1, seed=2)["film"])
    noise = helpers.rel_l2(nat_a, nat_b)          # two independent draws of the same estimator
    cross = helpers.rel_l2(nat_a, stream)
    print("noise band", noise, "native-vs-stream", cross)
    assert cross <= 1.35 * noise + 1e-3
    # and the means agree (unbiasedness), far tighter than the per-pixel noise
    assert abs(nat_a.mean() - stream.mean()) <= 0.03 * stream.mean()


def test_sharded_windows_draw_the_same_numbers():
    """Native samples are keyed by the pixel's position in the FULL window, so a shard draws what the whole draws."""
    scene = gs.load_scene("bunny", gs.config_overrides(resolution=(16, 16), spp=4, depth=2))
    o = ob.Oracle(scene)
    w = o.window()
    whole = o.native_samples(5).reshape(w[3] - w[2], w[1] - w[0], 4, -1)
    sub = (w[0] + 3, w[0] + 9, w[2] + 2, w[2] + 7)
    part = o.native_samples(5, window=sub).reshape(5, 6, 4, -1)
    np.testing.assert_array_equal(part, whole[2:7, 3:9])
