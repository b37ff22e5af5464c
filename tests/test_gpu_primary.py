"""The primary pass (kernels/packet.h, kernels_quad.hip primary_kernel): the camera rays of a call traced as per-pixel packets ahead
of the quad path kernel, whose paths then start at those hits.  A packet answers for a ray only when the answer does not depend on the
visiting order -- a lane tests the triangles of the leaves its OWN ray enters, a ray that met two triangles at exactly its closest
distance (under exact_ties also: a hit the reference's box tests might pass by) is handed back to the path kernel -- so the per-sample
radiance must be bit for bit what the path kernel gives on its own (GBL_PRIMARY=0), for whole frames, sub-windows, tile shards, sample
counts that leave a packet's last lanes empty, instanced scenes, and the scene whose every query ends in a tie.  Against the oracle the
kernels with the pass are covered like any other by the parity tests (the pass is on by default).
Match: Scene::intersect as PathTracer::Li calls it for the camera ray, /root/reference/src/GoblinPathtracer.cpp:58-60, GoblinBVH.cpp:234-280.
"""
import os

import numpy as np
import pytest

from goblin_amd import scene as gs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


CASES = [
    # scene, overrides, render kwargs
    ("bunny", gs.config_overrides(resolution=(96, 80), spp=64, depth=5), {}),
    ("bunny", gs.config_overrides(resolution=(64, 64), spp=24, depth=4), {}),                 # 24 of a packet's 64 lanes have a sample
    ("bunny", gs.config_overrides(resolution=(64, 64), spp=100, depth=4), {}),                # a second, partly filled packet per pixel
    ("bunny", gs.config_overrides(resolution=(96, 96), spp=16, depth=4), {"shard": (1, 3)}),   # every third tile
    ("bunny", gs.config_overrides(resolution=(96, 96), spp=16, depth=4), {"window": "inner"}),
    ("cornell", gs.config_overrides(resolution=(64, 64), spp=32, depth=6), {}),
    ("grid", gs.config_overrides(resolution=(64, 64), spp=32, depth=4), {}),                  # instances of one mesh
    ("ties", gs.config_overrides(resolution=(96, 96), spp=16, depth=4), {}),                  # nearly every camera ray ties
]


def _render(r, primary, exact, kwargs):
    if not primary:
        os.environ["GBL_PRIMARY"] = "0"
    try:
        kw = dict(kwargs)
        if kw.get("window") == "inner":
            x0, x1, y0, y1 = r.window
            kw["window"] = (x0 + 9, x1 - 14, y0 + 5, y1 - 3)   # not tile aligned: edge tiles with clipped pixels
        return r.render(seed=918273, want_li=True, schedule="megakernel", exact_ties=exact, **kw)
    finally:
        os.environ.pop("GBL_PRIMARY", None)


@pytest.mark.parametrize("exact", [False, True], ids=["lean", "exact_ties"])
@pytest.mark.parametrize("name,overrides,kwargs", CASES, ids=["%s-%d" % (c[0], i) for i, c in enumerate(CASES)])
def test_radiance_is_bit_identical_with_and_without_the_primary_pass(torch, name, overrides, kwargs, exact):
    from goblin_amd.renderer import HipPathTracer
    r = HipPathTracer(gs.load_scene(name, overrides), 0)
    with_pass = _render(r, True, exact, kwargs)
    without = _render(r, False, exact, kwargs)
    assert torch.isfinite(with_pass["li"]).all()
    assert torch.equal(with_pass["li"], without["li"])
    # (the film sums the same radiance; tiles add their halos with float atomics, whose order is free)
    np.testing.assert_allclose(with_pass["film"].numpy(), without["film"].numpy(), rtol=1e-5, atol=1e-6)
    assert float(with_pass["li"][:, :3].sum()) > 0.0


def test_the_pass_leaves_the_tied_camera_rays_to_the_path_kernel(torch):
    """ties.json: the lean kernel keeps whichever of two exactly tied triangles its own traversal meets last; with the pass on, such a
    camera ray must come back flagged (GBL_PRIM_TIED) and be traced by the path kernel -- otherwise the packet's visiting order
    (the lead lane's) would decide, and the frames above would differ.  Here: the exact-ties render with the pass equals the oracle's
    radiance sample by sample, i.e. the flagged rays went through the reference's tie rule."""
    import helpers
    import oracle_binding as ob
    from goblin_amd.renderer import HipPathTracer
    scene = gs.load_scene("ties", gs.config_overrides(resolution=(64, 64), spp=16, depth=4))
    o = ob.Oracle(scene)
    li_ref, _ = o.li_replay(o.native_samples(5150), threads=4)
    r = HipPathTracer(scene, 0)
    li = r.render(seed=5150, want_li=True, schedule="megakernel", exact_ties=True)["li"].cpu().numpy()
    assert helpers.li_mismatch_fraction(li, li_ref) == 0.0
    np.testing.assert_array_equal(li[:, :3], li_ref[:, :3])


def test_a_pass_that_does_not_fit_the_budget_is_left_out(torch):
    """The pass keeps 20 bytes per camera sample beside the 16 of the per-sample radiance, both under GBL_LI_BUDGET_MB (a quarter of
    the device by default): a call whose radiance fits and whose pass does not runs without the pass -- same radiance."""
    from goblin_amd.renderer import HipPathTracer
    scene = gs.load_scene("bunny", gs.config_overrides(resolution=(64, 64), spp=64, depth=4))
    samples = scene.num_paths()
    budget_mb = (samples * 18) >> 20                     # between 16 and 20 bytes per sample
    assert samples * 16 <= budget_mb << 20 < samples * 20
    ref = HipPathTracer(scene, 0).render(seed=77, want_li=True, schedule="megakernel")["li"]
    os.environ["GBL_LI_BUDGET_MB"] = str(budget_mb)
    try:
        small = HipPathTracer(scene, 0).render(seed=77, want_li=False, schedule="megakernel")   # (its own radiance buffer: the budget applies)
        again = HipPathTracer(scene, 0).render(seed=77, want_li=True, schedule="megakernel")["li"]
    finally:
        os.environ.pop("GBL_LI_BUDGET_MB", None)
    assert torch.equal(again, ref)
    assert np.isfinite(small["film"].numpy()).all()
