"""Exact-t ties, made common: goblin_amd/scenes/ties.json is a floor of coplanar triangles whose hit distances are bit-identical
(scenes/make_meshes.py ties_mesh), so nearly every query ends in a tie -- inside one two-triangle leaf of the reference's tree and
across leaves (GoblinTriangle.cpp:74-80 `t > ray.maxt`, GoblinBVH.cpp:106-118, 156-187).  Every kernel that follows the
reference's tie rule (kernels/trace.h tie_goes_to, GBL_TIE_DETECT / GBL_TIE_EXACT) must return the oracle's radiance sample by
sample -- and so must the build with the rule forced inline (lib/variants/libgoblin_hip_tieinl.so, -DGBL_TIE_INLINE): that is the
form a StructurizeCFG bug of this compiler miscompiled in round 3 (tools/compiler_bugs/structurizecfg_hoisted_phi.ll; on the
Cornell box it showed as 114 samples in 6.8e7, here it would be every other sample).  The oracle itself is pinned on this scene by
the `ties_pt` fixture captured from the compiled reference (tests/test_oracle_vs_reference.py).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
import oracle_binding as ob
from goblin_amd import scene as gs

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 40417
OVERRIDES = gs.config_overrides(resolution=(96, 96), spp=16, depth=4)
VARIANTS = [("megakernel", None), ("megakernel", "0"), ("wavefront", None)]   # (schedule, GBL_MK_QUAD)


def render_variants(scene):
    """li of every exact-tie kernel of the path tracer + the lean megakernel (no tie rule), as numpy arrays."""
    import torch
    from goblin_amd.renderer import HipPathTracer
    assert torch.cuda.is_available()
    out = {}
    for schedule, quad in VARIANTS:
        if quad is not None:
            os.environ["GBL_MK_QUAD"] = quad
        try:
            r = HipPathTracer(scene, 0)
            out["%s%s" % (schedule, "" if quad is None else "_quad" + quad)] = \
                r.render(seed=SEED, want_li=True, schedule=schedule, exact_ties=True)["li"].cpu().numpy()
        finally:
            os.environ.pop("GBL_MK_QUAD", None)
    r = HipPathTracer(scene, 0)
    out["lean"] = r.render(seed=SEED, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    out["replay"] = None
    return out, r


@pytest.fixture(scope="module")
def tie_scene():
    scene = gs.load_scene("ties", OVERRIDES)
    o = ob.Oracle(scene)
    samples = o.native_samples(SEED)
    li_ref, _ = o.li_replay(samples, threads=4)
    return scene, samples, li_ref


def test_every_tie_following_kernel_returns_the_oracles_radiance(tie_scene):
    scene, samples, li_ref = tie_scene
    got, r = render_variants(scene)
    for name, li in got.items():
        if name in ("lean", "replay"):
            continue
        bad = int((li[:, :3] != li_ref[:, :3]).any(axis=1).sum())
        print(name, "samples off the oracle:", bad, "of", li.shape[0])
        assert bad == 0, (name, bad)
    # replayed records (the TIES build every fixture test runs), both schedules
    for schedule in ("megakernel", "wavefront"):
        li = r.render(replay_samples=samples, want_li=True, schedule=schedule)["li"].cpu().numpy()
        assert np.array_equal(li[:, :3], li_ref[:, :3]), schedule
    # the scene does what it is for: without the rule (last triangle tested at a distance keeps it) a large share of the samples
    # comes out differently
    lean_off = float((got["lean"][:, :3] != li_ref[:, :3]).any(axis=1).mean())
    print("lean kernel (no tie rule) differs on %.1f %% of the samples" % (100 * lean_off))
    assert lean_off > 0.03, lean_off


def test_the_build_with_the_tie_rule_inlined_returns_it_too(tie_scene):
    """Same renders through lib/variants/libgoblin_hip_tieinl.so in a process of its own (the library is chosen at load time)."""
    scene, samples, li_ref = tie_scene
    lib = os.path.join(REPO, "goblin_amd", "lib", "variants", "libgoblin_hip_tieinl.so")
    assert os.path.exists(lib), "goblin_amd.build.build_tie_inline() did not run"
    ref_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ties_li_ref_%d.npy" % os.getpid())
    np.save(ref_path, li_ref)
    env = dict(os.environ, GOBLIN_HIP_LIB=lib, PYTHONPATH=os.pathsep.join([REPO, os.path.join(REPO, "tests"), os.environ.get("PYTHONPATH", "")]))
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), ref_path], env=env, capture_output=True, text=True, timeout=600)
    finally:
        os.remove(ref_path)
    assert p.returncode == 0, p.stdout + p.stderr
    res = json.loads(p.stdout.strip().splitlines()[-1])
    print(res)
    assert res["library"].endswith("libgoblin_hip_tieinl.so")
    for name, bad in res["off"].items():
        assert bad == 0, (name, bad)


if __name__ == "__main__":   # the child of the test above
    sys.path.insert(0, REPO)
    from goblin_amd import _abi
    li_ref = np.load(sys.argv[1])
    scene = gs.load_scene("ties", OVERRIDES)
    got, _ = render_variants(scene)
    off = {k: int((v[:, :3] != li_ref[:, :3]).any(axis=1).sum()) for k, v in got.items() if k not in ("lean", "replay")}
    print(json.dumps({"library": _abi.hip_lib()._name, "off": off}))
