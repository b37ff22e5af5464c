"""The boundary, running.  (1) The REFERENCE as host: oracle/_ref/g_ray_hipbind = the reference's own ContextLoader, Scene, Film and
writeImage with `HipPathTracer : Renderer` (tests/integration/) bound in behind the virtual Renderer::render
(/root/reference/src/GoblinRenderer.h:55-59, GoblinContextLoader.cpp:447-504); it renders BASELINE configs[0] (bunny.json 256x256,
16 spp, depth 4) with GBL_SAMPLES_STREAM and the Film it leaves in the reference's Film object must be the Film the compiled
reference's own PathTracer produced for the same file (the `bunny_config1` fixture): same sample stream, float summation order apart.
(2) This repository's stand-alone host, goblin_amd/lib/g_ray_hip (the g_ray.cpp:7-27 equivalent): the EXR it writes for the same
file, read back through gbl_host_read_image, is that Film normalised, to HALF precision (Film::writeImage, GoblinFilm.cpp:164-192)."""
import json
import os
import subprocess

import numpy as np
import pytest

import helpers
import integration_helpers as ih
import oracle_binding as ob
from goblin_amd import _abi

pytestmark = pytest.mark.gpu

CLI = os.path.join(ih.REPO, "goblin_amd", "lib", "g_ray_hip")


@pytest.fixture(scope="module")
def config1(golden, tmp_path_factory):
    meta, data = golden("bunny_config1")
    d = tmp_path_factory.mktemp("integration")
    js = str(d / "bunny_config1.json")
    ih.write_scene(meta["scene"], meta["overrides"], js)
    return meta, data, js, d


def test_reference_hosted_binding_leaves_the_references_film(config1):
    meta, data, js, d = config1
    assert os.path.exists(ih.HIPBIND), "oracle/_ref/g_ray_hipbind is missing: `make -C oracle hipbind` where /root/reference exists"
    dump = str(d / "film_bind.f32")
    p = subprocess.run([ih.HIPBIND, js, "--sampler", "stream", "--dump-film", dump], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    paths = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["paths"]
    assert paths == meta["paths"]
    ref = data["film"]
    film = np.fromfile(dump, np.float32).reshape(ref.shape)
    np.testing.assert_allclose(film[..., 3], ref[..., 3], rtol=1e-5, atol=1e-6)          # sums of filter-table values
    rel = helpers.rel_l2(ob.normalize_film(film), ob.normalize_film(ref))
    print("reference-hosted HipPathTracer (stream sampler) vs the reference's own PathTracer: film relL2", rel)
    assert rel <= 1e-5
    # the reference's writeImage ran on that Film: its EXR sits beside the scene file (default output path, ContextLoader.cpp:473-484)
    exr = js[:-5] + ".exr"
    assert os.path.exists(exr)
    img = _abi.read_image(exr)[..., :3]
    want = ob.normalize_film(film)
    np.testing.assert_allclose(img, want.astype(np.float16).astype(np.float32), rtol=2e-3, atol=1e-4)


def test_standalone_host_writes_that_film(config1):
    meta, data, js, d = config1
    assert os.path.exists(CLI)
    out = str(d / "cli.exr")
    p = subprocess.run([CLI, js, "--sampler", "stream", "--out", out], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "Render Complete" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    img = _abi.read_image(out)
    ref = ob.normalize_film(data["film"])
    assert img.shape[:2] == ref.shape[:2]
    # HALF has 11 significant bits: 2^-11 relative rounding, on top of the film's own 1e-5
    np.testing.assert_allclose(img[..., :3], ref, rtol=1.5e-3, atol=2e-4)
    rel = helpers.rel_l2(img[..., :3], ref)
    print("g_ray_hip's EXR vs the reference's normalised film: relL2", rel)
    assert rel <= 1e-3
