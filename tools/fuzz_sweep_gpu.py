"""Device against oracle on many round-2 fuzz scenes (tests/helpers.random_scene_r2), beyond the seeds the test suite runs:
    python tools/fuzz_sweep_gpu.py <first seed> <last seed + 1>"""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers, oracle_binding as ob
from goblin_amd import scene as gs, _abi
from goblin_amd.renderer import HipPathTracer
SCENE_DIR = os.path.dirname(gs.scene_path("bunny"))
bad = []
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for whitted in (False, True):
        doc, hetero = helpers.random_scene_r2(2000 + seed, whitted)
        scene = gs.load_scene_text(json.dumps(doc), SCENE_DIR)
        o = ob.Oracle(scene)
        samples = o.native_samples(99 + seed)
        li_ref, _ = o.li_replay(samples, threads=8)
        r = HipPathTracer(scene, 0)
        for schedule in (("megakernel",) if whitted else ("megakernel", "wavefront")):
            li = r.render(seed=99 + seed, want_li=True, schedule=schedule)["li"].cpu().numpy()
            # (a NaN radiance -- the reference produces one now and then with a medium -- must be NaN on the device too)
            n = int(np.any((li != li_ref) & ~(np.isnan(li) & np.isnan(li_ref)), axis=1).sum())
            if n:
                bad.append((seed, whitted, schedule, n))
                print("MISMATCH seed", seed, "whitted" if whitted else "pt", schedule, n, "of", li.shape[0], "volume", doc.get("volume", {}).get("type"),
                      "lights", [(l["type"], l.get("geometry")) for l in doc["lights"]], flush=True)
        if True:
            ref = o.render(threads=1)["film"]
            film = r.render(sampler="stream")["film"].numpy()
            # the accumulators themselves (a filter with negative lobes leaves weight sums near zero, where rgb / weight turns
            # the last bit of a float sum into 1e-4: seeds 537, 766, 775)
            rel = helpers.rel_l2(film, ref)
            if not (rel <= 2e-6) or not np.allclose(film[..., 3], ref[..., 3], rtol=1e-5, atol=1e-6):
                bad.append((seed, whitted, "stream", rel))
                print("STREAM MISMATCH seed", seed, "whitted" if whitted else "pt", "relL2 %.3g" % rel, flush=True)
    if seed % 10 == 9:
        print("... through seed", seed, flush=True)
print("swept", sys.argv[1], sys.argv[2], "bad", bad)
