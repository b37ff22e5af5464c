"""Debug helper for tests/test_gpu_fuzz.py's round-2 scenes: where do device and oracle differ?  python tools/fuzz_dbg.py <seed>"""
import os, sys, json, copy
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import helpers, oracle_binding as ob
from goblin_amd import scene as gs, _abi
from goblin_amd.renderer import HipPathTracer
SCENE_DIR = os.path.dirname(gs.scene_path("bunny"))
seed = int(sys.argv[1])
doc, hetero = helpers.random_scene_r2(2000 + seed)
def compare(doc, tag):
    scene = gs.load_scene_text(json.dumps(doc), SCENE_DIR)
    o = ob.Oracle(scene)
    samples = o.native_samples(99 + seed)
    li_ref, _ = o.li_replay(samples, threads=8)
    li = HipPathTracer(scene, 0).render(seed=99 + seed, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    bad = np.flatnonzero(np.any(li != li_ref, axis=1))
    print(tag, "differ", bad.size, "of", li.shape[0])
    kinds = {}
    for i in bad[:400]:
        r = o.camera_ray(float(samples[i, 0]), float(samples[i, 1]))
        h = o.intersect(r[0:3], r[3:6])
        inst = int(h[11]) if h is not None and h[0] >= 0 else -1
        mt = scene.desc.materials[scene.desc.instances[inst].material].type if inst >= 0 else -1
        kinds[(inst, mt)] = kinds.get((inst, mt), 0) + 1
    print("   first-hit (instance, material type) of differing samples:", kinds)
    for i in bad[:3]:
        print("   ", i, li[i, :3], li_ref[i, :3])
    return bad.size
compare(doc, "full")
d2 = copy.deepcopy(doc); d2.pop("volume", None)
compare(d2, "no volume")
d3 = copy.deepcopy(doc)
for m in d3["materials"]:
    m.pop("bumpmap", None); m.pop("normalmap", None)
compare(d3, "no bump")
print("lights", json.dumps(doc["lights"]))
d4 = copy.deepcopy(doc)
d4["materials"] = [m if m["type"] not in ("mask", "subsurface") else {"name": m["name"], "type": "lambert", "Kd": "c0"} for m in d4["materials"]]
compare(d4, "mask / sss -> lambert")
d5 = copy.deepcopy(doc)
d5["lights"] = [l for l in d5["lights"] if l["type"] != "area"]
compare(d5, "no area light")
d6 = copy.deepcopy(doc)
for l in d6["lights"]:
    if l["type"] == "area": l["geometry"] = "quad"
compare(d6, "area light on quad")
d7 = copy.deepcopy(doc)
d7["primitives"] = [p for p in d7["primitives"] if not (p["type"] == "instance" and p["name"] != "floor")]
compare(d7, "floor only")
