"""Oracle against the compiled reference (oracle/_ref/ref_harness) on many round-2 fuzz scenes, beyond the seeds the test suite runs:
    python tools/fuzz_sweep_reference.py <first seed> <last seed + 1>"""
import sys, json, os, subprocess, tempfile
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,REPO); sys.path.insert(0,os.path.join(REPO,'tests'))
import numpy as np
import helpers, oracle_binding as ob
from goblin_amd import scene as gs
SCENE_DIR=os.path.dirname(gs.scene_path("bunny")); HARNESS=os.path.join(REPO,'oracle','_ref','ref_harness')
bad=[]
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
  for whitted in (False, True):
    doc,_=helpers.random_scene_r2(2000+seed, whitted)
    doc["render_setting"]["thread_num"]=1
    ref_doc=json.loads(json.dumps(doc))
    for section in ("geometries","textures","lights"):
        for g in ref_doc[section]:
            if "file" in g: g["file"]=os.path.join(SCENE_DIR,g["file"])
    if "density_grid" in ref_doc.get("volume",{}): ref_doc["volume"]["density_grid"]=os.path.join(SCENE_DIR,ref_doc["volume"]["density_grid"])
    with tempfile.TemporaryDirectory() as tmp:
        jp=os.path.join(tmp,"s.json"); json.dump(ref_doc,open(jp,"w")); pre=os.path.join(tmp,"o")
        try:
            meta=json.loads(subprocess.check_output([HARNESS,"li",jp,pre,"1","100000"],stderr=subprocess.DEVNULL,timeout=300).decode())
        except Exception as e:
            print(seed, whitted, "harness failed", e); continue
        samples=np.fromfile(pre+".samples.f32",np.float32).reshape(-1,meta["dims"]); li_ref=np.fromfile(pre+".li.f32",np.float32).reshape(-1,4)
        film_ref=np.fromfile(pre+".film.f32",np.float32).reshape(meta["yres"],meta["xres"],4)
    sc=gs.load_scene_text(json.dumps(doc),SCENE_DIR); o=ob.Oracle(sc)
    film=o.render(threads=1)["film"]
    ok_f=np.array_equal(film,film_ref)
    ok_l=True
    if "volume" not in doc:
        li,_=o.li_replay(samples,threads=8); ok_l=np.array_equal(li,li_ref)
    if not (ok_f and ok_l): bad.append((seed,whitted)); print("MISMATCH", seed, whitted, "film", ok_f, "li", ok_l, flush=True)
print("swept", sys.argv[1], sys.argv[2], "bad", bad)
