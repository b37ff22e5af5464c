"""A fuzz scene whose stream-mode Film left the reference's (tools/fuzz_sweep_gpu.py): which pixels, and which of the oracle's own
stream samples the device renders differently when it replays them.
    python tools/fuzz_stream_hunt.py <seed> <0 | 1: whitted>"""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob, helpers
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
seed, whitted = int(sys.argv[1]), bool(int(sys.argv[2]))
SCENE_DIR = os.path.dirname(gs.scene_path("bunny"))
doc, hetero = helpers.random_scene_r2(2000 + seed, whitted)
scene = gs.load_scene_text(json.dumps(doc), SCENE_DIR)
print("seed", seed, "whitted", whitted, "volume", doc.get("volume", {}).get("type"), "lights", [(l["type"], l.get("geometry")) for l in doc["lights"]],
      "spp", scene.spp(), "integrator", doc["render_setting"], flush=True)
o = ob.Oracle(scene)
res = o.render(threads=1, want_samples=True)
tr = HipPathTracer(scene, 0)
film = tr.render(sampler="stream")["film"].numpy()
ref = res["film"]
print("relL2", helpers.rel_l2(ob.normalize_film(film), ob.normalize_film(ref)))
wd = np.abs(film[..., 3] - ref[..., 3]) / np.maximum(ref[..., 3], 1e-9)
ys, xs = np.nonzero(wd > 1e-4)
print("pixels whose weight differs (other samples):", len(ys), list(zip(xs.tolist(), ys.tolist()))[:12], flush=True)
cd = np.abs(film[..., :3] - ref[..., :3]).max(axis=2)
ys, xs = np.nonzero(cd > 1e-4 * max(1e-6, float(np.abs(ref[..., :3]).max())))
print("pixels whose colour differs:", len(ys), list(zip(xs.tolist(), ys.tolist()))[:12], flush=True)
x0, x1, y0, y1 = o.window()
S = scene.spp()
idx = helpers.tile_order_index((x0, x1, y0, y1), S)
recs = np.ascontiguousarray(res["samples"][idx])
li = tr.render(replay_samples=recs, want_li=True)["li"].cpu().numpy()
ref_li = res["li"][idx]
d = np.nonzero((li[:, :3] != ref_li[:, :3]).any(axis=1))[0]
W = x1 - x0
print("replayed samples that differ:", d.size)
for b in d[:8]:
    pix = b // S
    print("  pixel", (x0 + pix % W, y0 + pix // W), "tile", ((pix % W) // 8, (pix // W) // 8), "sample", b % S, "dev", li[b, :3], "ref", ref_li[b, :3])
print("filter", doc["camera"]["film"].get("filter"), "resolution", doc["camera"]["film"].get("resolution"), "crop", doc["camera"]["film"].get("crop"))
wd = np.abs(film[..., 3] - ref[..., 3]) / np.maximum(ref[..., 3], 1e-9)
ys, xs = np.nonzero(wd > 1e-4)
for y, x in list(zip(ys.tolist(), xs.tolist()))[:5]:
    print("  pixel", (x, y), "device", film[y, x], "reference", ref[y, x])
nf, nr = ob.normalize_film(film), ob.normalize_film(ref)
e = np.abs(nf - nr).max(axis=2)
ys, xs = np.unravel_index(np.argsort(e.ravel())[-3:], e.shape)
for y, x in zip(ys.tolist(), xs.tolist()):
    print("  worst normalised pixel", (x, y), "device", nf[y, x], "reference", nr[y, x], "accumulators", film[y, x], ref[y, x])
