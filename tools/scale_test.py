"""Scale check: a procedurally generated blob of N million triangles (written as OBJ under /tmp on the GPU box, not
committed), loaded through the host front end, built with the host SAH builder and with the device LBVH builder,
and traced.  Reports load / build times, tree depth, device memory and trace time."""
import sys, os, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np

NU, NV = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 1002)


def blob(nu, nv):
    j = np.arange(1, nv - 1)[:, None]
    i = np.arange(nu)[None, :]
    theta = np.pi * j / (nv - 1)
    phi = 2.0 * np.pi * i / nu
    st = np.sin(theta)
    r = 1.0 + 0.10 * st * st * np.sin(3 * phi + 0.5) * np.sin(2 * theta) + 0.05 * st * np.sin(7 * phi) * np.sin(5 * theta) \
        + 0.02 * st * np.sin(41 * phi) * np.sin(37 * theta)
    x, y, z = r * st * np.cos(phi), r * np.cos(theta) * np.ones_like(phi), r * st * np.sin(phi)
    ring = np.stack([x, y, z], -1).reshape(-1, 3)
    verts = np.concatenate([[[0, 1.0, 0]], ring, [[0, -1.0, 0]]]).astype(np.float32) * 0.08 + np.array([0, 0.11, 0], np.float32)
    faces = []
    a = 1 + (np.arange(nv - 3)[:, None] * nu + np.arange(nu)[None, :])
    b = 1 + (np.arange(nv - 3)[:, None] * nu + (np.arange(nu)[None, :] + 1) % nu)
    c, d = a + nu, b + nu
    quads = np.concatenate([np.stack([a, c, b], -1).reshape(-1, 3), np.stack([b, c, d], -1).reshape(-1, 3)])
    top = np.stack([np.zeros(nu, int), 1 + (np.arange(nu) + 1) % nu, 1 + np.arange(nu)], -1)
    last = len(verts) - 1
    base = 1 + (nv - 3) * nu
    bot = np.stack([np.full(nu, last), base + np.arange(nu), base + (np.arange(nu) + 1) % nu], -1)
    return verts, np.concatenate([top, quads, bot])


t0 = time.time()
verts, faces = blob(NU, NV)
path = "/tmp/blob_%dx%d.obj" % (NU, NV)
with open(path, "w") as f:
    np.savetxt(f, verts, fmt="v %.6f %.6f %.6f")
    np.savetxt(f, faces + 1, fmt="f %d %d %d")
gen_s = time.time() - t0
doc = json.load(open(os.path.join(REPO, "goblin_amd", "scenes", "bunny.json")))
for g in doc["geometries"]:
    if g["name"] == "bunny":
        g["file"] = path
    elif "file" in g:
        g["file"] = os.path.join(REPO, "goblin_amd", "scenes", g["file"])
doc["camera"]["film"]["resolution"] = [512, 512]
doc["render_setting"].update({"sample_per_pixel": 16, "max_ray_depth": 8})
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
import torch
t0 = time.time()
scene = gs.load_scene_text(json.dumps(doc), "/tmp")
load_s = time.time() - t0
print(json.dumps({"triangles": int(len(faces)), "obj_MB": round(os.path.getsize(path) / 1e6, 1), "generate_s": round(gen_s, 1),
                  "load_s": round(load_s, 2)}), flush=True)
for bvh in ("device", "host"):
    for sch in ("megakernel", "wavefront"):
        try:
            t0 = time.time()
            tr = HipPathTracer(scene, 0, bvh=bvh)
            create_s = time.time() - t0
            film = tr.new_film()
            best = 1e30
            for i in range(2):
                film.zero_()
                out = tr.render(film=film, seed=1, timed=True, schedule=sch)
                torch.cuda.synchronize()
                best = min(best, out["stats"]["kernel_ms"])
            print(json.dumps({"bvh": bvh, "schedule": sch, "build_ms": round(tr.info.build_ms, 1), "create_s": round(create_s, 2),
                              "blas_nodes": tr.info.blas_nodes, "blas_depth": tr.info.blas_depth, "scene_MB": round(tr.info.scene_bytes / 1e6, 1),
                              "trace_ms": round(best, 2), "mpaths_s": round(out["paths"] / best / 1e3, 1),
                              "mean": float(film.normalized().mean())}), flush=True)
            del tr
        except Exception as e:
            print(json.dumps({"bvh": bvh, "schedule": sch, "error": str(e)[:300]}), flush=True)
os.remove(path)
