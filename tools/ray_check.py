"""Debug: for the samples whose radiance differs between device and oracle, replay every scene query the oracle
issued for them on the device (gbl_selftest_trace) and report the first query that disagrees."""
import sys, os, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
res, spp, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = gs.load_scene(sys.argv[4] if len(sys.argv) > 4 else "bunny", gs.config_overrides(resolution=(res, res), spp=spp, depth=depth))
o = ob.Oracle(scene)
seed = 20261003
samples = o.native_samples(seed)
li_ref, _ = o.li_replay(samples, threads=8)
tr = HipPathTracer(scene, 0)
li = tr.render(seed=seed, want_li=True)["li"].cpu().numpy()
d = np.abs(li[:, :3] - li_ref[:, :3]).max(axis=1)
bad = np.nonzero(d > 0)[0]
print("samples", d.size, "differing", bad.size)
L = ob.lib()
L.orc_debug_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
for b in bad[:8]:
    rec = np.ascontiguousarray(samples[b])
    out = np.zeros((256, 16), np.float32)
    n = L.orc_debug_rays(o.h, C.byref(scene.desc.setting), rec.ctypes.data, out.ctypes.data, 256)
    rays = np.ascontiguousarray(out[:n, :9])
    rd = torch.from_numpy(rays).to(tr.device)
    od = torch.zeros((n, 8), dtype=torch.float32, device=tr.device)
    assert tr.lib.gbl_selftest_trace(tr.handle, rd.data_ptr(), od.data_ptr(), n) == 0
    dev = od.cpu().numpy()
    print("sample", b, "li dev", li[b, :3], "ref", li_ref[b, :3], "queries", n)
    for i in range(n):
        if dev[i, 0] != out[i, 9] or (out[i, 0] == 0.0 and out[i, 9] >= 0 and not np.array_equal(dev[i, 2:8], out[i, 10:16])):
            print("   query", i, "kind", int(out[i, 0]), "oracle", out[i, 9], "device", dev[i, 0], "inst", dev[i, 1],
                  "o", out[i, 1:4], "d", out[i, 4:7], "mint", out[i, 7], "maxt", out[i, 8],
                  "\n      frame oracle", out[i, 10:16], "\n      frame device", dev[i, 2:8])
            break
    else:
        print("   every query agrees")
    # spot-light cone test of every shadow query (kind 1): how close to the cone's edges is it?
    for li_ in range(scene.desc.num_lights):
        lt = scene.desc.lights[li_]
        if lt.type != 2: continue
        ax = np.array(list(lt.direction), np.float64); ax /= np.linalg.norm(ax)
        for i in range(n):
            if out[i, 0] == 1.0:
                c = float(np.dot(-out[i, 4:7].astype(np.float64), ax))
                print("   shadow query", i, "cos_t - cos_max %.3e  cos_t - cos_falloff %.3e  occluded(oracle) %d" % (c - lt.cos_theta_max, c - lt.cos_falloff_start, int(out[i, 9])))
