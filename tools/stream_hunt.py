"""Debug: find the samples where the device's stream-mode render leaves the reference's stream at the full config-2
size.  The oracle renders the whole frame with the reference's stream and keeps every record (25 GB of host memory at
256 spp); the device replays them; differing samples get the ray_check treatment."""
import sys, os, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob, helpers
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene = gs.load_scene("bunny", gs.config_overrides(resolution=(512, 512), spp=spp, depth=8))
o = ob.Oracle(scene)
res = o.render(threads=16, want_samples=True)
print("oracle rendered", res["li"].shape[0], "samples in %.1f s" % res["seconds"], flush=True)
tr = HipPathTracer(scene, 0)
film = tr.render(sampler="stream")["film"].numpy()
ref = res["film"]
wd = np.abs(film[..., 3] - ref[..., 3]) / np.maximum(ref[..., 3], 1e-9)
ys, xs = np.nonzero(wd > 1e-4)
print("pixels on other samples:", len(ys), list(zip(xs.tolist(), ys.tolist()))[:10], flush=True)
# replay the oracle's records tile row by tile row and look for radiance that differs
x0, x1, y0, y1 = o.window()
S = scene.spp()
idx = helpers.tile_order_index((x0, x1, y0, y1), S)       # pixel-major -> tile-order record index
W = x1 - x0
L = ob.lib()
L.orc_debug_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
bad_total = 0
for band in range(y0, y1, 8):
    b1 = min(band + 8, y1)
    sel = idx[(band - y0) * W * S:(b1 - y0) * W * S]
    recs = np.ascontiguousarray(res["samples"][sel])
    li = tr.render(window=(x0, x1, band, b1), replay_samples=recs, want_li=True)["li"].cpu().numpy()
    ref_li = res["li"][sel]
    d = np.nonzero((li[:, :3] != ref_li[:, :3]).any(axis=1))[0]
    bad_total += d.size
    for b in d[:3]:
        pix = b // S
        print("band", band, "pixel", (x0 + pix % W, band + pix // W), "sample", b % S, "dev", li[b, :3], "ref", ref_li[b, :3], flush=True)
        out = np.zeros((256, 16), np.float32)
        n = L.orc_debug_rays(o.h, C.byref(scene.desc.setting), recs[b].ctypes.data, out.ctypes.data, 256)
        rd = torch.from_numpy(np.ascontiguousarray(out[:n, :9])).to(tr.device)
        od = torch.zeros((n, 8), dtype=torch.float32, device=tr.device)
        tr.lib.gbl_selftest_trace(tr.handle, rd.data_ptr(), od.data_ptr(), n)
        dev = od.cpu().numpy()
        for i in range(n):
            if dev[i, 0] != out[i, 9] or (out[i, 0] == 0.0 and out[i, 9] >= 0 and not np.array_equal(dev[i, 2:8], out[i, 10:16])):
                print("   query", i, "kind", int(out[i, 0]), "oracle", out[i, 9], "device", dev[i, 0], "inst", dev[i, 1], "o", out[i, 1:4], "d", out[i, 4:7],
                      "mint", out[i, 7], "maxt", out[i, 8], "\n      frame oracle", out[i, 10:16], "\n      frame device", dev[i, 2:8], flush=True)
                break
        else:
            print("   every query agrees", flush=True)
print("differing samples:", bad_total)
