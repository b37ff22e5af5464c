"""Debug: stream-mode per-sample radiance of one tile row band against the oracle's stream render."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob, helpers
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
scene = gs.load_scene("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8))
o = ob.Oracle(scene)
res = o.render(threads=16, want_samples=True)
tr = HipPathTracer(scene, 0)
x0, x1, y0, y1 = o.window()
S = scene.spp(); W = x1 - x0
idx = helpers.tile_order_index((x0, x1, y0, y1), S)
band = 214
sel = idx[(band - y0) * W * S:(band + 8 - y0) * W * S]
out = tr.render(window=(x0, x1, band, band + 8), sampler="stream", want_li=True)
li = out["li"].cpu().numpy()
ref_li = res["li"][sel]
d = np.nonzero((li[:, :3] != ref_li[:, :3]).any(axis=1))[0]
print("differing samples in the band:", d.size)
for b in d[:6]:
    pix = b // S
    print("pixel", (x0 + pix % W, band + pix // W), "sample", b % S, "dev", li[b, :3], "ref", ref_li[b, :3])
if d.size:
    pix = d // S
    px = x0 + pix % W; py = band + pix // W
    print("pixels affected:", sorted(set(zip(px.tolist(), py.tolist())))[:12])
    # the first affected pixel's predecessor in tile order is where the draw count went wrong: print its paths' li
    first = d[0]; p0 = first // S
    prev = p0 - 1
    print("previous pixel", (x0 + prev % W, band + prev // W))
    # replay that previous pixel's oracle records and report draw-relevant facts
    recs = np.ascontiguousarray(res["samples"][sel][prev * S:(prev + 1) * S])
    import ctypes as C
    L = ob.lib(); L.orc_debug_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    for k in range(S):
        q = np.zeros((256, 16), np.float32)
        n = L.orc_debug_rays(o.h, C.byref(scene.desc.setting), recs[k].ctypes.data, q.ctypes.data, 256)
        rd = torch.from_numpy(np.ascontiguousarray(q[:n, :9])).to(tr.device)
        od = torch.zeros((n, 8), dtype=torch.float32, device=tr.device)
        tr.lib.gbl_selftest_trace(tr.handle, rd.data_ptr(), od.data_ptr(), n)
        dev = od.cpu().numpy()
        bad = [i for i in range(n) if dev[i, 0] != q[i, 9]]
        if bad:
            i = bad[0]
            print("  sample", k, "query", i, "kind", int(q[i, 0]), "oracle", q[i, 9], "device", dev[i, 0], "o", q[i, 1:4], "d", q[i, 4:7], "mint", q[i, 7], "maxt", q[i, 8])
