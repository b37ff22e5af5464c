"""For the differing samples of a pixel block: replay every scene query the oracle issued on the device, report the first disagreement.
    python tools/block_rays.py cornell 1024 1024 16 100 500 8 8"""
import sys, os, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
name, res, spp, depth, bx, by, bw, bh = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
seed = 20261003
scene = gs.load_scene(name, gs.config_overrides(resolution=(res, res), spp=spp, depth=depth))
tr = HipPathTracer(scene, 0)
x0, x1, y0, y1 = tr.window
sub = (x0 + bx, x0 + bx + bw, y0 + by, y0 + by + bh)
o = ob.Oracle(scene)
samples = o.native_samples(seed, window=sub)
li_ref, _ = o.li_replay(samples, threads=8)
li = tr.render(seed=seed, window=sub, want_li=True, replay_samples=samples, schedule="megakernel")["li"].cpu().numpy()
bad = np.nonzero((li != li_ref).any(axis=1))[0]
print("samples", li.shape[0], "differing", bad.size)
L = ob.lib()
L.orc_debug_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
np.set_printoptions(precision=9, floatmode="unique")
for b in bad[:4]:
    rec = np.ascontiguousarray(samples[b])
    out = np.zeros((512, 16), np.float32)
    n = L.orc_debug_rays(o.h, C.byref(scene.desc.setting), rec.ctypes.data, out.ctypes.data, 512)
    rays = np.ascontiguousarray(out[:n, :9])
    rd = torch.from_numpy(rays).to(tr.device)
    od = torch.zeros((n, 8), dtype=torch.float32, device=tr.device)
    assert tr.lib.gbl_selftest_trace(tr.handle, rd.data_ptr(), od.data_ptr(), n) == 0
    dev = od.cpu().numpy()
    print("sample", b, "li dev", li[b, :3], "ref", li_ref[b, :3], "queries", n)
    for i in range(n):
        agree = dev[i, 0] == out[i, 9] and (out[i, 0] != 0.0 or out[i, 9] < 0 or np.array_equal(dev[i, 2:8], out[i, 10:16]))
        print("   query", i, "kind", int(out[i, 0]), "oracle", out[i, 9], "device", dev[i, 0], "inst", dev[i, 1], "OK" if agree else "DIFFERS",
              "o", out[i, 1:4], "d", out[i, 4:7], "mint", out[i, 7], "maxt", out[i, 8])
        if not agree:
            print("      frame oracle", out[i, 10:16], "\n      frame device", dev[i, 2:8])
