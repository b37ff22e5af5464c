"""Debug: do the device's sqrtf / division / reciprocal round like IEEE (numpy float32)?"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
tr = HipPathTracer(gs.load_scene("bunny", gs.config_overrides(resolution=(16, 16), spp=1, depth=2)), 0)
rng = np.random.default_rng(1)
n = 20_000_000
for lo, hi in ((0.0, 1.0), (1e-6, 1e-3), (1.0, 1000.0), (1e-30, 1e-20)):
    a = rng.uniform(lo, hi, n).astype(np.float32) + np.float32(1e-38)
    b = rng.uniform(lo, hi, n).astype(np.float32) + np.float32(1e-38)
    ad, bd = torch.from_numpy(a).to(tr.device), torch.from_numpy(b).to(tr.device)
    od = torch.empty((n, 4), dtype=torch.float32, device=tr.device)
    assert tr.lib.gbl_selftest_arith(tr.handle, ad.data_ptr(), bd.data_ptr(), od.data_ptr(), n) == 0
    o = od.cpu().numpy()
    inv = np.float32(1.0) / np.sqrt(a * a + b * b + np.float32(0.25))
    print("range", lo, hi, "sqrt mism", int((o[:, 0] != np.sqrt(a)).sum()), "div mism", int((o[:, 1] != a / b).sum()),
          "rcp mism", int((o[:, 2] != np.float32(1.0) / a).sum()), "normalize.x mism", int((o[:, 3] != a * inv).sum()))
