"""Which blocks of a persistent 768-block grid (45 KB of LDS each, three per CU) share a CU / an XCD: gbl_selftest_placement."""
import sys, os, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
t = HipPathTracer(gs.load_scene("bunny", gs.config_overrides(resolution=(32, 32), spp=1, depth=2)), 0)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 768
out = np.zeros(2 * blocks, np.uint32)
assert t.lib.gbl_selftest_placement(t.handle, blocks, 46080, out.ctypes.data) == 0
xcc, hw = out[0::2] & 0xf, out[1::2]
cu_key = (hw >> 8) & 0xff      # CU_ID [11:8], SH_ID [12], SE_ID [15:13] on gfx9
print("xcc of blocks 0..23:", xcc[:24].tolist())
print("hw_id bits [15:8] of blocks 0..23:", cu_key[:24].tolist())
groups = collections.defaultdict(list)
for b in range(blocks):
    groups[(int(xcc[b]), int(cu_key[b]))].append(b)
sizes = collections.Counter(len(v) for v in groups.values())
print("distinct (xcc, cu) keys:", len(groups), "blocks per key:", dict(sizes))
for k in list(groups)[:12]:
    print(k, groups[k])
