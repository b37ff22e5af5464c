import sys, os, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
gdir = os.path.join(REPO, 'tests', 'golden')
manifest = json.load(open(os.path.join(gdir, 'manifest.json')))
case = sys.argv[1] if len(sys.argv) > 1 else 'bunny_pt_d8'
meta = manifest[case]; data = dict(np.load(os.path.join(gdir, case + '.npz')))
scene = gs.load_scene(meta['scene'], meta['overrides'])
r = HipPathTracer(scene, 0)
samples = data['samples']; spp = meta['spp']
n = (samples.shape[0] // spp) * spp
samples = samples[:n]; ref = data['li'][:n]
x0, x1, y0, y1 = r.window
npx = n // spp
for rows in range(1, 100):
    if npx % rows == 0 and npx // rows <= x1 - x0:
        win = (x0, x0 + npx // rows, y0, y0 + rows); break
print('window', win, 'records', n, 'spp', spp)
mk = r.render(window=win, replay_samples=samples, want_li=True, schedule='megakernel')['li'].cpu().numpy()
wf = r.render(window=win, replay_samples=samples, want_li=True, schedule='wavefront')['li'].cpu().numpy()
bad = np.where(np.any(np.abs(wf[:, :3] - mk[:, :3]) > 1e-6 + 1e-4 * np.abs(mk[:, :3]), axis=1))[0]
print('mk vs ref max', np.abs(mk - ref).max(), 'wf vs mk mismatches', len(bad))
print(bad[:64])
for i in bad[:10]:
    print(i, 'mk', mk[i], 'wf', wf[i], 'ref', ref[i])
