"""bench.py --gpus 2 on ONE GPU: two ranks under GBL_DIST_BACKEND=gloo (RCCL refuses two ranks on one device) shard the frame's
8x8 sample tiles, each traces its share on the device, one all-reduce(sum) of the film accumulators -- the reduced film must be
the film one rank renders alone (Film::mergeTile's sum over workers, /root/reference/src/GoblinFilm.cpp:140-153).  The N > 1 line's
fields are checked too: per-rank schedules (all equal, or bench.py exits), tile balance, the collective's description."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from goblin_amd import scene as gs

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_gpus_2_reduces_to_the_one_gpu_film(tmp_path):
    import torch
    assert torch.cuda.is_available()
    dump = str(tmp_path / "film_n2.npy")
    env = dict(os.environ, GBL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--resolution", "256", "256", "--spp", "16", "--dump-film", dump]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["collective"]["backend"].startswith("gloo") and line["collective"]["ranks"] == 2
    assert len(set(line["per_rank"]["schedule"])) == 1 and line["per_rank"]["schedule"][0] in ("megakernel", "wavefront")
    assert sum(line["per_rank"]["paths"]) == line["config"]["paths_per_step"]
    assert line["per_rank"]["trace_imbalance"] < 0.5
    film2 = np.load(dump)
    # the same frame on one rank, in this process: grid.json (the N > 1 default workload) at the same size and seed
    from goblin_amd.renderer import HipPathTracer
    scene = gs.load_scene("grid", gs.config_overrides(resolution=(256, 256), spp=16, depth=8))
    film1 = HipPathTracer(scene, 0).render(seed=20261003)["film"].numpy()
    assert film1.shape == film2.shape
    w1, w2 = film1[..., 3], film2[..., 3]
    np.testing.assert_allclose(w2, w1, rtol=1e-5, atol=1e-6)
    rel = float(np.linalg.norm(film2[..., :3].astype(np.float64) - film1[..., :3]) / np.linalg.norm(film1[..., :3].astype(np.float64)))
    print("two-rank film vs one-rank film: relL2", rel)
    assert rel <= 1e-5
