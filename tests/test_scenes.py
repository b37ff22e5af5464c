import hashlib
import os

import pytest

from goblin_amd import scene as gs
from goblin_amd.scenes import make_meshes


def sha(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def test_mesh_generator_reproduces_the_committed_models(tmp_path):
    make_meshes.main(str(tmp_path))
    committed = os.path.join(gs.SCENE_DIR, "models")
    for f in sorted(os.listdir(committed)):
        assert sha(os.path.join(committed, f)) == sha(str(tmp_path / f)), f


@pytest.mark.parametrize("name,tris,instances,lights", [("bunny", 69122, 2, 1), ("cornell", 69122 + 12, 9, 1), ("grid", 69122, 16, 2)])
def test_bundled_scenes_load(name, tris, instances, lights):
    s = gs.load_scene(name)
    assert s.desc.num_triangles == tris and s.desc.num_instances == instances and s.desc.num_lights == lights
    if name == "grid":   # config 4: ~1.04 M instanced triangles
        assert sum(s.desc.meshes[s.desc.instances[i].mesh].tri_count for i in range(instances)) == 15 * 69120 + 2
