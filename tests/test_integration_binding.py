"""The reference-side binding as code that runs (tests/integration/): `flattenSceneForHip` reads a scene the
REFERENCE's own ContextLoader built -- its PolygonMesh / Model / InstancedPrimitive / Material / Light / Camera / Film objects --
into a gbl_scene_desc.  Here, without a GPU: that description must describe the same scene as the one this repository's loader
(libgoblin_host.so, the host-side mirror of ContextLoader) derives from the same file -- meshes, instances, materials, lights,
camera, film, filter.  (The reference keeps no instance order -- Scene holds them in its BVH's order -- so instances are matched
by content.)  tests/test_gpu_integration.py then renders through the binding."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import integration_helpers as ih
from goblin_amd import _abi
from goblin_amd import scene as gs

pytestmark = pytest.mark.skipif(not os.path.exists(ih.HIPBIND), reason="oracle/_ref/g_ray_hipbind is built where /root/reference exists")


def mesh_key(desc_arrays, m):
    pos, nrm, uv, idx = desc_arrays
    v0, v1, t0, t1 = m.vertex_offset, m.vertex_offset + m.vertex_count, m.tri_offset, m.tri_offset + m.tri_count
    return (m.shape, m.has_normal, m.has_uv, pos[3 * v0:3 * v1].tobytes(), nrm[3 * v0:3 * v1].tobytes(), uv[2 * v0:2 * v1].tobytes(), idx[3 * t0:3 * t1].tobytes())


def material_key(m):
    key = [m.type, tuple(m.color)]
    if m.type == _abi.GBL_MAT_TRANSPARENT:
        key += [tuple(m.color2), m.index]
    elif m.type == _abi.GBL_MAT_MIRROR:
        key += [m.index, m.k]
    elif m.type == _abi.GBL_MAT_BLINN:
        # a Blinn lobe is a conductor iff k > 0 (createBlinnMaterial, GoblinMaterial.cpp:835-851); the file's missing "k" reaches
        # the two descriptions as different non-positive defaults
        key += [m.index, m.k if m.k > 0 else "dielectric"]
    if m.type == _abi.GBL_MAT_BLINN:
        key += [m.exponent]
    key += [m.tex_color, m.tex_color2, m.tex_exponent, m.masked_material, m.tex_bump, m.tex_normal]
    return tuple(key)


def trs_key(t):
    return (tuple(t.position), tuple(t.orientation), tuple(t.scale))


@pytest.mark.parametrize("scene,overrides", [("bunny", gs.config_overrides(resolution=(256, 256), spp=16, depth=4)),
                                              ("cornell", gs.config_overrides(resolution=(48, 48), spp=16, depth=6)),
                                              ("grid", gs.config_overrides(resolution=(48, 48), spp=4, depth=5))])
def test_flattened_reference_objects_describe_the_loaders_scene(tmp_path, scene, overrides):
    js = str(tmp_path / (scene + ".json"))
    ih.write_scene(scene, overrides, js)
    dump = str(tmp_path / "desc.bin")
    p = subprocess.run([ih.HIPBIND, js, "--dump-desc", dump], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    raw = ih.read_desc_dump(dump)
    ours_scene = gs.load_scene(js)   # (owns the arrays the description points into)
    ours = ours_scene.desc

    def arrays_of_dump():
        return (np.frombuffer(raw["positions"], np.float32), np.frombuffer(raw["normals"], np.float32), np.frombuffer(raw["uvs"], np.float32),
                np.frombuffer(raw["indices"], np.uint32))

    def arrays_of_desc(d):
        nv, nt = d.num_vertices, d.num_triangles
        return (np.ctypeslib.as_array(d.positions, (3 * nv,)), np.ctypeslib.as_array(d.normals, (3 * nv,)), np.ctypeslib.as_array(d.uvs, (2 * nv,)),
                np.ctypeslib.as_array(d.indices, (3 * nt,)))

    b_meshes = ih.structs(raw["meshes"], _abi.gbl_mesh)
    b_mats = ih.structs(raw["materials"], _abi.gbl_material)
    b_inst = ih.structs(raw["instances"], _abi.gbl_instance)
    b_lights = ih.structs(raw["lights"], _abi.gbl_light)
    assert len(b_inst) == ours.num_instances and len(b_lights) == ours.num_lights

    # lights: same order (SceneCache::getLights) and content
    for i, bl in enumerate(b_lights):
        ol = ours.lights[i]
        assert bl.type == ol.type and tuple(bl.color) == tuple(ol.color)
        if bl.type in (_abi.GBL_LIGHT_POINT, _abi.GBL_LIGHT_SPOT):
            assert tuple(bl.position) == tuple(ol.position)
        if bl.type == _abi.GBL_LIGHT_SPOT:
            # the binding reads the direction the constructor normalised (once more than the file's): same direction to a rounding
            a, b = np.array(bl.direction), np.array(ol.direction)
            np.testing.assert_allclose(a, b / np.linalg.norm(b), rtol=0, atol=2e-7)
            assert bl.cos_theta_max == ol.cos_theta_max and bl.cos_falloff_start == ol.cos_falloff_start
        if bl.type == _abi.GBL_LIGHT_AREA:
            assert trs_key(bl.to_world) == trs_key(ol.to_world) and bl.sample_num == ol.sample_num
            assert mesh_key(arrays_of_dump(), b_meshes[bl.mesh]) == mesh_key(arrays_of_desc(ours), ours.meshes[ol.mesh])

    # instances, matched by content: (mesh geometry, material, transform, which light they emit for)
    def inst_key(arrays, meshes, mats, inst):
        return (mesh_key(arrays, meshes[inst.mesh]), material_key(mats[inst.material]), trs_key(inst.to_world), inst.area_light)

    theirs = sorted(repr(inst_key(arrays_of_dump(), b_meshes, b_mats, x)) for x in b_inst)
    mine = sorted(repr(inst_key(arrays_of_desc(ours), ours.meshes, ours.materials, ours.instances[i])) for i in range(ours.num_instances))
    assert theirs == mine

    cam_b, cam_o = _abi.gbl_camera.from_buffer_copy(raw["camera"]), ours.camera
    for field in ("position", "orientation"):
        assert tuple(getattr(cam_b, field)) == tuple(getattr(cam_o, field)), field
    for field in ("fov_degrees", "near_plane", "far_plane", "lens_radius", "type"):
        assert getattr(cam_b, field) == getattr(cam_o, field), field
    film_b, film_o = _abi.gbl_film.from_buffer_copy(raw["film"]), ours.film
    assert (film_b.xres, film_b.yres, tuple(film_b.crop), film_b.filter_type, tuple(film_b.filter_width)) == \
           (film_o.xres, film_o.yres, tuple(film_o.crop), film_o.filter_type, tuple(film_o.filter_width))
    if film_b.filter_type == _abi.GBL_FILTER_GAUSSIAN:
        assert film_b.gaussian_falloff == film_o.gaussian_falloff
    set_b, set_o = _abi.gbl_render_setting.from_buffer_copy(raw["setting"]), ours.setting
    assert (set_b.integrator, set_b.sample_per_pixel, set_b.max_ray_depth, set_b.bssrdf_sample_num) == \
           (set_o.integrator, set_o.sample_per_pixel, set_o.max_ray_depth, set_o.bssrdf_sample_num)


def test_binding_header_is_what_integration_md_quotes():
    """INTEGRATION.md shows the binding; the file that is compiled and run is tests/integration/GoblinHipPathtracer.h: the document
    must quote it, not a copy that drifts."""
    with open(os.path.join(ih.REPO, "INTEGRATION.md")) as f:
        md = f.read()
    with open(os.path.join(ih.REPO, "tests", "integration", "GoblinHipPathtracer.h")) as f:
        code = f.read()
    body = code[code.index("namespace Goblin {"):code.rindex("#endif")].strip()
    assert body in md
