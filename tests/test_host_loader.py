"""libgoblin_host.so: same keys, defaults, strictness and error behaviour as the
reference's ContextLoader / ParamSet / PolygonMesh loader for the hot-path subset
(/root/reference/src/GoblinContextLoader.cpp, GoblinParamSet.cpp, GoblinPolygonMesh.cpp)."""
import json
import os

import numpy as np
import pytest

from goblin_amd import _abi
from goblin_amd import scene as gs

MODELS = os.path.join(gs.SCENE_DIR)


def load(doc):
    return gs.load_scene_text(json.dumps(doc), MODELS)


def minimal(**extra):
    doc = {
        "camera": {"position": [0, 1, -4], "fov": 50.0},
        "geometries": [{"name": "q", "type": "mesh", "file": "models/plane.obj"}],
        "textures": [{"name": "w", "type": "constant", "color": [0.5, 0.6, 0.7]}],
        "materials": [{"name": "m", "type": "lambert", "Kd": "w"}],
        "primitives": [{"type": "model", "name": "mq", "geometry": "q", "material": "m"},
                       {"type": "instance", "name": "i0", "model": "mq"}],
        "lights": [{"type": "point", "name": "p", "intensity": [1, 2, 3], "position": [0, 3, 0]}],
    }
    doc.update(extra)
    return doc


def test_defaults_match_the_reference():
    s = load(minimal())
    d = s.desc
    assert d.setting.integrator == _abi.GBL_INTEGRATOR_PATH          # unknown/absent render_method -> path tracing
    assert (d.setting.sample_per_pixel, d.setting.max_ray_depth, d.setting.bssrdf_sample_num, d.setting.ao_sample_num) == (1, 5, 4, 25)
    assert (d.film.xres, d.film.yres) == (512, 512) and list(d.film.crop) == [0, 1, 0, 1]
    assert d.film.filter_type == _abi.GBL_FILTER_GAUSSIAN and list(d.film.filter_width) == [1, 1] and d.film.gaussian_falloff == 2.0
    assert d.camera.fov_degrees == 50.0 and d.camera.near_plane == np.float32(0.1) and d.camera.far_plane == 1000.0
    assert list(d.camera.orientation) == [1, 0, 0, 0] and d.camera.lens_radius == 0.0 and d.camera.focal_distance == 1.0
    inst = d.instances[0]
    assert list(inst.to_world.scale) == [1, 1, 1] and list(inst.to_world.position) == [0, 0, 0] and inst.area_light == -1
    assert d.lights[0].type == _abi.GBL_LIGHT_POINT and list(d.lights[0].color) == [1, 2, 3]
    assert s.sample_window() == (-1, 513, -1, 513)                     # gaussian width 1: film padded by the radius


def test_integer_literals_are_invisible_to_float_params():
    """ParamSet is strictly typed: `"fov": 45` is an int and getFloat("fov") keeps its default (GoblinContextLoader.cpp:40-45)."""
    s = load(minimal(camera={"position": [0, 1, -4], "fov": 45, "filter": {"type": "gaussian", "width": [2, 2], "falloff": 3}}))
    assert s.desc.camera.fov_degrees == 60.0
    assert s.desc.film.gaussian_falloff == 2.0 and list(s.desc.film.filter_width) == [2, 2]   # arrays DO take ints
    s = load(minimal(render_setting={"sample_per_pixel": 16.0, "max_ray_depth": 0}))
    assert s.desc.setting.sample_per_pixel == 1                        # float literal invisible to getInt
    assert s.desc.setting.max_ray_depth == 1                           # max(1, ...)


def test_render_methods():
    s = load(minimal(render_setting={"render_method": "ao"}))          # keep the Scene alive: desc points into it
    assert s.desc.setting.integrator == _abi.GBL_INTEGRATOR_AO
    s = load(minimal(render_setting={"render_method": "no such thing"}))
    assert s.desc.setting.integrator == _abi.GBL_INTEGRATOR_PATH
    s = load(minimal(render_setting={"render_method": "whitted"}))
    assert s.desc.setting.integrator == _abi.GBL_INTEGRATOR_WHITTED
    for m in ("sppm", "bdpt", "light_tracing"):
        with pytest.raises(_abi.GoblinError) as e:
            load(minimal(render_setting={"render_method": m}))
        assert e.value.status == _abi.GBL_ERR_UNSUPPORTED
    with pytest.raises(_abi.GoblinError) as e:   # the shipped reference example selects sppm
        gs.load_scene_text(json.dumps({"render_setting": {"render_method": "sppm"}}), ".")
    assert e.value.status == _abi.GBL_ERR_UNSUPPORTED


def test_bump_and_normal_maps_load():
    """getBumpShaders (GoblinMaterial.cpp:813-824): "bumpmap" names a float texture, "normalmap" a colour texture, on every
    material type but the mask; constant textures become texture nodes too (a constant bump map still swaps the shading
    normal for cross(dpdu, dpdv)); an undefined name is an error like any other texture reference."""
    tex = [{"format": "color", "name": "w", "type": "constant", "color": [1, 1, 1]},
           {"format": "float", "name": "b", "type": "constant", "float": 0.25},
           {"format": "color", "name": "nm", "type": "constant", "color": [0.5, 0.5, 1.0]}]
    s = load(minimal(textures=tex, materials=[{"name": "m", "type": "lambert", "Kd": "w", "bumpmap": "b", "normalmap": "nm"}]))
    m = s.desc.materials[0]
    assert m.tex_bump >= 0 and m.tex_normal >= 0 and m.tex_bump != m.tex_normal
    tb, tn = s.desc.textures[m.tex_bump], s.desc.textures[m.tex_normal]
    assert tb.type == _abi.GBL_TEX_CONSTANT and tb.is_float == 1 and tb.value[0] == 0.25
    assert tn.type == _abi.GBL_TEX_CONSTANT and tn.is_float == 0 and list(tn.value) == [0.5, 0.5, 1.0]
    s = load(minimal(textures=tex, materials=[{"name": "m", "type": "lambert", "Kd": "w"}]))
    assert s.desc.materials[0].tex_bump == -1 and s.desc.materials[0].tex_normal == -1
    with pytest.raises(_abi.GoblinError):
        load(minimal(textures=tex, materials=[{"name": "m", "type": "lambert", "Kd": "w", "bumpmap": "missing"}]))


def test_image_textures_and_image_based_light():
    """getImageTextureParams (GoblinTexture.cpp:677-732) and createImageBasedLight (GoblinLight.cpp:681-691): defaults,
    fallbacks, the pyramid MIPMap's constructor builds (48 x 32 -> 64 x 32, 7 levels), per-format texel layout, the cache."""
    tex = [{"format": "color", "name": "w", "type": "image", "file": "images/tiles.exr"},
           {"format": "color", "name": "w2", "type": "image", "file": "images/tiles.exr", "filter": "EWA", "address": "border", "scale": [2.0, 3.0],
            "max_anisotropy": 4.0},
           {"format": "float", "name": "e", "type": "image", "file": "images/tiles.exr", "filter": "trilinear", "address": "clamp", "channel": "B",
            "gamma": 2.0, "max_anisotropy": 4.0},
           {"format": "float", "name": "odd", "type": "image", "file": "images/tiles.exr", "filter": "sharp", "address": "mirror", "channel": "luma"}]
    mats = [{"name": "m", "type": "blinn", "Kg": "w", "exponent": "e"}, {"name": "m2", "type": "blinn", "Kg": "w2", "exponent": "odd"}]
    prims = [{"type": "model", "name": "mq", "geometry": "q", "material": "m"}, {"type": "instance", "name": "i0", "model": "mq"},
             {"type": "model", "name": "mq2", "geometry": "q", "material": "m2"}, {"type": "instance", "name": "i1", "model": "mq2"}]
    s = load(minimal(textures=tex, materials=mats, primitives=prims,
                     lights=[{"type": "ibl", "name": "sky", "file": "images/env.exr", "filter": [1.0, 0.5, 0.25], "sample_num": 4},
                             {"type": "ibl", "name": "dark", "file": "images/env.exr"}]))
    d = s.desc
    by_mat = {i: d.materials[d.instances[i].material] for i in range(2)}
    w, e = d.textures[by_mat[0].tex_color], d.textures[by_mat[0].tex_exponent]
    w2, odd = d.textures[by_mat[1].tex_color], d.textures[by_mat[1].tex_exponent]
    assert w.type == _abi.GBL_TEX_IMAGE and (w.image_filter, w.address, w.mapping) == (_abi.GBL_IMAGE_FILTER_NONE, _abi.GBL_ADDRESS_REPEAT, _abi.GBL_MAP_UV)
    assert (w2.image_filter, w2.address) == (_abi.GBL_IMAGE_FILTER_EWA, _abi.GBL_ADDRESS_BORDER) and list(w2.uv_scale) == [2.0, 3.0]
    assert w2.max_anisotropy == 10.0 and e.max_anisotropy == 4.0     # createColorImageTexture does not forward it (:741-745)
    assert w2.image == w.image                                        # same file, gamma, channel: one MIPMap (ImageTexture::imageCache)
    assert (e.image_filter, e.address, e.is_float) == (_abi.GBL_IMAGE_FILTER_TRILINEAR, _abi.GBL_ADDRESS_CLAMP, 1)
    assert (odd.image_filter, odd.address) == (_abi.GBL_IMAGE_FILTER_NONE, _abi.GBL_ADDRESS_REPEAT)   # unrecognised strings fall back
    img, fimg = d.images[w.image], d.images[e.image]
    assert (img.width, img.height, img.levels, img.channels) == (64, 32, 7, 4) and (fimg.width, fimg.height, fimg.levels, fimg.channels) == (64, 32, 7, 1)
    texels = np.ctypeslib.as_array(d.texels, shape=(d.num_texels,))
    n = sum(max(1, 64 >> l) * max(1, 32 >> l) for l in range(7))
    assert fimg.texel_offset + n <= d.num_texels and img.texel_offset + 4 * n <= d.num_texels
    src = _abi.read_image(os.path.join(MODELS, "images", "tiles.exr"))
    assert src.shape == (32, 48, 4)                                   # not powers of two: resized up before level 0
    assert d.num_lights == 2 and d.lights[0].type == _abi.GBL_LIGHT_IBL and d.lights[0].sample_num == 4 and d.lights[1].sample_num == 1
    assert list(d.lights[0].color) == [1.0, 0.5, 0.25] and list(d.lights[1].color) == [0.0, 0.0, 0.0]   # "filter" defaults to black
    sky = d.images[d.lights[0].image]
    env = _abi.read_image(os.path.join(MODELS, "images", "env.exr"))
    lvl0 = texels[sky.texel_offset:sky.texel_offset + 4 * 64 * 32].reshape(32, 64, 4)
    np.testing.assert_array_equal(lvl0[..., :3], env[..., :3] * np.array([1.0, 0.5, 0.25], np.float32))   # buffer[i] *= filter
    with pytest.raises(_abi.GoblinError) as err:   # a missing image is an error here (the reference renders magenta)
        load(minimal(textures=[{"format": "color", "name": "w", "type": "image", "file": "images/none.exr"}]))
    assert err.value.status == _abi.GBL_ERR_IO


def test_shapes_cameras_and_directional_light():
    """SURVEY 8f rank 3 rows now on the device path: sphere / disk geometry (createGeometries,
    GoblinContextLoader.cpp:226-234), orthographic and thin-lens cameras (:146-187), directional light."""
    s = load(minimal(geometries=[{"name": "q", "type": "sphere", "radius": 2.5}]))
    m = s.desc.meshes[s.desc.instances[0].mesh]
    assert (m.shape, m.radius, m.tri_count) == (_abi.GBL_SHAPE_SPHERE, 2.5, 0)
    s = load(minimal(geometries=[{"name": "q", "type": "disk"}]))
    m = s.desc.meshes[s.desc.instances[0].mesh]
    assert (m.shape, m.radius) == (_abi.GBL_SHAPE_DISK, 1.0)          # "radius" defaults to 1
    s = load(minimal(geometries=[{"name": "q", "type": "torus"}]))
    assert s.desc.meshes[s.desc.instances[0].mesh].shape == _abi.GBL_SHAPE_SPHERE   # unknown type -> sphere (:232-234)
    s = load(minimal(camera={"type": "orthographic", "film_width": 12.0}))
    assert (s.desc.camera.type, s.desc.camera.film_width) == (_abi.GBL_CAMERA_ORTHOGRAPHIC, 12.0)
    s = load(minimal(camera={"type": "orthographic"}))
    assert s.desc.camera.film_width == 35.0
    # a lens adds a black-lambert Disk instance with the camera's transform BEFORE every other instance
    s = load(minimal(camera={"lens_radius": 0.25, "focal_distance": 3.0, "position": [1, 2, 3]}))
    d = s.desc
    assert d.num_instances == 2 and (d.camera.lens_radius, d.camera.focal_distance) == (0.25, 3.0)
    lens = d.instances[0]
    assert (d.meshes[lens.mesh].shape, d.meshes[lens.mesh].radius) == (_abi.GBL_SHAPE_DISK, 0.25)
    assert list(lens.to_world.position) == [1, 2, 3] and list(d.materials[lens.material].color) == [0, 0, 0]
    s = load(minimal(lights=[{"type": "directional", "name": "d", "radiance": [1, 2, 3], "direction": [0, -2, 0]}]))
    l = s.desc.lights[0]
    assert l.type == _abi.GBL_LIGHT_DIRECTIONAL and list(l.color) == [1, 2, 3] and list(l.direction) == [0, -2, 0]


def test_texture_graphs():
    """createTextures (GoblinContextLoader.cpp:245-307): checkerboard / scale textures reference earlier textures
    by name, per format; constants stay inline in the material, graphs become gbl_texture entries."""
    tex = [
        {"format": "color", "name": "a", "type": "constant", "color": [1, 0, 0]},
        {"format": "color", "name": "b", "type": "constant", "color": [0, 1, 0]},
        {"format": "float", "name": "s", "type": "constant", "float": 0.25},
        {"format": "color", "name": "w", "type": "checkerboard", "texture1": "a", "texture2": "b",
         "scale": [4.0, 2.0], "offset": [0.5, 0.0], "filter": True},
        {"format": "color", "name": "dim", "type": "scale", "texture": "w", "scale": "s"},
    ]
    s = load(minimal(textures=tex))
    d = s.desc
    m = d.materials[d.instances[0].material]
    assert m.tex_color >= 0 and (m.tex_color2, m.tex_exponent) == (-1, -1)
    w = d.textures[m.tex_color]
    assert (w.type, w.is_float, w.mapping, w.filter) == (_abi.GBL_TEX_CHECKERBOARD, 0, _abi.GBL_MAP_UV, 1)
    assert list(w.uv_scale) == [4.0, 2.0] and list(w.uv_offset) == [0.5, 0.0]
    c1, c2 = d.textures[w.child[0]], d.textures[w.child[1]]
    assert (c1.type, list(c1.value)) == (_abi.GBL_TEX_CONSTANT, [1, 0, 0]) and list(c2.value) == [0, 1, 0]
    doc = minimal(textures=tex)
    doc["materials"][0]["Kd"] = "dim"
    s = load(doc)
    d = s.desc
    t = d.textures[d.materials[d.instances[0].material].tex_color]
    assert t.type == _abi.GBL_TEX_SCALE and d.textures[t.child[0]].type == _abi.GBL_TEX_CHECKERBOARD
    assert d.textures[t.child[1]].is_float == 1 and d.textures[t.child[1]].value[0] == 0.25
    # a texture only sees the ones defined before it; color and float names live in separate maps
    bad = [dict(tex[3]), tex[0], tex[1]]
    with pytest.raises(_abi.GoblinError) as e:
        load(minimal(textures=bad))
    assert e.value.status == _abi.GBL_ERR_INVALID
    # constant materials stay inline
    s = load(minimal())
    assert s.desc.num_textures == 0 and s.desc.materials[0].tex_color == -1


def test_mask_material():
    """createMaskMaterial (GoblinMaterial.cpp:929-950): alpha defaults to 1, transparent_color to white, the wrapped
    material must already exist when the mask is created."""
    mats = [{"name": "inner", "type": "lambert", "Kd": "w"},
            {"name": "m", "type": "mask", "material": "inner", "alpha": "a"}]
    tex = [{"name": "w", "type": "constant", "color": [0.5, 0.6, 0.7]},
           {"format": "float", "name": "a", "type": "constant", "float": 0.25}]
    s = load(minimal(materials=mats, textures=tex))
    d = s.desc
    m = d.materials[d.instances[0].material]
    assert m.type == _abi.GBL_MAT_MASK and m.exponent == 0.25 and list(m.color) == [1, 1, 1]
    inner = d.materials[m.masked_material]
    assert inner.type == _abi.GBL_MAT_LAMBERT and inner.masked_material == -1
    s = load(minimal(materials=[mats[0], {"name": "m", "type": "mask", "material": "inner"}], textures=tex))
    m = s.desc.materials[s.desc.instances[0].material]
    assert m.exponent == 1.0                                            # "no feed in alpha"
    with pytest.raises(_abi.GoblinError) as e:                          # wrapped material defined later
        load(minimal(materials=[mats[1], mats[0]], textures=tex))
    assert e.value.status == _abi.GBL_ERR_INVALID
    with pytest.raises(_abi.GoblinError) as e:                          # mask of a mask
        load(minimal(materials=[mats[0], {"name": "m1", "type": "mask", "material": "inner"},
                                {"name": "m", "type": "mask", "material": "m1"}], textures=tex))
    assert e.value.status == _abi.GBL_ERR_UNSUPPORTED


def test_subsurface_material():
    """createSubsurfaceMaterial (GoblinMaterial.cpp:881-927): marble coefficients, index 1.5, g 0 and a white Kr by
    default; the Kd + mean_free_path form goes through BSSRDF::convertFromDiffuse on the host."""
    s = load(minimal(materials=[{"name": "m", "type": "subsurface"}]))
    m = s.desc.materials[s.desc.instances[0].material]
    assert m.type == _abi.GBL_MAT_SUBSURFACE and (m.index, m.k) == (1.5, 0.0)
    np.testing.assert_allclose(list(m.color), [0.0021, 0.0041, 0.0071], rtol=1e-6)
    np.testing.assert_allclose(list(m.color2), [2.19, 2.62, 3.00], rtol=1e-6)
    assert list(m.color3) == [1, 1, 1] and (m.tex_color, m.tex_color2, m.tex_color3) == (-1, -1, -1)
    tex = [{"name": "w", "type": "constant", "color": [0.5, 0.6, 0.7]},
           {"name": "c", "type": "checkerboard", "texture1": "w", "texture2": "w"}]
    s = load(minimal(materials=[{"name": "m", "type": "subsurface", "absorb": "w", "scatter_prime": "c", "Kr": "w", "index": 1.3, "g": 0.25}],
                     textures=tex))
    m = s.desc.materials[s.desc.instances[0].material]
    np.testing.assert_allclose(list(m.color), [0.5, 0.6, 0.7], rtol=1e-6)
    np.testing.assert_allclose(list(m.color3), [0.5, 0.6, 0.7], rtol=1e-6)
    assert m.tex_color == -1 and m.tex_color2 >= 0 and (np.float32(m.index), np.float32(m.k)) == (np.float32(1.3), np.float32(0.25))
    # diffuse form: sigma_tr = 1 / mean free path and the dipole's total diffuse reflectance reproduces Kd
    s = load(minimal(materials=[{"name": "m", "type": "subsurface", "Kd": [0.8, 0.5, 0.3], "mean_free_path": [0.5, 0.25, 0.125]}]))
    m = s.desc.materials[s.desc.instances[0].material]
    sa, ssp = np.array(list(m.color), np.float64), np.array(list(m.color2), np.float64)
    np.testing.assert_allclose(np.sqrt(3 * sa * (sa + ssp)), [2.0, 4.0, 8.0], rtol=1e-4)
    eta = 1.5
    fdr = -1.4399 / eta ** 2 + 0.7099 / eta + 0.6681 + 0.0636 * eta
    A = (1 + fdr) / (1 - fdr)
    alpha = ssp / (sa + ssp)
    root = np.sqrt(3 * (1 - alpha))
    np.testing.assert_allclose(0.5 * alpha * (1 + np.exp(-4.0 / 3.0 * A * root)) * np.exp(-root), [0.8, 0.5, 0.3], atol=1e-3)   # 16 bisection steps
    with pytest.raises(_abi.GoblinError) as e:                          # a mask around a subsurface material
        load(minimal(materials=[{"name": "inner", "type": "subsurface"}, {"name": "m", "type": "mask", "material": "inner"}]))
    assert e.value.status == _abi.GBL_ERR_UNSUPPORTED


def test_homogeneous_volume():
    """createVolume / createHomogeneousVolume (GoblinContextLoader.cpp:189-207, GoblinVolume.cpp:343-360)."""
    s = load(minimal())
    assert s.desc.volume.type == _abi.GBL_VOLUME_NONE
    s = load(minimal(volume={"type": "homogeneous", "attenuation": [0.1, 0.2, 0.3], "albedo": [0.5, 0.6, 0.7], "box_min": [-1.0, -2.0, -3.0],
                             "box_max": [1.0, 2.0, 3.0], "position": [0.0, 1.0, 0.0]}))
    v = s.desc.volume
    assert v.type == _abi.GBL_VOLUME_HOMOGENEOUS and (v.g, v.sample_num) == (0.0, 5)
    np.testing.assert_allclose(list(v.attenuation) + list(v.albedo), [0.1, 0.2, 0.3, 0.5, 0.6, 0.7], rtol=1e-6)
    assert list(v.box_min) == [-1, -2, -3] and list(v.box_max) == [1, 2, 3] and list(v.to_world.position) == [0, 1, 0]
    s = load(minimal(volume={"type": "no such medium"}))                  # unknown type -> homogeneous (:205-207)
    assert s.desc.volume.type == _abi.GBL_VOLUME_HOMOGENEOUS


def test_whitted_quota_follows_the_lights():
    """WhittedRenderer::querySampleQuota (GoblinWhitted.cpp:46-70): a LightSampleIndex and a BSDFSampleIndex of
    roundToSquare(getSamplesNum()) slots per light, one pick 1D, the BSSRDF block; "sample_num" is read by area lights only."""
    lights = [{"name": "a", "type": "area", "geometry": "q", "radiance": [1, 1, 1], "sample_num": 3},
              {"name": "p", "type": "point", "intensity": [1, 1, 1], "sample_num": 7}]
    s = load(minimal(render_setting={"render_method": "whitted", "bssrdf_sample_num": 4}, lights=lights))
    assert [s.desc.lights[i].sample_num for i in range(2)] == [3, 1]
    # light a: 4 slots -> 2 * (4 + 2 * 4); light p: 1 slot -> 2 * (1 + 2); pick 1; BSSRDF 4 * 4 + 2 * 2 * 4
    assert s.sample_dimension() == 4 + 24 + 6 + 1 + 32


def test_unused_out_of_scope_declarations_are_ignored():
    """bunny.json declares a sphere geometry nothing uses (examples/bunny.json:36-40)."""
    doc = minimal()
    doc["geometries"].append({"name": "ball", "type": "sphere", "radius": 0.05})
    doc["materials"].append({"name": "sss", "type": "subsurface"})
    s = load(doc)
    assert s.desc.num_instances == 1 and s.desc.num_meshes == 1


def test_missing_names_and_files():
    for doc, code in [
        (minimal(materials=[{"name": "m", "type": "lambert", "Kd": "nope"}]), _abi.GBL_ERR_INVALID),
        (minimal(primitives=[{"type": "instance", "name": "i", "model": "nope"}]), _abi.GBL_ERR_INVALID),
        (minimal(geometries=[{"name": "q", "type": "mesh", "file": "models/missing.obj"}]), _abi.GBL_ERR_IO),
    ]:
        with pytest.raises(_abi.GoblinError) as e:
            load(doc)
        assert e.value.status == code
    with pytest.raises(_abi.GoblinError) as e:
        gs.load_scene_text("{ not json", ".")
    assert e.value.status == _abi.GBL_ERR_IO
    with pytest.raises(_abi.GoblinError):
        _abi.host_lib()   # make sure the lib is loaded for the next line
        import ctypes as C
        h = C.c_void_p()
        st = _abi.host_lib().gbl_host_load_file(b"/no/such/scene.json", C.byref(h))
        raise _abi.GoblinError(st, _abi.host_lib().gbl_host_last_error().decode())


def test_first_definition_of_a_name_wins_and_only_instances_render():
    """SceneCache::add* use map::insert; models and instances share one name map (bunny.json names both 'bunny')."""
    doc = minimal()
    doc["textures"].append({"name": "w", "type": "constant", "color": [9, 9, 9]})
    doc["primitives"] = [{"type": "model", "name": "x", "geometry": "q", "material": "m"},
                         {"type": "instance", "name": "x", "model": "x", "position": [1, 2, 3]},
                         {"type": "whatever", "name": "y", "geometry": "q", "material": "m"}]   # unknown type -> model
    s = load(doc)
    assert s.desc.num_instances == 1 and list(s.desc.instances[0].to_world.position) == [1, 2, 3]
    assert list(s.desc.materials[s.desc.instances[0].material].color) == pytest.approx([0.5, 0.6, 0.7])


def test_area_light_auto_instances_an_emissive_black_model():
    doc = minimal(lights=[{"type": "area", "name": "L", "geometry": "q", "radiance": [5, 4, 3], "position": [0, 2, 0],
                           "orientation": [0, 1, 0, 0], "scale": [0.5, 0.5, 0.5]}])
    s = load(doc)
    d = s.desc
    assert d.num_instances == 2 and d.num_lights == 1
    emissive = d.instances[1]                       # appended after the scene's own instances
    assert emissive.area_light == 0 and list(d.materials[emissive.material].color) == [0, 0, 0]
    assert list(emissive.to_world.scale) == [0.5, 0.5, 0.5] and list(d.lights[0].to_world.orientation) == [0, 1, 0, 0]


def test_spot_light_target_and_cosines():
    doc = minimal(lights=[{"type": "spot", "name": "s", "intensity": [1, 1, 1], "position": [0, 4, 0], "target": [0, 0, 3],
                           "theta_max": 30.0, "falloff_start": 20.0}])
    s = load(doc)
    l = s.desc.lights[0]
    assert l.type == _abi.GBL_LIGHT_SPOT
    np.testing.assert_allclose(list(l.direction), [0, -0.8, 0.6], rtol=1e-6)
    np.testing.assert_allclose([l.cos_theta_max, l.cos_falloff_start], np.cos(np.radians([30.0, 20.0])), rtol=1e-6)


def test_euler_orientation():
    doc = minimal()
    doc["primitives"][1].update({"euler": [0.0, 90.0, 0.0]})
    s = load(doc)
    q = list(s.desc.instances[0].to_world.orientation)
    np.testing.assert_allclose(q, [np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0], atol=1e-6)


def test_obj_loader(tmp_path):
    """v/vn/vt/f, quads split (0,1,2)+(0,2,3), negative indices, (v,vn,vt) de-duplication, format from the first face."""
    (tmp_path / "a.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf -4//1 -3//1 -2//1 -1//1\n")
    (tmp_path / "b.obj").write_text("# tri soup\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\nf 3/3 2/2 4/1\n")
    doc = minimal(geometries=[{"name": "q", "type": "mesh", "file": str(tmp_path / "a.obj")},
                              {"name": "r", "type": "mesh", "file": str(tmp_path / "b.obj")}])
    doc["primitives"] += [{"type": "model", "name": "mr", "geometry": "r", "material": "m"},
                          {"type": "instance", "name": "i1", "model": "mr"}]
    s = load(doc)
    d = s.desc
    a, b = d.meshes[0], d.meshes[1]
    assert (a.vertex_count, a.tri_count, a.has_normal, a.has_uv) == (4, 2, 1, 0)
    assert list(d.indices[0:6]) == [0, 1, 2, 0, 2, 3]
    assert (b.vertex_count, b.tri_count, b.has_normal, b.has_uv) == (4, 2, 0, 1)    # (v4, vt1) is a new vertex
    assert list(d.indices[6:12]) == [0, 1, 2, 2, 1, 3]
    np.testing.assert_array_equal(np.ctypeslib.as_array(d.normals, (12,))[:3], [0, 0, 1])
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    with pytest.raises(_abi.GoblinError):
        load(minimal(geometries=[{"name": "q", "type": "mesh", "file": str(tmp_path / "bad.obj")}]))


def test_sample_bookkeeping_helpers():
    s = gs.load_scene("bunny", gs.config_overrides(resolution=(512, 512), spp=250, depth=8))
    assert s.spp() == 256 and s.sample_window() == (-2, 514, -2, 514)
    assert s.num_paths() == 516 * 516 * 256 == 68161536
    assert s.sample_dimension() == 92                                   # 4 + 7*8 + 32
    s = gs.load_scene("bunny", gs.config_overrides(resolution=(256, 256), spp=16, depth=4))
    assert s.sample_dimension() == 64 and s.num_paths() == 1081600
    s = gs.load_scene("bunny", gs.config_overrides(spp=4, method="ao", ao_samples=25))
    assert s.sample_dimension() == 4 + 50


def test_film_normalize_and_pfm(tmp_path):
    import ctypes as C
    acc = np.array([[[2, 4, 6, 2], [1, 1, 1, 4]]], np.float32)
    rgb = np.zeros((1, 2, 3), np.float32)
    _abi.host_lib().gbl_host_film_normalize(acc.ctypes.data_as(C.c_void_p), 2, 1, rgb.ctypes.data_as(C.c_void_p))
    np.testing.assert_array_equal(rgb, [[[1, 2, 3], [0.25, 0.25, 0.25]]])
    p = str(tmp_path / "x.pfm")
    assert _abi.host_lib().gbl_host_write_pfm(p.encode(), rgb.ctypes.data_as(C.c_void_p), 2, 1) == 0
    raw = open(p, "rb").read()
    assert raw.startswith(b"PF\n2 1\n-1.0\n") and len(raw) == len(b"PF\n2 1\n-1.0\n") + 24


def test_repeated_json_key_keeps_its_last_value_and_nesting_is_bounded():
    """The reference's nlohmann DOM parser assigns through operator[] (json.hpp:2899,2987): the last value of a repeated key
    is the one every later lookup sees.  (Distinct from repeated *names* of geometries / materials, where the first wins.)"""
    text = json.dumps(minimal(render_setting={"sample_per_pixel": 4}))
    text = text.replace('"sample_per_pixel": 4', '"sample_per_pixel": 4, "max_ray_depth": 3, "sample_per_pixel": 9', 1)
    assert text.count('"sample_per_pixel"') == 2
    s = gs.load_scene_text(text, MODELS)   # keep the Scene alive: desc points into it
    assert s.desc.setting.sample_per_pixel == 9 and s.desc.setting.max_ray_depth == 3
    # a pathologically nested file is a parse error, not a stack overflow
    with pytest.raises(_abi.GoblinError) as e:
        gs.load_scene_text("[" * 100000, MODELS)
    assert "nested too deeply" in str(e.value)


def test_heterogeneous_volume_and_vol_grid(tmp_path):
    """createHeterogeneousVolume + loadVolFile (GoblinVolume.cpp:222-257, 362-384): float32 .vol grids, the grid's bounding
    box as the region, step_size / sample_num defaults; a missing file, a wrong signature or another encoding falls back to the
    reference's one-cell grid of density 1 on [-1, 1]^3."""
    import struct
    data = np.arange(2 * 3 * 4 * 3, dtype=np.float32).reshape(4, 3, 2, 3) / 10.0   # z, y, x, channel
    with open(tmp_path / "g.vol", "wb") as f:
        f.write(b"VOL\x03" + struct.pack("<5i", 1, 2, 3, 4, 3) + struct.pack("<6f", -1, -2, -3, 1.5, 2.5, 3.5) + data.tobytes())
    with open(tmp_path / "half.vol", "wb") as f:   # float16 encoding: not loaded
        f.write(b"VOL\x03" + struct.pack("<5i", 2, 2, 3, 4, 1) + struct.pack("<6f", -1, -2, -3, 1.5, 2.5, 3.5) + bytes(2 * 24))
    vol = {"type": "heterogeneous", "density_grid": str(tmp_path / "g.vol"), "albedo": [0.5, 0.6, 0.7], "g": 0.2, "position": [1, 2, 3]}
    s = gs.load_scene_text(json.dumps(minimal(volume=vol)), MODELS)
    v = s.desc.volume
    assert v.type == _abi.GBL_VOLUME_HETEROGENEOUS and list(v.grid) == [2, 3, 4] and v.grid_channels == 3
    assert list(v.box_min) == [-1, -2, -3] and list(v.box_max) == [1.5, 2.5, 3.5]
    assert v.step_size == pytest.approx(0.1) and v.sample_num == 5 and v.g == pytest.approx(0.2)
    np.testing.assert_array_equal(np.ctypeslib.as_array(v.density, shape=(72,)), data.ravel())
    assert list(v.to_world.position) == [1, 2, 3]
    # hostile or truncated headers are not grids either (on the GPU they would be out-of-bounds reads): dimensions whose product
    # wraps size_t, a product beyond what 32-bit indices address, a file shorter than its header promises
    with open(tmp_path / "wrap.vol", "wb") as f:
        f.write(b"VOL\x03" + struct.pack("<5i", 1, 1 << 30, 1 << 30, 16, 1) + struct.pack("<6f", -1, -2, -3, 1.5, 2.5, 3.5) + bytes(64))
    with open(tmp_path / "huge.vol", "wb") as f:
        f.write(b"VOL\x03" + struct.pack("<5i", 1, 2048, 2048, 1024, 1) + struct.pack("<6f", -1, -2, -3, 1.5, 2.5, 3.5) + bytes(64))
    with open(tmp_path / "short.vol", "wb") as f:
        f.write(b"VOL\x03" + struct.pack("<5i", 1, 2, 3, 4, 3) + struct.pack("<6f", -1, -2, -3, 1.5, 2.5, 3.5) + data.tobytes()[:100])
    for bad in (str(tmp_path / "missing.vol"), str(tmp_path / "half.vol"), str(tmp_path / "wrap.vol"), str(tmp_path / "huge.vol"), str(tmp_path / "short.vol")):
        s = gs.load_scene_text(json.dumps(minimal(volume=dict(vol, density_grid=bad, step_size=0.25, sample_num=2))), MODELS)
        v = s.desc.volume
        assert list(v.grid) == [1, 1, 1] and v.grid_channels == 1 and v.density[0] == 1.0
        assert list(v.box_min) == [-1, -1, -1] and list(v.box_max) == [1, 1, 1] and v.step_size == 0.25 and v.sample_num == 2
