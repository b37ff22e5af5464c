"""ctypes mirror of include/goblin_hip.h and loaders for the two shared libraries.

Nothing here computes anything: it declares the C structs, finds
``libgoblin_host.so`` / ``libgoblin_hip.so`` next to the package (built in-tree by
``goblin_amd.build``), and turns non-zero status codes into exceptions.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "lib")

GBL_ABI_VERSION = 14
GBL_AUTO_WAVEFRONT_RAYS_PER_PATH, GBL_AUTO_WAVEFRONT_PATHS = 6.0, 1 << 22   # gbl_schedule AUTO thresholds (goblin_hip.h)
GBL_OK, GBL_ERR_INVALID, GBL_ERR_UNSUPPORTED, GBL_ERR_IO, GBL_ERR_DEVICE, GBL_ERR_OOM, GBL_ERR_INTERNAL = range(7)
STATUS_NAMES = {0: "GBL_OK", 1: "GBL_ERR_INVALID", 2: "GBL_ERR_UNSUPPORTED", 3: "GBL_ERR_IO",
                4: "GBL_ERR_DEVICE", 5: "GBL_ERR_OOM", 6: "GBL_ERR_INTERNAL"}

GBL_MAT_LAMBERT, GBL_MAT_BLINN, GBL_MAT_TRANSPARENT, GBL_MAT_MIRROR = range(4)
GBL_LIGHT_POINT, GBL_LIGHT_DIRECTIONAL, GBL_LIGHT_SPOT, GBL_LIGHT_AREA, GBL_LIGHT_IBL = 0, 1, 2, 3, 4
GBL_SHAPE_MESH, GBL_SHAPE_SPHERE, GBL_SHAPE_DISK = 0, 1, 2
GBL_CAMERA_PERSPECTIVE, GBL_CAMERA_ORTHOGRAPHIC = 0, 1
GBL_FILTER_BOX, GBL_FILTER_TRIANGLE, GBL_FILTER_GAUSSIAN, GBL_FILTER_MITCHELL = range(4)
GBL_INTEGRATOR_PATH, GBL_INTEGRATOR_AO, GBL_INTEGRATOR_WHITTED = 0, 1, 2
GBL_SAMPLES_NATIVE, GBL_SAMPLES_REPLAY, GBL_SAMPLES_STREAM = 0, 1, 2
GBL_SCHEDULE_AUTO, GBL_SCHEDULE_MEGAKERNEL, GBL_SCHEDULE_WAVEFRONT = 0, 1, 2


class GoblinError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


class gbl_trs(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("orientation", C.c_float * 4), ("scale", C.c_float * 3)]


class gbl_mesh(C.Structure):
    _fields_ = [("vertex_offset", C.c_uint32), ("vertex_count", C.c_uint32), ("tri_offset", C.c_uint32),
                ("tri_count", C.c_uint32), ("has_normal", C.c_uint32), ("has_uv", C.c_uint32),
                ("shape", C.c_uint32), ("radius", C.c_float)]


class gbl_texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("is_float", C.c_uint32), ("value", C.c_float * 3), ("child", C.c_int32 * 2),
                ("mapping", C.c_uint32), ("uv_scale", C.c_float * 2), ("uv_offset", C.c_float * 2), ("to_tex", gbl_trs),
                ("filter", C.c_uint32), ("image", C.c_int32), ("image_filter", C.c_uint32), ("address", C.c_uint32),
                ("max_anisotropy", C.c_float)]


class gbl_image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("levels", C.c_uint32), ("channels", C.c_uint32),
                ("texel_offset", C.c_uint64)]


GBL_TEX_CONSTANT, GBL_TEX_CHECKERBOARD, GBL_TEX_SCALE, GBL_TEX_IMAGE = 0, 1, 2, 3
GBL_IMAGE_FILTER_NONE, GBL_IMAGE_FILTER_BILINEAR, GBL_IMAGE_FILTER_TRILINEAR, GBL_IMAGE_FILTER_EWA = 0, 1, 2, 3
GBL_ADDRESS_REPEAT, GBL_ADDRESS_CLAMP, GBL_ADDRESS_BORDER = 0, 1, 2
GBL_MAT_LAMBERT, GBL_MAT_BLINN, GBL_MAT_TRANSPARENT, GBL_MAT_MIRROR, GBL_MAT_MASK, GBL_MAT_SUBSURFACE = 0, 1, 2, 3, 4, 5
GBL_MAP_UV, GBL_MAP_SPHERICAL = 0, 1


class gbl_material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color", C.c_float * 3), ("color2", C.c_float * 3), ("index", C.c_float),
                ("k", C.c_float), ("exponent", C.c_float), ("tex_color", C.c_int32), ("tex_color2", C.c_int32),
                ("tex_exponent", C.c_int32), ("masked_material", C.c_int32), ("color3", C.c_float * 3),
                ("tex_color3", C.c_int32), ("tex_bump", C.c_int32), ("tex_normal", C.c_int32)]


class gbl_instance(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("material", C.c_uint32), ("area_light", C.c_int32), ("to_world", gbl_trs)]


class gbl_light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color", C.c_float * 3), ("position", C.c_float * 3),
                ("direction", C.c_float * 3), ("cos_theta_max", C.c_float), ("cos_falloff_start", C.c_float),
                ("mesh", C.c_uint32), ("to_world", gbl_trs), ("sample_num", C.c_uint32), ("image", C.c_int32)]


class gbl_camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("orientation", C.c_float * 4), ("fov_degrees", C.c_float),
                ("near_plane", C.c_float), ("far_plane", C.c_float), ("lens_radius", C.c_float),
                ("focal_distance", C.c_float), ("type", C.c_uint32), ("film_width", C.c_float)]


class gbl_film(C.Structure):
    _fields_ = [("xres", C.c_int32), ("yres", C.c_int32), ("crop", C.c_float * 4), ("filter_type", C.c_uint32),
                ("filter_width", C.c_float * 2), ("gaussian_falloff", C.c_float), ("mitchell_b", C.c_float),
                ("mitchell_c", C.c_float), ("tone_mapping", C.c_uint32), ("bloom_radius", C.c_float),
                ("bloom_weight", C.c_float)]


class gbl_render_setting(C.Structure):
    _fields_ = [("integrator", C.c_uint32), ("sample_per_pixel", C.c_int32), ("max_ray_depth", C.c_int32),
                ("bssrdf_sample_num", C.c_int32), ("ao_sample_num", C.c_int32), ("thread_num", C.c_int32)]


class gbl_volume(C.Structure):
    _fields_ = [("type", C.c_uint32), ("attenuation", C.c_float * 3), ("albedo", C.c_float * 3), ("emission", C.c_float * 3),
                ("g", C.c_float), ("sample_num", C.c_int32), ("box_min", C.c_float * 3), ("box_max", C.c_float * 3),
                ("to_world", gbl_trs), ("step_size", C.c_float), ("grid", C.c_int32 * 3), ("grid_channels", C.c_int32),
                ("density", C.POINTER(C.c_float))]


GBL_VOLUME_NONE, GBL_VOLUME_HOMOGENEOUS, GBL_VOLUME_HETEROGENEOUS = 0, 1, 2


class gbl_scene_desc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32),
                ("num_vertices", C.c_uint32), ("positions", C.POINTER(C.c_float)),
                ("normals", C.POINTER(C.c_float)), ("uvs", C.POINTER(C.c_float)),
                ("num_triangles", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
                ("num_meshes", C.c_uint32), ("meshes", C.POINTER(gbl_mesh)),
                ("num_materials", C.c_uint32), ("materials", C.POINTER(gbl_material)),
                ("num_textures", C.c_uint32), ("textures", C.POINTER(gbl_texture)),
                ("num_images", C.c_uint32), ("images", C.POINTER(gbl_image)),
                ("num_texels", C.c_uint64), ("texels", C.POINTER(C.c_float)),
                ("num_instances", C.c_uint32), ("instances", C.POINTER(gbl_instance)),
                ("num_lights", C.c_uint32), ("lights", C.POINTER(gbl_light)),
                ("camera", gbl_camera), ("film", gbl_film), ("setting", gbl_render_setting), ("volume", gbl_volume)]


class gbl_render_params(C.Structure):
    _fields_ = [("integrator", C.c_uint32), ("sample_per_pixel", C.c_int32), ("max_ray_depth", C.c_int32),
                ("ao_sample_num", C.c_int32), ("bssrdf_sample_num", C.c_int32), ("window", C.c_int32 * 4),
                ("tile_shard_index", C.c_int32), ("tile_shard_count", C.c_int32),
                ("sample_mode", C.c_uint32), ("seed", C.c_uint64), ("replay_samples", C.c_void_p),
                ("li_out", C.c_void_p), ("russian_roulette", C.c_uint32), ("collect_stats", C.c_uint32),
                ("schedule", C.c_uint32), ("exact_ties", C.c_uint32), ("stream", C.c_void_p)]


class gbl_stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("nodes", C.c_uint64), ("tris", C.c_uint64), ("splats", C.c_uint64), ("dims", C.c_uint64),
                ("kernel_ms", C.c_double), ("schedule", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class gbl_timing(C.Structure):
    _fields_ = [("main_kernel_ms", C.c_double), ("total_ms", C.c_double)]


class gbl_info(C.Structure):
    _fields_ = [("xres", C.c_int32), ("yres", C.c_int32), ("window", C.c_int32 * 4), ("blas_nodes", C.c_uint64),
                ("tlas_nodes", C.c_uint64), ("triangles", C.c_uint64), ("instances", C.c_uint64),
                ("scene_bytes", C.c_uint64), ("instanced_triangles", C.c_uint64), ("build_ms", C.c_double),
                ("blas_depth", C.c_int32), ("tlas_depth", C.c_int32)]


HOST_SYMBOLS = ["gbl_host_load_file", "gbl_host_load_string", "gbl_host_desc", "gbl_host_free",
                "gbl_host_last_error", "gbl_host_sample_window", "gbl_host_round_to_square",
                "gbl_host_sample_dimension", "gbl_host_sample_dimension_scene", "gbl_host_film_normalize", "gbl_host_write_pfm", "gbl_host_output_path",
                "gbl_host_bloom", "gbl_host_tone_map", "gbl_host_write_ppm", "gbl_host_write_exr", "gbl_host_write_image",
                "gbl_host_read_image", "gbl_host_free_image"]
GBL_CREATE_DEVICE_BVH = 1
HIP_SYMBOLS = ["gbl_create", "gbl_create_ex", "gbl_update_instances", "gbl_render", "gbl_film_allreduce", "gbl_film_resolve", "gbl_get_info", "gbl_destroy",
               "gbl_last_error", "gbl_abi_version", "gbl_get_timings", "gbl_selftest_sincos", "gbl_selftest_trace", "gbl_selftest_arith", "gbl_selftest_libm", "gbl_selftest_valu_issue"]

_host = None
_hip = None


def host_lib():
    """libgoblin_host.so (g++ only; loads on any box)."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libgoblin_host.so")
        if not os.path.exists(path):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or goblin_amd.build.build_all()) first" % path)
        lib = C.CDLL(path)
        lib.gbl_host_load_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.gbl_host_load_string.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.gbl_host_desc.argtypes = [C.c_void_p]
        lib.gbl_host_desc.restype = C.POINTER(gbl_scene_desc)
        lib.gbl_host_free.argtypes = [C.c_void_p]
        lib.gbl_host_free.restype = None
        lib.gbl_host_last_error.restype = C.c_char_p
        lib.gbl_host_sample_window.argtypes = [C.POINTER(gbl_film), C.POINTER(C.c_int32)]
        lib.gbl_host_sample_window.restype = None
        lib.gbl_host_round_to_square.argtypes = [C.c_int32]
        lib.gbl_host_sample_dimension.argtypes = [C.POINTER(gbl_render_setting)]
        lib.gbl_host_sample_dimension_scene.argtypes = [C.POINTER(gbl_scene_desc), C.POINTER(gbl_render_setting)]
        lib.gbl_host_film_normalize.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        lib.gbl_host_film_normalize.restype = None
        lib.gbl_host_write_pfm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        lib.gbl_host_output_path.argtypes = [C.c_void_p]
        lib.gbl_host_output_path.restype = C.c_char_p
        lib.gbl_host_bloom.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float]
        lib.gbl_host_bloom.restype = None
        lib.gbl_host_tone_map.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.gbl_host_tone_map.restype = None
        lib.gbl_host_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        lib.gbl_host_write_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        lib.gbl_host_write_image.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        lib.gbl_host_read_image.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.gbl_host_free_image.argtypes = [C.POINTER(C.c_float)]
        lib.gbl_host_free_image.restype = None
        _host = lib
    return _host


def read_image(path):
    """Goblin::loadImage through libgoblin_host.so: (H, W, 4) float32."""
    import numpy as np
    lib = host_lib()
    p = C.POINTER(C.c_float)()
    w, h = C.c_int32(), C.c_int32()
    st = lib.gbl_host_read_image(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h))
    if st != GBL_OK:
        raise GoblinError(st, lib.gbl_host_last_error().decode())
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib.gbl_host_free_image(p)


def hip_lib():
    """libgoblin_hip.so (hipcc, gfx950).  There is NO fallback: if the HIP
    library is missing the device path fails here, loudly."""
    global _hip
    if _hip is None:
        path = os.environ.get("GOBLIN_HIP_LIB") or os.path.join(LIB_DIR, "libgoblin_hip.so")
        if not os.path.exists(path):
            raise ImportError("%s is missing: the device integrator has no CPU fallback. Build it with "
                              "`python -c 'import __graft_entry__ as g; g.build()'`" % path)
        # PyTorch-ROCm ships its own libamdhip64; film buffers and streams come from
        # torch, so the integrator must bind to THAT runtime, not to a second copy
        # from /opt/rocm.  Importing torch first makes the loader reuse it (same soname).
        import torch  # noqa: F401
        lib = C.CDLL(path)
        lib.gbl_create.argtypes = [C.POINTER(gbl_scene_desc), C.c_int, C.POINTER(C.c_void_p)]
        lib.gbl_create_ex.argtypes = [C.POINTER(gbl_scene_desc), C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.gbl_update_instances.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(gbl_trs)]
        lib.gbl_render.argtypes = [C.c_void_p, C.POINTER(gbl_render_params), C.c_void_p, C.POINTER(gbl_stats)]
        lib.gbl_film_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.gbl_film_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.gbl_get_info.argtypes = [C.c_void_p, C.POINTER(gbl_info)]
        lib.gbl_get_timings.argtypes = [C.c_void_p, C.c_int, C.POINTER(gbl_timing)]
        lib.gbl_selftest_sincos.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        lib.gbl_selftest_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        lib.gbl_selftest_arith.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        lib.gbl_selftest_libm.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        lib.gbl_selftest_valu_issue.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
        lib.gbl_destroy.argtypes = [C.c_void_p]
        lib.gbl_destroy.restype = None
        lib.gbl_last_error.argtypes = [C.c_void_p]
        lib.gbl_last_error.restype = C.c_char_p
        _hip = lib
    return _hip
