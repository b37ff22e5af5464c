"""The C ABI: every function include/goblin_hip.h declares is exported by the library
that implements it, the ctypes mirror has the C layout, and the device library refuses
to work without a GPU instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import pytest

from goblin_amd import _abi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "goblin_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gbl_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    host = [n for n in names if n.startswith("gbl_host_")]
    hip = [n for n in names if not n.startswith("gbl_host_")]
    assert sorted(host) == sorted(_abi.HOST_SYMBOLS)
    assert sorted(hip) == sorted(_abi.HIP_SYMBOLS)
    hl = _abi.host_lib()
    for n in host:
        assert hasattr(hl, n), n
    dl = _abi.hip_lib()          # dlopen only; no compute without a GPU
    for n in hip:
        assert hasattr(dl, n), n
    assert dl.gbl_abi_version() == _abi.GBL_ABI_VERSION


def test_ctypes_mirror_has_the_c_layout(tmp_path):
    structs = ["gbl_trs", "gbl_mesh", "gbl_texture", "gbl_image", "gbl_material", "gbl_instance", "gbl_light", "gbl_camera", "gbl_film", "gbl_volume",
               "gbl_render_setting", "gbl_scene_desc", "gbl_render_params", "gbl_stats", "gbl_info"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "goblin_hip.h"\nint main(void){\n' +
                   "".join('printf("%s %%zu\\n", sizeof(%s));\n' % (s, s) for s in structs) +
                   'printf("off_film %zu\\n", offsetof(gbl_scene_desc, film));\n'
                   'printf("off_seed %zu\\n", offsetof(gbl_render_params, seed));\nreturn 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])   # plain C
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for s in structs:
        assert C.sizeof(getattr(_abi, s)) == int(out[s]), s
    assert _abi.gbl_scene_desc.film.offset == int(out["off_film"])
    assert _abi.gbl_render_params.seed.offset == int(out["off_seed"])


def test_device_library_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    from goblin_amd import scene as gs
    from goblin_amd.renderer import HipPathTracer
    scene = gs.load_scene("bunny", gs.config_overrides(resolution=(16, 16), spp=1, depth=2))
    with pytest.raises(RuntimeError):
        HipPathTracer(scene, 0)
    h = C.c_void_p()
    st = _abi.hip_lib().gbl_create(scene.desc_ptr, 0, C.byref(h))
    assert st == _abi.GBL_ERR_DEVICE and not h
    assert b"no CPU fallback" in _abi.hip_lib().gbl_last_error(None)


def test_bad_descriptions_are_rejected_before_touching_the_device():
    from goblin_amd import scene as gs
    scene = gs.load_scene("bunny", gs.config_overrides(resolution=(16, 16), spp=1, depth=2))
    desc = _abi.gbl_scene_desc.from_buffer_copy(scene.desc)
    desc.abi_version = 99
    h = C.c_void_p()
    assert _abi.hip_lib().gbl_create(C.byref(desc), 0, C.byref(h)) == _abi.GBL_ERR_INVALID
    desc = _abi.gbl_scene_desc.from_buffer_copy(scene.desc)
    desc.film.filter_width[0] = 9.0     # LDS film tile halo limit
    assert _abi.hip_lib().gbl_create(C.byref(desc), 0, C.byref(h)) == _abi.GBL_ERR_UNSUPPORTED


def test_degenerate_march_steps_are_rejected_not_hung():
    """A heterogeneous medium is ray marched in steps of step_size (kernels/medium.h): zero, negative, NaN, or a step so small
    that the region is more than 10^6 steps across would never come back from the device.  pack_scene refuses them."""
    from goblin_amd import scene as gs
    scene = gs.load_scene("hetero", gs.config_overrides(resolution=(16, 16), spp=1, depth=2))
    h = C.c_void_p()
    for step in (0.0, -0.1, float("nan"), float("inf"), 1e-12):
        desc = _abi.gbl_scene_desc.from_buffer_copy(scene.desc)
        desc.volume.step_size = step
        assert _abi.hip_lib().gbl_create(C.byref(desc), 0, C.byref(h)) == _abi.GBL_ERR_INVALID, step
        assert b"step_size" in _abi.hip_lib().gbl_last_error(None)
    desc = _abi.gbl_scene_desc.from_buffer_copy(scene.desc)
    desc.volume.grid[0] = 1 << 16
    desc.volume.grid[1] = 1 << 16
    assert _abi.hip_lib().gbl_create(C.byref(desc), 0, C.byref(h)) == _abi.GBL_ERR_INVALID
    assert b"2^31" in _abi.hip_lib().gbl_last_error(None)


def test_product_never_touches_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(REPO, "goblin_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(root, f), errors="ignore").read()
                for needle in ("oracle_binding", "liboracle", "orc_", "oracle/"):
                    if needle in text:
                        # comments that merely point at the oracle's restated definition are fine; code use is not
                        for line in text.splitlines():
                            if needle in line:
                                stripped = line.strip()
                                assert stripped.startswith(("//", "#", "*", '"""')) or "oracle/goblin_oracle.cpp" in line, (f, line)


def test_integration_binding_compiles_against_the_reference_headers(tmp_path):
    """INTEGRATION.md's `HipPathTracer : Renderer` is the binding a Goblin maintainer would add: it must at least be
    well-formed C++ against the reference's own headers and this repository's include/ (flags = oracle/Makefile's)."""
    import re
    import subprocess
    ref = "/root/reference/src"
    if not os.path.exists(os.path.join(ref, "GoblinRenderer.h")) or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs the reference's headers and the HIP runtime API header")
    with open(os.path.join(REPO, "INTEGRATION.md")) as f:
        code = re.findall(r"```cpp\n(.*?)```", f.read(), re.S)[0]
    assert "class HipPathTracer : public Renderer" in code
    src = tmp_path / "GoblinHipPathtracer.cpp"
    src.write_text(code)
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-w", "-include", "math.h", "-include", "condition_variable", "-include", "random",
                        "-Duniform_real=uniform_real_distribution", "-Duniform_int=uniform_int_distribution", "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-I" + ref, "-I" + os.path.join(REPO, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:4000]


def test_source_stamp_counts_code_not_comments():
    """goblin_amd/build.py source_stamp ties the committed rocprof counters to the tree they were collected from (bench.py only
    quotes counters whose stamp is the running tree's).  It hashes code: a note beside a kernel must not orphan the counters,
    a changed token must."""
    from goblin_amd import build
    a = 'int f(int x) {   // adds one\n    return x + 1; /* really */\n}\nconst char* s = "// not a comment /* nor this */";\n'
    b = 'int f(int x) {\n  return x + 1;\n}\n\n// a later note\nconst char* s = "// not a comment /* nor this */";'
    assert build._code_only(a) == build._code_only(b)
    assert build._code_only(a) != build._code_only(a.replace("x + 1", "x + 2"))
    assert build._code_only(a) != build._code_only(a.replace("// not a comment", "// no comment"))   # (inside a literal: code)
    stamp = build.source_stamp()
    assert len(stamp) == 16 and int(stamp, 16) >= 0
    # the committed counter summaries carry the stamp of the tree they came from
    import json
    for name in ("pmc_bunny_megakernel", "pmc_cornell_wavefront", "pmc_grid_megakernel", "pmc_ao_megakernel"):
        with open(os.path.join(REPO, "profiles", name + ".json")) as f:
            assert len(json.load(f)["source_stamp"]) == 16
