"""In-tree builds of the two shared libraries (explicit compiler invocations; the
built .so files live in goblin_amd/lib/, git-ignored but shipped to the GPU box).

    python -m goblin_amd.build            # build everything that is stale
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HOST_SOURCES = [os.path.join(CSRC, "host", f) for f in ("scene_loader.cpp", "image_io.cpp", "exr_reader.cpp", "mipmap.cpp")]
HOST_DEPS = HOST_SOURCES + [os.path.join(CSRC, "host", "json_lite.h"), os.path.join(CSRC, "abi_guard.h"), os.path.join(REPO, "include", "goblin_hip.h")]
# libgoblin_hip.so: the host side of the C ABI and one translation unit per kernel family, compiled side by side
# (a single unit took 3.6 minutes; these take about one on 8 cores, and an edit rebuilds only the units that include
# what changed -- hipcc's depfiles decide).
HIP_SOURCES = [os.path.join(CSRC, f) for f in
               ("gbl_api.hip", "kernels_path.hip", "kernels_quad.hip", "kernels_stream.hip", "kernels_wavefront.hip",
                "kernels_whitted.hip", "kernels_aux.hip", "scene_prep.cpp")]
OBJ = os.path.join(LIB, "obj")

# -ffp-contract=off: every add/mul in the integrator rounds like the reference's
# CPU build; the BVH slab tests opt back in with explicit __builtin_fmaf.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-munsafe-fp-atomics"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _depfile_deps(path):
    """Prerequisites listed in a compiler-written depfile (make syntax), or None when there is none yet."""
    if not os.path.exists(path):
        return None
    text = open(path).read().replace("\\\n", " ")
    deps = []
    for rule in text.split("\n"):
        if ":" in rule:
            deps += rule.split(":", 1)[1].split()
    return deps


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libgoblin_host.so")
    if force or _stale(out, HOST_DEPS):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-o", out] + HOST_SOURCES)
    return out


def _hip_object(src, force, extra_flags):
    obj = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")
    dep = obj + ".d"
    deps = _depfile_deps(dep)
    if force or deps is None or _stale(obj, [src] + deps):
        _run([HIPCC] + HIP_FLAGS + list(extra_flags) + ["-c", src, "-o", obj, "-MD", "-MF", dep])
        return obj, True
    return obj, False


def build_hip(force=False, extra_flags=(), jobs=None):
    """hipcc cross-compiles gfx950 code objects without a GPU present."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)
    out = os.path.join(LIB, "libgoblin_hip.so")
    jobs = jobs or int(os.environ.get("GBL_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        results = list(pool.map(lambda s: _hip_object(s, force, extra_flags), HIP_SOURCES))
    objs = [o for o, _ in results]
    if force or any(rebuilt for _, rebuilt in results) or _stale(out, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"])
    return out


def build_cli(force=False):
    """g_ray_hip: the stand-alone `g_ray scene.json` equivalent, linked against both libraries."""
    out = os.path.join(LIB, "g_ray_hip")
    src = os.path.join(CSRC, "host", "g_ray_hip.cpp")
    if force or _stale(out, [src, os.path.join(REPO, "include", "goblin_hip.h"), os.path.join(LIB, "libgoblin_hip.so"),
                             os.path.join(LIB, "libgoblin_host.so")]):
        _run([HIPCC, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", src, "-o", out, "-L" + LIB, "-lgoblin_hip",
              "-lgoblin_host", "-Wl,-rpath,$ORIGIN"])
    return out


TIE_INLINE_UNITS = ("kernels_path", "kernels_quad", "kernels_wavefront")


def build_tie_inline(force=False):
    """lib/variants/libgoblin_hip_tieinl.so: the library with the tie rule FORCED inline into the traversal loops of the path
    tracer's kernels (-DGBL_TIE_INLINE, kernels/trace.h) -- the form a StructurizeCFG bug of this compiler once miscompiled
    (tools/compiler_bugs/).  Test infrastructure: tests/test_gpu_ties.py renders a scene where every hit is a tie with it and
    with the shipped library; both must equal the oracle.  The other units are the regular build's objects."""
    from concurrent.futures import ThreadPoolExecutor
    vdir = os.path.join(LIB, "variants")
    odir = os.path.join(LIB, "obj_tieinl")
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    out = os.path.join(vdir, "libgoblin_hip_tieinl.so")

    def one(unit):
        src = os.path.join(CSRC, unit + ".hip")
        obj = os.path.join(odir, unit + ".o")
        dep = obj + ".d"
        deps = _depfile_deps(dep)
        if force or deps is None or _stale(obj, [src] + deps):
            _run([HIPCC] + HIP_FLAGS + ["-DGBL_TIE_INLINE", "-c", src, "-o", obj, "-MD", "-MF", dep])
        return obj
    with ThreadPoolExecutor(max_workers=len(TIE_INLINE_UNITS)) as pool:
        vobjs = list(pool.map(one, TIE_INLINE_UNITS))
    rest = [os.path.join(OBJ, os.path.splitext(os.path.basename(s))[0] + ".o") for s in HIP_SOURCES
            if os.path.splitext(os.path.basename(s))[0] not in TIE_INLINE_UNITS]
    if force or _stale(out, vobjs + rest):
        _run([HIPCC, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC", "-o", out] + vobjs + rest + ["-ldl", "-lpthread"])
    return out


def build_all(force=False):
    return build_host(force), build_hip(force), build_cli(force), build_tie_inline(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)


def _code_only(text):
    """C / C++ source without its comments and with runs of white space collapsed (string literals kept as they are)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def source_stamp():
    """sha256 (first 16 hex digits) over the CODE of every source file the HIP library is built from -- comments and white
    space do not count, so a note added beside a kernel does not orphan the counters collected from it: profiles/ records
    the stamp with the counters it collects, bench.py only quotes counters whose stamp is the running tree's."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for root, _, names in os.walk(CSRC):
        files += [os.path.join(root, n) for n in names if n.endswith((".h", ".hip", ".cpp"))]
    files.append(os.path.join(REPO, "include", "goblin_hip.h"))
    for f in sorted(files):
        h.update(os.path.relpath(f, REPO).encode())
        with open(f, "r", encoding="utf-8", errors="replace") as fh:
            h.update(_code_only(fh.read()).encode())
    return h.hexdigest()[:16]
