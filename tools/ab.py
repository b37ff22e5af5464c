"""A/B of libgoblin_hip builds on one box: best-of-N kernel ms on the BASELINE scenes, both schedules, one subprocess per library.
    python tools/ab.py [--quick] base.so ... (names under goblin_amd/lib/variants/, or 'main' for the shipped library)"""
import sys, os, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch, os
sys.path.insert(0, %r)
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
quick = os.environ.get("AB_QUICK") == "1"
cases = [("bunny", dict(resolution=(512, 512), spp=256, depth=8), ["megakernel"] if quick else ["megakernel", "wavefront"]),
         ("cornell", dict(resolution=(512, 512), spp=64, depth=16), ["megakernel", "wavefront"]),
         ("grid", dict(resolution=(512, 512), spp=64, depth=8), ["megakernel", "wavefront"]),
         ("bunny", dict(resolution=(1024, 1024), spp=16, method="ao", ao_samples=25), ["auto"])]
if not quick:
    cases += [("grid", dict(resolution=(1024, 1024), spp=16, method="ao", ao_samples=25), ["auto"]),
              ("shapes", dict(resolution=(512, 512), spp=64), ["megakernel"]), ("masked", dict(resolution=(512, 512), spp=64), ["megakernel"])]
row = {}
for sc_name, kw, scheds in cases:
    tr = HipPathTracer(gs.load_scene(sc_name, gs.config_overrides(**kw)), 0)
    film = tr.new_film()
    for sch in scheds:
        best = 1e30
        for i in range(4):
            film.zero_()
            out = tr.render(film=film, seed=1, stats=False, timed=True, schedule=sch)
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
        row["%%s%%s/%%s" %% (sc_name, "-ao" if "method" in kw else "", sch[:2])] = round(best, 2)
        row.setdefault("_mean", []).append(round(float(film.normalized().mean()), 6))
print(json.dumps(row), flush=True)
''' % REPO
args = [a for a in sys.argv[1:] if a != "--quick"]
for lib in args or ["main"]:
    env = dict(os.environ, AB_QUICK="1" if "--quick" in sys.argv else "0")
    if lib != "main":
        env["GOBLIN_HIP_LIB"] = os.path.join(REPO, "goblin_amd", "lib", "variants", lib if lib.endswith(".so") else "libgoblin_hip_%s.so" % lib)
    sys.stdout.write("%-12s " % lib)
    sys.stdout.flush()
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
