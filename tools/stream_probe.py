"""Phase shares of the stream sampler's tile walk on configs[1].
    instrumented build (its own probes):   python tools/stream_probe.py
    the lean build:   python tools/build_variant.py streamtm kernels_stream -DGBL_STREAM_TM   (here), then on the GPU box
                      GOBLIN_HIP_LIB=goblin_amd/lib/variants/libgoblin_hip_streamtm.so python tools/stream_probe.py lean"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
lean = len(sys.argv) > 1 and sys.argv[1] == "lean"
os.environ["GBL_PHASE_CLOCK" if lean else "GBL_PROBE"] = "1"
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
scene = gs.load_scene("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8))
tr = HipPathTracer(scene, 0)
for i in range(2):
    out = tr.render(sampler="stream", stats=not lean, timed=True, schedule="megakernel")
print(out["stats"])
