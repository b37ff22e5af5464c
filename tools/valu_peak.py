"""VALU issue peak of this GPU, measured (gbl_selftest_valu_issue: independent register chains, no memory): for every
instruction kind the traversal kernels are made of, 1-4 resident waves per SIMD on all CUs -> profiles/valu_issue_peak.json.
bench.py prices the VALU roofline against the v_fma_f32 figure of this file (cycles per wave64 instruction per SIMD)."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer

OPS = ["v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_perm_b32", "v_mov_b32_dpp quad_perm", "v_cndmask_b32",
       "v_and_b32", "v_rcp_f32", "v_med3_f32", "v_cmp_lt_f32"]
t = HipPathTracer(gs.load_scene("bunny", gs.config_overrides(resolution=(32, 32), spp=1, depth=2)), 0)
import torch
cus = torch.cuda.get_device_properties(0).multi_processor_count
simds = cus * 4
iters = 8192
res = {"note": "every CU runs one workgroup of 4*w waves (w per SIMD); each wave issues iters*64 instructions of one kind over 16 "
               "independent chains; ms = HIP events around the launch; ticks = s_memtime per wave, first to last instruction. "
               "cycles_per_instruction_per_simd = launch time x clock / (instructions each SIMD issued); the clock is taken as "
               "ticks_per_wave / launch time of the same launch (s_memtime counts shader cycles only if that ratio is ~2.4 GHz)",
       "cus": cus, "iters": iters, "ops": {}}
for op, name in enumerate(OPS):
    rows = []
    for w in (1, 2, 3, 4):
        r = t.valu_issue(op, w, iters)
        per_simd = r["wave_instructions"] / simds
        tick_hz = r["ticks_per_wave"] / (r["ms"] * 1e-3)
        rows.append({"waves_per_simd": w, "ms": round(r["ms"], 4), "wave_instructions": int(r["wave_instructions"]),
                     "g_wave_instructions_per_s": round(r["wave_instructions"] / (r["ms"] * 1e-3) * 1e-9, 2),
                     "ticks_per_instruction_of_a_wave": round(r["ticks_per_instruction"], 3),
                     "ticks_per_instruction_per_simd": round(r["ticks_per_instruction"] / w, 3),
                     "tick_rate_ghz": round(tick_hz * 1e-9, 4),
                     "ns_per_instruction_per_simd": round(r["ms"] * 1e6 / per_simd, 4)})
        print(name, rows[-1], flush=True)
    res["ops"][name] = rows
out = os.path.join(REPO, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/valu_issue_peak.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print("wrote", out)
