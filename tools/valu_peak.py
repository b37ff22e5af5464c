"""VALU issue peak of this GPU, measured (gbl_selftest_valu_issue: independent register chains, no memory): for the instruction
kinds the traversal kernels are made of, 1-4 resident waves per SIMD on all CUs -> profiles/valu_issue_peak.json.

Method (MI355X_MICROARCH.md, DVFS item 6): every figure is read from ONE launch of >= 10 ms made after two seconds of the same
launch back to back.  The clock is s_memtime / s_memrealtime x 100 MHz inside the waves.  The SIMD's rate is

    cycles_per_instruction_per_simd = LONGEST span of a wave (s_memrealtime) x clock / instructions the SIMD's waves issued

-- the longest span, not the average one: the waves of a SIMD start together, but the arbiter serves the older wave first, so they
finish one after the other and the average span of a wave is up to a third shorter than the time the SIMD was busy.  (Round 3
divided the average span by the number of waves and read 2.5 cycles for v_fma_f32 at four waves and a "throttled" 1.5 GHz clock;
the launch's own wall time, the cross-check printed beside each figure here, never agreed with that.)
    python tools/valu_peak.py [out.json] [--ops=fma,add,...]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer

OPS = ["v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_perm_b32", "v_mov_b32_dpp quad_perm", "v_cndmask_b32",
       "v_and_b32", "v_rcp_f32", "v_med3_f32", "v_cmp_lt_f32", "v_mul_f32", "v_fmac_f32", "v_max_f32", "v_mov_b32", "v_add_u32", "v_lshlrev_b32",
       "v_add_f32_e64 (VOP3 encoding)", "v_fma_f32 (two source registers)", "v_mul_lo_u32", "v_mul_u32_u24", "v_mul_hi_u32", "v_xor_b32", "v_lshrrev_b32",
       "v_cndmask_b32 (sgpr pair)"]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
only = None
for a in sys.argv[1:]:
    if a.startswith("--ops="):
        only = a[6:].split(",")
t = HipPathTracer(gs.load_scene("bunny", gs.config_overrides(resolution=(32, 32), spp=1, depth=2)), 0)
import torch
cus = torch.cuda.get_device_properties(0).multi_processor_count
simds = cus * 4
res = {"method": "one launch of >= 10 ms after 2 s of the same launch back to back; clock = s_memtime / s_memrealtime x 100 MHz inside the waves; "
                 "cycles_per_instruction_per_simd = longest span of a wave x clock / (w x instructions per wave); every CU runs one workgroup of 4 w waves "
                 "(w per SIMD), each wave iters x 64 instructions of one kind over 16 independent chains",
       "cus": cus, "ops": {}}
for op, name in enumerate(OPS):
    if only and not any(o in name for o in only):
        continue
    rows = []
    for w in (1, 2, 3, 4):
        iters = 65536 // w * 2          # >= 10 ms per launch at any of the rates seen
        r = t.valu_issue(op, w, iters)
        per_simd = r["wave_instructions"] / simds
        cyc = r["longest_span_us"] * 1e-6 * r["clock_ghz"] * 1e9 / per_simd
        row = {"waves_per_simd": w, "iters": iters, "ms": round(r["ms"], 3), "wave_instructions": int(r["wave_instructions"]),
               "clock_ghz": round(r["clock_ghz"], 4),
               "ticks_per_instruction_per_simd": round(cyc, 3),
               "from_wall_time": round(r["ms"] * 1e-3 * r["clock_ghz"] * 1e9 / per_simd, 3),
               "from_average_span_of_a_wave": round(r["ticks_per_instruction"] / w, 3),
               "span_us": {"longest": round(r["longest_span_us"], 1), "shortest": round(r["shortest_span_us"], 1)},
               "g_wave_instructions_per_s": round(r["wave_instructions"] / (r["ms"] * 1e-3) * 1e-9, 2)}
        rows.append(row)
        print(name, row, flush=True)
    res["ops"][name] = rows
out = os.path.join(REPO, args[0] if args else "gpurun_out/valu_issue_peak.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print("wrote", out)
