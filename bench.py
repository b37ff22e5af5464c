#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the HIP path tracer on N MI355X GPUs.

    python bench.py                      # 1 GPU: BASELINE configs[1], 5 steps, 1 warmup
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over the workload with the scene already resident in HBM: zero the film,
trace every camera path of the sample window, splat, and -- for N > 1 -- sum the film over the ranks with one RCCL
all-reduce.  Rank 0 prints ONE JSON line.

Workloads (--workload):
  bunny  BASELINE configs[1]: bunny.json, 512x512 film (516x516 sampled px), 256 spp, max_ray_depth 8 = 68,161,536
         paths per step.  The default at N = 1: the configuration the headline metric is quoted on.
  grid   BASELINE configs[3]: grid.json, 15 instances of the 69k-triangle bunny (1.04 M instanced triangles), 1024x1024
         film, 256 spp, max_ray_depth 8 = 270,536,704 paths per step.  The default for N > 1, as north_star words it:
         "image tiles shard across the 8 GPUs of one node with a final RCCL reduce of Film tiles".
  cornell / ao  BASELINE configs[2] / configs[4].
At N = 1 the default run also renders ONE full-size step of each of the other three configurations after the headline's
timed region (`other_configs` in the line), so that every BASELINE configuration has a number the driver saw.

Roofline (`roofline` in the line): SURVEY.md 8(d)'s figure -- algorithmic bytes of the dominant kernel over its measured
launch time against the 8 TB/s HBM peak -- with the HBM bytes that really moved (`traffic`, PMC) beside it.  The scenes are
cache resident, so the kernel is not HBM bound; what the counters say does bound it is under `roofline.issue`: VALU
wave-instructions per second against the MEASURED issue peak of this chip (profiles/valu_issue_peak.json, tools/valu_peak.py)
and the share of the waves' cycles spent parked on memory.

Multi-GPU work split (goblin_amd/distributed.py), --scaling:
  strong (default for N > 1) rank r traces every N-th 8x8 sample tile of the ONE frame (Film::mergeTile's sum over
         per-thread full-film tiles, GoblinFilm.cpp:140-153, becomes one all-reduce of the W x H float4 accumulators).
  weak   every rank traces the whole window with its own sample set (seed + rank): the reduced film holds N x spp.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
SIMDS = 256 * 4


def survey_8d_bytes(st):
    """SURVEY.md 8(d), literally: per ray 32 B per BVH node visited (the reference's CompactBVHNode, GoblinBVH.h:8-16) + 48 B
    per triangle tested + 48 B (32 B ray in, 16 B hit out); per path 4 B per sample dimension consumed + 16 B of radiance
    written.  st["nodes"] counts child boxes tested, 4 per visit of this build's 4-wide nodes.  (The film splat's 16 B per
    touched pixel belongs to the splat kernel.)"""
    rays = st["extension_rays"] + st["shadow_rays"]
    return 32 * (st["nodes"] // 4) + 48 * st["tris"] + 48 * rays + 4 * st["dims"] + 16 * st["paths"]


def build_bytes(st):
    """The same with this build's real node: one visit fetches 64 B (four 16-B quantised child boxes)."""
    rays = st["extension_rays"] + st["shadow_rays"]
    return 16 * st["nodes"] + 48 * st["tris"] + 48 * rays + 4 * st["dims"] + 16 * st["paths"]


def _template_arg(kernel_name, index):
    """The index-th template argument of a demangled kernel name ('void wf_trace<false, true, ...>(...)' -> 'true' for 1)."""
    a = kernel_name.find("<")
    b = kernel_name.find(">", a)
    args = [x.strip() for x in kernel_name[a + 1:b].split(",")] if a >= 0 and b > a else []
    return args[index] if index < len(args) else None


# the un-instrumented lean kernels a standard run launches, by (integrator, schedule)
# (a path-tracing step under the megakernel schedule is two launches: the primary pass -- the camera rays as per-pixel packets,
#  kernels/packet.h -- and the path kernel whose paths start at those hits; the first name is the one the step spends its time in)
LEAN_KERNELS = {("path", "megakernel"): ("void path_trace_kernel<0, false, false, true, false, true>(DevScene, RenderArgs)",
                                         "void primary_kernel<false>(DevScene, RenderArgs, HIP_vector_type<float, 4u>*, int*)"),
                ("ao", "megakernel"): ("void ao_kernel<0, false, false, true, false>(DevScene, RenderArgs)",)}


def pmc_for(workload, schedule, integrator, frame_spp=None):
    """Counters of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/pmc_<workload>_<schedule>.json,
    written by tools/summarize_profile.py from tools/profile_gpu.sh's passes) -- only if they were collected from THIS source
    tree (the stamp is recorded on the GPU box at collection time)."""
    from goblin_amd import build
    path = os.path.join(REPO, "profiles", "pmc_%s_%s.json" % (workload, schedule))
    if not os.path.exists(path):
        return None, "no profiles/pmc_%s_%s.json" % (workload, schedule)
    with open(path) as f:
        d = json.load(f)
    stamp = build.source_stamp()
    if d.get("source_stamp") != stamp:
        return None, "%s was collected from source stamp %s, this tree is %s" % (os.path.basename(path), d.get("source_stamp"), stamp)
    # a frame profiled at fewer samples per pixel (the Cornell and AO frames take seconds per pass at full size): the same
    # kernels over the same window, so the extensive counters and the kernel time scale with the samples; the ratios do not
    factor = 1.0
    if d.get("spp_profiled") and frame_spp and d["spp_profiled"] != frame_spp:
        factor = frame_spp / float(d["spp_profiled"])
    keys = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES",
            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_INSTS_LDS")
    if schedule == "wavefront":
        # a step is hundreds of launches of three kernels: the counters of one step = sum over the un-instrumented wavefront
        # kernels of (counters per launch x launches per render); their kernel time likewise, from the same kernel trace
        calls = d.get("kernel_calls", {})
        renders = sum(n for k, n in calls.items() if "wf_splat<" in k and _template_arg(k, 1) == "false")
        agg, kernels = {}, []
        for name, c in d.get("counters_per_launch", {}).items():
            if not name.startswith("void wf_") or _template_arg(name, 1) != "false" or name not in calls or not renders:
                continue
            per_render = calls[name] / renders
            kernels.append("%s x %g" % (name.split("(")[0].replace("void ", ""), per_render))
            for key in keys:
                if key in c:
                    agg[key] = agg.get(key, 0.0) + c[key] * per_render
            agg["kernel_avg_ms"] = agg.get("kernel_avg_ms", 0.0) + c.get("kernel_avg_ms", 0.0) * per_render
        if agg.get("SQ_INSTS_VALU"):
            agg = {k: v * factor for k, v in agg.items()}
            # resident waves per SIMD of the kernels the step spends its time in: the two trace kernels (register allocation
            # as the profiler saw it at launch)
            regs = kernel_registers()
            tr = [regs[k] for k in d.get("counters_per_launch", {}) if k.startswith("void wf_trace<") and _template_arg(k, 1) == "false" and k in regs]
            waves = min(waves_per_simd_of(v, a) for v, a in tr) if tr else None
            return dict(agg, kernel="one step: " + ", ".join(kernels), file=os.path.relpath(path, REPO), source_stamp=stamp, waves_per_simd=waves,
                        scaled_from_spp=d.get("spp_profiled") if factor != 1.0 else None), None
        return None, "no wavefront kernels in %s" % os.path.basename(path)
    names = LEAN_KERNELS.get((integrator, schedule), ())
    per_launch = d.get("counters_per_launch", {})
    if names and all(n in per_launch for n in names):
        # one launch of each per step: the step's counters and kernel time are their sums (ratios are taken of the sums)
        c = {}
        for n in names:
            for k, v in per_launch[n].items():
                if isinstance(v, (int, float)) and k in keys + ("kernel_avg_ms",):
                    c[k] = c.get(k, 0.0) + v * factor
        li = kernel_registers().get(names[0])
        return dict(c, kernel=" + ".join(n.split("(")[0].replace("void ", "") for n in names), file=os.path.relpath(path, REPO), source_stamp=stamp,
                    waves_per_simd=waves_per_simd_of(li[0], li[1]) if li else None,
                    scaled_from_spp=d.get("spp_profiled") if factor != 1.0 else None), None
    return None, "not every kernel of %r in %s" % (names, os.path.basename(path))


def valu_issue_peak(waves_per_simd):
    """The measured VALU issue peak (tools/valu_peak.py -> profiles/valu_issue_peak.json): v_fma_f32 and v_add_f32, independent
    chains, `waves_per_simd` resident waves on every SIMD of every CU; one >= 10 ms launch after 2 s of the same launch back to
    back, cycles from s_memtime, the clock from s_memtime / s_memrealtime."""
    path = os.path.join(REPO, "profiles", "valu_issue_peak.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    out = {"file": "profiles/valu_issue_peak.json", "waves_per_simd": waves_per_simd}
    for op, tag in (("v_fma_f32", "fma"), ("v_add_f32", "add")):
        rows = [r for r in d["ops"].get(op, []) if r["waves_per_simd"] == waves_per_simd]
        if not rows:
            return None
        out[tag + "_cycles_per_instruction_per_simd"] = rows[0]["ticks_per_instruction_per_simd"]
        out[tag + "_clock_ghz_under_that_load"] = rows[0].get("clock_ghz", rows[0].get("tick_rate_ghz"))
    return out


_KERNEL_REGS = None


def kernel_registers():
    """{demangled kernel name: (vgpr, agpr)} of the library this process runs, read from its code objects' metadata
    (tools/kernel_resources.py).  (rocprofv3's VGPR_Count column is in allocation units of two registers: not used.)"""
    global _KERNEL_REGS
    if _KERNEL_REGS is None:
        _KERNEL_REGS = {}
        try:
            lib = os.environ.get("GOBLIN_HIP_LIB") or os.path.join(REPO, "goblin_amd", "lib", "libgoblin_hip.so")
            out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "kernel_resources.py"), "--json", lib], capture_output=True, text=True, timeout=120).stdout
            for r in json.loads(out):
                _KERNEL_REGS[r["name"]] = (int(r.get("vgpr_count", 0)), int(r.get("agpr_count", 0)))
        except Exception as e:
            print("kernel_registers: %s" % e, file=sys.stderr)
    return _KERNEL_REGS


def waves_per_simd_of(vgpr, agpr=0):
    """Resident waves per SIMD a kernel's register allocation allows (MI355X_MICROARCH.md: 512 registers per lane per SIMD,
    allocated in steps of 8, at most 8 waves)."""
    alloc = max(8, (int(vgpr) + int(agpr) + 7) // 8 * 8)
    return max(1, min(8, 512 // alloc))


def issue_reading(r, issue):
    """What the counters of this line say bounds the kernel -- composed from them, so that no sentence can contradict the
    field beside it."""
    parts = []
    toa = r.get("traffic_over_algorithmic")
    if toa is not None:
        if toa < 0.1:
            parts.append("cache resident: %.1f %% of the algorithmic bytes reach HBM (%.0f GB/s, %.1f %% of the 8 TB/s peak), so HBM does not bound this kernel"
                         % (100 * toa, r.get("traffic_gbps", 0.0), 100 * r.get("traffic_gbps", 0.0) / HBM_PEAK_GBPS))
        else:
            parts.append("HBM traffic is %.2f x the algorithmic bytes (%.0f GB/s, %.0f %% of the 8 TB/s peak): the path state streams through HBM every iteration"
                         % (toa, r.get("traffic_gbps", 0.0), 100 * r.get("traffic_gbps", 0.0) / HBM_PEAK_GBPS))
    if "l2_hit_rate" in issue:
        parts.append("L2 hit rate %.2f" % issue["l2_hit_rate"])
    if "wait_frac_pmc" in issue:
        parts.append("of the waves' cycles %.0f %% are parked on memory (s_waitcnt), %.0f %% issue-stalled, %.0f %% issuing"
                     % (100 * issue["wait_frac_pmc"], 100 * issue.get("issue_stall_frac_pmc", 0.0), 100 * issue.get("active_frac_pmc", 0.0)))
    cpi = issue.get("cycles_per_instruction_per_simd")
    peak = issue.get("measured_peak")
    if cpi and peak:
        parts.append("a SIMD issues one wave64 VALU instruction per %.2f cycles at %d waves per SIMD where the microbenchmark sustains one per %.2f"
                     % (cpi, issue["waves_per_simd"], peak["fma_cycles_per_instruction_per_simd"]))
    elif cpi:
        parts.append("a SIMD issues one wave64 VALU instruction per %.2f cycles at %d waves per SIMD" % (cpi, issue["waves_per_simd"]))
    if "lane_util_pmc" in issue:
        parts.append("%.0f %% of the lanes of an issued VALU instruction are switched on" % (100 * issue["lane_util_pmc"]))
    verdict = None
    issue_frac = (peak["fma_cycles_per_instruction_per_simd"] / cpi) if (cpi and peak) else None
    if issue_frac is not None and issue_frac >= 0.8:
        verdict = ("VALU issue bound: the SIMDs issue at %.2f of what they can (what one wave waits for, the other waves' instructions fill) -- with %.0f %% of "
                   "the lanes on, the lever is lane utilisation and instructions per ray, not memory" % (issue_frac, 100 * issue.get("lane_util_pmc", 0.0)))
    elif toa is not None and toa >= 0.5 and issue.get("wait_frac_pmc", 0) >= 0.45:
        verdict = "bound by memory: state traffic and the latency of the traversal's dependent fetches"
    elif issue.get("wait_frac_pmc", 0) >= 0.35:
        verdict = "latency bound (dependent node / triangle fetches) under low lane utilisation, not bandwidth bound"
    if verdict:
        parts.append(verdict)
    return "; ".join(parts)


def roofline_object(counted, kernel_ms, pmc, pmc_note, kernel_label, waves_per_simd=3):
    """SURVEY 8(d)'s HBM-roofline object for one step's dominant kernel(s), with the issue-side evidence beside it."""
    b8d, bb = survey_8d_bytes(counted), build_bytes(counted)
    sec = kernel_ms * 1e-3
    ach = b8d / sec / 1e9
    traffic = int((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024) if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc else None
    r = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
         "traffic": traffic,
         "formula": "SURVEY 8(d): 32 B x node visits + 48 B x triangles tested + 48 B x rays + 4 B x sample dimensions + 16 B x paths, over the "
                    "kernel's measured launch time",
         "algorithmic_bytes_per_launch": int(b8d),
         "with_this_builds_64B_nodes": {"bytes_per_launch": int(bb), "gbps": round(bb / sec / 1e9, 2), "frac": round(bb / sec / 1e9 / HBM_PEAK_GBPS, 4)},
         "kernel": kernel_label, "kernel_ms_avg": round(kernel_ms, 3),
         "counters": {k: int(v) for k, v in counted.items() if k not in ("kernel_ms", "reserved")}}
    if traffic:
        r["traffic_gbps"] = round(traffic / sec / 1e9, 1)
        r["traffic_over_algorithmic"] = round(traffic / b8d, 4)
    if pmc and pmc.get("SQ_INSTS_VALU") and pmc.get("GRBM_GUI_ACTIVE"):
        clock_hz = pmc["GRBM_GUI_ACTIVE"] / 8.0 / (pmc["kernel_avg_ms"] * 1e-3)
        ach_i = pmc["SQ_INSTS_VALU"] / sec * 1e-9
        issue = {"valu_wave_instructions_per_launch": int(pmc["SQ_INSTS_VALU"]), "achieved_g_wave_instructions_per_s": round(ach_i, 1),
                 "cycles_per_instruction_per_simd": round(SIMDS * clock_hz * sec / pmc["SQ_INSTS_VALU"], 3),
                 "clock_ghz_pmc": round(clock_hz * 1e-9, 3),
                 "pmc": {"file": pmc["file"], "source_stamp": pmc["source_stamp"], "kernel": pmc["kernel"], "kernel_avg_ms_rocprof": pmc["kernel_avg_ms"],
                         "scaled_from_spp": pmc.get("scaled_from_spp")}}
        if "TCC_HIT_sum" in pmc:
            issue["l2_hit_rate"] = round(pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), 4)
        if pmc.get("SQ_WAVE_CYCLES"):
            for key, out in (("SQ_WAIT_ANY", "wait_frac_pmc"), ("SQ_WAIT_INST_ANY", "issue_stall_frac_pmc"), ("SQ_ACTIVE_INST_ANY", "active_frac_pmc")):
                if key in pmc:
                    issue[out] = round(pmc[key] / pmc["SQ_WAVE_CYCLES"], 4)
        if pmc.get("SQ_THREAD_CYCLES_VALU") and pmc.get("SQ_ACTIVE_INST_VALU"):
            issue["lane_util_pmc"] = round(pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"]), 4)
        waves_per_simd = pmc.get("waves_per_simd") or waves_per_simd
        issue["waves_per_simd"] = int(waves_per_simd)
        peak = valu_issue_peak(min(4, waves_per_simd))   # (the microbenchmark covers 1-4 resident waves per SIMD)
        if peak:
            # in cycles: what a SIMD with this many waves can issue (measured, independent v_fma_f32) against what the kernel's
            # instruction stream took per instruction, each at the clock its own launch held
            issue["measured_peak"] = peak
            issue["frac_of_fma_issue_peak_in_cycles"] = round(peak["fma_cycles_per_instruction_per_simd"] / issue["cycles_per_instruction_per_simd"], 4)
        issue["reading"] = issue_reading(r, issue)
        r["issue"] = issue
    else:
        r["pmc_note"] = pmc_note
    return r


def cpu_baseline(workload_overrides, spp_sample, cores, scene_name="bunny", what="bunny.json 512x512", full_spp=256, port=True):
    """Time the CPU path on a bounded sample of the same workload (same scene, film,
    filter and depth; fewer samples per pixel).  Prefers the REAL reference built into
    oracle/_ref (kind "reference"); falls back to the oracle port."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from goblin_amd import scene as gs
    ov = json.loads(json.dumps(workload_overrides))
    ov.setdefault("render_setting", {})["sample_per_pixel"] = spp_sample
    harness = os.path.join(REPO, "oracle", "_ref", "ref_harness")
    sample = "%s, %d spp (of %d), %d threads" % (what, spp_sample, full_spp, cores)
    if spp_sample >= full_spp:
        sample = "%s, %d spp (the whole workload), %d threads" % (what, full_spp, cores)
    out = None
    if os.path.exists(harness):
        try:
            src = gs.scene_path(scene_name)
            with open(src) as f:
                doc = json.load(f)
            gs._merge(doc, ov)
            doc["render_setting"]["thread_num"] = cores
            for g in doc.get("geometries", []):
                if "file" in g:
                    g["file"] = os.path.join(os.path.dirname(src), g["file"])
            with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as tf:
                json.dump(doc, tf)
            prefix = tf.name[:-5]
            try:
                res = json.loads(subprocess.check_output([harness, "film", tf.name, prefix, str(cores)], timeout=600).decode())
                import numpy as np
                ref_film = np.fromfile(prefix + ".film.f32", np.float32).reshape(res["yres"], res["xres"], 4)
            finally:
                os.unlink(tf.name)
                if os.path.exists(prefix + ".film.f32"):
                    os.unlink(prefix + ".film.f32")
            out = {"value": round(res["mpaths_per_s"], 4), "unit": "Mpaths/s", "cores": cores, "kind": "reference",
                   "sample": sample + ", %d paths in %.2f s (oracle/_ref/ref_harness = /root/reference/src compiled as-is)"
                   % (res["paths"], res["seconds"]), "_film": ref_film}
        except Exception as e:  # the prebuilt binary may be absent or unusable on this box
            print("cpu_baseline: reference harness failed (%s); using the oracle port" % e, file=sys.stderr)
    if not port and out is not None:
        return out
    try:
        import oracle_binding as ob
        ov_port = json.loads(json.dumps(ov))
        ov_port["render_setting"]["sample_per_pixel"] = min(spp_sample, 16)   # the port is a side note: keep it short
        scene = gs.load_scene(scene_name, ov_port)
        oracle = ob.Oracle(scene)
        res = oracle.render(threads=cores, ref_faithful=1)
        port = scene.num_paths() / res["seconds"] * 1e-6
        if out is None:
            out = {"value": round(port, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
                   "sample": sample + ", %d paths in %.2f s (oracle port incl. the reference's redundant traversals)"
                   % (scene.num_paths(), res["seconds"])}
        else:
            out["port_value"] = round(port, 4)
    except Exception as e:
        print("cpu_baseline: oracle port failed: %s" % e, file=sys.stderr)
    return out


def l2_vs_reference(tracer, ref_film, spp_sample):
    """Per-pixel L2 of the normalised film against the Film the compiled reference itself rendered for the
    cpu_baseline leg: the device regenerates the reference's own sample stream (GBL_SAMPLES_STREAM), so the two
    films hold the same samples and differ by float summation order only."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_binding as ob
    from goblin_amd import _abi
    s = _abi.gbl_render_setting.from_buffer_copy(tracer.scene.desc.setting)
    s.sample_per_pixel = spp_sample
    gpu = tracer.render(setting=s, sampler="stream", schedule="megakernel")["film"].numpy()
    a, b = ob.normalize_film(gpu).astype(np.float64), ob.normalize_film(ref_film).astype(np.float64)
    wdiff = np.abs(gpu[..., 3] - ref_film[..., 3]) / np.maximum(ref_film[..., 3], 1e-9)
    return {"rel_l2": float(np.linalg.norm(a - b) / np.linalg.norm(b)),
            "rmse": float(np.sqrt(np.mean((a - b) ** 2))),
            "pixels_on_other_samples": int((wdiff > 1e-4).sum()),
            "pixels": int(wdiff.size),
            "sample": "512x512 film, %d spp: oracle/_ref/ref_harness's Film vs the device rendering the reference's own "
                      "mt19937 sample stream" % spp_sample,
            "note": "a pixel counts as on other samples when its filter-weight sum differs: a path that hits the shared "
                    "edge of two triangles at exactly equal t resolves the tie by BVH visiting order, draws a different "
                    "number of floats and shifts the rest of its 8x8 tile's stream (DESIGN.md 5)"}


def l2_vs_cpu(tracer, workload_overrides, spp_sample, cores, seed):
    """Per-pixel L2 of the normalised film against the CPU oracle on identical samples
    (native sampler, same seed), at the bounded spp."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_binding as ob
    from goblin_amd import _abi
    scene = tracer.scene
    s = _abi.gbl_render_setting.from_buffer_copy(scene.desc.setting)
    s.sample_per_pixel = spp_sample
    oracle = ob.Oracle(scene)
    import helpers
    res = oracle.render(setting=s, threads=cores, sampler=1, seed=seed, want_samples=True)
    cpu = res["film"]
    # the same records on the device: replayed, so that the kernel is the one with the reference's tie rule compiled in
    # (the native sampler's lean kernel leaves exact-t ties to its own tree)
    idx = helpers.tile_order_index(oracle.window(), int(np.ceil(np.sqrt(np.float32(spp_sample)))) ** 2)
    gpu = tracer.render(setting=s, replay_samples=res["samples"][idx])["film"].numpy()
    a, b = ob.normalize_film(gpu).astype(np.float64), ob.normalize_film(cpu).astype(np.float64)
    return {"rel_l2": float(np.linalg.norm(a - b) / np.linalg.norm(b)),
            "rmse": float(np.sqrt(np.mean((a - b) ** 2))),
            "sample": "512x512 film, %d spp, the oracle's counter-based samples replayed on the device" % spp_sample}


WORKLOADS = {
    # name: (scene, resolution, spp, depth, what it is[, extra overrides])
    "bunny": ("bunny", (512, 512), 256, 8, "BASELINE configs[1]: bunny.json, glass stand-in bunny (69120 tris) on a plane, spot light"),
    "grid": ("grid", (1024, 1024), 256, 8, "BASELINE configs[3]: grid.json, 15 instances of the bunny BLAS (1.04 M instanced triangles)"),
    "cornell": ("cornell", (1024, 1024), 1024, 16, "BASELINE configs[2]: cornell.json, Cornell box + glass bunny, area light, divergent BSDF mix"),
    "ao": ("bunny", (2048, 2048), 4096, 8, "BASELINE configs[4]: bunny.json under the AO integrator (1 closest hit + 25 any-hit rays per camera sample)",
           {"method": "ao", "ao_samples": 25}),
}


class Workload:
    """One configuration resident on the device: scene, tracer, film, and the band structure of a frame."""

    def __init__(self, name, device_index, res=None, spp=None, depth=None):
        from goblin_amd import scene as gs
        from goblin_amd.renderer import HipPathTracer
        w = WORKLOADS[name]
        self.name, self.scene_name, self.text = name, w[0], w[4]
        self.extra = w[5] if len(w) > 5 else {}
        self.res = tuple(res) if res else w[1]
        self.spp = spp or w[2]
        self.depth = depth or w[3]
        self.standard = (self.res, self.spp, self.depth) == w[1:4]
        self.integrator = "ao" if self.extra.get("method") == "ao" else "path"
        self.overrides = gs.config_overrides(resolution=self.res, spp=self.spp, depth=self.depth, **self.extra)
        self.scene = gs.load_scene(self.scene_name, self.overrides)
        self.tracer = HipPathTracer(self.scene, device_index)
        self.film = self.tracer.new_film()
        # a call takes fewer than 2^32 camera samples (configs[4] at its 4096 spp has 1.7e10): such a frame is rendered in
        # bands of 8-pixel tile rows, one call each, all of them inside the timed step
        x0, x1, y0, y1 = self.tracer.window
        rows = max(8, ((1 << 31) // max(1, (x1 - x0) * self.scene.spp())) // 8 * 8)
        self.bands = [(x0, x1, y, min(y1, y + rows)) for y in range(y0, y1, rows)]
        if len(self.bands) == 1:
            self.bands = [None]

    def render_frame(self, target=None, setting=None, **kw):
        out = None
        for w in self.bands:
            r = self.tracer.render(film=target or self.film, window=w, setting=setting, **kw)
            if out is None:
                out = r
            elif r["stats"]:
                for k, v in r["stats"].items():
                    if k not in ("schedule", "reserved"):   # (what the call ran under: not a counter)
                        out["stats"][k] += v
        return out

    def describe(self):
        wpx = (self.tracer.window[1] - self.tracer.window[0], self.tracer.window[3] - self.tracer.window[2])
        return "%s -- %dx%d film (%dx%d sampled px), %d spp, %s, gaussian r=2" % (
            self.text, self.res[0], self.res[1], wpx[0], wpx[1], self.scene.spp(),
            "%d occlusion rays per camera sample" % self.extra["ao_samples"] if self.integrator == "ao" else "max_ray_depth %d" % self.depth)

    def kernel_label(self, resolved):
        if resolved == "wavefront":
            return "wf_trace / wf_shade (all wavefront kernels of a step)"
        if self.integrator == "ao":
            return "ao_kernel<native sampler, lean, quad-per-ray queries>"
        return "primary_kernel (camera rays as per-pixel packets) + path_trace_kernel<native sampler, lean, quad-per-ray queries, paths start at the primary hits>"


def run_steps(render, zero_film, allreduce, barrier, sync, steps, warmup, world, make_event=None):
    """The timed region of the contract: W untimed warmup steps, then exactly K steps bracketed by barrier + device sync on
    both sides.  `render()` traces this rank's share into its film, `allreduce()` sums the films (N > 1).
    Returns (elapsed seconds on this rank, per-step (render_ms, reduce_ms) from device events or None)."""
    def step(ev=None):
        zero_film()
        if ev:
            ev[0].record()
        render()
        if ev:
            ev[1].record()
        if world > 1:
            allreduce()
        if ev:
            ev[2].record()

    for _ in range(warmup):
        step()
    events = [tuple(make_event() for _ in range(3)) for _ in range(steps)] if make_event else [None] * steps
    barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(events[i])
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = [(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])) for e in events] if make_event else None
    return elapsed, per_step


def counters_of(wl, seed, shard, schedule, scale_from_spp=None):
    """Node / triangle / ray counters of one step (an instrumented launch of the same deterministic work).  A frame whose
    instrumented launch would take tens of seconds is counted at `scale_from_spp` samples per pixel and scaled: the sampler is
    stratified per pixel, so the per-path averages of a lower spp are the frame's to well under a per cent."""
    from goblin_amd import _abi
    if scale_from_spp and scale_from_spp < wl.scene.spp():
        s = _abi.gbl_render_setting.from_buffer_copy(wl.scene.desc.setting)
        s.sample_per_pixel = scale_from_spp
        x0, x1, y0, y1 = wl.tracer.window
        st = wl.tracer.render(film=wl.tracer.new_film(), setting=s, seed=seed, shard=shard, stats=True, schedule=schedule)["stats"]
        k = wl.scene.spp() / float(scale_from_spp)
        out = {key: (int(round(v * k)) if key not in ("schedule", "reserved", "kernel_ms") else v) for key, v in st.items()}
        out["_scaled_from_spp"] = scale_from_spp
        return out
    return wl.render_frame(seed=seed, shard=shard, stats=True, schedule=schedule)["stats"]


def one_config(name, device_index, seed, torch, steps=1, cpu_cores=0):
    """`steps` full-size steps of another BASELINE configuration (N = 1, after the headline's timed region): the median step's
    device time and rates, the schedule AUTO resolved to, counters, SURVEY 8(d)'s fraction, and -- with cpu_cores -- the compiled
    reference on a bounded sample of the same frame."""
    from goblin_amd import _abi
    wl = Workload(name, device_index)
    # warm-up (allocates the radiance buffer / the wavefront pool): a full step where that is a few seconds, 16 spp for the
    # AO frame (25 s at its 4096 spp)
    if wl.integrator == "ao":
        s = _abi.gbl_render_setting.from_buffer_copy(wl.scene.desc.setting)
        s.sample_per_pixel = 16
        wl.tracer.render(film=wl.film, setting=s, seed=seed)
    else:
        wl.render_frame(seed=seed)
    torch.cuda.synchronize()
    runs = []
    for _ in range(steps):
        wl.film.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        full = wl.render_frame(seed=seed, timed=True)   # (timed: the library also reports the schedule AUTO resolved this call to)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        timings = wl.tracer.timings(len(wl.bands))
        runs.append((ms, sum(x[0] for x in timings) if timings else ms))
    runs.sort()
    ms, kernel_ms = runs[len(runs) // 2]
    resolved = {1: "megakernel", 2: "wavefront"}.get(full["stats"]["schedule"], "megakernel")
    counted = counters_of(wl, seed, None, resolved, scale_from_spp=64 if wl.scene.spp() > 256 else None)
    paths = counted["paths"]
    rays = counted["extension_rays"] + counted["shadow_rays"]
    pmc, note = pmc_for(name, resolved, wl.integrator, wl.scene.spp())
    roof = roofline_object({k: v for k, v in counted.items() if not k.startswith("_")}, kernel_ms, pmc, note, wl.kernel_label(resolved),
                           waves_per_simd=5 if resolved == "wavefront" else 3)
    out = {"workload": wl.describe(), "schedule": resolved, "steps": steps, "ms_per_step": round(ms, 2), "ms_all_steps": [round(r[0], 2) for r in runs],
           "kernel_ms": round(kernel_ms, 2),
           "paths_per_step": int(paths), "value": round(paths / ms * 1e-3, 2), "unit": "Mpaths/s",
           "rays_per_path": round(rays / max(1, paths), 3), "grays_per_s": round(rays / ms * 1e-6, 3), "roofline": roof}
    if steps > 1:
        out["note"] = "median of %d steps" % steps
    if "_scaled_from_spp" in counted:
        out["counters_note"] = "counted by an instrumented launch at %d spp and scaled to the frame's %d" % (counted["_scaled_from_spp"], wl.scene.spp())
    if cpu_cores:
        # the compiled reference on the same frame at a few samples per pixel (its rate does not depend on the count: the
        # sampler is stratified per pixel); ~5-15 s each on a many-core host
        cpu_spp = {"cornell": 16, "grid": 16, "ao": 4}[name]
        try:
            cb = cpu_baseline(wl.overrides, cpu_spp, cpu_cores, scene_name=wl.scene_name,
                              what="%s.json %dx%d%s" % (wl.scene_name, wl.res[0], wl.res[1], ", AO" if wl.integrator == "ao" else ""),
                              full_spp=wl.scene.spp(), port=False)
            if cb:
                cb.pop("_film", None)
                out["cpu_baseline"] = cb
                out["gpu_over_cpu"] = round(out["value"] / cb["value"], 1) if cb.get("value") else None
        except Exception as e:
            print("other_configs[%s] cpu_baseline failed: %s" % (name, e), file=sys.stderr)
    del wl
    torch.cuda.empty_cache()
    return out


FEATURE_SCENES = ("whitted", "hetero", "masked")


def feature_scenes(device_index, seed, torch):
    """One step each of three scenes outside the headline feature set: the Whitted integrator, a
    heterogeneous participating medium, mask materials -- 512x512 film, 64 spp (17 M camera paths), AUTO schedule.  The best of
    three steps; no roofline (their EXT kernels are shading bound)."""
    from goblin_amd import scene as gs
    from goblin_amd.renderer import HipPathTracer
    out = {}
    for name in FEATURE_SCENES:
        try:
            scene = gs.load_scene(name, gs.config_overrides(resolution=(512, 512), spp=64))
            tr = HipPathTracer(scene, device_index)
            film = tr.new_film()
            best, paths = 1e30, 0
            for _ in range(3):
                film.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = tr.render(film=film, seed=seed)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e3)
                paths = r["paths"]
            integ = {0: "path_tracing", 1: "ao", 2: "whitted"}.get(int(scene.desc.setting.integrator), "?")
            out[name] = {"workload": "%s.json 512x512, 64 spp, %s" % (name, integ), "paths_per_step": int(paths), "ms_per_step": round(best, 2),
                         "value": round(paths / best * 1e-3, 1), "unit": "Mpaths/s", "steps": 3, "film_mean": round(float(film.normalized().mean()), 6)}
            del tr, film
        except Exception as e:
            out[name] = {"error": str(e)}
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None, help="default: bunny at N = 1, grid for N > 1")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="default for N > 1: strong (tile shards of one frame)")
    ap.add_argument("--resolution", type=int, nargs=2, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / l2 legs")
    ap.add_argument("--no-others", action="store_true", help="skip the full-size steps of the other BASELINE configurations")
    ap.add_argument("--schedule", choices=["auto", "wavefront", "megakernel"], default="auto")
    ap.add_argument("--dump-film", default=None, help="rank 0 saves the film accumulators of the last step (reduced, N > 1) as .npy (tests)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from goblin_amd import distributed as gd

    rank, local_rank, world = gd.init()
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run --nproc-per-node %d"
                  % (args.gpus, world, args.gpus), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback on the product path)")
    ndev = torch.cuda.device_count()
    device_index = local_rank % max(1, ndev)   # one GPU per rank on a full node; ranks share GPUs only in the gloo rehearsal
    torch.cuda.set_device(device_index)

    wl_name = args.workload or ("bunny" if world == 1 else "grid")
    wl = Workload(wl_name, device_index, args.resolution, args.spp, args.depth)
    tracer, film, scene = wl.tracer, wl.film, wl.scene
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    base_seed = 20261003
    part = gd.shard_for(rank, world, "samples" if scaling == "weak" else "tiles", base_seed)

    # counters for the roofline (one instrumented launch, outside the timed region;
    # the sampler is counter-based so every timed launch does exactly this work)
    counted = counters_of(wl, part["seed"], part["shard"], args.schedule,
                          scale_from_spp=64 if (scene.spp() > 256 and wl.standard) else None)
    counted_note = counted.pop("_scaled_from_spp", None)
    my_paths = counted["paths"]

    one_gpu_ms = None
    if world > 1 and rank == 0 and scaling == "strong":
        # the WHOLE frame on one GPU, outside the timed region: the N-GPU line then carries its own strong-scaling baseline
        whole = tracer.new_film()
        wl.render_frame(whole, seed=base_seed, schedule=args.schedule)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        whole.zero_()
        wl.render_frame(whole, seed=base_seed, schedule=args.schedule)
        torch.cuda.synchronize()
        one_gpu_ms = (time.perf_counter() - t1) * 1e3
        del whole

    if world > 1:   # set the communicator up outside the timed region even when --warmup is 0
        gd.allreduce_film(torch.zeros(16, device=tracer.device))
        gd.barrier()
    elapsed, per_step = run_steps(
        render=lambda: wl.render_frame(seed=part["seed"], shard=part["shard"], schedule=args.schedule),
        zero_film=film.zero_, allreduce=lambda: gd.allreduce_film(film.accum), barrier=gd.barrier, sync=torch.cuda.synchronize,
        steps=args.steps, warmup=args.warmup, world=world, make_event=lambda: torch.cuda.Event(enable_timing=True))
    my_elapsed = elapsed
    if args.dump_film and rank == 0:
        import numpy as np
        np.save(args.dump_film, film.accum.detach().cpu().numpy())
    call_ms = [a for a, _ in per_step]          # device events around gbl_render (the launch stream)
    reduce_ms = [b for _, b in per_step]
    my_trace_ms = sum(call_ms) / len(call_ms)

    red_dev = tracer.device if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    paths = torch.tensor([float(my_paths)], dtype=torch.float64, device=red_dev)
    per_rank = None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(paths, op=dist.ReduceOp.SUM)
        # per rank: device time of the trace, of the reduce, paths traced and the schedule the library resolved AUTO to
        mine = torch.tensor([my_trace_ms, sum(reduce_ms) / len(reduce_ms), float(my_paths), float(counted.get("schedule", 0)), my_elapsed * 1e3 / args.steps],
                            dtype=torch.float64, device=red_dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = [g.cpu().tolist() for g in gathered]
    elapsed = float(t.item())
    job_paths = float(paths.item())   # paths all ranks traced in one step
    if per_rank is not None:
        # every rank resolves GBL_SCHEDULE_AUTO by itself (its own pilot over its own tiles): they must all have come to the same
        # schedule, or the ranks' times would not be comparable and the film would mix two summation orders
        resolved_all = [int(p[3]) for p in per_rank]
        if len(set(resolved_all)) != 1:
            if rank == 0:
                print("bench.py: the ranks resolved GBL_SCHEDULE_AUTO differently: %s (1 = megakernel, 2 = wavefront); pass --schedule"
                      % resolved_all, file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)

    # what AUTO resolved to: the library reports the schedule a call ran under (gbl_stats.schedule)
    resolved = args.schedule if args.schedule != "auto" else {1: "megakernel", 2: "wavefront"}.get(counted.get("schedule"), "megakernel")
    if rank == 0:
        timings = tracer.timings(args.steps * len(wl.bands))                      # HIP events inside the library, per kernel class and call
        main_ms = [x[0] for x in timings] or call_ms
        avg_kernel_ms = sum(main_ms) / args.steps if timings else sum(main_ms) / len(main_ms)   # per step (a step is len(bands) calls)
        pmc, pmc_note = (pmc_for(wl_name, resolved, wl.integrator, scene.spp()) if (wl.standard and world == 1) else (None, "non-standard run: no counters quoted"))
        rays = counted["extension_rays"] + counted["shadow_rays"]
        value = job_paths * args.steps / elapsed * 1e-6
        roofline = roofline_object(counted, avg_kernel_ms, pmc, pmc_note, wl.kernel_label(resolved), waves_per_simd=5 if resolved == "wavefront" else 3)
        roofline["call_ms_avg"] = round(sum(call_ms) / len(call_ms), 3)
        if counted_note:
            roofline["counters_note"] = "counted by an instrumented launch at %d spp and scaled to the frame's %d" % (counted_note, scene.spp())
        line = {
            "metric": "Mpaths/sec at 512x512x256spp (GoblinPathtracer hot path, bunny.json, max_ray_depth 8)" if wl_name == "bunny" and wl.standard
                      else "Mpaths/sec at %dx%dx%dspp (GoblinPathtracer hot path, %s.json, max_ray_depth %d)" % (wl.res[0], wl.res[1], scene.spp(), wl.scene_name, wl.depth),
            "value": round(value, 3),
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl.describe(),
                "schedule": resolved,
                "paths_per_step": int(job_paths),
                "rays_per_path": round(rays / max(1, my_paths), 3),
                "mrays_per_s": round(value * rays / max(1, my_paths), 2),
                "sampler": "native counter-based, reference stratification law",
                "sharding": ({"weak": "whole window per rank, seed+rank, one film all-reduce",
                              "strong": "8x8 tiles interleaved over ranks (rank r owns tiles t = r mod N), one film all-reduce"}[scaling]
                             if world > 1 else "single GPU"),
            },
            "roofline": roofline,
        }
        if world > 1:
            try:
                import torch.cuda.nccl as tnccl
                rccl_version = ".".join(str(v) for v in tnccl.version()) if dist.get_backend() == "nccl" else None
            except Exception:
                rccl_version = None
            tr = [p[0] for p in per_rank]
            line["collective"] = {"backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (CPU rehearsal)"),
                                  "rccl_version": rccl_version,
                                  "ranks": dist.get_world_size(), "op": "all_reduce(sum) of the film accumulators",
                                  "bytes": int(film.accum.numel() * 4),
                                  "ms_avg": round(sum(reduce_ms) / len(reduce_ms), 3), "trace_ms_avg": round(sum(call_ms) / len(call_ms), 3),
                                  "note": "ms_avg is rank 0's device time between the end of its trace and the end of the all-reduce: it "
                                          "includes waiting for the slowest rank's trace (per_rank below tells the two apart)"}
            line["per_rank"] = {"trace_ms": [round(x, 3) for x in tr], "trace_ms_min": round(min(tr), 3), "trace_ms_max": round(max(tr), 3),
                                "trace_imbalance": round(max(tr) / max(1e-9, min(tr)) - 1.0, 4),
                                "reduce_ms": [round(p[1], 3) for p in per_rank], "paths": [int(p[2]) for p in per_rank],
                                "schedule": [{1: "megakernel", 2: "wavefront"}.get(int(p[3]), "?") for p in per_rank],
                                "step_ms": [round(p[4], 3) for p in per_rank]}
            if one_gpu_ms is not None:
                line["config"]["one_gpu_same_workload"] = {"ms_per_step": round(one_gpu_ms, 3), "mpaths_per_s": round(job_paths / one_gpu_ms * 1e-3, 2),
                                                           "speedup": round(one_gpu_ms / (elapsed / args.steps * 1e3), 3),
                                                           "note": "rank 0 alone, whole frame, outside the timed region"}
        if world == 1 and wl.standard and resolved == "megakernel":
            # the same steps with the reference's exact-t tie rule kept in the lean kernel (gbl_render_params.exact_ties): what the
            # headline would post if it resolved the ~5 ties per 10^7 paths as the replay / stream kernels do
            try:
                for _ in range(1):
                    wl.render_frame(seed=part["seed"], schedule=args.schedule, exact_ties=True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n_tie = 3
                for _ in range(n_tie):
                    film.zero_()
                    wl.render_frame(seed=part["seed"], schedule=args.schedule, exact_ties=True)
                torch.cuda.synchronize()
                tie_ms = (time.perf_counter() - t1) * 1e3 / n_tie
                line["value_tie_exact"] = {"value": round(my_paths / tie_ms * 1e-3, 3), "unit": "Mpaths/s", "ms_per_step": round(tie_ms, 3), "steps": n_tie,
                                           "note": "gbl_render_params.exact_ties = 1: the lean quad kernel following the reference's exact-t tie rule and reachability test (trace.h GBL_TIE_DETECT: a flag in the loops, one check per query, an exact retrace of the rare ray)"}
            except Exception as e:
                print("value_tie_exact leg failed: %s" % e, file=sys.stderr)
        if world == 1 and wl_name == "bunny":
            # the same frame with the reference's own mt19937 sample stream generated on the device (GBL_SAMPLES_STREAM):
            # its Film is the reference binary's; reported beside the headline, never as `value`
            try:
                sfilm = tracer.new_film()
                tracer.render(film=sfilm, sampler="stream", schedule="megakernel")
                torch.cuda.synchronize()
                ts = time.perf_counter()
                sfilm.zero_()
                tracer.render(film=sfilm, sampler="stream", schedule="megakernel")
                torch.cuda.synchronize()
                sms = (time.perf_counter() - ts) * 1e3
                line["reference_stream_sampler"] = {"value": round(my_paths / sms * 1e-3, 2), "unit": "Mpaths/s",
                                                    "ms_per_step": round(sms, 2),
                                                    "note": "bit-faithful sampler: per-tile mt19937 + Sampler::requestSamples on the device"}
                del sfilm
            except Exception as e:
                print("reference_stream_sampler leg failed: %s" % e, file=sys.stderr)
        if world == 1 and wl_name == "bunny" and wl.standard and not args.no_others:
            # every other BASELINE configuration, one full-size step each, outside the headline's timed region
            line["other_configs"] = {}
            host_cores = max(1, len(os.sched_getaffinity(0)))
            for other in ("cornell", "grid", "ao"):
                try:
                    line["other_configs"][other] = one_config(other, device_index, base_seed, torch, steps=1 if other == "ao" else 3,
                                                              cpu_cores=host_cores if (host_cores >= 12 and not args.no_cpu) else 0)
                except Exception as e:
                    line["other_configs"][other] = {"error": str(e)}
                    print("other_configs[%s] failed: %s" % (other, e), file=sys.stderr)
        if world == 1 and wl_name == "bunny" and wl.standard and not args.no_others:
            line["feature_scenes"] = feature_scenes(device_index, base_seed, torch)
        if world == 1 and wl_name == "bunny" and not args.no_cpu:
            # every host core this process may run on (the reference defaults to hardware_concurrency, GoblinThreadPool.cpp:5-10)
            cores = max(1, len(os.sched_getaffinity(0)))
            # the CPU leg renders the FULL workload when the box has the cores to do it in ~15-30 s (68 M paths at
            # ~0.3 Mpaths/s per thread), so that its Film can be compared at BASELINE's own size; fewer samples otherwise
            cpu_spp = wl.spp if cores >= 12 else (64 if cores >= 4 else 16)
            line["cpu_baseline"] = cpu_baseline(wl.overrides, cpu_spp, cores)
            if line["cpu_baseline"]:
                line["cpu_baseline"]["host"] = {"os_cpu_count": os.cpu_count(), "affinity": cores}
            ref_film = line["cpu_baseline"].pop("_film", None) if line["cpu_baseline"] else None
            if ref_film is not None:
                try:
                    line["l2_vs_reference"] = l2_vs_reference(tracer, ref_film, cpu_spp)
                except Exception as e:
                    print("l2_vs_reference failed: %s" % e, file=sys.stderr)
            try:
                line["l2_vs_cpu"] = l2_vs_cpu(tracer, wl.overrides, 16, min(cores, 32), base_seed)
            except Exception as e:
                print("l2_vs_cpu failed: %s" % e, file=sys.stderr)
            if line["cpu_baseline"]:
                line["config"]["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
