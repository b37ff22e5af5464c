#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the HIP path tracer on N MI355X GPUs.

    python bench.py                      # 1 GPU: BASELINE.json configs[1], 5 steps, 1 warmup
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


def algorithmic_bytes(st):
    """Algorithmic bytes of one launch of the tracing kernel (with the build's real node size): per ray
    64 B per BVH node VISITED (one 4-wide node = four 16-B quantised child boxes; st["nodes"] counts child boxes tested,
    4 per visit) + 48 B per triangle tested + 48 B (32 B ray in, 16 B hit out); per path 4 B per sample dimension
    consumed + 16 B of radiance written.  The film splat (16 B per touched pixel) belongs to the splat kernel."""
    rays = st["extension_rays"] + st["shadow_rays"]
    return 16 * st["nodes"] + 48 * st["tris"] + 48 * rays + 4 * st["dims"] + 16 * st["paths"]


def survey_literal_bytes(st):
    """SURVEY 8d read literally on node VISITS: 32 B (the reference's CompactBVHNode) per node visited."""
    rays = st["extension_rays"] + st["shadow_rays"]
    return 32 * (st["nodes"] // 4) + 48 * st["tris"] + 48 * rays + 4 * st["dims"] + 16 * st["paths"]


def pmc_for(kernel_tag, workload, schedule):
    """Counters of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/pmc_<workload>_<schedule>.json,
    written by tools/summarize_profile.py) -- only if they were collected from THIS source tree (build stamp)."""
    from goblin_amd import build
    path = os.path.join(REPO, "profiles", "pmc_%s_%s.json" % (workload, schedule))
    if not os.path.exists(path):
        return None, "no profiles/pmc_%s_%s.json" % (workload, schedule)
    with open(path) as f:
        d = json.load(f)
    stamp = build.source_stamp()
    if d.get("source_stamp") != stamp:
        return None, "%s was collected from source stamp %s, this tree is %s" % (os.path.basename(path), d.get("source_stamp"), stamp)
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
            for key in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum", "FETCH_SIZE", "WRITE_SIZE"):
                if key in c:
                    agg[key] = agg.get(key, 0.0) + c[key] * per_render
            agg["kernel_avg_ms"] = agg.get("kernel_avg_ms", 0.0) + c.get("kernel_avg_ms", 0.0) * per_render
        if agg.get("SQ_INSTS_VALU"):
            return dict(agg, kernel="one step: " + ", ".join(kernels), file=os.path.relpath(path, REPO), source_stamp=stamp), None
        return None, "no wavefront kernels in %s" % os.path.basename(path)
    # the lean instantiation: the named kernel with every template argument false (the list grows with the experiments) but
    # the one that selects the quad-per-ray steps (kernels/quadtrace.h), which is what a lean scene runs by default
    base = kernel_tag.split("<")[0]
    quad_arg = {"path_trace_kernel": 7, "ao_kernel": 4}.get(base)
    best = None
    for name, c in d.get("counters_per_launch", {}).items():
        if (" " + base + "<") not in (" " + name) or _template_arg(name, 0) != "false":
            continue
        if not all(_template_arg(name, i) in ("false", None) for i in range(12) if i != quad_arg):
            continue
        if best is None or _template_arg(name, quad_arg) == "true":
            best = (name, c)
    if best is not None:
        return dict(best[1], kernel=best[0], file=os.path.relpath(path, REPO), source_stamp=stamp), None
    return None, "no kernel matching %r in %s" % (kernel_tag, os.path.basename(path))


def _template_arg(kernel_name, index):
    """The index-th template argument of a demangled kernel name ('void wf_trace<false, true, ...>(...)' -> 'true' for 1)."""
    a = kernel_name.find("<")
    b = kernel_name.find(">", a)
    args = [x.strip() for x in kernel_name[a + 1:b].split(",")] if a >= 0 and b > a else []
    return args[index] if index < len(args) else None


def cpu_baseline(workload_overrides, spp_sample, cores):
    """Time the CPU path on a bounded sample of the same workload (same scene, film,
    filter and depth; fewer samples per pixel).  Prefers the REAL reference built into
    oracle/_ref (kind "reference"); falls back to the oracle port."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from goblin_amd import scene as gs
    ov = json.loads(json.dumps(workload_overrides))
    ov.setdefault("render_setting", {})["sample_per_pixel"] = spp_sample
    harness = os.path.join(REPO, "oracle", "_ref", "ref_harness")
    sample = "bunny.json 512x512, %d spp (of 256), max_ray_depth 8, %d threads" % (spp_sample, cores)
    if spp_sample >= 256:
        sample = "bunny.json 512x512, 256 spp (the whole workload), max_ray_depth 8, %d threads" % cores
    out = None
    if os.path.exists(harness):
        try:
            src = gs.scene_path("bunny")
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
    try:
        import oracle_binding as ob
        ov_port = json.loads(json.dumps(ov))
        ov_port["render_setting"]["sample_per_pixel"] = min(spp_sample, 16)   # the port is a side note: keep it short
        scene = gs.load_scene("bunny", ov_port)
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
    import ctypes as C
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
    # name: (scene, resolution, spp, depth, what it is)
    "bunny": ("bunny", (512, 512), 256, 8, "BASELINE configs[1]: bunny.json, glass stand-in bunny (69120 tris) on a plane, spot light"),
    "grid": ("grid", (1024, 1024), 256, 8, "BASELINE configs[3]: grid.json, 15 instances of the bunny BLAS (1.04 M instanced triangles)"),
    # the other two configurations, for profiles (tools/profile_gpu.sh): run them with --spp to bound the time, the kernels'
    # per-launch behaviour does not depend on it
    "cornell": ("cornell", (1024, 1024), 1024, 16, "BASELINE configs[2]: cornell.json, Cornell box + glass bunny, area light, divergent BSDF mix"),
    "ao": ("bunny", (2048, 2048), 4096, 8, "BASELINE configs[4]: bunny.json under the AO integrator (1 closest hit + 25 any-hit rays per camera sample)",
           {"method": "ao", "ao_samples": 25}),
}
VALU_CYCLES_PER_WAVE_INSTRUCTION = 4   # MI355X_MICROARCH.md: one wave64 f32 VALU instruction holds its SIMD's issue for 4 cycles
SIMDS = 256 * 4


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
    ap.add_argument("--schedule", choices=["auto", "wavefront", "megakernel", "wavepool"], default="auto")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from goblin_amd import distributed as gd
    from goblin_amd import scene as gs
    from goblin_amd import _abi
    from goblin_amd.renderer import HipPathTracer

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
    scene_name, res, spp, depth, wl_text = WORKLOADS[wl_name][:5]
    wl_extra = WORKLOADS[wl_name][5] if len(WORKLOADS[wl_name]) > 5 else {}
    res = tuple(args.resolution) if args.resolution else res
    spp = args.spp or spp
    depth = args.depth or depth
    standard = (res, spp, depth) == WORKLOADS[wl_name][1:4]
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    overrides = gs.config_overrides(resolution=res, spp=spp, depth=depth, **wl_extra)
    scene = gs.load_scene(scene_name, overrides)
    tracer = HipPathTracer(scene, device_index)
    film = tracer.new_film()
    base_seed = 20261003
    part = gd.shard_for(rank, world, "samples" if scaling == "weak" else "tiles", base_seed)

    # a call takes fewer than 2^32 camera samples (configs[4] at its 4096 spp has 1.7e10): such a frame is rendered in
    # bands of 8-pixel tile rows, one call each, all of them inside the timed step
    x0, x1, y0, y1 = tracer.window
    rows_per_band = max(8, ((1 << 31) // max(1, (x1 - x0) * scene.spp())) // 8 * 8)
    bands = [(x0, x1, y, min(y1, y + rows_per_band)) for y in range(y0, y1, rows_per_band)]
    if len(bands) == 1:
        bands = [None]

    def render_frame(target, **kw):
        out = None
        for w in bands:
            r = tracer.render(film=target, window=w, **kw)
            if out is None:
                out = r
            elif r["stats"]:
                for k, v in r["stats"].items():
                    if k not in ("schedule", "reserved"):   # (what the call ran under: not a counter)
                        out["stats"][k] += v
        return out

    # counters for the roofline (one instrumented launch, outside the timed region;
    # the sampler is counter-based so every timed launch does exactly this work)
    counted = render_frame(film, seed=part["seed"], shard=part["shard"], stats=True, schedule=args.schedule)["stats"]
    my_paths = counted["paths"]

    one_gpu_ms = None
    if world > 1 and rank == 0 and scaling == "strong":
        # the WHOLE frame on one GPU, outside the timed region: the N-GPU line then carries its own strong-scaling baseline
        whole = tracer.new_film()
        render_frame(whole, seed=base_seed, schedule=args.schedule)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        whole.zero_()
        render_frame(whole, seed=base_seed, schedule=args.schedule)
        torch.cuda.synchronize()
        one_gpu_ms = (time.perf_counter() - t1) * 1e3
        del whole

    if world > 1:   # set the communicator up outside the timed region even when --warmup is 0
        gd.allreduce_film(torch.zeros(16, device=tracer.device))
        gd.barrier()
    elapsed, per_step = run_steps(
        render=lambda: render_frame(film, seed=part["seed"], shard=part["shard"], schedule=args.schedule),
        zero_film=film.zero_, allreduce=lambda: gd.allreduce_film(film.accum), barrier=gd.barrier, sync=torch.cuda.synchronize,
        steps=args.steps, warmup=args.warmup, world=world, make_event=lambda: torch.cuda.Event(enable_timing=True))

    red_dev = tracer.device if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    paths = torch.tensor([float(my_paths)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(paths, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    job_paths = float(paths.item())   # paths all ranks traced in one step

    # what AUTO resolved to: the library reports the schedule a call ran under (gbl_stats.schedule)
    resolved = args.schedule if args.schedule != "auto" else {1: "megakernel", 2: "wavefront", 3: "wavepool"}.get(counted.get("schedule"), "megakernel")
    if rank == 0:
        call_ms = [a for a, _ in per_step]                                    # device events around gbl_render (the launch stream)
        reduce_ms = [b for _, b in per_step]
        timings = tracer.timings(args.steps * len(bands))                      # HIP events inside the library, per kernel class and call
        main_ms = [x[0] for x in timings] or call_ms
        avg_kernel_ms = sum(main_ms) / args.steps if timings else sum(main_ms) / len(main_ms)   # per step (a step is len(bands) calls)
        alg_bytes = algorithmic_bytes(counted)
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        kernel_tag = {"megakernel": "path_trace_kernel<false, ...>", "wavepool": "wp_kernel<false, false, false>",
                      "wavefront": "wf_trace<false, false, false, false, false>"}[resolved]
        if wl_extra.get("method") == "ao":
            kernel_tag = "ao_kernel<false, false, false, false>"
        pmc, pmc_note = (pmc_for(kernel_tag, wl_name, resolved) if (standard and world == 1) else (None, "non-standard run: no counters quoted"))
        rays = counted["extension_rays"] + counted["shadow_rays"]
        value = job_paths * args.steps / elapsed * 1e-6
        window_px = (tracer.window[1] - tracer.window[0], tracer.window[3] - tracer.window[2])
        hbm = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBPS, 4),
               "algorithmic_bytes_per_launch": int(alg_bytes),
               "survey_8d_literal": {"bytes_per_launch": int(survey_literal_bytes(counted)),
                                     "frac": round(survey_literal_bytes(counted) / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
               "note": "algorithmic bytes (64 B per 4-wide node visited, 48 B per triangle, 48 B per ray, 4 B per sample "
                       "dimension, 16 B per path) over the kernel's measured time: a cache-served figure -- the scene is "
                       "L2-resident, `traffic` below is what reaches HBM"}
        roofline = dict(hbm)
        if pmc and pmc.get("SQ_INSTS_VALU") and pmc.get("GRBM_GUI_ACTIVE"):
            # the kernel is VALU-issue bound: its instruction stream over what the 1024 SIMDs can issue in the measured time.
            # Instruction count from the stamped PMC pass of this very build; time and clock measured here / there.
            clock_hz = pmc["GRBM_GUI_ACTIVE"] / 8.0 / (pmc["kernel_avg_ms"] * 1e-3)
            peak = SIMDS * clock_hz / VALU_CYCLES_PER_WAVE_INSTRUCTION * 1e-9
            ach = pmc["SQ_INSTS_VALU"] / (avg_kernel_ms * 1e-3) * 1e-9
            traffic = int((2 * pmc.get("FETCH_SIZE", 0.0) + pmc.get("WRITE_SIZE", 0.0)) * 1024) if "FETCH_SIZE" in pmc else None
            roofline = {"bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s",
                        "frac": round(ach / peak, 4), "traffic": traffic,
                        "valu_wave_instructions_per_launch": int(pmc["SQ_INSTS_VALU"]),
                        "valu_busy_pmc": round(pmc["SQ_ACTIVE_INST_VALU"] * 4 / (pmc["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS), 4) if "SQ_ACTIVE_INST_VALU" in pmc else None,
                        "clock_ghz_pmc": round(clock_hz * 1e-9, 3),
                        "l2_hit_rate": round(pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in pmc else None,
                        "pmc": {"file": pmc["file"], "source_stamp": pmc["source_stamp"], "kernel": pmc["kernel"],
                                "kernel_avg_ms_rocprof": pmc["kernel_avg_ms"]},
                        "hbm": dict(hbm, traffic=traffic, traffic_gbps=round(traffic / (avg_kernel_ms * 1e-3) / 1e9, 1) if traffic else None),
                        "note": "peak = 1024 SIMDs x clock / 4 cycles per wave64 VALU instruction; frac = the share of the "
                                "SIMDs' issue slots the kernel's own instruction stream fills -- high means issue bound, and "
                                "the headroom is in the lanes (see DESIGN.md 4.1: lane utilisation), not in this fraction"}
        else:
            roofline["traffic"] = None
            roofline["pmc_note"] = pmc_note
        roofline.update({"kernel": kernel_tag if resolved != "wavefront" else "wf_trace / wf_shade (all wavefront kernels of a step)",
                         "kernel_ms_avg": round(avg_kernel_ms, 3), "call_ms_avg": round(sum(call_ms) / len(call_ms), 3),
                         "counters": {k: int(v) for k, v in counted.items() if k != "kernel_ms"}})
        line = {
            "metric": "Mpaths/sec at 512x512x256spp (GoblinPathtracer hot path, bunny.json, max_ray_depth 8)" if wl_name == "bunny" and standard
                      else "Mpaths/sec at %dx%dx%dspp (GoblinPathtracer hot path, %s.json, max_ray_depth %d)" % (res[0], res[1], scene.spp(), scene_name, depth),
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
                "workload": "%s -- %dx%d film (%dx%d sampled px), %d spp, max_ray_depth %d, gaussian r=2" % (
                    wl_text, res[0], res[1], window_px[0], window_px[1], scene.spp(), depth),
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
            line["collective"] = {"backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (CPU rehearsal)"),
                                  "ranks": dist.get_world_size(), "op": "all_reduce(sum) of the film accumulators",
                                  "bytes": int(film.accum.numel() * 4),
                                  "ms_avg": round(sum(reduce_ms) / len(reduce_ms), 3), "trace_ms_avg": round(sum(call_ms) / len(call_ms), 3)}
            if one_gpu_ms is not None:
                line["config"]["one_gpu_same_workload"] = {"ms_per_step": round(one_gpu_ms, 3), "mpaths_per_s": round(job_paths / one_gpu_ms * 1e-3, 2),
                                                           "speedup": round(one_gpu_ms / (elapsed / args.steps * 1e3), 3),
                                                           "note": "rank 0 alone, whole frame, outside the timed region"}
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
        if world == 1 and wl_name == "bunny" and not args.no_cpu:
            # every host core this process may run on (the reference defaults to hardware_concurrency, GoblinThreadPool.cpp:5-10)
            cores = max(1, len(os.sched_getaffinity(0)))
            # the CPU leg renders the FULL workload when the box has the cores to do it in ~15-30 s (68 M paths at
            # ~0.3 Mpaths/s per thread), so that its Film can be compared at BASELINE's own size; fewer samples otherwise
            cpu_spp = spp if cores >= 12 else (64 if cores >= 4 else 16)
            line["cpu_baseline"] = cpu_baseline(overrides, cpu_spp, cores)
            if line["cpu_baseline"]:
                line["cpu_baseline"]["host"] = {"os_cpu_count": os.cpu_count(), "affinity": cores}
            ref_film = line["cpu_baseline"].pop("_film", None) if line["cpu_baseline"] else None
            if ref_film is not None:
                try:
                    line["l2_vs_reference"] = l2_vs_reference(tracer, ref_film, cpu_spp)
                except Exception as e:
                    print("l2_vs_reference failed: %s" % e, file=sys.stderr)
            try:
                line["l2_vs_cpu"] = l2_vs_cpu(tracer, overrides, 16, min(cores, 32), base_seed)
            except Exception as e:
                print("l2_vs_cpu failed: %s" % e, file=sys.stderr)
            if line["cpu_baseline"]:
                line["config"]["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
