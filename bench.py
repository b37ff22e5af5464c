#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the HIP path tracer on BASELINE.json configs[1]
(bunny.json, 512x512 film, 256 spp, max_ray_depth 8) on N MI355X GPUs.

    python bench.py                      # 1 GPU, 5 steps, 1 warmup
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over the workload with the scene already
resident in HBM: zero the film, trace every camera path of the sample window
(516x516 pixels x 256 spp = 68,161,536 paths), splat, and -- for N > 1 -- sum the
film over the ranks with one RCCL all-reduce.  Rank 0 prints ONE JSON line.

Multi-GPU work split (goblin_amd/distributed.py):
  --scaling weak   (default) every rank traces the whole window with its own sample
                   set (seed + rank); the reduced film holds N x 256 spp.  Per-GPU
                   work is fixed as N grows.
  --scaling strong rank r traces every N-th 8x8 sample tile of the one 256-spp frame.
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


def algorithmic_bytes(st, main_kernel_only=True):
    """Algorithmic bytes of one launch: per ray 32 B per BVH box node tested +
    48 B per triangle tested + 48 B (32 B ray in, 16 B hit out); per path 4 B per sample
    dimension consumed + 16 B of radiance written.  The film splat (16 B per touched pixel)
    belongs to the separate splat kernel and is excluded from the dominant kernel's figure."""
    rays = st["extension_rays"] + st["shadow_rays"]
    b = 32 * st["nodes"] + 48 * st["tris"] + 48 * rays + 4 * st["dims"] + 16 * st["paths"]
    if not main_kernel_only:
        b += 16 * st["splats"]
    return b


def requested_bytes(st):
    """Bytes the dominant kernel's loads/stores actually request for the same work: the 4-wide node is
    64 B for four quantised boxes (16 B per box tested instead of 32)."""
    rays = st["extension_rays"] + st["shadow_rays"]
    return 16 * st["nodes"] + 48 * st["tris"] + 48 * rays + 4 * st["dims"] + 16 * st["paths"]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--resolution", type=int, nargs=2, default=[512, 512])
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / l2 legs")
    ap.add_argument("--schedule", choices=["auto", "wavefront", "megakernel", "wavepool"], default="auto")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from goblin_amd import distributed as gd
    from goblin_amd import scene as gs
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

    overrides = gs.config_overrides(resolution=tuple(args.resolution), spp=args.spp, depth=args.depth)
    scene = gs.load_scene("bunny", overrides)
    tracer = HipPathTracer(scene, device_index)
    film = tracer.new_film()
    base_seed = 20261003
    part = gd.shard_for(rank, world, "samples" if args.scaling == "weak" else "tiles", base_seed)

    # counters for the roofline (one instrumented launch, outside the timed region;
    # the sampler is counter-based so every timed launch does exactly this work)
    counted = tracer.render(film=film, seed=part["seed"], shard=part["shard"], stats=True, schedule=args.schedule)["stats"]
    my_paths = counted["paths"]

    def step(ev=None):
        film.zero_()
        if ev is not None:
            ev[0].record()
        tracer.render(film=film, seed=part["seed"], shard=part["shard"], schedule=args.schedule)
        if ev is not None:
            ev[1].record()
        if world > 1:
            gd.allreduce_film(film.accum)

    if world > 1:   # set the communicator up outside the timed region even when --warmup is 0
        gd.allreduce_film(torch.zeros(16, device=tracer.device))
        gd.barrier()
    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    gd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    torch.cuda.synchronize()
    gd.barrier()
    elapsed = time.perf_counter() - t0

    red_dev = tracer.device if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    paths = torch.tensor([float(my_paths)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(paths, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    job_paths = float(paths.item())   # paths all ranks traced in one step

    resolved = args.schedule if args.schedule != "auto" else ("wavefront" if args.depth >= 12 else "megakernel")
    if rank == 0:
        call_ms = sorted(e0.elapsed_time(e1) for e0, e1 in events)          # torch events around gbl_render
        timings = tracer.timings(args.steps)                                   # HIP events inside, per kernel class
        main_ms = [t[0] for t in timings] or call_ms
        avg_kernel_ms = sum(main_ms) / len(main_ms)
        alg_bytes = algorithmic_bytes(counted)
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        req_bytes = requested_bytes(counted)
        traffic, pmc = None, {}
        tpath = os.path.join(REPO, "profiles", "traffic_r01.json")
        if os.path.exists(tpath) and args.spp == 256 and args.resolution == [512, 512] and args.depth == 8 and world == 1 \
                and resolved == "megakernel":
            with open(tpath) as f:
                pmc = json.load(f)
            traffic = pmc.get("hbm_bytes_per_launch")
        rays = counted["extension_rays"] + counted["shadow_rays"]
        value = job_paths * args.steps / elapsed * 1e-6
        line = {
            "metric": "Mpaths/sec at 512x512x256spp (GoblinPathtracer hot path, bunny.json, max_ray_depth 8)",
            "value": round(value, 3),
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "bunny.json %dx%d film (%dx%d sampled px), %d spp, max_ray_depth %d, gaussian r=2, glass "
                            "stand-in bunny (69120 tris) + spot light" % (
                                args.resolution[0], args.resolution[1], tracer.window[1] - tracer.window[0],
                                tracer.window[3] - tracer.window[2], scene.spp(), args.depth),
                "schedule": resolved,
                "paths_per_step": int(job_paths),
                "rays_per_path": round(rays / max(1, my_paths), 3),
                "mrays_per_s": round(value * rays / max(1, my_paths), 2),
                "sampler": "native counter-based, reference stratification law",
                "sharding": {"weak": "whole window per rank, seed+rank, RCCL film all-reduce",
                             "strong": "8x8 tiles interleaved over ranks, RCCL film all-reduce"}[args.scaling]
                if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "kernel": "path_trace_kernel<false,false,false,false>" if resolved != "wavefront" else "wf_trace/wf_shade (all wavefront kernels of a step)",
                "kernel_ms_avg": round(avg_kernel_ms, 3),
                "call_ms_avg": round(sum(call_ms) / len(call_ms), 3),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "requested_bytes_per_launch": int(req_bytes),
                "achieved_requested": round(req_bytes / (avg_kernel_ms * 1e-3) / 1e9, 2),
                "note": "algorithmic = SURVEY 8d convention (32 B per box node tested); requested = what the 4-wide "
                        "quantised node layout really loads (16 B per box); traffic = HBM bytes from rocprofv3 PMC "
                        "(profiles/), the scene is cache resident",
                "counters": {k: int(v) for k, v in counted.items() if k != "kernel_ms"},
                # from the committed rocprofv3 PMC passes of this command (profiles/): what really bounds the kernel
                "valu_busy_frac": pmc.get("valu_busy_frac"),
                "l2_hit_rate": pmc.get("l2_hit_rate"),
            },
        }
        if world == 1:
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
        if world == 1 and not args.no_cpu:
            cores = max(1, min(16, len(os.sched_getaffinity(0))))
            # the CPU leg renders the FULL workload when the box has the cores to do it in ~15 s (68 M paths at
            # ~5 Mpaths/s on 16 threads), so that its Film can be compared at BASELINE's own size; fewer samples otherwise
            cpu_spp = args.spp if cores >= 12 else (64 if cores >= 4 else 16)
            line["cpu_baseline"] = cpu_baseline(overrides, cpu_spp, cores)
            ref_film = line["cpu_baseline"].pop("_film", None) if line["cpu_baseline"] else None
            if ref_film is not None:
                try:
                    line["l2_vs_reference"] = l2_vs_reference(tracer, ref_film, cpu_spp)
                except Exception as e:
                    print("l2_vs_reference failed: %s" % e, file=sys.stderr)
            try:
                line["l2_vs_cpu"] = l2_vs_cpu(tracer, overrides, 16, cores, base_seed)
            except Exception as e:
                print("l2_vs_cpu failed: %s" % e, file=sys.stderr)
            if line["cpu_baseline"]:
                line["config"]["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
