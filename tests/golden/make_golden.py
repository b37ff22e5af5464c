#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the authoring container: it needs oracle/_ref/ref_harness, i.e. the
reference compiled from /root/reference/src by oracle/Makefile (`make -C oracle
ref`).  The fixtures are data only -- inputs (scene name + overrides, Sample
records) and the reference's outputs (Film accumulators, per-sample Li, known
answer tables).  No reference source is stored.

    python tests/golden/make_golden.py

Each case is rendered with thread_num = 1 so the Film accumulation order is the
deterministic one (GoblinThreadPool.cpp:63-71).
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from goblin_amd import scene as gs  # noqa: E402

HARNESS = os.path.join(REPO, "oracle", "_ref", "ref_harness")


def ov(resolution, spp, depth=None, method=None, ao=None, filt=None, crop=None, geometries=None, camera=None):
    o = gs.config_overrides(resolution=resolution, spp=spp, depth=depth, method=method, ao_samples=ao, filter=filt)
    if crop:
        o["camera"]["film"]["crop"] = crop
    if camera:
        o["camera"].update(camera)
    if geometries:
        o["geometries"] = geometries
    return o


VN_GEOMS = [{"name": "bunny", "type": "mesh", "file": "models/bunny_vn.obj"},
            {"name": "plane", "type": "mesh", "file": "models/plane.obj"}]

SPOT = [{"name": "spot", "type": "spot", "intensity": [30.0, 32.0, 40.0], "position": [-2.4, 2.6, -1.2], "target": [0.2, 0.6, 0.4],
         "theta_max": 28.0, "falloff_start": 20.0}]

def whitted_sss_materials():
    """whitted.json's materials with the red lambert and the blinn plastic replaced by subsurface.json's skin and jade."""
    with open(gs.scene_path("whitted")) as f:
        mats = json.load(f)["materials"]
    with open(gs.scene_path("subsurface")) as f:
        sss = {m["name"]: m for m in json.load(f)["materials"] if m["type"] == "subsurface"}
    out = []
    for m in mats:
        if m["name"] == "red":
            m = dict(sss["skin"], name="red")
        elif m["name"] == "plastic":
            m = dict(sss["jade"], name="plastic")
        out.append(m)
    return out


def whitted_sss_textures():
    with open(gs.scene_path("whitted")) as f:
        tex = json.load(f)["textures"]
    with open(gs.scene_path("subsurface")) as f:
        have = {t["name"] for t in tex}
        tex += [t for t in json.load(f)["textures"] if t["name"] not in have]
    return tex


WHITTED_SSS_MATERIALS = whitted_sss_materials()
WHITTED_SSS_TEXTURES = whitted_sss_textures()

# name -> (scene, overrides, number of (Sample -> Li) records to keep, with_kat)
CASES = {
    "bunny_pt": ("bunny", ov((64, 64), 16, 4), 2048, True),
    # BASELINE configs[0] at its full size (1 081 600 paths), Film only
    "bunny_config1": ("bunny", ov((256, 256), 16, 4), 0, False),
    "bunny_pt_d8": ("bunny", ov((32, 32), 4, 8), 1024, False),
    "cornell_pt": ("cornell", ov((48, 48), 16, 6), 2048, True),
    # BASELINE configs[2]'s shape: max_ray_depth 16 (148-float records; AUTO picks the wavefront schedule from depth 12), 64 spp
    "cornell_pt_d16": ("cornell", ov((32, 32), 64, 16), 2048, False),
    "grid_pt": ("grid", ov((48, 48), 4, 5), 1024, True),
    "bunny_ao": ("bunny", ov((48, 48), 4, method="ao", ao=9), 1024, False),
    "bunny_vn_box": ("bunny", ov((48, 48), 9, 5, filt={"type": "box", "width": [0.5, 0.5]}, geometries=VN_GEOMS), 512, False),
    "cornell_mitchell": ("cornell", ov((40, 30), 5, 3, filt={"type": "mitchell", "width": [2.0, 2.0], "b": 0.33, "c": 0.33}), 0, False),
    "cornell_triangle_crop": ("cornell", ov((40, 30), 4, 3, filt={"type": "triangle", "width": [1.5, 1.5]},
                                            crop=[0.25, 0.75, 0.1, 0.9]), 0, False),
    # sphere / disk geometry (also as area lights), directional light, thin-lens and orthographic cameras
    "shapes_pt": ("shapes", ov((64, 64), 9, 5), 2048, False),
    "shapes_thinlens": ("shapes", ov((48, 48), 9, 4, camera={"lens_radius": 0.12, "focal_distance": 4.6}), 1024, False),
    "shapes_ortho": ("shapes", ov((48, 48), 4, 4, camera={"type": "orthographic", "film_width": 5.0}), 1024, False),
    "shapes_ao": ("shapes", ov((40, 40), 4, method="ao", ao=4), 512, False),
    # checkerboard (uv + spherical mapping, filtered through the camera ray's differentials), scale and float textures
    "textured_pt": ("textured", ov((64, 64), 9, 5), 2048, False),
    "textured_ortho": ("textured", ov((48, 48), 4, 4, camera={"type": "orthographic", "film_width": 6.0}), 1024, False),
    # mask materials: BSDFnullptr punch-through, isOpaque-filtered shadow / MIS queries, evalAttenuation walks
    "masked_pt": ("masked", ov((64, 64), 9, 6), 2048, False),
    # subsurface materials -- Lsubsurface at the camera hit (single scattering + dipole diffusion), the
    # Fresnel mirror lobe of SubsurfaceMaterial, and its BSDFAll type under the isOpaque / notOpaque filters
    # the Whitted renderer: multiSampleLd over every light (per-light sample counts 4, 1, 2 -> 4, 1, 4 slots), specular tree
    "whitted": ("whitted", ov((64, 64), 9, 5), 2048, False),
    "whitted_d2": ("whitted", ov((48, 48), 4, 2), 1024, False),
    # a homogeneous participating medium around the camera ray: RenderTask's tr * L + Lv (Film only: the probes log Li)
    "volume_pt": ("volume", ov((64, 64), 9, 5), 0, False),
    "volume_ao": ("volume", ov((48, 48), 4, method="ao", ao=4), 0, False),
    "volume_spot": ("volume", dict(ov((64, 64), 9, 5), lights=SPOT), 0, False),
    # ... under the other two renderers with the delta light alone (the medium's samples are then exact):
    # AORenderer::Li draws nothing from the tile's generator, WhittedRenderer::Li 6 floats per (light, slot) and per level
    "volume_ao_spot": ("volume", dict(ov((48, 48), 4, method="ao", ao=4), lights=SPOT), 0, False),
    "volume_whitted_spot": ("volume", dict(ov((48, 48), 4, 3, method="whitted"), lights=SPOT), 0, False),
    # a heterogeneous medium (HeterogeneousVolumeRegion: .vol density grid, trilinear lookups, ray-marched transmittance and
    # Lv with a data-dependent number of random numbers per sample); one-channel grid under the path tracer, three-channel
    # grid in a rotated region under Whitted and AO
    "hetero_pt": ("hetero", ov((48, 48), 4, 4), 0, False),
    "hetero_spot": ("hetero", dict(ov((48, 48), 4, 4), lights=SPOT), 0, False),
    "hetero_tint_whitted": ("hetero_tint", ov((40, 40), 4, 3, method="whitted"), 0, False),
    "hetero_tint_ao": ("hetero_tint", ov((40, 40), 4, method="ao", ao=4), 0, False),
    # bump and normal maps (BumpShaders: Material::perturb at every closest hit): image / scaled / checkerboard / constant
    # float bump maps, image and constant normal maps, on lambert, blinn and glass, and behind a mask
    "bumpy_pt": ("bumpy", ov((64, 64), 9, 5), 2048, False),
    "bumpy_whitted": ("bumpy", ov((48, 48), 4, 3, method="whitted"), 1024, False),
    "bumpy_ao": ("bumpy", ov((40, 40), 4, method="ao", ao=4), 512, False),
    "subsurface_pt": ("subsurface", ov((64, 64), 9, 5), 2048, False),
    # subsurface materials under the Whitted renderer: Lsubsurface at every level of the recursion (here also behind the
    # mirror and the glass), SubsurfaceMaterial's BSDFAll lobe never matching the non-specular / specular requests
    "subsurface_whitted": ("subsurface", ov((48, 48), 4, 3, method="whitted"), 1024, False),
    "whitted_sss": ("whitted", dict(ov((48, 48), 4, 3), materials=WHITTED_SSS_MATERIALS, textures=WHITTED_SSS_TEXTURES), 1024, False),
    # mask materials under the Whitted renderer: opaque to every query; the BSDFnullptr lobe only shows up in estimateLd's
    # BSDF sample (straight through, towards an area light); masked mirror / glass answer the specular requests times alpha
    "masked_whitted": ("masked", ov((48, 48), 4, 3, method="whitted"), 1024, False),
    # SURVEY 8f rank 3, the last of it: image textures (MIPMap: nearest / bilinear / trilinear / EWA, repeat / clamp / border,
    # gamma, channel picks, uv and spherical mapping, colour and float formats) and the image based light (Le of escaping
    # rays, sampleL through its CDF2D, pdf, the two IBL branches of PathTracer::Li; estimateLd's under Whitted)
    "imagetex_pt": ("imagetex", ov((64, 64), 9, 5), 2048, False),
    "ibl_pt": ("ibl", ov((64, 64), 9, 5), 2048, False),
    "ibl_whitted": ("ibl", ov((48, 48), 4, 3, method="whitted"), 1024, False),
    # a floor of coplanar triangles with bit-identical hit distances (goblin_amd/scenes/make_meshes.py ties_mesh): nearly every
    # hit is an exact-t tie, within one two-triangle leaf of the reference's tree or across leaves (GoblinTriangle.cpp:74-80,
    # GoblinBVH.cpp:106-118, 156-187) -- what the oracle's and the device's tie rule restate
    "ties_pt": ("ties", ov((64, 64), 9, 4), 4096, False),
    "subsurface_n9": ("subsurface", dict(ov((40, 40), 4, 4), render_setting=dict(ov((40, 40), 4, 4)["render_setting"], bssrdf_sample_num=7)), 1024, False),
}


def absolute_scene(scene, overrides, path):
    src = gs.scene_path(scene)
    with open(src) as f:
        doc = json.load(f)
    gs._merge(doc, overrides)
    doc.setdefault("render_setting", {})["thread_num"] = 1
    for section in ("geometries", "textures", "lights"):   # meshes, image textures, environment maps
        for g in doc.get(section, []):
            if "file" in g:
                g["file"] = os.path.join(os.path.dirname(src), g["file"])
    if "density_grid" in doc.get("volume", {}):   # the heterogeneous medium's .vol grid
        doc["volume"]["density_grid"] = os.path.join(os.path.dirname(src), doc["volume"]["density_grid"])
    with open(path, "w") as f:
        json.dump(doc, f)


def parse_kat(path):
    out = {}
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        parts = lines[i].split()
        i += 1
        if not parts:
            continue
        key = parts[0]
        if key == "sample_range":
            out[key] = np.array([int(x) for x in parts[1:]], np.int32)
            continue
        n = int(parts[1])
        rows = [[float(x) for x in lines[i + k].split()] for k in range(n)]
        i += n
        out[key] = np.array(rows, np.float32)
    return out


def main():
    if not os.path.exists(HARNESS):
        sys.exit("oracle/_ref/ref_harness is missing: run `make -C oracle ref` (needs /root/reference)")
    manifest = {}
    with tempfile.TemporaryDirectory() as tmp:
        only = set(sys.argv[1:])
        if only and os.path.exists(os.path.join(HERE, "manifest.json")):
            with open(os.path.join(HERE, "manifest.json")) as f:
                manifest = json.load(f)
        for name, (scene, overrides, nrec, with_kat) in CASES.items():
            if only and name not in only:
                continue
            jp = os.path.join(tmp, name + ".json")
            absolute_scene(scene, overrides, jp)
            prefix = os.path.join(tmp, name)
            # count the Li calls first so the kept records are spread over the whole image
            meta = json.loads(subprocess.check_output([HARNESS, "li", jp, prefix, "1", "0"]).decode())
            calls = meta["li_calls"]
            stride = max(1, calls // nrec) if nrec else 1
            meta = json.loads(subprocess.check_output([HARNESS, "li", jp, prefix, str(stride), str(nrec)]).decode())
            film = np.fromfile(prefix + ".film.f32", np.float32).reshape(meta["yres"], meta["xres"], 4)
            arrays = {"film": film}
            if nrec:
                dims = meta["dims"]
                arrays["samples"] = np.fromfile(prefix + ".samples.f32", np.float32).reshape(-1, dims)
                arrays["li"] = np.fromfile(prefix + ".li.f32", np.float32).reshape(-1, 4)
                assert arrays["samples"].shape[0] == arrays["li"].shape[0] == meta["records"]
            if with_kat:
                subprocess.check_output([HARNESS, "kat", jp, prefix])
                for k, v in parse_kat(prefix + ".kat.txt").items():
                    arrays["kat_" + k] = v
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
            manifest[name] = {"scene": scene, "overrides": overrides, "window": meta["window"], "spp": meta["spp"],
                              "paths": meta["paths"], "records": meta["records"], "record_stride": stride,
                              "dims": meta["dims"]}
            print(name, meta["paths"], "paths,", meta["records"], "records, film", film.shape)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
