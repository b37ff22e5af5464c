"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def tile_order_index(window, spp, tile=8):
    """Index array mapping the oracle's tile-then-pixel record order to the
    pixel-major order the C ABI's replay mode expects.

    out[pixel_major_record] = tile_order_record
    """
    x0, x1, y0, y1 = window
    w, h = x1 - x0, y1 - y0
    order = np.zeros((h, w), np.int64)
    n = 0
    for ty in range(y0, y1, tile):
        for tx in range(x0, x1, tile):
            tw, th = min(tile, x1 - tx), min(tile, y1 - ty)
            blk = n + np.arange(tw * th).reshape(th, tw)
            order[ty - y0:ty - y0 + th, tx - x0:tx - x0 + tw] = blk
            n += tw * th
    pix = order.reshape(-1)
    return (pix[:, None] * spp + np.arange(spp)[None, :]).reshape(-1)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def li_mismatch_fraction(a, b, rtol=1e-3, atol=1e-5):
    """Fraction of samples whose radiance differs by more than rtol (a 'flipped' sample:
    a discrete decision -- reflect/refract pick, hit/miss at an edge -- went the other way)."""
    a = np.asarray(a, np.float64)[:, :3]
    b = np.asarray(b, np.float64)[:, :3]
    bad = np.any(np.abs(a - b) > atol + rtol * np.abs(b), axis=1)
    return float(bad.mean())


# ---------------------------------------------------------------------------
# Random scenes over the supported feature set (tests/test_gpu_fuzz.py, tests/test_oracle_fuzz.py)
# ---------------------------------------------------------------------------
def quat(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return [float(x) for x in q]


def random_scene(seed):
    rng = np.random.default_rng(seed)
    U = lambda a, b: float(rng.uniform(a, b))
    tex = [
        {"format": "color", "name": "c0", "type": "constant", "color": [U(.1, .9), U(.1, .9), U(.1, .9)]},
        {"format": "color", "name": "c1", "type": "constant", "color": [U(.1, .9), U(.1, .9), U(.1, .9)]},
        {"format": "color", "name": "c2", "type": "constant", "color": [U(.5, 1), U(.5, 1), U(.5, 1)]},
        {"format": "float", "name": "f0", "type": "constant", "float": U(2, 60)},
        {"format": "float", "name": "f1", "type": "constant", "float": U(.2, .9)},
        {"format": "float", "name": "one", "type": "constant", "float": 1.0},
        {"format": "float", "name": "zero", "type": "constant", "float": 0.0},
        {"format": "color", "name": "chk", "type": "checkerboard", "texture1": "c0", "texture2": "c1",
         "scale": [U(2, 9), U(2, 9)], "filter": bool(rng.integers(2))},
        {"format": "color", "name": "sph", "type": "checkerboard", "texture1": "c1", "texture2": "c2", "mapping": "spherical",
         "position": [U(-1, 1), U(0, 1), U(-1, 1)], "orientation": quat(rng), "scale": [U(.1, .4)] * 3, "filter": bool(rng.integers(2))},
        {"format": "color", "name": "dim", "type": "scale", "texture": "chk", "scale": "f1"},
        {"format": "float", "name": "cut", "type": "checkerboard", "texture1": "one", "texture2": "zero", "scale": [U(2, 6), U(2, 6)]},
    ]
    kd = lambda: str(rng.choice(["c0", "c1", "chk", "sph", "dim"]))
    mats = [
        {"name": "floor", "type": "lambert", "Kd": kd()},
        {"name": "m0", "type": "lambert", "Kd": kd()},
        {"name": "m1", "type": "blinn", "Kg": kd(), "exponent": "f0", "index": U(1.2, 2.0)},
        {"name": "m2", "type": "transparent", "Kr": "c2", "Kt": str(rng.choice(["c2", "chk"])), "index": U(1.2, 1.8)},
        {"name": "m3", "type": "mirror", "Kr": "c2"},
    ]
    use_mask = bool(rng.integers(2))
    if use_mask:
        mats.append({"name": "m4", "type": "mask", "material": str(rng.choice(["m0", "m1", "m2", "m3"])),
                     "alpha": str(rng.choice(["f1", "cut"])), "transparent_color": "c2"})
    rng2 = np.random.default_rng(seed + 7919)   # its own stream: the scenes of earlier seeds keep their other draws
    use_sss = bool(rng2.integers(2))
    if use_sss:
        V = lambda a, b: float(rng2.uniform(a, b))
        if rng2.integers(2):
            mats.append({"name": "m5", "type": "subsurface", "Kd": [V(.3, .9), V(.3, .9), V(.3, .9)],
                         "mean_free_path": [V(.05, .3), V(.05, .3), V(.05, .3)], "index": V(1.2, 1.6)})
        else:
            tex.append({"format": "color", "name": "sa", "type": "constant", "color": [V(.5, 3), V(.5, 3), V(.5, 3)]})
            tex.append({"format": "color", "name": "ss", "type": "constant", "color": [V(40, 120), V(40, 120), V(40, 120)]})
            mats.append({"name": "m5", "type": "subsurface", "absorb": "sa", "scatter_prime": str(rng2.choice(["ss", "chk"])),
                         "Kr": "c2", "index": V(1.2, 1.6), "g": V(0.0, 0.6)})
    geoms = [
        {"name": "quad", "type": "mesh", "file": "models/plane.obj"},
        {"name": "cube", "type": "mesh", "file": "models/cube.obj"},
        {"name": "bunny", "type": "mesh", "file": "models/bunny_vn.obj"},
        {"name": "ball", "type": "sphere", "radius": U(.3, 1.0)},
        {"name": "plate", "type": "disk", "radius": U(.4, 1.0)},
    ]
    prims = [{"type": "model", "name": "mfloor", "geometry": "quad", "material": "floor"},
             {"type": "instance", "name": "floor", "model": "mfloor", "position": [0, 0, 0], "orientation": [1, 0, 0, 0], "scale": [4, 4, 4]}]
    names = [m["name"] for m in mats if m["name"] != "floor"]
    for i in range(int(rng.integers(3, 7))):
        g = str(rng.choice(["cube", "bunny", "ball", "plate"]))
        prims.append({"type": "model", "name": "mod%d" % i, "geometry": g, "material": str(rng.choice(names))})
        sc = U(.25, .6) * (6.0 if g == "bunny" else 1.0)
        scale = [sc * U(.7, 1.3), sc * U(.7, 1.3), sc * U(.7, 1.3)] if rng.integers(2) else [sc] * 3
        prims.append({"type": "instance", "name": "inst%d" % i, "model": "mod%d" % i,
                      "position": [U(-1.3, 1.3), U(.3, 1.2) - (0.9 if g == "bunny" else 0.0), U(-1.0, 1.4)], "orientation": quat(rng), "scale": scale})
    lights = []
    for i in range(int(rng.integers(1, 4))):
        t = str(rng.choice(["point", "spot", "area", "directional"]))
        if t == "point":
            lights.append({"name": "l%d" % i, "type": "point", "intensity": [U(3, 9)] * 3, "position": [U(-2, 2), U(1.5, 3), U(-2, 1)]})
        elif t == "spot":
            lights.append({"name": "l%d" % i, "type": "spot", "intensity": [U(8, 20)] * 3, "position": [U(-2, 2), U(2, 3), U(-2, 1)],
                           "target": [U(-.5, .5), 0.2, U(-.5, .5)], "theta_max": U(25, 50), "falloff_start": U(5, 20)})
        elif t == "directional":
            d = np.array([U(-.5, .5), -1.0, U(-.5, .5)])
            d /= np.linalg.norm(d)
            lights.append({"name": "l%d" % i, "type": "directional", "radiance": [U(.3, 1)] * 3, "direction": [float(x) for x in d]})
        else:
            s = U(.15, .5)
            lights.append({"name": "l%d" % i, "type": "area", "geometry": str(rng.choice(["quad", "ball", "plate"])),
                           "radiance": [U(5, 20), U(5, 20), U(5, 20)], "position": [U(-1.5, 1.5), U(2.0, 3.0), U(-1, 1)],
                           "orientation": [0, 1, 0, 0] if rng.integers(2) else [0.70710678, 0.70710678, 0, 0], "scale": [s, s, s]})
    cam = {"position": [U(-.5, .5), U(1.0, 1.6), U(-4.6, -3.8)], "orientation": [0.9961947, 0.0871557, 0, 0], "fov": U(35, 55),
           "film": {"resolution": [40, 32]}, "filter": {"type": str(rng.choice(["gaussian", "box", "triangle", "mitchell"])), "width": [U(.6, 2.0)] * 2}}
    c = int(rng.integers(3))
    if c == 1:
        cam.update({"type": "orthographic", "film_width": U(4, 7)})
    elif c == 2:
        cam.update({"lens_radius": U(.03, .15), "focal_distance": U(3.5, 5)})
    doc = {"render_setting": {"render_method": "path_tracing", "sample_per_pixel": 4, "max_ray_depth": int(rng.integers(3, 9)),
                              "bssrdf_sample_num": int(rng2.integers(1, 6))},
           "camera": cam, "geometries": geoms, "textures": tex, "materials": mats, "primitives": prims, "lights": lights}
    return doc, use_mask or use_sss


def random_scene_r2(seed, whitted=False):
    """random_scene(seed) plus the features of round 2, drawn from a stream of their own: image textures (every filter and
    address mode, colour and float, uv and spherical mapping), bump and normal maps, an image based light, a homogeneous
    or heterogeneous participating medium.  Returns (doc, has_heterogeneous_medium)."""
    doc, _ = (random_whitted_scene if whitted else random_scene)(seed)
    rng = np.random.default_rng(seed + 15485863)
    U = lambda a, b: float(rng.uniform(a, b))
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    tex, mats = doc["textures"], doc["materials"]
    # (no "bilinear" here: MIPMap::lookupBilinear rounds the level and MIPMap::lookup clamps it to [0, levels] -- one past the
    #  pyramid -- so a footprint wider than ~0.7 of the texture reads mPyramid[levels] out of bounds in the reference
    #  (GoblinTexture.cpp:104-110, 276-277; SIGBUS on some of these scenes); the hand-made imagetex fixture covers it)
    filters, modes = ["nearest", "trilinear", "EWA"], ["repeat", "clamp", "border"]
    tex.append({"format": "color", "name": "img", "type": "image", "file": "images/tiles.exr", "filter": pick(filters), "address": pick(modes),
                "scale": [U(1, 6), U(1, 6)], "offset": [U(-.5, .5), U(-.5, .5)], "gamma": pick([1.0, 2.2])})
    tex.append({"format": "color", "name": "img_sph", "type": "image", "file": "images/tiles.exr", "filter": pick(filters), "address": "repeat",
                "mapping": "spherical", "position": [U(-1, 1), U(0, 1), U(-1, 1)], "orientation": quat(rng), "scale": [U(.1, .4)] * 3})
    tex.append({"format": "float", "name": "imgf", "type": "image", "file": "images/tiles.exr", "filter": pick(filters), "address": pick(modes),
                "channel": pick(["R", "G", "B"]), "scale": [U(2, 8), U(2, 8)], "gamma": 1.0})
    tex.append({"format": "float", "name": "amp", "type": "constant", "float": U(.02, .1)})
    tex.append({"format": "float", "name": "imgf_small", "type": "scale", "texture": "imgf", "scale": "amp"})
    tex.append({"format": "color", "name": "flatn", "type": "constant", "color": [0.5, 0.5, 1.0]})
    plain = [m for m in mats if m["type"] in ("lambert", "blinn", "transparent", "mirror")]
    for m in plain:
        if rng.integers(3) == 0:
            key = {"lambert": "Kd", "blinn": "Kg", "transparent": "Kt", "mirror": "Kr"}[m["type"]]
            m[key] = pick(["img", "img_sph"])
        r = int(rng.integers(4))
        if r == 0:
            m["bumpmap"] = pick(["imgf_small", "cut", "f1"])
        elif r == 1:
            m["normalmap"] = pick(["img", "flatn"])
        elif r == 2:
            m["bumpmap"] = pick(["imgf_small", "amp"])
            m["normalmap"] = pick(["img", "flatn"])
    if rng.integers(3) == 0:
        doc["lights"].append({"name": "sky", "type": "ibl", "file": "images/env.exr", "filter": [U(.5, 1), U(.5, 1), U(.5, 1)],
                              "orientation": quat(rng), "sample_num": int(rng.integers(1, 5))})
    hetero = False
    v = int(rng.integers(4))
    if v == 1:
        doc["volume"] = {"type": "homogeneous", "attenuation": [U(.1, .5), U(.1, .5), U(.1, .5)], "albedo": [U(.5, .9)] * 3, "emission": [0.0, 0.0, 0.0],
                         "g": U(0, .6), "sample_num": int(rng.integers(1, 4)), "box_min": [-1, -1, -1], "box_max": [1, 1, 1],
                         "position": [U(-.5, .5), U(.8, 1.3), U(-.3, .6)], "orientation": quat(rng), "scale": [U(1, 2), U(.8, 1.4), U(1, 2)]}
    elif v == 2:
        hetero = True
        doc["volume"] = {"type": "heterogeneous", "density_grid": pick(["volumes/puff.vol", "volumes/tint.vol"]), "albedo": [U(.5, .9), U(.5, .9), U(.5, .9)],
                         "g": U(0, .6), "step_size": U(.15, .4), "sample_num": int(rng.integers(1, 4)),
                         "position": [U(-.5, .5), U(.8, 1.3), U(-.3, .6)], "orientation": quat(rng), "scale": [U(1, 2), U(.8, 1.4), U(1, 2)]}
    return doc, hetero


def random_whitted_scene(seed):
    """random_scene(seed) under the Whitted renderer: shallow recursion, per-light sample counts on the area lights."""
    doc, special = random_scene(seed)
    rng = np.random.default_rng(seed + 104729)
    doc["render_setting"]["render_method"] = "whitted"
    doc["render_setting"]["max_ray_depth"] = int(rng.integers(1, 4))
    for l in doc["lights"]:
        if l["type"] == "area":
            l["sample_num"] = int(rng.integers(1, 6))
    return doc, special

