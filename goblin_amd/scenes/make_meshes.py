#!/usr/bin/env python3
"""Deterministic generator for the stand-in meshes used by every scene.

The reference's examples/models/bunny.obj is absent from the snapshot
(/root/reference/.MISSING_LARGE_BLOBS), so every "bunny" configuration uses
this analytic closed blob instead: a lat-long sphere with a smooth radial
displacement (body lobes + two ear-like bumps), affinely fitted to the
Stanford bunny's object-space bounding box so that bunny.json's transforms
(scale 5, floor at y=-0.835065) still make sense.

No RNG is used; coordinates are printed with '%.6f' so the text (and therefore
the float32 values every consumer parses) is bit-reproducible.  The generated
files are committed; tests/test_scenes.py checks this script reproduces them.
"""
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
NU = 192      # segments around
NV = 182      # rings pole..pole inclusive -> 2*NU*(NV-2) = 69120 triangles
# Stanford bunny object-space bbox (what bunny.json's transform was tuned for)
BB_MIN = (-0.094690, 0.032987, -0.061874)
BB_MAX = (0.061009, 0.187321, 0.058800)


def radius(theta, phi):
    """theta in [0,pi] from +y pole, phi in [0,2pi)."""
    st = math.sin(theta)
    r = 1.0
    r += 0.10 * st * st * math.sin(3.0 * phi + 0.5) * math.sin(2.0 * theta)
    r += 0.05 * st * math.sin(7.0 * phi) * math.sin(5.0 * theta)
    r += 0.025 * st * st * math.cos(11.0 * phi + 1.0) * math.cos(9.0 * theta)
    # two "ears" near the top
    for (t0, p0) in ((0.45, 0.9), (0.45, 2.3)):
        dp = math.atan2(math.sin(phi - p0), math.cos(phi - p0))
        d2 = (theta - t0) ** 2 + (st * dp) ** 2
        r += 0.55 * math.exp(-d2 / 0.02)
    return r


def build(NU=NU, NV=NV):
    verts = []
    # north pole, rings, south pole
    verts.append((0.0, radius(0.0, 0.0), 0.0))
    for j in range(1, NV - 1):
        theta = math.pi * j / (NV - 1)
        for i in range(NU):
            phi = 2.0 * math.pi * i / NU
            r = radius(theta, phi)
            verts.append((r * math.sin(theta) * math.cos(phi),
                          r * math.cos(theta),
                          r * math.sin(theta) * math.sin(phi)))
    verts.append((0.0, -radius(math.pi, 0.0), 0.0))
    # fit to the bunny bbox
    lo = [min(v[k] for v in verts) for k in range(3)]
    hi = [max(v[k] for v in verts) for k in range(3)]
    out = []
    for v in verts:
        out.append(tuple(BB_MIN[k] + (v[k] - lo[k]) / (hi[k] - lo[k]) *
                         (BB_MAX[k] - BB_MIN[k]) for k in range(3)))
    faces = []
    south = len(out) - 1

    def ring(j, i):
        return 1 + (j - 1) * NU + (i % NU)

    for i in range(NU):
        faces.append((0, ring(1, i + 1), ring(1, i)))
    for j in range(1, NV - 2):
        for i in range(NU):
            a, b = ring(j, i), ring(j, i + 1)
            c, d = ring(j + 1, i), ring(j + 1, i + 1)
            faces.append((a, b, d))
            faces.append((a, d, c))
    for i in range(NU):
        faces.append((south, ring(NV - 2, i), ring(NV - 2, i + 1)))
    return out, faces


def vertex_normals(verts, faces):
    acc = [[0.0, 0.0, 0.0] for _ in verts]
    for (a, b, c) in faces:
        pa, pb, pc = verts[a], verts[b], verts[c]
        e1 = [pb[k] - pa[k] for k in range(3)]
        e2 = [pc[k] - pa[k] for k in range(3)]
        n = (e1[1] * e2[2] - e1[2] * e2[1],
             e1[2] * e2[0] - e1[0] * e2[2],
             e1[0] * e2[1] - e1[1] * e2[0])
        for idx in (a, b, c):
            for k in range(3):
                acc[idx][k] += n[k]
    out = []
    for n in acc:
        l = math.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
        out.append((n[0] / l, n[1] / l, n[2] / l))
    return out


def write_obj(path, verts, faces, normals=None):
    with open(path, "w") as f:
        f.write("# stand-in for the missing Stanford bunny: analytic blob, %d tris\n" % len(faces))
        for v in verts:
            f.write("v %.6f %.6f %.6f\n" % v)
        if normals:
            for n in normals:
                f.write("vn %.6f %.6f %.6f\n" % n)
            for (a, b, c) in faces:
                f.write("f %d//%d %d//%d %d//%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1))
        else:
            for (a, b, c) in faces:
                f.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))


QUAD = """# 2x2 quad in the xz plane, +y normal, with uv
v -1.0 0.0 1.0
v 1.0 0.0 1.0
v -1.0 0.0 -1.0
v 1.0 0.0 -1.0
vt 0.0 0.0
vt 1.0 0.0
vt 0.0 1.0
vt 1.0 1.0
vn 0.0 1.0 0.0
f 1/1/1 2/2/1 3/3/1
f 3/3/1 2/2/1 4/4/1
"""

# the same quad without vt/vn: exercises the face-normal / default-uv branch
QUAD_BARE = """# 2x2 quad in the xz plane, no vt / vn
v -1.0 0.0 1.0
v 1.0 0.0 1.0
v -1.0 0.0 -1.0
v 1.0 0.0 -1.0
f 1 2 3 4
"""

# a unit cube out of quads (tri+quad OBJ faces), no normals: 12 triangles
CUBE = """# unit cube [-0.5,0.5]^3 from quads
v -0.5 -0.5 -0.5
v 0.5 -0.5 -0.5
v 0.5 0.5 -0.5
v -0.5 0.5 -0.5
v -0.5 -0.5 0.5
v 0.5 -0.5 0.5
v 0.5 0.5 0.5
v -0.5 0.5 0.5
f 1 4 3 2
f 5 6 7 8
f 1 2 6 5
f 4 8 7 3
f 1 5 8 4
f 2 3 7 6
"""


def ties_mesh(n=16, h=0.25):
    """A floor on which (almost) every hit is an EXACT t tie (tests/test_gpu_ties.py, trace.h tie_goes_to).

    Per cell of an n x n grid in the plane y = 0, from the cell's corner p0: triangle A = (p0, p0 + e1, p0 + e2), its twin A' (the
    same three positions again -- same centre, so the reference's builder leaves the two in ONE leaf: GoblinBVH.cpp:106-118) and
    B = (p0, p0 + 2 e1, p0 + 2 e2).  Moller-Trumbore on B is the arithmetic on A with every edge doubled: powers of two go through
    every rounding, so wherever a ray passes A it gets bit-identical distances from all three -- and B, centred elsewhere, sits in
    another leaf, so its tie with A / A' goes through the visiting order and the strict box test (GoblinBVH.cpp:156-187).  Each
    triangle has its own tilted vertex normals, so the radiance tells which one the traversal kept.  All coordinates are
    multiples of 1/8: exact in float32."""
    verts, normals, faces = [], [], []
    tilt = ((0.3, 1.0, 0.0), (-0.3, 1.0, 0.2), (0.0, 1.0, -0.35))
    for k, t in enumerate(tilt):
        l = math.sqrt(sum(c * c for c in t))
        normals.append(tuple(c / l for c in t))
    for j in range(n):
        for i in range(n):
            x0, z0 = -0.5 * n * h + i * h, -0.5 * n * h + j * h
            for kind, s in ((0, 1.0), (1, 1.0), (2, 2.0)):
                base = len(verts)
                verts += [(x0, 0.0, z0), (x0, 0.0, z0 + s * h), (x0 + s * h, 0.0, z0)]   # e1 = +z, e2 = +x: the face looks up
                faces.append((base, base + 1, base + 2, kind))
    return verts, normals, faces


def write_ties(path, n=16, h=0.25):
    verts, normals, faces = ties_mesh(n, h)
    with open(path, "w") as f:
        f.write("# %d coplanar triangles, three per cell with bit-identical hit distances (make_meshes.py ties_mesh)\n" % len(faces))
        for v in verts:
            f.write("v %.6f %.6f %.6f\n" % v)
        for nn in normals:
            f.write("vn %.6f %.6f %.6f\n" % nn)
        for (a, b, c, k) in faces:
            f.write("f %d//%d %d//%d %d//%d\n" % (a + 1, k + 1, b + 1, k + 1, c + 1, k + 1))


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    write_ties(os.path.join(outdir, "ties.obj"))
    verts, faces = build()
    write_obj(os.path.join(outdir, "bunny.obj"), verts, faces)
    # low-res variant WITH vertex normals: exercises the interpolated-normal
    # branch of the triangle test (GoblinTriangle.cpp:85-90)
    verts, faces = build(64, 50)
    rv = [tuple(float("%.6f" % c) for c in v) for v in verts]
    write_obj(os.path.join(outdir, "bunny_vn.obj"), verts, faces, vertex_normals(rv, faces))
    with open(os.path.join(outdir, "plane.obj"), "w") as f:
        f.write(QUAD)
    with open(os.path.join(outdir, "quad_bare.obj"), "w") as f:
        f.write(QUAD_BARE)
    with open(os.path.join(outdir, "cube.obj"), "w") as f:
        f.write(CUBE)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "models"))
