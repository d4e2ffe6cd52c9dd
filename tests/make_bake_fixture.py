#!/usr/bin/env python3
"""Writes tests/golden/bake/model.obj + model.mtl (a small OBJ that exercises the bake path: quads split along either
diagonal, negative indices, 'o' and 'g' shapes, duplicate and nearly-duplicate vertices around the weld tolerance,
mirrored texture coordinates, materials with and without each map) and, where /root/reference and
oracle/_ref/ref_fixtures exist, tests/golden/ref_objmodel.json = what the reference's rapidobj + make_indexed_mesh
pipeline makes of it (oracle/ref_fixtures.cpp, command objmodel)."""
import os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "golden", "bake")


def main():
    os.makedirs(D, exist_ok=True)
    rng = np.random.default_rng(0x0B4E)
    L = ["# bake fixture", "mtllib model.mtl"]
    # --- a 5x5-vertex height field of quads, smooth normals, one v/vt/vn triple per grid vertex
    L.append("o floor")
    L.append("usemtl mat_all_maps")
    n = 5
    h = rng.uniform(-0.4, 0.4, (n, n))
    for j in range(n):
        for i in range(n):
            L.append("v %.6f %.6f %.6f" % (i * 0.5, h[j, i], -j * 0.5))
    for j in range(n):
        for i in range(n):
            L.append("vt %.6f %.6f" % (i / (n - 1) * 2.0, j / (n - 1) * 2.0))
    for j in range(n):
        for i in range(n):
            dx = h[j, min(i + 1, n - 1)] - h[j, max(i - 1, 0)]
            dz = h[min(j + 1, n - 1), i] - h[max(j - 1, 0), i]
            v = np.array([-dx, 1.0, dz]); v /= np.linalg.norm(v)
            L.append("vn %.6f %.6f %.6f" % tuple(v))
    for j in range(n - 1):
        for i in range(n - 1):
            a, b, c, d = j * n + i + 1, j * n + i + 2, (j + 1) * n + i + 2, (j + 1) * n + i + 1
            L.append("f %d/%d/%d %d/%d/%d %d/%d/%d %d/%d/%d" % (a, a, a, b, b, b, c, c, c, d, d, d))
    # --- a cube of quads with face normals, addressed with negative indices, mirrored u on two faces
    L.append("g cube")
    L.append("usemtl mat_alpha")
    cube = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    for x, y, z in cube:
        L.append("v %.6f %.6f %.6f" % (3 + 0.5 * x, 0.5 * y, -1 + 0.5 * z))
    for u, v in ((0, 0), (1, 0), (1, 1), (0, 1), (-1, 0), (-1, 1)):
        L.append("vt %.6f %.6f" % (u, v))
    for nx, ny, nz in ((0, 0, -1), (0, 0, 1), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0)):
        L.append("vn %.6f %.6f %.6f" % (nx, ny, nz))
    faces = [((0, 3, 2, 1), 0, (0, 3, 2, 1)), ((4, 5, 6, 7), 1, (0, 1, 2, 3)), ((0, 4, 7, 3), 2, (0, 4, 5, 3)),
             ((1, 2, 6, 5), 3, (0, 3, 2, 1)), ((0, 1, 5, 4), 4, (0, 1, 2, 3)), ((3, 7, 6, 2), 5, (0, 4, 5, 3))]
    for vs, nn, ts in faces:
        L.append("f " + " ".join("%d/%d/%d" % (v - 8, t - 6, nn - 6) for v, t in zip(vs, ts)))
    # --- loose triangles: exact duplicates, near duplicates inside and outside the 1e-5 tolerance
    L.append("o shards")
    L.append("usemtl mat_plain")
    base = len(cube) + n * n
    tb, nb = n * n + 6, n * n + 6
    pts = [(5.0, 0.0, 0.0), (6.0, 0.0, 0.0), (5.5, 1.0, 0.0),
           (5.0, 0.0, 0.0), (5.5, 1.0, 0.0), (4.5, 1.0, 0.25),            # shares two vertices exactly
           (6.0 + 4e-6, 0.0, 0.0), (7.0, 0.0, 0.0), (6.5, 1.0, 0.0),       # first vertex within tolerance of (6,0,0)
           (7.0 + 3e-5, 0.0, 0.0), (8.0, 0.0, 0.0), (7.5, 1.0, 0.5),       # first vertex outside tolerance of (7,0,0)
           (5.5, 1.0, 0.0), (6.5, 1.0, 0.0), (6.0, 2.0, -0.3)]             # same positions, different uv: no merge
    uvs = [(0, 0), (1, 0), (0.5, 1), (0, 0), (0.5, 1), (-0.5, 1), (1, 0), (2, 0), (1.5, 1), (2, 0), (3, 0), (2.5, 1),
           (0.25, 0.75), (0.75, 0.75), (0.5, 0.1)]
    for p in pts:
        L.append("v %.7f %.7f %.7f" % p)
    for t in uvs:
        L.append("vt %.6f %.6f" % t)
    L.append("vn 0.000000 0.000000 1.000000")
    for k in range(0, len(pts), 3):
        L.append("f " + " ".join("%d/%d/%d" % (base + k + c + 1, tb + k + c + 1, nb + 1) for c in range(3)))
    open(os.path.join(D, "model.obj"), "w").write("\n".join(L) + "\n")
    open(os.path.join(D, "model.mtl"), "w").write("""# bake fixture materials
newmtl mat_all_maps
	Kd 1.000000 1.000000 1.000000
	Pm 1.000000
	Pr 0.552786
	map_Kd tex/a_BaseColor.png
	map_Pr tex/r-a_Specular.png
	map_Pm tex/m-a_Specular.png
	norm tex/a_Normal.png
	map_Ke tex/a_Emissive.png

newmtl mat_alpha
	Kd 1.000000 1.000000 1.000000
	map_Kd tex/leaf_BaseColor.png
	map_d tex/leaf_BaseColor.png
	map_Pr tex/r-a_Specular.png
	norm -bm 1.000000 tex/leaf_Normal.png

newmtl mat_plain
	Kd 0.500000 0.250000 0.125000
	Pm 0.000000
	Pr 1.000000
""")
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_fixtures")
    if os.path.exists(ref):
        with open(os.path.join(ROOT, "tests", "golden", "ref_objmodel.json"), "w") as f:
            subprocess.check_call([ref, "objmodel", "tests/golden/bake/model.obj", "model"], stdout=f, cwd=ROOT)
    else:
        print("oracle/_ref/ref_fixtures missing: ref_objmodel.json not regenerated", file=sys.stderr)


if __name__ == "__main__":
    main()
