"""Procedural scenes in the reference's BakedModel shape.

The Sun Temple geometry blob is not available offline, so measurement and
parity run on "SunTemple-proxy v1": a hall that encloses the reference's start
camera, its forward light (main/main.cpp:1314) and all 21 brazier lights
(main/main.cpp:1321-1347), with 48 materials (3 alpha-masked, mirroring
assets-src/main/suntemple.mtl) and four textures per material.  A small
"micro" scene with the same features is used by the parity tests.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .abi import NO_TEXTURE, PackedScene
from .baked_model import pack_tbn

PROXY_SEED = 0x53554E54  # "SUNT"

# brazier lights, main/main.cpp:1321-1347
BRAZIERS = np.array([
    [8.0377, -1.1, -15.8845], [5.6632, -1.1, -12.25], [2.1255, -1.1, -10.1], [-2.0867, -1.1, -10.25],
    [-5.7367, -1.1, -12.2], [-7.8244, -1.1, -15.85], [3.0632, -1.1, -25.85], [-3.0632, -1.1, -25.85],
    [-7.4867, -1.1, -36.05], [-2.4367, -3.1, -46.55], [2.3632, -3.1, -46.55], [-0.0514, -3.9, -49.6],
    [-7.2367, -3.1, -61.25], [7.1088, -3.1, -61.25], [-2.0867, -3.1, -66.55], [2.0867, -3.1, -66.55],
    [-2.0867, -3.1, -68.9], [2.0867, -3.1, -68.9], [0.0, -4.1, -76.0], [-1.6867, -3.1, -88.85],
    [1.6867, -3.1, -88.85]], np.float32)


# --------------------------------------------------------------------------
# mesh helpers
# --------------------------------------------------------------------------
def _finish(pos, nrm, uv, tan, idx, material):
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
    tan4 = np.concatenate([np.asarray(tan, np.float32).reshape(-1, 3), np.ones((len(pos), 1), np.float32)], axis=1)
    return dict(materialId=int(material), positions=pos, normals=nrm,
                texcoords=np.ascontiguousarray(uv, np.float32).reshape(-1, 2), tangents=tan4,
                tangentsComp=pack_tbn(tan4, nrm), indices=np.ascontiguousarray(idx, np.uint32).reshape(-1))


def _grid_indices(nu, nv, flip=False):
    """(nu+1) x (nv+1) vertices, vertex (i,j) at j*(nu+1)+i; two triangles per cell."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    a = (j * (nu + 1) + i).reshape(-1)
    b = a + 1
    c = a + (nu + 1) + 1
    d = a + (nu + 1)
    if flip:
        tri = np.stack([a, c, b, a, d, c], axis=1)
    else:
        tri = np.stack([a, b, c, a, c, d], axis=1)
    return tri.reshape(-1).astype(np.uint32)


def grid_surface(origin, du, dv, nu, nv, uv0, uv_du, uv_dv, material):
    """Planar grid O + s*du + t*dv, s,t in [0,1]; front face towards du x dv."""
    origin, du, dv = [np.asarray(x, np.float64) for x in (origin, du, dv)]
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t, indexing="xy")
    pos = origin[None, None, :] + S[..., None] * du + T[..., None] * dv
    n = np.cross(du, dv)
    n /= np.linalg.norm(n)
    uv = np.asarray(uv0, np.float64)[None, None, :] + S[..., None] * np.asarray(uv_du) + T[..., None] * np.asarray(uv_dv)
    tdir = du / np.linalg.norm(du)
    V = (nu + 1) * (nv + 1)
    return _finish(pos, np.tile(n, (V, 1)), uv, np.tile(tdir, (V, 1)), _grid_indices(nu, nv), material)


def param_surface(fn, nu, nv, material, wrap_u=False):
    """fn(U, V) -> (pos, normal, tangent, uv) arrays over a (nv+1, nu+1) grid of
    parameters in [0,1]; winding chosen so the front face looks along +normal."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="xy")
    pos, nrm, tan, uv = fn(U, V)
    # orientation test on the first non-degenerate cell
    idx = _grid_indices(nu, nv)
    P = pos.reshape(-1, 3)
    tri = idx.reshape(-1, 3)
    fn_ = np.cross(P[tri[:, 1]] - P[tri[:, 0]], P[tri[:, 2]] - P[tri[:, 0]])
    nn = nrm.reshape(-1, 3)[tri[:, 0]]
    if np.sum(np.einsum("ij,ij->i", fn_, nn)) < 0:
        idx = _grid_indices(nu, nv, flip=True)
    return _finish(pos, nrm, uv, tan, idx, material)


def fluted_column(cx, cz, y0, y1, radius, nseg, nring, material, flutes=16, depth=0.06):
    def fn(U, V):
        ang = U * 2.0 * np.pi
        r = radius * (1.0 - depth * (0.5 + 0.5 * np.cos(flutes * ang))) * (1.0 - 0.08 * V)
        pos = np.stack([cx + r * np.cos(ang), y0 + (y1 - y0) * V, cz + r * np.sin(ang)], axis=-1)
        nrm = np.stack([np.cos(ang), np.zeros_like(ang), np.sin(ang)], axis=-1)
        tan = np.stack([-np.sin(ang), np.zeros_like(ang), np.cos(ang)], axis=-1)
        uv = np.stack([U * 4.0, V * (y1 - y0) / 2.0], axis=-1)
        return pos, nrm, tan, uv
    return param_surface(fn, nseg, nring, material)


def sphere_bowl(center, radius, nseg, nring, material):
    c = np.asarray(center, np.float64)

    def fn(U, V):
        ang = U * 2.0 * np.pi
        th = 0.05 + V * (np.pi - 0.1)  # polar angle, poles cut off
        d = np.stack([np.sin(th) * np.cos(ang), np.cos(th), np.sin(th) * np.sin(ang)], axis=-1)
        pos = c + radius * d
        tan = np.stack([-np.sin(ang), np.zeros_like(ang), np.cos(ang)], axis=-1)
        uv = np.stack([U * 2.0, V], axis=-1)
        return pos, d, tan, uv
    return param_surface(fn, nseg, nring, material)


# --------------------------------------------------------------------------
# procedural textures
# --------------------------------------------------------------------------
def _noise_grids(rng, size, octaves=5, base=4):
    """The random lattices of one value-noise image (everything that consumes the generator)."""
    return [rng.random((min(base << o, size),) * 2, dtype=np.float32) for o in range(octaves)]


def _value_noise(grids, size):
    """Tileable value noise in [0,1], size x size, from the lattices of _noise_grids."""
    out = np.zeros((size, size), np.float32)
    amp, tot = 1.0, 0.0
    for g in grids:
        n = g.shape[0]
        rep = size // n
        # bilinear upsample with wrap
        gx = np.concatenate([g, g[:, :1]], axis=1)
        gx = np.concatenate([gx, gx[:1, :]], axis=0)
        f = (np.arange(size, dtype=np.float32) % rep) / rep
        i0 = np.arange(size) // rep
        a = gx[i0][:, i0] * (1 - f)[None, :] + gx[i0][:, i0 + 1] * f[None, :]
        b = gx[i0 + 1][:, i0] * (1 - f)[None, :] + gx[i0 + 1][:, i0 + 1] * f[None, :]
        out += amp * (a * (1 - f)[:, None] + b * f[:, None])
        tot += amp
        amp *= 0.5
    return out / tot


def material_draw(rng, size):
    """The random numbers of one material's textures: the only part that must run in sequence."""
    return _noise_grids(rng, size), _noise_grids(rng, size, octaves=3)


def material_textures(draw, size, tint, alpha_mask=False):
    """Four (five with the mask) RGBA8 level-0 images for one material, laid out
    as stb returns them for the asset kinds of the real scene: RGB(A) base
    colour, grey roughness / metalness replicated to RGB with alpha 255, RGB
    tangent-space normal map.  `draw` comes from material_draw."""
    n0 = _value_noise(draw[0], size)
    n1 = np.roll(n0, (size // 3, size // 5), axis=(0, 1))
    n2 = _value_noise(draw[1], size)
    base = np.empty((size, size, 4), np.uint8)
    shade = 0.35 + 0.65 * n0
    for c in range(3):
        base[..., c] = np.clip(255.0 * tint[c] * shade * (0.9 + 0.2 * n1), 0, 255).astype(np.uint8)
    base[..., 3] = 255
    rough = np.empty((size, size, 4), np.uint8)
    rv = np.clip(255.0 * (0.3 + 0.6 * n1), 0, 255).astype(np.uint8)
    rough[..., 0] = rough[..., 1] = rough[..., 2] = rv
    rough[..., 3] = 255
    metal = np.empty((size, size, 4), np.uint8)
    mv = np.where(n2 > 0.62, 255, 0).astype(np.uint8)
    metal[..., 0] = metal[..., 1] = metal[..., 2] = mv
    metal[..., 3] = 255
    # normal map from the height field n0
    hx = np.roll(n0, -1, axis=1) - np.roll(n0, 1, axis=1)
    hy = np.roll(n0, -1, axis=0) - np.roll(n0, 1, axis=0)
    k = size / 64.0
    nx, ny, nz = -hx * k, -hy * k, np.ones_like(n0)
    ln = np.sqrt(nx * nx + ny * ny + nz * nz)
    nmap = np.empty((size, size, 4), np.uint8)
    nmap[..., 0] = np.clip((nx / ln * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    nmap[..., 1] = np.clip((ny / ln * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    nmap[..., 2] = np.clip((nz / ln * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    nmap[..., 3] = 255
    tex = [dict(rgba8=base, space=1, channels=4), dict(rgba8=rough, space=0, channels=1),
           dict(rgba8=metal, space=0, channels=1), dict(rgba8=nmap, space=0, channels=3)]
    if alpha_mask:
        mask = base.copy()
        mask[..., 3] = np.where(n2 > 0.45, 255, 0).astype(np.uint8)
        tex.append(dict(rgba8=mask, space=1, channels=4))
    return tex


def _add_material(rng, textures, materials, size, tint, alpha_mask=False, built=None):
    t0 = len(textures)
    textures.extend(built if built is not None else material_textures(material_draw(rng, size), size, tint, alpha_mask))
    # BakedMaterialInfo order: baseColor, roughness, metalness, alphaMask, normalMap, emissive
    materials.append((t0, t0 + 1, t0 + 2, (t0 + 4) if alpha_mask else NO_TEXTURE, t0 + 3, t0))
    return len(materials) - 1


def foliage_cluster(rng, center, count, size, material, spread=0.8):
    """Double-sided alpha-masked quads with random orientations."""
    meshes = []
    pos, nrm, uv, tan, idx = [], [], [], [], []
    for q in range(count):
        c = np.asarray(center) + rng.normal(0, spread, 3) * np.array([1.0, 0.6, 1.0])
        a = rng.random() * 2 * np.pi
        tilt = (rng.random() - 0.5) * 0.8
        du = np.array([np.cos(a), 0.0, np.sin(a)]) * size
        dv = np.array([-np.sin(a) * np.sin(tilt), np.cos(tilt), np.cos(a) * np.sin(tilt)]) * size
        n = np.cross(du, dv)
        n /= np.linalg.norm(n)
        base = len(pos)
        for (s, t) in ((0, 0), (1, 0), (1, 1), (0, 1)):
            pos.append(c + (s - 0.5) * du + (t - 0.5) * dv)
            nrm.append(n)
            uv.append([s, t])
            tan.append(du / np.linalg.norm(du))
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    meshes.append(_finish(np.array(pos), np.array(nrm), np.array(uv), np.array(tan), np.array(idx), material))
    return meshes


# --------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------
ODD_SIZES = [(24, 20), (17, 9), (5, 7), (16, 16), (12, 10), (31, 13), (3, 32), (1, 6)]


def micro_scene(tex_size=16, seed=7, odd_textures=False):
    """A few dozen triangles with every feature the parity tests need: a floor
    that passes under the camera (near-plane clipping), a tiled back wall
    (minification), a column, a sphere, two alpha-masked double-sided quads,
    an opaque mesh listed after the alpha-masked one (submission order differs
    from file order)."""
    rng = np.random.default_rng(seed)
    textures, materials = [], []
    m_floor = _add_material(rng, textures, materials, tex_size, (0.8, 0.7, 0.6))
    m_wall = _add_material(rng, textures, materials, tex_size * 2, (0.6, 0.65, 0.8))
    m_leaf = _add_material(rng, textures, materials, tex_size, (0.3, 0.8, 0.3), alpha_mask=True)
    m_metal = _add_material(rng, textures, materials, tex_size, (0.9, 0.8, 0.4))
    meshes = []
    meshes.append(grid_surface((-4, -1.5, 2), (8, 0, 0), (0, 0, -14), 4, 6, (0, 0), (6, 0), (0, 10), m_floor))
    meshes.append(grid_surface((-4, -1.5, -12), (8, 0, 0), (0, 5, 0), 3, 2, (0, 0), (9, 0), (0, 6), m_wall))
    meshes.append(grid_surface((-4, -1.5, 2), (0, 0, -14), (0, 5, 0), 3, 2, (0, 0), (3, 0), (0, 1), m_wall))
    meshes += foliage_cluster(rng, (1.2, -0.6, -5.0), 3, 1.6, m_leaf)
    meshes.append(fluted_column(-1.5, -6.0, -1.5, 2.5, 0.5, 10, 3, m_metal, flutes=5, depth=0.1))
    meshes.append(sphere_bowl((0.6, -0.9, -3.5), 0.6, 10, 6, m_metal))
    # a coplanar duplicate of part of the floor: equal-depth fragments, later primitive must win
    meshes.append(grid_surface((-1, -1.5, -2), (2, 0, 0), (0, 0, -2), 1, 1, (0, 0), (1, 0), (0, 1), m_wall))
    if odd_textures:
        # non-power-of-two sizes, different for every texture of a material (crops of the generated ones)
        for i, t in enumerate(textures):
            h, w = ODD_SIZES[i % len(ODD_SIZES)]
            src = t["rgba8"]
            reps = (-(-h // src.shape[0]), -(-w // src.shape[1]), 1)
            t["rgba8"] = np.ascontiguousarray(np.tile(src, reps)[:h, :w])
    return PackedScene(textures, materials, meshes)


def proxy_scene(detail=1.0, tex_size=1024, seed=PROXY_SEED, materials_total=48, foliage=True):
    """SunTemple-proxy v1.  detail=1.0 gives about 5.2e5 triangles; smaller
    values thin the tessellation for tests."""
    rng = np.random.default_rng(seed)
    textures, materials = [], []
    tints = rng.random((materials_total, 3)) * 0.6 + 0.35
    alpha_ids = {15, 37, 40}  # three alpha-masked materials, as suntemple.mtl:131,330,349
    # the lattices are drawn in sequence (they fix the scene's bytes); the images are then built on a thread pool
    draws = [material_draw(rng, tex_size) for _ in range(materials_total)]
    workers = max(1, min(16, len(os.sched_getaffinity(0)), materials_total)) if tex_size >= 256 else 1
    build = lambda i: material_textures(draws[i], tex_size, tints[i], i in alpha_ids)
    if workers > 1:
        with ThreadPoolExecutor(workers) as pool:
            built = list(pool.map(build, range(materials_total)))
    else:
        built = [build(i) for i in range(materials_total)]
    for i in range(materials_total):
        _add_material(rng, textures, materials, tex_size, tints[i], alpha_mask=(i in alpha_ids), built=built[i])
    opaque = [i for i in range(materials_total) if i not in alpha_ids]
    alpha = sorted(alpha_ids)
    meshes = []
    x0, x1, y0, y1, z0, z1 = -12.0, 12.0, -5.0, 10.0, 5.0, -95.0
    cell = 0.25 / max(detail, 1e-3) ** 0.5
    nsec = 10
    zs = np.linspace(z0, z1, nsec + 1)
    k = 0

    def nx(length):
        return max(1, int(round(abs(length) / cell)))

    for s in range(nsec):
        za, zb = zs[s], zs[s + 1]
        L = zb - za  # negative
        # floor (normal +y): du = +x, dv = -z  -> du x dv = (1,0,0)x(0,0,-1) = (0,1,0)
        meshes.append(grid_surface((x0, y0, za), (x1 - x0, 0, 0), (0, 0, L), nx(x1 - x0), nx(L),
                                   (0, -za / 4), ((x1 - x0) / 4, 0), (0, -L / 4), opaque[k % len(opaque)])); k += 1
        # ceiling (normal -y): du = +x, dv = +z direction reversed
        meshes.append(grid_surface((x0, y1, zb), (x1 - x0, 0, 0), (0, 0, -L), nx(x1 - x0), nx(L),
                                   (0, 0), ((x1 - x0) / 4, 0), (0, -L / 4), opaque[k % len(opaque)])); k += 1
        # left wall x = x0 (normal +x): du = -z... du=(0,0,L) dv=(0,h,0): du x dv = (0,0,L)x(0,h,0) = (-L*h,0,0) -> +x since L<0
        meshes.append(grid_surface((x0, y0, za), (0, 0, L), (0, y1 - y0, 0), nx(L), nx(y1 - y0),
                                   (0, 0), (-L / 4, 0), (0, (y1 - y0) / 4), opaque[k % len(opaque)])); k += 1
        # right wall x = x1 (normal -x)
        meshes.append(grid_surface((x1, y0, zb), (0, 0, -L), (0, y1 - y0, 0), nx(L), nx(y1 - y0),
                                   (0, 0), (-L / 4, 0), (0, (y1 - y0) / 4), opaque[k % len(opaque)])); k += 1
    # end walls
    meshes.append(grid_surface((x1, y0, z0), (x0 - x1, 0, 0), (0, y1 - y0, 0), nx(x1 - x0), nx(y1 - y0),
                               (0, 0), (6, 0), (0, 4), opaque[k % len(opaque)])); k += 1  # behind camera, normal -z
    meshes.append(grid_surface((x0, y0, z1), (x1 - x0, 0, 0), (0, y1 - y0, 0), nx(x1 - x0), nx(y1 - y0),
                               (0, 0), (6, 0), (0, 4), opaque[k % len(opaque)])); k += 1  # far end, normal +z
    # two rows of 24 fluted columns
    nseg = max(8, int(round(64 * detail ** 0.5)))
    nring = max(2, int(round(40 * detail ** 0.5)))
    for row, cx in enumerate((-9.5, 9.5)):
        for c in range(24):
            cz = 2.0 - (c + 0.5) * (97.0 / 24.0)
            meshes.append(fluted_column(cx, cz, y0, y1, 0.7, nseg, nring, opaque[k % len(opaque)])); k += 1
    # braziers: pedestal + bowl under each light
    bseg = max(6, int(round(32 * detail ** 0.5)))
    bring = max(3, int(round(16 * detail ** 0.5)))
    for i, lp in enumerate(BRAZIERS):
        bowl_c = (float(lp[0]), float(lp[1]) - 0.45, float(lp[2]))
        meshes.append(sphere_bowl(bowl_c, 0.35, bseg, bring, opaque[k % len(opaque)])); k += 1
        meshes.append(fluted_column(bowl_c[0], bowl_c[2], y0, bowl_c[1] - 0.3, 0.18, max(6, bseg // 2), max(2, bring // 2),
                                    opaque[k % len(opaque)], flutes=6, depth=0.15)); k += 1
    # foliage: three clusters of alpha-masked quads
    fcount = max(4, int(round(500 * detail)))
    for i, c in enumerate(((-6.0, -3.5, -8.0), (6.5, -3.5, -30.0), (-5.0, -3.5, -55.0)) if foliage else ()):
        meshes += foliage_cluster(rng, c, fcount, 0.6, alpha[i % len(alpha)], spread=1.3)
    return PackedScene(textures, materials, meshes)
