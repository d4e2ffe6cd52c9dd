"""GPU parity on adversarial random geometry: the HIP renderer against the CPU oracle on seeded triangle soups built to
sit on the rasteriser's edge cases - slivers, fans whose edges all meet in one vertex, coplanar duplicates (equal depth:
the later primitive must win), vertices that project onto pixel centres and pixel corners, triangles through the near
plane and behind the camera, triangles far outside the guard band, sub-pixel and zero-area triangles, alpha-tested and
opaque meshes interleaved.  The fixed-function rules under test are those of the reference's pipelines
(main/main.cpp:2298-2327: cull mode, LESS_OR_EQUAL, no depth clamp) as the oracle restates them.
Integer buffers bit-exact, colour within 1/255."""
import numpy as np
import pytest

from conftest import IDENTITY, pose_matrix

pytestmark = pytest.mark.gpu

SIZES = [(97, 61), (160, 90), (333, 187), (64, 64)]


def _mesh(pos, uv, idx, material, rng=None):
    """A mesh with a constant tangent frame, or (with rng) a random orthonormal one per vertex."""
    from vulkanrenderer_amd.baked_model import pack_tbn
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    n = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(pos), 1))
    t = np.tile(np.array([[1.0, 0.0, 0.0, 1.0]], np.float32), (len(pos), 1))
    if rng is not None:
        n = rng.normal(size=(len(pos), 3))
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        r = rng.normal(size=(len(pos), 3))
        tv = np.cross(n, r)
        tv /= np.linalg.norm(tv, axis=1, keepdims=True)
        n = n.astype(np.float32)
        t = np.concatenate([tv, rng.choice([-1.0, 1.0], size=(len(pos), 1))], axis=1).astype(np.float32)
    return dict(materialId=int(material), positions=pos, normals=n, texcoords=np.ascontiguousarray(uv, np.float32).reshape(-1, 2),
                tangents=t, tangentsComp=pack_tbn(t, n), indices=np.ascontiguousarray(idx, np.uint32).reshape(-1))


def _soup(rng, n, lo, hi):
    pos = rng.uniform(lo, hi, size=(n * 3, 3))
    uv = rng.uniform(-2.0, 3.0, size=(n * 3, 2))
    return pos, uv, np.arange(n * 3)


def fuzz_scene(seed, W, H, scale=1):
    """A scene of a few hundred triangles (times `scale`) per family; camera at the origin looking down -z
    (camera2world = identity)."""
    from vulkanrenderer_amd import scenes
    from vulkanrenderer_amd.abi import PackedScene
    rng = np.random.default_rng(seed)
    textures, materials = [], []
    m_a = scenes._add_material(rng, textures, materials, 16, (0.8, 0.6, 0.5))
    m_b = scenes._add_material(rng, textures, materials, 8, (0.5, 0.7, 0.9))
    m_leaf = scenes._add_material(rng, textures, materials, 16, (0.3, 0.8, 0.3), alpha_mask=True)
    meshes = []
    # 1. plain soup, part of it crossing the near plane or behind the camera
    p, uv, idx = _soup(rng, 150 * scale, (-6, -4, -14), (6, 4, -2.5))
    meshes.append(_mesh(p, uv, idx, m_a, rng))
    a = rng.uniform((-1, -1, -3), (1, 1, 1.5), size=(16, 3))
    p = np.stack([a, a + rng.normal(0, 0.4, size=a.shape), a + rng.normal(0, 0.4, size=a.shape)], axis=1).reshape(-1, 3)
    meshes.append(_mesh(p, rng.uniform(0, 1, size=(48, 2)), np.arange(48), m_b))
    # 2. slivers: two vertices close together, the third far away
    a = rng.uniform((-4, -3, -10), (4, 3, -2), size=(120 * scale, 3))
    b = a + rng.normal(0, 0.004, size=a.shape)
    c = a + rng.uniform(-5, 5, size=a.shape)
    p = np.stack([a, b, c], axis=1).reshape(-1, 3)
    meshes.append(_mesh(p, rng.uniform(0, 1, size=(len(p), 2)), np.arange(len(p)), m_b))
    # 3. alpha-tested soup in between (submission order puts it behind the opaque meshes, file order does not)
    p, uv, idx = _soup(rng, 80 * scale, (-3, -2, -9), (3, 2, -1))
    meshes.append(_mesh(p, uv, idx, m_leaf, rng))
    # 4. fans: many triangles around one vertex, every edge shared by two of them (each pixel covered exactly once)
    for _ in range(3):
        centre = rng.uniform((-2, -2, -8), (2, 2, -3))
        k = int(rng.integers(5, 40))
        ang = np.sort(rng.uniform(0, 2 * np.pi, size=k))
        rim = centre + np.stack([np.cos(ang), np.sin(ang), rng.uniform(-0.5, 0.5, size=k)], axis=1) * rng.uniform(0.5, 3.0)
        pos = np.concatenate([centre[None], rim])
        tri = np.stack([np.zeros(k, int), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k], axis=1)
        meshes.append(_mesh(pos, rng.uniform(0, 4, size=(k + 1, 2)), tri, m_a))
    # 5. coplanar duplicates: the same triangles twice in one mesh and once more in a later mesh
    p, uv, idx = _soup(rng, 12, (-3, -2, -7), (3, 2, -4))
    meshes.append(_mesh(p, uv, np.concatenate([idx, idx]), m_b))
    meshes.append(_mesh(p, uv, idx, m_a))
    # 6. vertices on pixel centres and pixel corners: x_ndc = 2 (i + f) / W - 1 with f in {0, 0.5}, at z = -2^k so that
    #    the perspective division is exact
    fy = 1.0 / np.tan(np.float32(np.radians(60.0)) / 2.0)  # the projection of make_uniforms: fovy 60 degrees
    fx = fy * H / W
    tris = []
    for _ in range(60):
        z = -float(2 ** rng.integers(0, 4))
        v = []
        for _ in range(3):
            i = rng.integers(-2, W + 2) + rng.choice([0.0, 0.5])
            j = rng.integers(-2, H + 2) + rng.choice([0.0, 0.5])
            xn, yn = 2.0 * i / W - 1.0, 2.0 * j / H - 1.0
            v.append([xn * -z / fx, -yn * -z / fy, z])
        tris.append(v)
    p = np.array(tris).reshape(-1, 3)
    meshes.append(_mesh(p, rng.uniform(0, 1, size=(len(p), 2)), np.arange(len(p)), m_b))
    # 7. far outside the guard band, and huge triangles that cover the whole frame
    p, uv, idx = _soup(rng, 4, (-4000, -4000, -40), (4000, 4000, -16))
    meshes.append(_mesh(p, uv, idx, m_a))
    a = rng.uniform((-3, -2, -12), (3, 2, -3), size=(10, 3))
    b = a + rng.normal(0, 0.3, size=a.shape)
    c = rng.uniform((-4000, -4000, -12), (4000, 4000, 3), size=(10, 3))  # spikes that leave the guard band, some through z = 0
    p = np.stack([a, b, c], axis=1).reshape(-1, 3)
    meshes.append(_mesh(p, rng.uniform(0, 1, size=(30, 2)), np.arange(30), m_b))
    # 8. sub-pixel triangles and zero-area ones (repeated vertices, collinear vertices, repeated indices)
    a = rng.uniform((-3, -2, -6), (3, 2, -3), size=(150, 3))
    p = np.stack([a, a + rng.normal(0, 0.01, size=a.shape), a + rng.normal(0, 0.01, size=a.shape)], axis=1).reshape(-1, 3)
    idx = np.arange(450)
    idx[3:6] = idx[3]                     # one vertex three times
    p[7] = p[6]                           # two equal positions
    p[11] = 0.5 * (p[9] + p[10])          # collinear
    meshes.append(_mesh(p, rng.uniform(0, 1, size=(450, 2)), idx, m_leaf))
    # 9. a backdrop so that most pixels hold something
    meshes.append(_mesh([[-30, -20, -20], [30, -20, -20], [30, 20, -20], [-30, 20, -20]], [[0, 0], [7, 0], [7, 5], [0, 5]],
                        [0, 1, 2, 0, 2, 3], m_a))
    return PackedScene(textures, materials, meshes)


def _render_both(sc, osc, p, want):
    return osc.render(p, want=want, threads=8), sc.render(p, want=want)


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_soups_against_the_oracle(seed):
    from oracle import oracle
    from vulkanrenderer_amd import abi, renderer
    W, H = SIZES[seed % len(SIZES)]
    packed = fuzz_scene(seed, W, H, scale=1 if seed <= 8 else 6)
    ctx = renderer.Context(0)
    sc, osc = renderer.Scene(ctx, packed), oracle.OracleScene(packed)
    try:
        poses = [IDENTITY, pose_matrix(0.3, -0.2, -1.0, yaw=0.15 * (seed - 3), pitch=0.1)]
        for pi, pose in enumerate(poses):
            for name, kw, want in (
                    ("deferred", dict(deferred=True), ("rgba8", "tri_id", "depth_bits", "gbuf_normal", "gbuf_albedo")),
                    ("forward", dict(debug=1, shadow_size=128), ("rgba8", "tri_id", "depth_bits", "shadow_map")),
                    ("mip", dict(debug=2), ("rgba8", "tri_id", "depth_bits", "mip_level")),
                    ("overdraw", dict(debug=5), ("rgba8", "stencil")),
                    ("overshade", dict(debug=6), ("rgba8", "stencil", "depth_bits"))):
                p = abi.make_params(W, H, renderer.make_uniforms(W, H, pose, kw.get("debug", 1)), **kw)
                ref, got = _render_both(sc, osc, p, want)
                label = "seed %d pose %d %s" % (seed, pi, name)
                for k in ("tri_id", "depth_bits", "stencil", "mip_level"):
                    if k in want:
                        assert np.array_equal(ref[k], got[k]), "%s: %s differs at %d pixels" % (label, k, int((ref[k] != got[k]).sum()))
                if "shadow_map" in want:
                    assert np.array_equal(ref["shadow_map"].view(np.uint32), got["shadow_map"].view(np.uint32)), label + ": shadow map"
                d = np.abs(ref["rgba8"].astype(np.int32) - got["rgba8"].astype(np.int32))
                assert d.max() <= 1, "%s: colour off by %d at %d pixels" % (label, int(d.max()), int((d > 1).sum()))
                # the scene must actually exercise what it is built for
                if name == "overdraw":
                    assert ref["stencil"].max() >= 4
                if name == "deferred" and pi == 0:
                    assert (ref["tri_id"] != 0xFFFFFFFF).mean() > 0.3
    finally:
        sc.close(); osc.close(); ctx.close()


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_partitions_of_random_soups_reassemble(seed):
    """Bands, interleaved stripes, two frames in flight and the allocation-free entry point on the adversarial scenes:
    every way of rendering the frame gives the bytes of the frame rendered in one piece (triangles that end on a
    tile or stripe border, spikes through several shares, coplanar duplicates across shares)."""
    from vulkanrenderer_amd import abi, renderer
    from vulkanrenderer_amd.dist import band_rect, stripe_rows
    W, H = [(200, 136), (333, 187), (160, 96)][seed % 3]
    packed = fuzz_scene(seed, W, H, scale=2)
    ctx = renderer.Context(0)
    sc = renderer.Scene(ctx, packed)
    try:
        pose = pose_matrix(0.1, 0.1, -0.5, yaw=0.1 * (seed - 22))
        for name, kw, want in (("deferred", dict(deferred=True), ("rgba8", "tri_id", "depth_bits")),
                               ("forward", dict(debug=1, shadow_size=128), ("rgba8", "tri_id", "depth_bits")),
                               ("overdraw", dict(debug=5), ("rgba8", "stencil")),
                               ("overshade", dict(debug=6), ("rgba8", "stencil"))):
            def params(**extra):
                return abi.make_params(W, H, renderer.make_uniforms(W, H, pose, kw.get("debug", 1)), **dict(kw, **extra))
            full = sc.render(params(), want=want)
            for world, sh in ((3, 32), (5, 32)):
                got = {k: np.zeros_like(v) for k, v in full.items()}
                for r in range(world):
                    rows = stripe_rows(H, world, r, sh)
                    part = sc.render(params(stripes=abi.stripes(sh, world, r)), want=want)
                    for k in want:
                        got[k][rows] = part[k]
                for k in want:
                    assert np.array_equal(full[k], got[k]), "seed %d %s stripes/%d: %s" % (seed, name, world, k)
            bands = [sc.render(params(rect=band_rect(r, 4, W, H)), want=want) for r in range(4)]
            for k in want:
                assert np.array_equal(np.concatenate([b[k] for b in bands]), full[k]), "seed %d %s bands: %s" % (seed, name, k)
            static = sc.render(params(), want=want, static=True)
            for k in want:
                assert np.array_equal(static[k], full[k]), "seed %d %s allocation-free: %s" % (seed, name, k)
    finally:
        sc.close(); ctx.close()


@pytest.mark.parametrize("count", [1, 2, 31, 33, 63, 64, 65, 129, 255, 257])
def test_triangle_counts_around_wavefront_and_workgroup_sizes(count):
    """One mesh of `count` triangles (a wavefront holds 64, a workgroup 256: triangle setup hands its records out half a
    wavefront at a time), a second, alpha-tested one of 5 behind it, half of the first mesh's triangles facing away or
    off screen: ids, depths and counters bit-exact against the oracle."""
    from oracle import oracle
    from vulkanrenderer_amd import abi, renderer, scenes
    from vulkanrenderer_amd.abi import PackedScene
    rng = np.random.default_rng(100 + count)
    textures, materials = [], []
    m_a = scenes._add_material(rng, textures, materials, 8, (0.8, 0.6, 0.5))
    m_leaf = scenes._add_material(rng, textures, materials, 8, (0.3, 0.8, 0.3), alpha_mask=True)
    p, uv, idx = _soup(rng, count, (-7, -4, -12), (7, 4, -2.5))
    q, uvq, idxq = _soup(rng, 5, (-3, -2, -13), (3, 2, -9))
    packed = PackedScene(textures, materials, [_mesh(p, uv, idx, m_a, rng), _mesh(q, uvq, idxq, m_leaf)])
    W, H = 120, 68
    ctx = renderer.Context(0)
    sc, osc = renderer.Scene(ctx, packed), oracle.OracleScene(packed)
    try:
        for kw, want in ((dict(deferred=True), ("tri_id", "depth_bits")), (dict(debug=1, shadow_size=64), ("tri_id", "depth_bits", "shadow_map")),
                         (dict(debug=6), ("stencil", "depth_bits"))):
            prm = abi.make_params(W, H, renderer.make_uniforms(W, H, IDENTITY, kw.get("debug", 1)), **kw)
            ref, got = osc.render(prm, want=want), sc.render(prm, want=want)
            for k in want:
                assert np.array_equal(ref[k].view(np.uint32) if k == "shadow_map" else ref[k], got[k].view(np.uint32) if k == "shadow_map" else got[k]), (count, kw, k)
    finally:
        sc.close(); osc.close(); ctx.close()
