"""The bake step (libvkr_bake.so) against the file the reference's own bake tool writes for the same Zstandard-compressed
OBJ (tests/golden/bake/ref_suntemple.comp5892mesh, made by oracle/_ref/main-bake - main-bake/*.cpp compiled from the
reference's sources - through tests/make_bake_fixture.py), the library's Zstandard decoder against frames made by
libzstd, and properties of the welder."""
import hashlib
import json
import os
import shutil

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fnv1a(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def test_library_exports_every_declared_symbol():
    import re
    from vulkanrenderer_amd import bake
    hdr = open(os.path.join(os.path.dirname(GOLD), "..", "include", "vkr_bake.h")).read()
    declared = set(re.findall(r"VKR_BAKE_API[^;(]*?(vkr_\w+)\s*\(", hdr))
    assert declared == set(bake.EXPORTS)
    for name in declared:
        assert hasattr(bake.lib(), name)


def _bake_like_the_reference(tmp_path, source):
    """Lays the fixture out where the reference tool expects its input (main-bake/main.cpp:131-134) and bakes it."""
    from vulkanrenderer_amd import bake
    src = tmp_path / "assets-src" / "main"
    src.mkdir(parents=True)
    shutil.copy(os.path.join(GOLD, "bake", source), str(src / "suntemple.obj-zstd"))
    shutil.copy(os.path.join(GOLD, "bake", "model.mtl"), str(src / "model.mtl"))
    cwd = os.getcwd()
    os.chdir(str(tmp_path))   # texture paths are the OBJ's directory + the MTL's names, relative to the working directory
    try:
        bake.bake_obj("assets-src/main/suntemple.obj-zstd", "assets/main/suntemple.comp5892mesh")
    finally:
        os.chdir(cwd)
    return (tmp_path / "assets" / "main" / "suntemple.comp5892mesh").read_bytes()


def test_bake_obj_equals_the_reference_tools_file(tmp_path):
    """Zstandard stream in, .comp5892mesh out: the same bytes as the reference's bake tool wrote - texture and material
    tables, the mesh order of a shape with three materials (iteration order of its std::unordered_set), quads split
    along the shorter diagonal, polygons of five to eight corners cut into the same ears, welded vertex numbering,
    tangents and packed TBN words."""
    from vulkanrenderer_amd import baked_model
    want = open(os.path.join(GOLD, "bake", "ref_suntemple.comp5892mesh"), "rb").read()
    got = _bake_like_the_reference(tmp_path, "model.obj-zstd")
    assert got == want
    tex, mats, meshes, trailing = baked_model.read_baked_model(str(tmp_path / "assets" / "main" / "suntemple.comp5892mesh"))
    assert trailing == 0 and len(mats) == 3 and len(meshes) == 7   # floor, cube, shards, polygons, mixed x 3 materials
    # the shards: two exact duplicates and one vertex 4e-6 away merge, the one 3e-5 away and the ones with other uvs do not
    assert len(meshes[2]["positions"]) == 12 and len(meshes[2]["indices"]) == 15
    # pentagon + hexagon + octagon = 3 + 4 + 6 triangles
    assert len(meshes[3]["indices"]) == 3 * 13


def test_plain_text_and_stored_frames_bake_the_same(tmp_path):
    """The OBJ as plain text and as a frame of stored blocks give the file the compressed stream gives."""
    from vulkanrenderer_amd import bake
    want = open(os.path.join(GOLD, "bake", "ref_suntemple.comp5892mesh"), "rb").read()
    assert _bake_like_the_reference(tmp_path / "a", "model.obj") == want
    stored = tmp_path / "stored.obj-zstd"
    stored.write_bytes(bake.zstd_store(open(os.path.join(GOLD, "bake", "model.obj"), "rb").read(), block=1000, checksum=True))
    (tmp_path / "b").mkdir()
    shutil.copy(str(stored), os.path.join(GOLD, "bake", "_stored.tmp"))
    try:
        assert _bake_like_the_reference(tmp_path / "b", "_stored.tmp") == want
    finally:
        os.remove(os.path.join(GOLD, "bake", "_stored.tmp"))


def test_zstd_decoder_against_libzstd_frames():
    """vkr_zstd_decode on frames libzstd made at levels 1 to 22 (tests/make_zstd_fixtures.py): raw, RLE and compressed
    blocks, Huffman literals in one and four streams with direct and FSE-compressed weights, treeless literals, all
    sequence table modes, repeat offsets, multi-block frames."""
    from vulkanrenderer_amd import bake
    index = json.load(open(os.path.join(GOLD, "zstd", "index.json")))
    assert len(index) >= 10
    for name, info in index.items():
        data = bake.zstd_decode(open(os.path.join(GOLD, "zstd", name + ".zst"), "rb").read())
        assert len(data) == info["length"] and hashlib.sha256(data).hexdigest() == info["sha256"], name
    # concatenated frames and a skippable frame in between
    a = open(os.path.join(GOLD, "zstd", "tiny_l3.zst"), "rb").read()
    b = open(os.path.join(GOLD, "zstd", "obj_l19.zst"), "rb").read()
    skip = b"\x50\x2a\x4d\x18" + (5).to_bytes(4, "little") + b"12345"
    both = bake.zstd_decode(a + skip + b)
    assert both[:5] == b"hello" and hashlib.sha256(both[5:]).hexdigest() == index["obj_l19"]["sha256"]


def test_zstd_decoder_rejects_damaged_frames():
    from vulkanrenderer_amd import bake
    good = open(os.path.join(GOLD, "zstd", "obj_l19.zst"), "rb").read()
    with pytest.raises(bake.BakeError, match="zstd"):
        bake.zstd_decode(b"not a frame at all")
    with pytest.raises(bake.BakeError, match="zstd"):
        bake.zstd_decode(good[: len(good) // 2])
    rng = np.random.default_rng(3)
    rejected = 0
    for _ in range(200):   # flipped bytes: an error or (rarely) other content, never a crash
        bad = bytearray(good)
        bad[int(rng.integers(6, len(bad)))] ^= int(rng.integers(1, 256))
        try:
            bake.zstd_decode(bytes(bad))
        except bake.BakeError:
            rejected += 1
    assert rejected > 100
    stored = bytearray(bake.zstd_store(b"some content to protect" * 10, checksum=True))
    stored[20] ^= 1
    with pytest.raises(bake.BakeError, match="checksum"):
        bake.zstd_decode(bytes(stored))


def test_zstd_decoder_bounds_what_a_small_input_can_ask_for():
    """Hand-made frames whose headers promise far more than they hold: rejected, and without the memory they claim."""
    import resource
    import struct
    from vulkanrenderer_amd import bake
    magic = struct.pack("<I", 0xFD2FB528)
    # single-segment frame, 8-byte content size of 1.5 GiB, one raw block of 4 bytes
    frame = magic + bytes([0xE0]) + struct.pack("<Q", 3 << 29) + struct.pack("<I", (4 << 3) | 1)[:3] + b"abcd"
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with pytest.raises(bake.BakeError, match="content size"):
        bake.zstd_decode(frame)
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < (1 << 20)   # KiB: less than 1 GiB more
    # an RLE block that claims 2 MiB (blocks regenerate at most 128 KiB)
    frame = magic + bytes([0x00, 0x58]) + struct.pack("<I", ((2 << 20) - 1) << 3 | (1 << 1) | 1)[:3] + b"x"
    with pytest.raises(bake.BakeError, match="block maximum"):
        bake.zstd_decode(frame)
    # the largest legal RLE block is fine
    frame = magic + bytes([0x00, 0x58]) + struct.pack("<I", (128 << 10) << 3 | (1 << 1) | 1)[:3] + b"x"
    assert bake.zstd_decode(frame) == b"x" * (128 << 10)
    # a window smaller than 128 KiB lowers the block maximum with it: 1 KiB window (descriptor 0x00), 2 KiB RLE block
    frame = magic + bytes([0x00, 0x00]) + struct.pack("<I", (2 << 10) << 3 | (1 << 1) | 1)[:3] + b"x"
    with pytest.raises(bake.BakeError, match="block maximum"):
        bake.zstd_decode(frame)
    frame = magic + bytes([0x00, 0x00]) + struct.pack("<I", (1 << 10) << 3 | (1 << 1) | 1)[:3] + b"x"
    assert bake.zstd_decode(frame) == b"x" * (1 << 10)


def test_baked_file_loads_in_the_reference_loader_format(tmp_path):
    """The writer's output passes the reader that mirrors load_baked_model (magic, variant, counts, no trailing bytes)
    and its packed TBN equals the packer that is pinned separately (test_pinned_fixtures)."""
    from vulkanrenderer_amd import bake, baked_model
    out = str(tmp_path / "m.comp5892mesh")
    bake.bake_obj(os.path.join(GOLD, "bake", "model.obj"), out)
    _, _, meshes, _ = baked_model.read_baked_model(out)
    degenerate = 0
    for m in meshes:
        # the octagon of the fixture maps to a line in texture space: tgen gives it no tangent (NaN), as in the reference's
        # file; everywhere else the frame is orthonormal and packs to the word the pinned packer gives
        ok = np.isfinite(m["tangents"]).all(axis=1)
        degenerate += int((~ok).sum())
        assert np.array_equal(baked_model.pack_tbn(m["tangents"][ok], m["normals"][ok]), m["tangentsComp"][ok])
        assert np.abs(np.linalg.norm(m["tangents"][ok, :3], axis=1) - 1).max() < 1e-5
        assert np.abs((m["tangents"][ok, :3] * m["normals"][ok]).sum(1)).max() < 1e-5   # orthogonalised against the normal
        assert m["indices"].max() == len(m["positions"]) - 1
    assert 0 < degenerate <= 12


def test_weld_properties():
    from vulkanrenderer_amd import bake
    rng = np.random.default_rng(5)
    # a soup made from an indexed grid: welding finds the grid back, numbered by first use
    n = 9
    gx, gy = np.meshgrid(np.arange(n), np.arange(n))
    P = np.stack([gx, rng.uniform(0, 1, (n, n)), gy], -1).reshape(-1, 3).astype(np.float32)
    T = (P[:, [0, 2]] / (n - 1)).astype(np.float32)
    N = np.tile(np.array([0, 1, 0], np.float32), (n * n, 1))
    quads = [(j * n + i, j * n + i + 1, (j + 1) * n + i + 1, (j + 1) * n + i) for j in range(n - 1) for i in range(n - 1)]
    tri = np.array([(a, b, c, a, c, d) for a, b, c, d in quads]).reshape(-1)
    m = bake.index_mesh(P[tri], N[tri], T[tri], 1e-5)
    assert len(m["positions"]) == n * n and len(m["indices"]) == len(tri)
    assert np.array_equal(m["positions"][m["indices"]], P[tri]) and np.array_equal(m["texcoords"][m["indices"]], T[tri])
    first = {}
    for k, v in enumerate(tri):
        first.setdefault(int(v), len(first))
    assert m["indices"].tolist() == [first[int(v)] for v in tri]
    # idempotent: indexing the de-indexed result changes nothing
    m2 = bake.index_mesh(m["positions"][m["indices"]], m["normals"][m["indices"]], m["texcoords"][m["indices"]], 1e-5)
    assert all(np.array_equal(m[k], m2[k]) for k in m)
    # tolerance 0 merges exact duplicates only; a different normal or uv keeps vertices apart
    Pz = np.zeros((6, 3), np.float32); Nz = np.tile(np.array([0, 0, 1], np.float32), (6, 1)); Tz = np.zeros((6, 2), np.float32)
    Pz[1] = (1, 0, 0); Pz[2] = (0, 1, 0); Pz[4] = (1, 0, 0); Pz[5] = (0, 1, 0)
    Tz[1] = Tz[4] = (1, 0); Tz[2] = (0, 1); Tz[5] = (0, 0.5)
    mz = bake.index_mesh(Pz, Nz, Tz, 0.0)
    assert mz["indices"].tolist() == [0, 1, 2, 0, 1, 3]
    with pytest.raises(bake.BakeError):
        bake.index_mesh(Pz[:4], Nz[:4], Tz[:4])


def test_bake_errors(tmp_path):
    from vulkanrenderer_amd import bake
    with pytest.raises(bake.BakeError):
        bake.bake_obj(str(tmp_path / "missing.obj"), str(tmp_path / "o.comp5892mesh"))
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")   # no vt / vn / material
    with pytest.raises(bake.BakeError):
        bake.bake_obj(str(p), str(tmp_path / "o.comp5892mesh"))


REAL_MTL = os.path.join(os.environ.get("VKR_REFERENCE_ASSETS", "/root/reference/assets-src/main"), "suntemple.mtl")


@pytest.mark.skipif(not os.path.exists(REAL_MTL), reason="the reference's material library exists in the build container only")
def test_real_material_library_bakes_like_the_reference_tool(tmp_path):
    """The real suntemple.mtl (48 materials) under one triangle per material (tests/golden/bake/mtl_materials.obj): the
    library writes the bytes the reference's own bake tool wrote for it (tests/make_asset_fixtures.py) - material order,
    the unique-texture table in its unordered_map order, colour spaces and channel counts, the fall-back textures of
    materials without maps (main-bake/main.cpp:236-251, 426-457) - and the alpha mask sits on exactly the three
    materials that carry a map_d line (the ones that start at suntemple.mtl:131, 330, 349; main/main.cpp:1080-1081)."""
    from vulkanrenderer_amd import bake, baked_model
    src = tmp_path / "assets-src" / "main"
    src.mkdir(parents=True)
    shutil.copy(os.path.join(GOLD, "bake", "mtl_materials.obj"), str(src / "suntemple.obj-zstd"))
    os.symlink(REAL_MTL, str(src / "suntemple.mtl"))
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        bake.bake_obj("assets-src/main/suntemple.obj-zstd", "assets/main/suntemple.comp5892mesh")
    finally:
        os.chdir(cwd)
    out = tmp_path / "assets" / "main" / "suntemple.comp5892mesh"
    assert out.read_bytes() == open(os.path.join(GOLD, "bake", "ref_suntemple_mtl.comp5892mesh"), "rb").read()
    tex, mats, meshes, trailing = baked_model.read_baked_model(str(out))
    assert trailing == 0 and len(mats) == 48 and len(meshes) == 48
    # the materials with a map_d line, by their position in the library
    names, masked, lines, started = [], [], [], 0
    for no, line in enumerate(open(REAL_MTL), 1):
        if line.startswith("newmtl"):
            names.append(line.split(None, 1)[1].strip()); started = no
        elif line.split()[:1] == ["map_d"]:
            masked.append(len(names) - 1); lines.append(started)
    assert lines == [131, 330, 349] and len(names) == 48   # the three double-sided foliage materials
    # the material table keeps the library's order; mesh k is the triangle drawn with material k
    with_mask = [i for i, m in enumerate(mats) if m[3] != 0xFFFFFFFF]
    assert with_mask == masked
    assert sorted(m["materialId"] for m in meshes) == list(range(48))
    # 193 scene textures + the fall-backs in use; the reader prefixes the file's directory to the stored names
    assert 190 <= len(tex) <= 196 and all("/suntemple-tex/" in t[0] for t in tex)
