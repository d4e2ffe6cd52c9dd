"""Reader / writer for the reference's baked model file (".comp5892mesh",
variant "21-tan") and the TBN quaternion packing stored in it.

File layout as read by load_baked_model_ (main/baked_model.cpp:70-174) and
written by write_model_data_ (main-bake/main.cpp:276-389); little-endian, no
padding:
  char[16] magic "\\0\\0COMP5892Mmesh", char[16] variant "21-tan" (zero padded)
  u32 U, U x { u32 len (counts the NUL), char[len], u8 space, u8 channels }
  u32 M, M x 6 u32 { baseColor, roughness, metalness, alphaMask, normalMap, emissive }
  u32 N, N x { u32 material, u32 V, u32 I, V x vec3 pos, V x vec3 normal,
               V x vec2 uv, V x vec4 tangent, V x u32 tangentComp, I x u32 index }
"""
import os
import struct

import numpy as np

FILE_MAGIC = b"\0\0COMP5892Mmesh"      # main/baked_model.cpp:12 (16 bytes with the implicit NUL)
FILE_VARIANT = b"21-tan"               # main/baked_model.cpp:13
MAX_STRING = 32 * 1024                 # main/baked_model.cpp:15

assert len(FILE_MAGIC) == 15


class BakedModelError(Exception):
    pass


def write_baked_model(path, textures, materials, meshes):
    """textures: list of (relative_path:str, space:int, channels:int)
    materials: list of 6 ints; meshes: list of dict with materialId, positions,
    normals, texcoords, tangents, tangentsComp, indices."""
    with open(path, "wb") as f:
        f.write(FILE_MAGIC.ljust(16, b"\0"))
        f.write(FILE_VARIANT.ljust(16, b"\0"))
        f.write(struct.pack("<I", len(textures)))
        for p, space, ch in textures:
            b = p.encode("utf-8") + b"\0"  # length counts the terminating NUL (main-bake/main.cpp:264-273)
            f.write(struct.pack("<I", len(b)))
            f.write(b)
            f.write(struct.pack("<BB", space, ch))
        f.write(struct.pack("<I", len(materials)))
        for m in materials:
            f.write(struct.pack("<6I", *[int(x) & 0xFFFFFFFF for x in m]))
        f.write(struct.pack("<I", len(meshes)))
        for m in meshes:
            pos = np.ascontiguousarray(m["positions"], np.float32).reshape(-1, 3)
            V = pos.shape[0]
            nrm = np.ascontiguousarray(m["normals"], np.float32).reshape(V, 3)
            uv = np.ascontiguousarray(m["texcoords"], np.float32).reshape(V, 2)
            tan = np.ascontiguousarray(m["tangents"], np.float32).reshape(V, 4)
            tc = np.ascontiguousarray(m["tangentsComp"], np.uint32).reshape(V)
            idx = np.ascontiguousarray(m["indices"], np.uint32).reshape(-1)
            f.write(struct.pack("<3I", int(m["materialId"]), V, idx.shape[0]))
            for a in (pos, nrm, uv, tan, tc, idx):
                f.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())


def read_baked_model(path):
    """Returns (textures, materials, meshes) in the shapes write_baked_model
    takes; texture paths are prefixed with the file's directory like
    main/baked_model.cpp:75-81,101 does."""
    prefix = path[: path.rfind("/") + 1] if "/" in path else ""
    with open(path, "rb") as f:
        data = f.read()
    off = 0

    def take(n):
        nonlocal off
        if off + n > len(data):
            raise BakedModelError("expected %d bytes, got %d" % (n, len(data) - off))
        b = data[off:off + n]
        off += n
        return b

    def u32():
        return struct.unpack("<I", take(4))[0]

    if take(16) != FILE_MAGIC.ljust(16, b"\0"):
        raise BakedModelError("%s: invalid file signature" % path)
    variant = take(16)
    if variant != FILE_VARIANT.ljust(16, b"\0"):
        raise BakedModelError("%s: file variant is %r, expected %r" % (path, variant.rstrip(b"\0"), FILE_VARIANT))
    textures = []
    for _ in range(u32()):
        n = u32()
        if n >= MAX_STRING:
            raise BakedModelError("unexpectedly long string (%u bytes)" % n)
        p = take(n).rstrip(b"\0").decode("utf-8")
        space, ch = struct.unpack("<BB", take(2))
        textures.append((prefix + p, space, ch))
    materials = [struct.unpack("<6I", take(24)) for _ in range(u32())]
    meshes = []
    for _ in range(u32()):
        mat, V, I = struct.unpack("<3I", take(12))
        pos = np.frombuffer(take(V * 12), "<f4").reshape(V, 3).copy()
        nrm = np.frombuffer(take(V * 12), "<f4").reshape(V, 3).copy()
        uv = np.frombuffer(take(V * 8), "<f4").reshape(V, 2).copy()
        tan = np.frombuffer(take(V * 16), "<f4").reshape(V, 4).copy()
        tc = np.frombuffer(take(V * 4), "<u4").copy()
        idx = np.frombuffer(take(I * 4), "<u4").copy()
        meshes.append(dict(materialId=mat, positions=pos, normals=nrm, texcoords=uv, tangents=tan,
                           tangentsComp=tc, indices=idx))
    trailing = len(data) - off
    return textures, materials, meshes, trailing


def _normalize3(v):
    v = v.astype(np.float32)
    d = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
    inv = np.float32(1.0) / np.sqrt(d, dtype=np.float32)
    return v * inv[:, None]


def _cross(x, y):
    return np.stack([x[:, 1] * y[:, 2] - y[:, 1] * x[:, 2],
                     x[:, 2] * y[:, 0] - y[:, 2] * x[:, 0],
                     x[:, 0] * y[:, 1] - y[:, 0] * x[:, 1]], axis=1).astype(np.float32)


def pack_tbn(tangent4, normal):
    """Vectorised restatement of the TBN packer of main-bake/index_mesh.cpp:178-220:
    T = normalize(tangent.xyz * tangent.w), N = normalize(normal), B = normalize(N x T),
    quat_cast(mat3(T,B,N)) normalised, largest component dropped (sign made
    positive), the other three mapped from [-1/sqrt2, 1/sqrt2] to [0,1023] by
    truncation, packed as (index << 30) | (s0 << 20) | (s1 << 10) | s2."""
    t4 = np.asarray(tangent4, np.float32).reshape(-1, 4)
    T = _normalize3(t4[:, :3] * t4[:, 3:4])
    N = _normalize3(np.asarray(normal, np.float32).reshape(-1, 3))
    B = _normalize3(_cross(N, T))
    m00, m01, m02 = T[:, 0], T[:, 1], T[:, 2]
    m10, m11, m12 = B[:, 0], B[:, 1], B[:, 2]
    m20, m21, m22 = N[:, 0], N[:, 1], N[:, 2]
    fx = m00 - m11 - m22
    fy = m11 - m00 - m22
    fz = m22 - m00 - m11
    fw = m00 + m11 + m22
    big = np.zeros(len(T), np.int32)
    fb = fw.copy()
    for k, f in ((1, fx), (2, fy), (3, fz)):
        sel = f > fb
        fb = np.where(sel, f, fb)
        big = np.where(sel, k, big)
    bv = np.sqrt(fb + np.float32(1.0), dtype=np.float32) * np.float32(0.5)
    mult = np.float32(0.25) / bv
    w = np.select([big == 0, big == 1, big == 2, big == 3], [bv, (m12 - m21) * mult, (m20 - m02) * mult, (m01 - m10) * mult])
    x = np.select([big == 0, big == 1, big == 2, big == 3], [(m12 - m21) * mult, bv, (m01 + m10) * mult, (m20 + m02) * mult])
    y = np.select([big == 0, big == 1, big == 2, big == 3], [(m20 - m02) * mult, (m01 + m10) * mult, bv, (m12 + m21) * mult])
    z = np.select([big == 0, big == 1, big == 2, big == 3], [(m01 - m10) * mult, (m20 + m02) * mult, (m12 + m21) * mult, bv])
    w, x, y, z = [a.astype(np.float32) for a in (w, x, y, z)]
    ln = np.sqrt((w * w + x * x) + (y * y + z * z), dtype=np.float32)
    ok = ln > 0
    ool = np.float32(1.0) / np.where(ok, ln, np.float32(1.0))
    q = np.stack([np.where(ok, x * ool, 0), np.where(ok, y * ool, 0), np.where(ok, z * ool, 0),
                  np.where(ok, w * ool, 1)], axis=1).astype(np.float32)  # x y z w
    maxi = np.argmax(np.abs(q), axis=1)  # first maximum, like std::max_element
    neg = q[np.arange(len(q)), maxi] < 0
    q = np.where(neg[:, None], -q, q)
    inv_sqrt2 = 1.0 / np.sqrt(2.0)
    mapped = ((q.astype(np.float64) + inv_sqrt2) / np.sqrt(2.0) * np.float64(np.float32(1023.0)))
    mapped = mapped.astype(np.int64).astype(np.uint32)  # truncation
    # gather the three kept components in order
    keep = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], np.int64)[maxi]
    s = np.take_along_axis(mapped, keep, axis=1)
    out = (maxi.astype(np.uint32) << np.uint32(30)) | (s[:, 0] << np.uint32(20)) | (s[:, 1] << np.uint32(10)) | s[:, 2]
    return out.astype(np.uint32)


def load_scene_arrays(path, image_loader=None):
    """Reads a baked model and decodes its textures the way the reference does
    (stbi_load forced to 4 channels, flipped vertically, utils/vkimage.cpp:62-65).
    Returns the (textures, materials, meshes) lists PackedScene takes."""
    if image_loader is None:
        from PIL import Image

        def image_loader(p):
            im = Image.open(p).convert("RGBA")
            return np.ascontiguousarray(np.asarray(im)[::-1])
    tex, mats, meshes, _ = read_baked_model(path)
    textures = [dict(rgba8=image_loader(p), space=space, channels=ch) for (p, space, ch) in tex]
    return textures, [tuple(m) for m in mats], meshes
