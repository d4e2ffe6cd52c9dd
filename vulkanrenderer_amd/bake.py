"""ctypes binding of libvkr_bake.so (include/vkr_bake.h): the asset bake step,
main-bake/main.cpp + index_mesh.cpp + tgen of the reference, as a library."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvkr_bake.so")
EXPORTS = ["vkr_index_mesh", "vkr_indexed_mesh_free", "vkr_bake_obj", "vkr_bake_last_error", "vkr_zstd_decode", "vkr_bake_free"]
_lib = None


class BakeError(RuntimeError):
    pass


class IndexedMesh(C.Structure):
    _fields_ = [("vertexCount", C.c_uint32), ("indexCount", C.c_uint32), ("positions", C.POINTER(C.c_float)),
                ("normals", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)), ("tangents", C.POINTER(C.c_float)),
                ("tangentsComp", C.POINTER(C.c_uint32)), ("indices", C.POINTER(C.c_uint32))]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BakeError("bake library not built: %s (run `make -C vulkanrenderer_amd/csrc`)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.vkr_bake_last_error.restype = C.c_char_p
        L.vkr_index_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.POINTER(IndexedMesh)]
        L.vkr_indexed_mesh_free.argtypes = [C.POINTER(IndexedMesh)]
        L.vkr_indexed_mesh_free.restype = None
        L.vkr_bake_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.vkr_zstd_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.vkr_bake_free.argtypes = [C.c_void_p]
        L.vkr_bake_free.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc:
        raise BakeError(lib().vkr_bake_last_error().decode("utf-8", "replace"))


def index_mesh(positions, normals, texcoords, tolerance=1e-6):
    """Triangle soup (float32[n,3], [n,3], [n,2], n a multiple of 3) -> dict of the indexed mesh's arrays
    (make_indexed_mesh, main-bake/index_mesh.cpp:76-228; its default tolerance is 1e-6, the bake tool passes 1e-5)."""
    p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    if not (len(p) == len(n) == len(t)):
        raise BakeError("positions, normals and texcoords differ in length")
    m = IndexedMesh()
    _check(lib().vkr_index_mesh(p.ctypes.data, n.ctypes.data, t.ctypes.data, len(p), tolerance, C.byref(m)))
    try:
        V, I = m.vertexCount, m.indexCount
        arr = lambda ptr, count, shape: np.ctypeslib.as_array(ptr, shape=(max(count, 1),))[:count].copy().reshape(shape)
        return {"positions": arr(m.positions, V * 3, (V, 3)), "normals": arr(m.normals, V * 3, (V, 3)),
                "texcoords": arr(m.texcoords, V * 2, (V, 2)), "tangents": arr(m.tangents, V * 4, (V, 4)),
                "tangentsComp": arr(m.tangentsComp, V, (V,)), "indices": arr(m.indices, I, (I,))}
    finally:
        lib().vkr_indexed_mesh_free(C.byref(m))


def bake_obj(obj_path, out_path, copy_textures=False):
    """OBJ (a Zstandard stream, as the reference's suntemple.obj-zstd, or plain text) + its MTL -> out_path (.comp5892mesh), as main-bake's process_model_ (main-bake/main.cpp:142-231)."""
    _check(lib().vkr_bake_obj(os.fsencode(obj_path), os.fsencode(out_path), int(copy_textures)))


def zstd_decode(data):
    """Decodes a buffer of Zstandard frames with the library's own decoder (what the reference's ZStdIStream,
    main-bake/zstdistream.cpp:35-117, does through libzstd)."""
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    _check(lib().vkr_zstd_decode(bytes(data), len(data), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value)
    finally:
        lib().vkr_bake_free(out)


def zstd_store(data, block=100000, checksum=False):
    """A Zstandard frame of raw (stored) blocks around `data`: valid input for any zstd decoder, made without a
    compressor.  Used to hand test OBJ files to the bake step the way the reference's source asset is packaged."""
    import struct
    data = bytes(data)
    out = bytearray(struct.pack("<I", 0xFD2FB528))
    out.append(0xE0 | (0x04 if checksum else 0))           # 8-byte content size, single segment, optional checksum
    out += struct.pack("<Q", len(data))
    if not data:
        out += struct.pack("<I", 1)[:3]                    # one empty raw block, last
    for at in range(0, len(data), block):
        chunk = data[at:at + block]
        last = at + block >= len(data)
        out += struct.pack("<I", (len(chunk) << 3) | int(last))[:3]
        out += chunk
    if checksum:
        import xxhash
        out += struct.pack("<I", xxhash.xxh64(data).intdigest() & 0xFFFFFFFF)
    return bytes(out)
