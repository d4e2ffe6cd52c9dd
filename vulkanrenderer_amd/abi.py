"""ctypes mirror of include/vkr_renderer.h.

The struct layouts follow the reference's own types: BakedModel
(main/baked_model.hpp:80-115) for the scene hand-off and struct Uniforms
(main/main.cpp:203-240) for the per-frame inputs.
"""
import ctypes as C

import numpy as np

NO_TEXTURE = 0xFFFFFFFF
NO_TRIANGLE = 0xFFFFFFFF
NUM_LIGHTS = 21


class TextureDesc(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("space", C.c_uint8), ("channels", C.c_uint8), ("_pad", C.c_uint8 * 6)]


class MaterialDesc(C.Structure):
    _fields_ = [("baseColorTextureId", C.c_uint32), ("roughnessTextureId", C.c_uint32),
                ("metalnessTextureId", C.c_uint32), ("alphaMaskTextureId", C.c_uint32),
                ("normalMapTextureId", C.c_uint32), ("emissiveTextureId", C.c_uint32)]


class MeshDesc(C.Structure):
    _fields_ = [("materialId", C.c_uint32), ("vertexCount", C.c_uint32), ("indexCount", C.c_uint32),
                ("_pad", C.c_uint32), ("positions", C.c_void_p), ("texcoords", C.c_void_p),
                ("normals", C.c_void_p), ("tangents", C.c_void_p), ("tangentsComp", C.c_void_p),
                ("indices", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [("textureCount", C.c_uint32), ("materialCount", C.c_uint32), ("meshCount", C.c_uint32),
                ("_pad", C.c_uint32), ("textures", C.POINTER(TextureDesc)),
                ("materials", C.POINTER(MaterialDesc)), ("meshes", C.POINTER(MeshDesc))]


class SceneUniform(C.Structure):
    _fields_ = [("camera", C.c_float * 16), ("projection", C.c_float * 16), ("projCam", C.c_float * 16),
                ("camPos", C.c_float * 4)]


class LightUniform(C.Structure):
    _fields_ = [("lightPos", C.c_float * 4), ("lightColour", C.c_float * 4)]


class DebugUniform(C.Structure):
    _fields_ = [("debug", C.c_int32)]


class MultipleLights(C.Structure):
    _fields_ = [("light", LightUniform * NUM_LIGHTS)]


class DepthMVP(C.Structure):
    _fields_ = [("depthMVP", C.c_float * 16)]


class Uniforms(C.Structure):
    _fields_ = [("sceneUniforms", SceneUniform), ("lightUniforms", LightUniform),
                ("debugUniforms", DebugUniform), ("multipleLightsUniform", MultipleLights),
                ("depthMVPUniform", DepthMVP)]


class FrameParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("shadow_size", C.c_uint32),
                ("debugVisualisation", C.c_int32), ("mosaicEffect", C.c_int32), ("deferredShading", C.c_int32),
                ("rect_x0", C.c_uint32), ("rect_y0", C.c_uint32), ("rect_x1", C.c_uint32), ("rect_y1", C.c_uint32),
                ("flags", C.c_uint32), ("stripes", C.c_uint32), ("uniforms", Uniforms)]


class FrameOut(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("tri_id", C.c_void_p), ("depth_bits", C.c_void_p),
                ("mip_level", C.c_void_p), ("stencil", C.c_void_p), ("gbuf_normal", C.c_void_p),
                ("gbuf_albedo", C.c_void_p), ("shadow_map", C.c_void_p)]


INPUT_KEYS = ("forward", "backward", "strafeLeft", "strafeRight", "levitate", "sink", "fast", "slow", "mousing")


class UserState(C.Structure):
    """vkr_user_state: the reference's UserState (main/main.cpp:111-125)."""
    _fields_ = [("inputMap", C.c_uint8 * 9), ("wasMousing", C.c_uint8), ("_pad", C.c_uint8 * 2),
                ("mouseX", C.c_float), ("mouseY", C.c_float), ("previousX", C.c_float), ("previousY", C.c_float),
                ("camera2world", C.c_float * 16),
                ("debugVisualisation", C.c_int32), ("mosaicEffect", C.c_int32), ("deferredShading", C.c_int32),
                ("shadows", C.c_int32)]


assert C.sizeof(UserState) == 108
assert C.sizeof(SceneUniform) == 208 and C.sizeof(LightUniform) == 32
assert C.sizeof(MultipleLights) == 672 and C.sizeof(DepthMVP) == 64
assert C.sizeof(TextureDesc) == 24 and C.sizeof(MaterialDesc) == 24 and C.sizeof(MeshDesc) == 64


class PackedScene:
    """Keeps the numpy arrays of a scene alive next to the ctypes SceneDesc that
    points into them."""

    def __init__(self, textures, materials, meshes):
        """textures: list of dict(rgba8=uint8[h,w,4], space=0|1, channels=int)
        materials: list of 6-tuples in BakedMaterialInfo order
        meshes: list of dict(materialId, positions f32[V,3], texcoords f32[V,2],
                normals f32[V,3]|None, tangents f32[V,4]|None, tangentsComp u32[V], indices u32[I])"""
        self.textures = textures
        self.materials = materials
        self.meshes = meshes
        self._keep = []
        nt, nm, nme = len(textures), len(materials), len(meshes)
        self._tex = (TextureDesc * max(nt, 1))()
        for i, t in enumerate(textures):
            a = np.ascontiguousarray(t["rgba8"], dtype=np.uint8)
            assert a.ndim == 3 and a.shape[2] == 4
            self._keep.append(a)
            self._tex[i].rgba8 = a.ctypes.data
            self._tex[i].height, self._tex[i].width = a.shape[0], a.shape[1]
            self._tex[i].space = int(t.get("space", 0))
            self._tex[i].channels = int(t.get("channels", 4))
        self._mat = (MaterialDesc * max(nm, 1))()
        for i, m in enumerate(materials):
            (self._mat[i].baseColorTextureId, self._mat[i].roughnessTextureId, self._mat[i].metalnessTextureId,
             self._mat[i].alphaMaskTextureId, self._mat[i].normalMapTextureId, self._mat[i].emissiveTextureId) = \
                [int(x) & 0xFFFFFFFF for x in m]
        self._mesh = (MeshDesc * max(nme, 1))()
        for i, m in enumerate(meshes):
            def arr(key, dt, cols):
                v = m.get(key)
                if v is None:
                    return None
                a = np.ascontiguousarray(v, dtype=dt)
                if cols:
                    a = a.reshape(-1, cols)
                self._keep.append(a)
                return a
            pos = arr("positions", np.float32, 3)
            uv = arr("texcoords", np.float32, 2)
            nrm = arr("normals", np.float32, 3)
            tan = arr("tangents", np.float32, 4)
            tc = arr("tangentsComp", np.uint32, 0)
            idx = arr("indices", np.uint32, 0)
            assert uv.shape[0] == pos.shape[0] == tc.shape[0]
            d = self._mesh[i]
            d.materialId = int(m["materialId"])
            d.vertexCount = pos.shape[0]
            d.indexCount = idx.shape[0]
            d.positions = pos.ctypes.data
            d.texcoords = uv.ctypes.data
            d.normals = nrm.ctypes.data if nrm is not None else None
            d.tangents = tan.ctypes.data if tan is not None else None
            d.tangentsComp = tc.ctypes.data
            d.indices = idx.ctypes.data
        self.desc = SceneDesc()
        self.desc.textureCount, self.desc.materialCount, self.desc.meshCount = nt, nm, nme
        self.desc.textures = C.cast(self._tex, C.POINTER(TextureDesc))
        self.desc.materials = C.cast(self._mat, C.POINTER(MaterialDesc))
        self.desc.meshes = C.cast(self._mesh, C.POINTER(MeshDesc))

    @property
    def triangle_count(self):
        return sum(len(m["indices"]) // 3 for m in self.meshes)


FRAME_SHADOW_PASS = 1  # VKR_FRAME_SHADOW_PASS
FRAME_FUSED_LIGHTING = 2  # VKR_FRAME_FUSED_LIGHTING


def stripes(height, count, index):
    """VKR_STRIPES(height, count, index)"""
    return int(height) | (int(count) << 16) | (int(index) << 24)


def make_params(width, height, uniforms, debug=1, mosaic=False, deferred=False, shadow_size=0, rect=None, flags=0, stripes=0):
    p = FrameParams()
    p.flags = flags
    p.stripes = stripes
    p.width, p.height, p.shadow_size = width, height, shadow_size
    p.debugVisualisation, p.mosaicEffect, p.deferredShading = int(debug), int(bool(mosaic)), int(bool(deferred))
    if rect is not None:
        p.rect_x0, p.rect_y0, p.rect_x1, p.rect_y1 = rect
    p.uniforms = uniforms
    return p


def mat4_to_c(m):
    """numpy 4x4 (math convention m[row, col]) -> glm column-major float[16]."""
    a = np.asarray(m, dtype=np.float32)
    return (C.c_float * 16)(*a.T.reshape(-1).tolist())
