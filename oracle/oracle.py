"""ctypes wrapper of the CPU oracle (oracle/vkr_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from vulkanrenderer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libvkr_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vkr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "_build/libvkr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ora_scene_create.restype = C.c_void_p
        L.ora_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        L.ora_scene_destroy.argtypes = [C.c_void_p]
        L.ora_render.argtypes = [C.c_void_p, C.POINTER(abi.FrameParams), C.POINTER(abi.FrameOut)]
        L.ora_render_rows.argtypes = [C.c_void_p, C.POINTER(abi.FrameParams), C.POINTER(abi.FrameOut), C.c_uint32, C.c_uint32]
        L.ora_render_bands.argtypes = [C.c_void_p, C.POINTER(abi.FrameParams), C.POINTER(abi.FrameOut), C.c_uint32]
        L.ora_set_band_threads.argtypes = [C.c_int]
        L.ora_texture_level.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ora_make_uniforms.argtypes = [C.POINTER(abi.Uniforms), C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_float), C.c_int32]
        L.ora_update_scene_uniforms.argtypes = [C.POINTER(abi.SceneUniform), C.c_uint32, C.c_uint32,
                                                C.POINTER(C.c_float)]
        L.ora_update_depth_mvp_uniforms.argtypes = [C.POINTER(abi.DepthMVP), C.POINTER(C.c_float)]
        L.ora_update_user_state.argtypes = [C.POINTER(abi.UserState), C.c_float]
        L.ora_update_user_state.restype = None
        L.ora_default_lights.argtypes = [C.POINTER(abi.LightUniform), C.POINTER(abi.MultipleLights)]
        L.ora_decode_tbn.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        L.ora_pack_tbn.restype = C.c_uint32
        L.ora_pack_tbn.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ora_float_to_half.restype = C.c_uint16
        L.ora_float_to_half.argtypes = [C.c_float]
        L.ora_half_to_float.restype = C.c_float
        L.ora_half_to_float.argtypes = [C.c_uint16]
        L.ora_srgb_encode.restype = C.c_uint8
        L.ora_srgb_encode.argtypes = [C.c_float]
        L.ora_srgb_decode.restype = C.c_float
        L.ora_srgb_decode.argtypes = [C.c_uint8]
        L.ora_linearise_depth.restype = C.c_float
        L.ora_linearise_depth.argtypes = [C.c_float]
        L.ora_lod_split.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.ora_brdf.argtypes = [C.POINTER(C.c_float)] * 4 + [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.ora_mat4_inverse.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ora_mat4_mul.argtypes = [C.POINTER(C.c_float)] * 3
        L.ora_xform.argtypes = [C.POINTER(C.c_float)] * 3
        L.ora_coverage.restype = C.c_int
        L.ora_coverage.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _fa(values):
    return (C.c_float * len(values))(*[float(v) for v in values])


def make_uniforms(width, height, camera2world_c16, debug=1):
    """camera2world_c16: 16 floats, glm column-major."""
    u = abi.Uniforms()
    lib().ora_make_uniforms(C.byref(u), width, height, _fa(camera2world_c16), debug)
    return u


class OracleScene:
    def __init__(self, packed):
        self.packed = packed
        self.h = lib().ora_scene_create(C.byref(packed.desc))
        if not self.h:
            raise RuntimeError("ora_scene_create failed")

    def close(self):
        if self.h:
            lib().ora_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def texture_level(self, tex, level):
        w, h = C.c_uint32(), C.c_uint32()
        if lib().ora_texture_level(self.h, tex, level, None, C.byref(w), C.byref(h)):
            raise IndexError("no such texture level")
        out = np.zeros((h.value, w.value, 4), np.uint8)
        lib().ora_texture_level(self.h, tex, level, out.ctypes.data, C.byref(w), C.byref(h))
        return out

    def render(self, params, want=("rgba8", "tri_id", "depth_bits", "mip_level", "stencil", "gbuf_normal",
                                   "gbuf_albedo", "shadow_map"), threads=1):
        """threads > 1: the frame is drawn as horizontal bands shared by that many threads (ora_render_bands: vertices
        transformed and the light's depth map drawn once, every band its own rows); the result is the same frame."""
        W, H = params.width, params.height
        S = params.shadow_size or 2048
        bufs = {
            "rgba8": np.zeros((H, W, 4), np.uint8), "tri_id": np.zeros((H, W), np.uint32),
            "depth_bits": np.zeros((H, W), np.uint32), "mip_level": np.zeros((H, W), np.uint8),
            "stencil": np.zeros((H, W), np.uint8), "gbuf_normal": np.zeros((H, W, 4), np.uint16),
            "gbuf_albedo": np.zeros((H, W, 4), np.uint16),
        }
        if "shadow_map" in want:
            bufs["shadow_map"] = np.zeros((S, S), np.float32)
        out = abi.FrameOut()
        for k, v in bufs.items():
            if k in want:
                setattr(out, k, v.ctypes.data)
        if threads <= 1:
            rc = lib().ora_render(self.h, C.byref(params), C.byref(out))
        else:
            lib().ora_set_band_threads(int(threads))
            # more bands than threads: the cost of a band depends on what it shows
            rc = lib().ora_render_bands(self.h, C.byref(params), C.byref(out), max(8, -(-H // (threads * 4))))
        if rc:
            raise RuntimeError("oracle render failed: %d" % rc)
        return {k: v for k, v in bufs.items() if k in want}


def pack_tbn(T, N):
    return int(lib().ora_pack_tbn(_fa(T), _fa(N)))


def decode_tbn(packed):
    m = (C.c_float * 9)()
    lib().ora_decode_tbn(C.c_uint32(packed), m)
    return np.array(list(m), np.float32).reshape(3, 3).T  # [row, col]
