"""Thin ctypes binding of libvkr_hip.so (include/vkr_renderer.h).

There is no fallback path: if the HIP library is missing or fails to load the
import of this module's users fails loudly.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# VKR_HIP_LIB: another build of the same library (kernel experiments); there is still no fallback if it is missing
LIB_PATH = os.environ.get("VKR_HIP_LIB") or os.path.join(_HERE, "lib", "libvkr_hip.so")
_lib = None

EXPORTS = [
    "vkr_last_error", "vkr_context_create", "vkr_context_destroy", "vkr_scene_create", "vkr_scene_load",
    "vkr_scene_destroy", "vkr_scene_stats", "vkr_scene_read_texture_level", "vkr_render", "vkr_wait",
    "vkr_update_scene_uniforms", "vkr_update_depth_mvp_uniforms", "vkr_default_lights", "vkr_make_uniforms",
    "vkr_profile_enable", "vkr_profile_reset", "vkr_stage_count", "vkr_stage_name", "vkr_profile_get",
    "vkr_context_stream", "vkr_decode_png_file", "vkr_free", "vkr_write_png_file",
    "vkr_user_state_init", "vkr_update_user_state", "vkr_context_set_frames_in_flight",
    "vkr_wait_next_lane", "vkr_user_state_key", "vkr_user_state_button", "vkr_user_state_motion",
    "vkr_frame_params_set_modes", "vkr_render_static", "vkr_context_device", "vkr_sortfirst_rows",
    "vkr_composite_unique_id", "vkr_compositor_create", "vkr_compositor_destroy", "vkr_composite_gather",
    "vkr_device_malloc", "vkr_device_free", "vkr_device_copy", "vkr_composite_gather_rows",
    "vkr_context_frame_stream", "vkr_context_frame_lane",
]


class VkrError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VkrError("HIP library not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.vkr_last_error.restype = C.c_char_p
        L.vkr_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.vkr_context_destroy.argtypes = [C.c_void_p]
        L.vkr_context_destroy.restype = None
        L.vkr_scene_create.argtypes = [C.c_void_p, C.POINTER(abi.SceneDesc), C.POINTER(C.c_void_p)]
        L.vkr_scene_load.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.vkr_scene_destroy.argtypes = [C.c_void_p]
        L.vkr_scene_destroy.restype = None
        L.vkr_scene_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.vkr_scene_read_texture_level.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.vkr_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.FrameParams), C.POINTER(abi.FrameOut)]
        L.vkr_render_static.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.FrameParams), C.POINTER(abi.FrameOut)]
        L.vkr_wait.argtypes = [C.c_void_p]
        L.vkr_update_scene_uniforms.argtypes = [C.POINTER(abi.SceneUniform), C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        L.vkr_update_scene_uniforms.restype = None
        L.vkr_update_depth_mvp_uniforms.argtypes = [C.POINTER(abi.DepthMVP), C.POINTER(C.c_float)]
        L.vkr_update_depth_mvp_uniforms.restype = None
        L.vkr_default_lights.argtypes = [C.POINTER(abi.LightUniform), C.POINTER(abi.MultipleLights)]
        L.vkr_default_lights.restype = None
        L.vkr_make_uniforms.argtypes = [C.POINTER(abi.Uniforms), C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.c_int32]
        L.vkr_make_uniforms.restype = None
        L.vkr_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.vkr_profile_reset.argtypes = [C.c_void_p]
        L.vkr_stage_name.restype = C.c_char_p
        L.vkr_stage_name.argtypes = [C.c_int]
        L.vkr_profile_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.vkr_decode_png_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.vkr_write_png_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.vkr_user_state_init.argtypes = [C.POINTER(abi.UserState)]
        L.vkr_user_state_init.restype = None
        L.vkr_update_user_state.argtypes = [C.POINTER(abi.UserState), C.c_float]
        L.vkr_update_user_state.restype = None
        L.vkr_user_state_key.argtypes = [C.POINTER(abi.UserState), C.c_int, C.c_int]
        L.vkr_user_state_button.argtypes = [C.POINTER(abi.UserState), C.c_int, C.c_int]
        L.vkr_user_state_button.restype = None
        L.vkr_user_state_motion.argtypes = [C.POINTER(abi.UserState), C.c_double, C.c_double]
        L.vkr_user_state_motion.restype = None
        L.vkr_frame_params_set_modes.argtypes = [C.POINTER(abi.FrameParams), C.POINTER(abi.UserState)]
        L.vkr_frame_params_set_modes.restype = None
        L.vkr_free.argtypes = [C.c_void_p]
        L.vkr_free.restype = None
        L.vkr_context_stream.restype = C.c_void_p
        L.vkr_context_stream.argtypes = [C.c_void_p]
        L.vkr_context_frame_stream.restype = C.c_void_p
        L.vkr_context_frame_stream.argtypes = [C.c_void_p]
        L.vkr_context_frame_lane.argtypes = [C.c_void_p]
        L.vkr_context_set_frames_in_flight.argtypes = [C.c_void_p, C.c_uint32]
        L.vkr_context_device.argtypes = [C.c_void_p]
        L.vkr_sortfirst_rows.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
        L.vkr_composite_unique_id.argtypes = [C.c_char_p]
        L.vkr_compositor_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.vkr_compositor_destroy.argtypes = [C.c_void_p]
        L.vkr_compositor_destroy.restype = None
        L.vkr_composite_gather.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.vkr_composite_gather_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.vkr_wait_next_lane.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc:
        raise VkrError("vkr error %d: %s" % (rc, lib().vkr_last_error().decode("utf-8", "replace")))


def make_uniforms(width, height, camera2world_c16, debug=1):
    """camera2world_c16: 16 floats, glm column-major (what UserState::camera2world holds)."""
    u = abi.Uniforms()
    lib().vkr_make_uniforms(C.byref(u), width, height, (C.c_float * 16)(*[float(v) for v in camera2world_c16]), debug)
    return u


def replay_inputs(trace, state=None):
    """Turns a recorded input trace into the stream of camera poses the
    reference's frame loop would have rendered (update_user_state,
    main/main.cpp:3643-3679, called once per frame at main/main.cpp:1302).
    trace: iterable of (held_keys, mouseX, mouseY, dt) with held_keys a set of
    names from abi.INPUT_KEYS or a bit mask in that order.  Returns float32[n, 16]
    column-major camera2world matrices (one per step) and the final state."""
    L = lib()
    if state is None:
        state = abi.UserState()
        L.vkr_user_state_init(C.byref(state))
    poses = []
    for keys, mx, my, dt in trace:
        if not isinstance(keys, int):
            keys = sum(1 << abi.INPUT_KEYS.index(k) for k in keys)
        for i in range(9):
            state.inputMap[i] = (keys >> i) & 1
        state.mouseX, state.mouseY = mx, my
        L.vkr_update_user_state(C.byref(state), dt)
        poses.append(np.array(state.camera2world, np.float32))
    return np.array(poses, np.float32).reshape(-1, 16), state


# GLFW's values (GLFW/glfw3.h), as the reference's callbacks receive them
GLFW_RELEASE, GLFW_PRESS, GLFW_REPEAT = 0, 1, 2
GLFW_MOUSE_BUTTON_RIGHT = 1
GLFW_KEY_ESCAPE, GLFW_KEY_LEFT_SHIFT, GLFW_KEY_LEFT_CONTROL, GLFW_KEY_RIGHT_SHIFT, GLFW_KEY_RIGHT_CONTROL = 256, 340, 341, 344, 345


def replay_events(events, state=None):
    """Replays a recorded GLFW event stream through the reference's three callbacks and its per-frame update
    (main/main.cpp:1422-1537, 3643-3679).  events: ("key", key, action) with key an int or a one-character string,
    ("button", button, action), ("motion", x, y), ("frame", dt).  Returns one (camera2world float32[16], dict of the
    three mode fields) per "frame" event, whether escape was pressed, and the final state."""
    L = lib()
    if state is None:
        state = abi.UserState()
        L.vkr_user_state_init(C.byref(state))
    frames, closed = [], False
    for ev in events:
        kind = ev[0]
        if kind == "key":
            key = ord(ev[1].upper()) if isinstance(ev[1], str) else int(ev[1])
            closed = bool(L.vkr_user_state_key(C.byref(state), key, int(ev[2]))) or closed
        elif kind == "button":
            L.vkr_user_state_button(C.byref(state), int(ev[1]), int(ev[2]))
        elif kind == "motion":
            L.vkr_user_state_motion(C.byref(state), float(ev[1]), float(ev[2]))
        elif kind == "frame":
            L.vkr_update_user_state(C.byref(state), float(ev[1]))
            frames.append((np.array(state.camera2world, np.float32),
                           dict(debug=state.debugVisualisation, mosaic=bool(state.mosaicEffect), deferred=bool(state.deferredShading))))
        else:
            raise ValueError("unknown event %r" % (kind,))
    return frames, closed, state


def decode_png(path):
    """uint8[h, w, 4], bottom row first, as the reference's texture loader sees it."""
    p, w, h = C.POINTER(C.c_uint8)(), C.c_uint32(), C.c_uint32()
    _check(lib().vkr_decode_png_file(path.encode(), C.byref(p), C.byref(w), C.byref(h)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib().vkr_free(p)


def write_png(path, rgba8, flip=False):
    """rgba8: uint8[h, w, 4]; row 0 = top unless flip."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    _check(lib().vkr_write_png_file(path.encode(), a.ctypes.data, a.shape[1], a.shape[0], int(flip)))


SIDE = ("rgba8", "tri_id", "depth_bits", "mip_level", "stencil", "gbuf_normal", "gbuf_albedo", "shadow_map")


def sortfirst_rows(height, world, rank, stripe_height=0):
    """(first_row, row_count) of `rank`'s share of a sort-first frame: a band (stripe_height 0) or its interleaved
    stripes (row_count rows stored back to back, the first stripe starting at first_row)."""
    a, b = C.c_uint32(), C.c_uint32()
    _check(lib().vkr_sortfirst_rows(height, world, rank, stripe_height, C.byref(a), C.byref(b)))
    return a.value, b.value


def composite_unique_id():
    """The blob one rank creates and every rank passes to Compositor (an ncclUniqueId)."""
    buf = C.create_string_buffer(128)
    _check(lib().vkr_composite_unique_id(buf))
    return buf.raw


class Compositor:
    """vkr_compositor: the RCCL exchange of sort-first rows on the context's stream."""

    def __init__(self, ctx, unique_id, rank, world):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.h = C.c_void_p()
        _check(lib().vkr_compositor_create(ctx.h, unique_id, rank, world, C.byref(self.h)))

    def gather(self, width, height, local_ptr, frame_ptr, root=0, stripe_height=0):
        _check(lib().vkr_composite_gather(self.h, width, height, stripe_height, local_ptr, frame_ptr, root))

    def gather_rows(self, row_bytes, height, local_ptr, frame_ptr, root=0, stripe_height=0):
        """The same exchange for a side buffer: rows of `row_bytes` bytes (tri_id: 4 per pixel, stencil: 1, ...)."""
        _check(lib().vkr_composite_gather_rows(self.h, row_bytes, height, stripe_height, local_ptr, frame_ptr, root))

    def close(self):
        if self.h:
            lib().vkr_compositor_destroy(self.h)
            self.h = C.c_void_p()


# Handles still open when the interpreter exits are released here, scenes first: a destructor that runs during
# interpreter teardown would call into a HIP runtime that may already be gone.
_live_scenes = weakref.WeakSet()
_live_contexts = weakref.WeakSet()


@atexit.register
def _close_live_handles():
    for s in list(_live_scenes):
        s.close()
    for c in list(_live_contexts):
        c.close()


class Context:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _check(lib().vkr_context_create(device, C.byref(self.h)))
        self.device = device
        self._scenes = weakref.WeakSet()
        _live_contexts.add(self)

    def close(self):
        if self.h:
            for s in list(self._scenes):  # a scene must not outlive the context whose stream and tables it uses
                s.close()
            lib().vkr_context_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def wait(self):
        _check(lib().vkr_wait(self.h))

    def wait_next_lane(self):
        """Waits until the lane the next frame goes to is idle: the buffers handed over n frames ago are complete."""
        _check(lib().vkr_wait_next_lane(self.h))

    def set_frames_in_flight(self, n):
        """n lanes (stream + scratch + frame buffers each); consecutive frames go to consecutive lanes and may
        overlap on the GPU, so n consecutive frames need n distinct output buffers."""
        _check(lib().vkr_context_set_frames_in_flight(self.h, int(n)))

    @property
    def stream(self):
        return lib().vkr_context_stream(self.h)

    def profile(self, on=True):
        _check(lib().vkr_profile_enable(self.h, int(on)))
        _check(lib().vkr_profile_reset(self.h))

    def profile_report(self):
        """{stage: (total_ms, launches)} for stages that ran."""
        out = {}
        for s in range(lib().vkr_stage_count()):
            ms, n = C.c_double(), C.c_uint64()
            _check(lib().vkr_profile_get(self.h, s, C.byref(ms), C.byref(n)))
            if n.value:
                out[lib().vkr_stage_name(s).decode()] = (ms.value, n.value)
        return out


class Scene:
    def __init__(self, ctx, packed=None, path=None):
        self.ctx = ctx
        self.h = C.c_void_p()
        if packed is not None:
            _check(lib().vkr_scene_create(ctx.h, C.byref(packed.desc), C.byref(self.h)))
        else:
            _check(lib().vkr_scene_load(ctx.h, path.encode(), C.byref(self.h)))
        ctx._scenes.add(self)
        _live_scenes.add(self)

    def close(self):
        if self.h:
            lib().vkr_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        t, v, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().vkr_scene_stats(self.h, C.byref(t), C.byref(v), C.byref(b)))
        return dict(triangles=t.value, vertices=v.value, texture_bytes=b.value)

    def texture_level(self, tex, level):
        w, h = C.c_uint32(), C.c_uint32()
        _check(lib().vkr_scene_read_texture_level(self.h, tex, level, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value, 4), np.uint8)
        _check(lib().vkr_scene_read_texture_level(self.h, tex, level, out.ctypes.data, C.byref(w), C.byref(h)))
        return out

    def submit(self, params, out):
        """Asynchronous vkr_render with a caller-built FrameOut (host or device pointers)."""
        _check(lib().vkr_render(self.ctx.h, self.h, C.byref(params), C.byref(out)))

    def submit_static(self, params, out):
        """vkr_render_static: the allocation-free variant (device buffers from the library's static scratch array)."""
        _check(lib().vkr_render_static(self.ctx.h, self.h, C.byref(params), C.byref(out)))

    def render(self, params, want=SIDE, static=False):
        """Renders one frame into fresh numpy arrays and waits for it."""
        W, H = params.width, params.height
        x0, y0, x1, y1 = params.rect_x0, params.rect_y0, params.rect_x1, params.rect_y1
        if (x0, y0, x1, y1) == (0, 0, 0, 0):
            x1, y1 = W, H
        rw, rh = x1 - x0, y1 - y0
        if params.stripes and ((params.stripes >> 16) & 0xFF) > max(1, params.stripes >> 24) and (params.stripes & 0xFFFF):
            rh = sortfirst_rows(H, (params.stripes >> 16) & 0xFF, params.stripes >> 24, params.stripes & 0xFFFF)[1]
        S = params.shadow_size or 2048
        bufs = {
            "rgba8": np.zeros((rh, rw, 4), np.uint8), "tri_id": np.full((rh, rw), 0xFFFFFFFF, np.uint32),
            "depth_bits": np.zeros((rh, rw), np.uint32), "mip_level": np.full((rh, rw), 0xFF, np.uint8),
            "stencil": np.zeros((rh, rw), np.uint8), "gbuf_normal": np.zeros((rh, rw, 4), np.uint16),
            "gbuf_albedo": np.zeros((rh, rw, 4), np.uint16),
        }
        if "shadow_map" in want:
            bufs["shadow_map"] = np.zeros((S, S), np.float32)
        out = abi.FrameOut()
        for k, v in bufs.items():
            if k in want:
                setattr(out, k, v.ctypes.data)
        (self.submit_static if static else self.submit)(params, out)
        self.ctx.wait()
        return {k: v for k, v in bufs.items() if k in want}
