"""vkr_render_static, the allocation-free variant of vkr_render: same frames, bit for bit, with every device buffer cut
from the library's static scratch array.  Called through the C ABI (ctypes), two frames in a row without a fence in
between, the second smaller than the first, so that the second runs on an array that still holds the first one's
contents."""
import ctypes as C

import numpy as np
import pytest

from conftest import IDENTITY, pose_matrix

pytestmark = pytest.mark.gpu


def _params(W, H, pose, **kw):
    from vulkanrenderer_amd import abi, renderer
    shadow = kw.pop("shadow", 512)
    return abi.make_params(W, H, renderer.make_uniforms(W, H, pose, kw.get("debug", 1)), shadow_size=shadow, **kw)


def _outputs(p):
    from vulkanrenderer_amd import abi
    H, W, S = p.height, p.width, p.shadow_size or 2048
    bufs = {"rgba8": np.zeros((H, W, 4), np.uint8), "tri_id": np.full((H, W), 0xFFFFFFFF, np.uint32), "depth_bits": np.zeros((H, W), np.uint32),
            "mip_level": np.full((H, W), 0xFF, np.uint8), "stencil": np.zeros((H, W), np.uint8), "gbuf_normal": np.zeros((H, W, 4), np.uint16),
            "gbuf_albedo": np.zeros((H, W, 4), np.uint16), "shadow_map": np.zeros((S, S), np.float32)}
    fo = abi.FrameOut()
    for k, v in bufs.items():
        setattr(fo, k, v.ctypes.data)
    return bufs, fo


def test_static_entry_point_equals_vkr_render():
    from vulkanrenderer_amd import abi, renderer, scenes
    L = renderer.lib()
    packed = scenes.proxy_scene(detail=0.25, tex_size=64)
    ctx = renderer.Context(0)
    sc = renderer.Scene(ctx, packed)
    try:
        big = _params(1280, 720, pose_matrix(0.0, -1.0, -20.0, yaw=0.2), deferred=True)
        small = _params(320, 180, IDENTITY)                       # forward + light pass, a smaller frame
        pairs = [(big, small), (_params(960, 540, pose_matrix(0.0, -1.0, -30.0), debug=6), _params(200, 120, IDENTITY, mosaic=True)),
                 (_params(640, 360, IDENTITY, debug=2), _params(128, 72, IDENTITY, deferred=True, flags=abi.FRAME_FUSED_LIGHTING | abi.FRAME_SHADOW_PASS))]
        for first, second in pairs:
            ref = [sc.render(p) for p in (first, second)]             # vkr_render, the existing entry point
            outs = [_outputs(p) for p in (first, second)]
            for p, (_, fo) in zip((first, second), outs):              # two calls in a row, no fence between them
                rc = L.vkr_render_static(ctx.h, sc.h, C.byref(p), C.byref(fo))
                assert rc == 0, L.vkr_last_error()
            assert L.vkr_wait(ctx.h) == 0, L.vkr_last_error()
            for r, (bufs, _) in zip(ref, outs):
                for k, v in r.items():
                    assert np.array_equal(v, bufs[k]), "%s differs between vkr_render and vkr_render_static" % k
        # a frame that does not fit in the array is refused with a clear error, and the next call still works
        huge = _params(3840, 2160, IDENTITY, deferred=True)
        bufs, fo = _outputs(huge)
        assert L.vkr_render_static(ctx.h, sc.h, C.byref(huge), C.byref(fo)) == 5     # VKR_E_LIMIT
        assert b"static scratch" in L.vkr_last_error()
        again = sc.render(small, static=True)
        for k, v in sc.render(small).items():
            assert np.array_equal(v, again[k]), k
    finally:
        sc.close(); ctx.close()


def test_the_static_array_has_one_owner_per_gpu():
    """Two contexts of one GPU share the static scratch array: while one has allocation-free frames pending the other
    is refused (VKR_E_LIMIT) instead of overwriting them, and the fence of the first hands the array over."""
    from vulkanrenderer_amd import renderer, scenes
    L = renderer.lib()
    packed = scenes.micro_scene()
    a, b = renderer.Context(0), renderer.Context(0)
    sa, sb = renderer.Scene(a, packed), renderer.Scene(b, packed)
    try:
        p = _params(256, 144, IDENTITY, deferred=True)
        ref = sa.render(p)
        bufs_a, fo_a = _outputs(p)
        bufs_b, fo_b = _outputs(p)
        assert L.vkr_render_static(a.h, sa.h, C.byref(p), C.byref(fo_a)) == 0, L.vkr_last_error()
        assert L.vkr_render_static(b.h, sb.h, C.byref(p), C.byref(fo_b)) == 5          # VKR_E_LIMIT: a's frame is pending
        assert b"another context" in L.vkr_last_error()
        assert L.vkr_wait(a.h) == 0, L.vkr_last_error()
        for k, v in ref.items():
            assert np.array_equal(v, bufs_a[k]), k
        assert L.vkr_render_static(b.h, sb.h, C.byref(p), C.byref(fo_b)) == 0, L.vkr_last_error()
        assert L.vkr_render_static(a.h, sa.h, C.byref(p), C.byref(fo_a)) == 5          # and the other way round
        assert L.vkr_wait(b.h) == 0, L.vkr_last_error()
        for k, v in ref.items():
            assert np.array_equal(v, bufs_b[k]), k
        # a context destroyed with frames pending gives the array back
        assert L.vkr_render_static(a.h, sa.h, C.byref(p), C.byref(fo_a)) == 0, L.vkr_last_error()
        sa.close(); a.close(); sa = a = None
        assert L.vkr_render_static(b.h, sb.h, C.byref(p), C.byref(fo_b)) == 0, L.vkr_last_error()
        assert L.vkr_wait(b.h) == 0, L.vkr_last_error()
    finally:
        for x in (sa, sb, a, b):
            if x is not None:
                x.close()
