"""A few hundred frames of every kind in random order on one context - modes, extents, bands, stripes, shadow sizes, fused
lighting, the allocation-free entry point, one and two frames in flight - and every repetition of a frame must give the
bytes its first rendering gave: the per-pass counter sets, the bins that grow and shrink, the lanes and the helper
stream leave nothing behind from one frame to the next."""
import hashlib

import numpy as np
import pytest

from conftest import IDENTITY, pose_matrix

pytestmark = pytest.mark.gpu


def test_random_frame_sequence_is_repeatable():
    from vulkanrenderer_amd import abi, renderer, scenes
    from vulkanrenderer_amd.dist import band_rect
    rng = np.random.default_rng(11)
    ctx = renderer.Context(0)
    sc = renderer.Scene(ctx, scenes.proxy_scene(detail=0.12, tex_size=32))
    poses = [IDENTITY, pose_matrix(0.0, -1.0, -30.0), pose_matrix(0.5, -2.0, -60.0, yaw=2.5), pose_matrix(0.0, 0.5, -10.0, yaw=-0.6, pitch=0.2)]
    sizes = [(320, 180), (640, 360), (97, 61), (1280, 720), (256, 256)]
    kinds = [dict(deferred=True), dict(deferred=True, flags=abi.FRAME_FUSED_LIGHTING), dict(deferred=True, flags=abi.FRAME_SHADOW_PASS, shadow_size=256),
             dict(debug=1, shadow_size=512), dict(debug=1, shadow_size=128), dict(mosaic=True, shadow_size=256), dict(debug=2), dict(debug=3),
             dict(debug=4), dict(debug=5), dict(debug=6)]
    want = ("rgba8", "tri_id", "depth_bits", "stencil", "mip_level")
    seen = {}
    repeats = 0
    try:
        for step in range(900):
            if step == 450:
                ctx.wait(); ctx.set_frames_in_flight(2)   # the second half with two lanes
            W, H = sizes[int(rng.integers(len(sizes)))]
            ki, pi = int(rng.integers(len(kinds))), int(rng.integers(len(poses)))
            kw = dict(kinds[ki])
            part = int(rng.integers(4))   # 0 whole, 1 band, 2 stripes, 3 allocation-free
            if part == 1:
                kw["rect"] = band_rect(int(rng.integers(3)), 3, W, H)
            elif part == 2 and not kw.get("mosaic"):
                kw["stripes"] = abi.stripes(32, 3, int(rng.integers(3)))
            static = part == 3 and W * H <= 640 * 360
            p = abi.make_params(W, H, renderer.make_uniforms(W, H, poses[pi], kw.get("debug", 1)), **kw)
            out = sc.render(p, want=want, static=static)
            h = hashlib.sha256()
            for k in want:
                h.update(np.ascontiguousarray(out[k]).tobytes())
            key = (W, H, ki, pi, part if part in (1, 2) else 0, kw.get("rect"), kw.get("stripes"))
            if key in seen:
                repeats += 1
                assert seen[key] == h.hexdigest(), "step %d: frame %r changed between two renderings" % (step, key)
            else:
                seen[key] = h.hexdigest()
        assert repeats >= 300
    finally:
        ctx.wait(); sc.close(); ctx.close()
