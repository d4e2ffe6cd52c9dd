#!/usr/bin/env python3
"""Per-stage GPU times of one sort-first share of a 4K deferred frame (a band of 1/8 of the rows, or every 8th stripe)
beside the whole frame: what a rank of an eight-GPU sort-first run spends in the geometry passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes
from vulkanrenderer_amd.dist import band_rect

W, H, N = 3840, 2160, 8
packed = scenes.proxy_scene(tex_size=256)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
fo = abi.FrameOut()
for pose in (0, 128, 256):
    u = renderer.make_uniforms(W, H, fly_pose(pose), 1)
    for name, kw in (("whole", {}), ("band 3 of 8", dict(rect=band_rect(3, N, W, H))), ("band 7 of 8", dict(rect=band_rect(7, N, W, H))),
                     ("stripes 3 of 8", dict(stripes=abi.stripes(64, N, 3)))):
        p = abi.make_params(W, H, u, deferred=True, **kw)
        sc.submit(p, fo); ctx.wait()
        ctx.profile(True)
        for _ in range(5):
            sc.submit(p, fo)
        ctx.wait()
        rep = ctx.profile_report()
        ctx.profile(False)
        print("pose %3d %-15s total %4.0f us: " % (pose, name, sum(v[0] / 5 for v in rep.values()) * 1e3) +
              "  ".join("%s=%.0f" % (k.replace("k_", ""), v[0] / v[1] * 1e3) for k, v in rep.items()))
