#!/usr/bin/env python3
"""Per-stage GPU times (HIP events) for a few fly-through poses and scene variants."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--tex-size", type=int, default=256)
ap.add_argument("--no-foliage", action="store_true")
ap.add_argument("--mode", default="deferred")
ap.add_argument("--poses", default="0,64,128,256,400,511")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--fused", action="store_true", help="deferred: resolve and lighting in one kernel")
a = ap.parse_args()
packed = scenes.proxy_scene(tex_size=a.tex_size, foliage=not a.no_foliage)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
W, H = a.width, a.height
kw = dict(flags=abi.FRAME_FUSED_LIGHTING if a.fused else 0, deferred=a.mode == "deferred", mosaic=a.mode == "mosaic", debug={"deferred": 1, "forward": 1, "mosaic": 1}.get(a.mode, 1) if not a.mode.isdigit() else int(a.mode))
fo = abi.FrameOut()  # no outputs requested: frames stay in the context's device buffers
for pi in [int(x) for x in a.poses.split(",")]:
    u = renderer.make_uniforms(W, H, fly_pose(pi), kw["debug"])
    p = abi.make_params(W, H, u, **kw)
    sc.submit(p, fo); ctx.wait()
    ctx.profile(True)
    for _ in range(a.reps):
        sc.submit(p, fo)
    ctx.wait()
    rep = ctx.profile_report()
    tot = sum(v[0] / a.reps for v in rep.values())
    print("pose %3d total %.0f us: " % (pi, tot * 1e3) + "  ".join("%s=%.0f" % (k.replace("k_", ""), v[0] / v[1] * 1e3) for k, v in rep.items()))
