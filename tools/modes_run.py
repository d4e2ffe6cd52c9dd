#!/usr/bin/env python3
"""Renders a few frames of every render mode of the proxy scene (fly-through poses), for kernel-level profiling:

  rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/modes_run.py --width 1920 --height 1080
"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (its HIP runtime has to be the first one initialised in the process)
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--tex-size", type=int, default=256)
a = ap.parse_args()
ctx = renderer.Context(0)
sc = renderer.Scene(ctx, scenes.proxy_scene(detail=1.0, tex_size=a.tex_size))
out = torch.empty((a.height, a.width, 4), dtype=torch.uint8, device="cuda:0")
fo = abi.FrameOut(); fo.rgba8 = out.data_ptr()
modes = [("deferred", dict(deferred=True)), ("forward", dict()), ("mosaic", dict(mosaic=True)), ("mip", dict(debug=2)),
         ("depth", dict(debug=3)), ("deriv", dict(debug=4)), ("overdraw", dict(debug=5)), ("overshade", dict(debug=6))]
for name, kw in modes:
    for i in range(a.frames):
        u = renderer.make_uniforms(a.width, a.height, fly_pose((i * 32) % 512), kw.get("debug", 1))
        sc.submit(abi.make_params(a.width, a.height, u, **kw), fo)
    ctx.wait()
    print(name, "ok", flush=True)
sc.close(); ctx.close()
