#!/usr/bin/env python3
"""Run under `rocprofv3 --hip-trace` to list the HIP API calls vkr_render_static makes: after set-up (context, scene, one
vkr_render frame) the script calls hipDeviceSynchronize twice as a marker, then renders allocation-free frames (device
output buffers) and calls the marker again; tools/static_trace_summary.py reads the trace between the markers."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes
hip = C.CDLL("libamdhip64.so")
packed = scenes.proxy_scene(detail=0.25, tex_size=64)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
W, H = 1280, 720
img = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
fo = abi.FrameOut(); fo.rgba8 = img.data_ptr()
modes = [dict(deferred=True), dict(), dict(debug=6), dict(mosaic=True), dict(debug=2)]
for kw in modes:   # set-up and a warm vkr_render frame per mode
    sc.submit(abi.make_params(W, H, renderer.make_uniforms(W, H, fly_pose(3), kw.get("debug", 1)), **kw), fo)
ctx.wait()
hip.hipDeviceSynchronize(); hip.hipDeviceSynchronize()
for i in range(10):
    kw = modes[i % len(modes)]
    sc.submit_static(abi.make_params(W, H, renderer.make_uniforms(W, H, fly_pose(i), kw.get("debug", 1)), **kw), fo)
ctx.wait()
hip.hipDeviceSynchronize(); hip.hipDeviceSynchronize()
sc.close(); ctx.close()
