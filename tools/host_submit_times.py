import os, sys, time
sys.path.insert(0, os.getcwd())
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes
packed = scenes.proxy_scene(tex_size=64)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
fo = abi.FrameOut()
for (W, H, kw, name) in ((3840, 2160, dict(deferred=True), "4K deferred"), (1920, 1080, dict(deferred=True), "1080p deferred"), (1920, 1080, dict(debug=3), "1080p depth view"),
                         (1280, 720, dict(debug=1, shadow_size=2048), "720p forward+shadow"), (1920, 1080, dict(mosaic=True), "1080p mosaic")):
    ps = [abi.make_params(W, H, renderer.make_uniforms(W, H, fly_pose(i), kw.get("debug", 1)), **kw) for i in range(64)]
    for p in ps[:8]:
        sc.submit(p, fo)
    ctx.wait()
    t0 = time.perf_counter()
    for p in ps:
        sc.submit(p, fo)
    t1 = time.perf_counter()
    ctx.wait()
    t2 = time.perf_counter()
    print("%-22s host submit %.1f us/frame, submit+GPU %.1f us/frame" % (name, (t1 - t0) / 64 * 1e6, (t2 - t0) / 64 * 1e6))
