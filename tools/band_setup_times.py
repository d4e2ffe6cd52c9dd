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

# Throughput of one rank's share with one and two frames in flight: consecutive fly-through frames of the same share,
# each followed by the exchange of its rows on its own lane (a communicator of one rank: the rows travel through RCCL
# to this GPU itself), wall clock over 64 frames.  With two lanes a frame's launch chain and its exchange run under
# the other frame's kernels.
import ctypes as C, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import DeviceBuffer
comp = renderer.Compositor(ctx, renderer.composite_unique_id(), 0, 1)
FR = 64
for name, kw, rows in (("whole", {}, H), ("band 3 of 8", dict(rect=band_rect(3, N, W, H)), band_rect(3, N, W, H)[3] - band_rect(3, N, W, H)[1]),
                       ("stripes 3 of 8", dict(stripes=abi.stripes(64, N, 3)), renderer.sortfirst_rows(H, N, 3, 64)[1])):
    res = {}
    for lanes in (1, 2):
        ctx.wait(); ctx.set_frames_in_flight(lanes)
        bufs = [(DeviceBuffer(W * rows * 4), DeviceBuffer(W * rows * 4)) for _ in range(lanes)]
        fos = []
        for l, _ in bufs:
            f = abi.FrameOut(); f.rgba8 = l.ptr; fos.append(f)
        ps = [abi.make_params(W, H, renderer.make_uniforms(W, H, fly_pose(i * 4), 1), deferred=True, **kw) for i in range(FR)]
        def go():
            for i, p in enumerate(ps):
                sc.submit(p, fos[i % lanes])
                # the share's rows as a frame of `rows` rows of its own: every byte goes through ncclSend / ncclRecv
                comp.gather(W, rows, bufs[i % lanes][0].ptr, bufs[i % lanes][1].ptr, root=0)
            ctx.wait()
        go()
        t0 = time.perf_counter(); go(); res[lanes] = (time.perf_counter() - t0) / FR * 1e6
        for l, f in bufs:
            l.free(); f.free()
    print("throughput %-15s one lane %4.0f us per frame, two lanes %4.0f us (x%.2f)" % (name, res[1], res[2], res[1] / res[2]))
ctx.wait(); ctx.set_frames_in_flight(1); comp.close()
