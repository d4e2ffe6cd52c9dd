#!/usr/bin/env python3
"""Throughput with K independent contexts (own stream, own scratch, own scene copy) fed round-robin."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes
W, H = 3840, 2160
packed = scenes.proxy_scene(tex_size=256)
for K in (1, 2, 3):
    ctxs = [renderer.Context(0) for _ in range(K)]
    scs = [renderer.Scene(c, packed) for c in ctxs]
    outs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(K)]
    fos = []
    for o in outs:
        fo = abi.FrameOut(); fo.rgba8 = o.data_ptr(); fos.append(fo)
    def frame(i):
        u = renderer.make_uniforms(W, H, fly_pose(i % 512), 1)
        scs[i % K].submit(abi.make_params(W, H, u, deferred=True), fos[i % K])
    for i in range(8): frame(i)
    for c in ctxs: c.wait()
    for c in ctxs:
        c.profile(True); renderer._check(renderer.lib().vkr_profile_enable(c.h, 2))
    t0 = time.perf_counter()
    N = 96
    for i in range(N): frame(8 + i)
    for c in ctxs: c.wait()
    dt = time.perf_counter() - t0
    rep = [c.profile_report().get("k_deferred_light") for c in ctxs]
    print("contexts %d: %.3f ms/frame, %.0f frames/s; light kernel avg %s us" % (K, dt / N * 1e3, N / dt, ["%.1f" % (r[0] / r[1] * 1e3) for r in rep]))
    for s in scs: s.close()
    for c in ctxs: c.close()
