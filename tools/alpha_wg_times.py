#!/usr/bin/env python3
"""Measurement build only (make EXTRA=-DVKR_WG_TIMES): where the workgroups of the last k_raster_alpha launch spent their time."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--poses", default="0,40,128")
a = ap.parse_args()
packed = scenes.proxy_scene(tex_size=256)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
W, H = a.width, a.height
L = renderer.lib()
L.vkr_debug_alpha_times.argtypes = [C.c_void_p, C.c_size_t]
fo = abi.FrameOut()
names = ("header + seed", "staging", "rank + lists", "walk", "store")
for pose in [int(x) for x in a.poses.split(",")]:
    p = abi.make_params(W, H, renderer.make_uniforms(W, H, fly_pose(pose), 1), deferred=True)
    zero = np.zeros((65536, 8), np.uint64)
    for _ in range(2):
        sc.submit(p, fo); ctx.wait()
    buf = np.zeros((65536, 8), np.uint64)
    assert L.vkr_debug_alpha_times(buf.ctypes.data, buf.nbytes) == 0
    nq = min(65536, ((W + 31) // 32) * ((H + 31) // 32) * 4 + 3584)
    b = buf[:nq].astype(np.int64)
    act = b[:, 6] > 0   # workgroups whose tile has entries in the alpha-tested / many-tile sections
    ph = b[act, :5] / 100.0
    tot = ph.sum(axis=1)
    t0 = b[act, 5].min()
    print("pose %d: %d active workgroups of %d, summed time %.0f us, median %.1f, p90 %.1f, max %.1f us; span %.1f us" % (
        pose, act.sum(), nq, tot.sum(), np.median(tot), np.percentile(tot, 90), tot.max(), ((b[act, 5] - t0) / 100.0 + tot).max()))
    print("   share of the summed time: " + "  ".join("%s %.0f%%" % (names[k], 100 * ph[:, k].sum() / tot.sum()) for k in range(5)))
    for lo, hi in ((1, 16), (16, 64), (64, 256), (256, 10 ** 9)):
        m = (b[act, 6] >= lo) & (b[act, 6] < hi)
        if m.any():
            print("   %4d..%-5s entries: %5d workgroups, mean %.1f us: " % (lo, hi if hi < 10 ** 9 else "", m.sum(), tot[m].mean()) +
                  "  ".join("%s %.1f" % (names[k], ph[m, k].mean()) for k in range(5)))
