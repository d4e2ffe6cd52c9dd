#!/usr/bin/env python3
"""Measurement build only (make EXTRA=-DVKR_WG_TIMES): per-tile times of the last k_overshade launch (debug view 6) -
the in-kernel sort of the tile's bin and the ordered walk - by bin length."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--poses", default="0,128")
a = ap.parse_args()
packed = scenes.proxy_scene(tex_size=64)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
L = renderer.lib()
if not hasattr(L, "vkr_debug_overshade_times"):
    sys.exit("library built without -DVKR_WG_TIMES")
L.vkr_debug_overshade_times.argtypes = [C.c_void_p, C.c_size_t]
W, H = a.width, a.height
nt = min(32768, ((W + 31) // 32) * ((H + 31) // 32))
fo = abi.FrameOut()
for pose in [int(x) for x in a.poses.split(",")]:
    p = abi.make_params(W, H, renderer.make_uniforms(W, H, fly_pose(pose), 6), debug=6)
    for _ in range(3):
        sc.submit(p, fo); ctx.wait()
    buf = np.zeros((32768, 8), np.uint64)
    assert L.vkr_debug_overshade_times(buf.ctypes.data, buf.nbytes) == 0
    b = buf[:nt].astype(np.int64)
    t0 = b[:, 0].min()
    sort, walk, n = (b[:, 1] - b[:, 0]) / 100.0, (b[:, 2] - b[:, 1]) / 100.0, b[:, 3]
    print("pose %d: %d tiles, kernel span %.0f us, last start %.0f us; bin entries median %d, p99 %d, max %d; summed sort %.0f us, summed walk %.0f us"
          % (pose, nt, (b[:, 2].max() - t0) / 100.0, (b[:, 0].max() - t0) / 100.0, np.median(n), np.percentile(n, 99), n.max(), sort.sum(), walk.sum()))
    for lo, hi in ((0, 64), (64, 256), (256, 1024), (1024, 4096), (4096, 16384), (16384, 10 ** 9)):
        m = (n >= lo) & (n < hi)
        if m.any():
            print("   bins of %5d..%-6s entries: %5d tiles, sort mean %.1f max %.1f us, walk mean %.1f max %.1f us"
                  % (lo, hi if hi < 10 ** 9 else "", m.sum(), sort[m].mean(), sort[m].max(), walk[m].mean(), walk[m].max()))
    # the ten tiles that take longest to walk: entries, entries walked per wavefront (mean over the 16, most), box hits culled by depth
    for t in np.argsort(-walk)[:10]:
        print("   tile %4d (%2d,%2d): %4d entries, sort %.0f us, walk %.0f us (staging %.0f); walked per wavefront %.0f (most %d), culled by depth %d, %.2f us per walked entry of the busiest wavefront"
              % (t, t % ((W + 31) // 32), t // ((W + 31) // 32), n[t], sort[t], walk[t], b[t, 7] / 100.0, b[t, 4] / 16.0, b[t, 5], b[t, 6], walk[t] / max(1, b[t, 5])))
