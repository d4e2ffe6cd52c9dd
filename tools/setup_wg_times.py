#!/usr/bin/env python3
"""Measurement build only (make EXTRA=-DVKR_WG_TIMES): per-workgroup phase times of the last k_setup launch of a
deferred frame (the camera pass)."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--pose", type=int, default=0)
a = ap.parse_args()
packed = scenes.proxy_scene(tex_size=64)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
W, H = a.width, a.height
u = renderer.make_uniforms(W, H, fly_pose(a.pose), 1)
p = abi.make_params(W, H, u, deferred=True)
fo = abi.FrameOut()
for _ in range(3):
    sc.submit(p, fo); ctx.wait()
L = renderer.lib()
L.vkr_debug_setup_times.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((4096, 8), np.uint64)
assert L.vkr_debug_setup_times(buf.ctypes.data, buf.nbytes) == 0
ntri = sum(len(m["indices"]) // 3 for m in packed.meshes)
nwg = min((ntri + 255) // 256, 4096)
b = buf[:nwg].astype(np.int64)
t0 = b[:, 0].min()
T = (b[:, :6] - t0) / 100.0
dur = T[:, 5] - T[:, 0]
print("k_setup: %d workgroups, span %.1f us; workgroup time median %.1f, p90 %.1f, p99 %.1f, max %.1f us; last start %.1f us; started within 2 us: %d"
      % (nwg, T[:, 5].max(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), T[:, 0].max(), int((T[:, 0] <= 2.0).sum())))
names = ("load+transform+prepare", "queue+recTiles+stage-out", "LDS table", "flush atomics", "finish")
for lo, hi in ((0, 1), (1, 64), (64, 192), (192, 257)):
    m = (b[:, 6] >= lo) & (b[:, 6] < hi)
    if m.any():
        print("   %3d..%-3d emitted: %5d workgroups, mean %.1f us (tiles in table %.0f): " % (lo, hi - 1, m.sum(), dur[m].mean(), b[m, 7].mean()) +
              "  ".join("%s %.1f" % (names[k], (T[m, k + 1] - T[m, k]).mean()) for k in range(5)))
ls = np.sort(T[T[:, 0] > 2.0, 0])
print("   starts after 2 us (%d): " % len(ls) + " ".join("%.1f" % v for v in ls[:: max(1, len(ls) // 24)]))
ends = np.sort(T[:, 5])
print("   ends: first %.1f, 100th %.1f, 300th %.1f, 1000th %.1f us" % (ends[0], ends[99], ends[299], ends[999]))
for x in (0, 3):
    idx = np.arange(nwg)[np.arange(nwg) % 8 == x]           # workgroup i runs on XCD i % 8
    first, rest = idx[idx < 2048], idx[idx >= 2048]
    print("   XCD %d: its first ends %s; its late workgroups start %s" % (x, " ".join("%.1f" % v for v in np.sort(T[first, 5])[:14]),
                                                                      " ".join("%.1f" % v for v in np.sort(T[rest, 0]))))
late = np.argsort(T[:, 5])[-8:]
print("   the eight last to finish: " + ", ".join("wg %d start %.1f end %.1f emitted %d tiles %d" % (i, T[i, 0], T[i, 5], b[i, 6], b[i, 7]) for i in late))
