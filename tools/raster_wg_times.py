#!/usr/bin/env python3
"""Measurement build only (make EXTRA=-DVKR_WG_TIMES): per-workgroup times of the last k_raster launch."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1280); ap.add_argument("--height", type=int, default=720)
ap.add_argument("--shadow", type=int, default=2048); ap.add_argument("--pose", type=int, default=-1)
ap.add_argument("--deferred", action="store_true", help="a deferred frame: the camera pass alone on the GPU (in a forward frame the light's pass runs beside it)")
a = ap.parse_args()
IDENT = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
packed = scenes.proxy_scene(tex_size=64)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
W, H = a.width, a.height
u = renderer.make_uniforms(W, H, IDENT if a.pose < 0 else fly_pose(a.pose), 1)
p = abi.make_params(W, H, u, debug=1, shadow_size=a.shadow, deferred=a.deferred)
fo = abi.FrameOut()
for _ in range(3):
    sc.submit(p, fo); ctx.wait()
L = renderer.lib()
L.vkr_debug_wg_times.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
for mode, name, nt in ((0, "camera pass", ((W + 31) // 32) * ((H + 31) // 32)), (1, "light pass", ((a.shadow + 31) // 32) ** 2))[:1 if a.deferred else 2]:
    buf = np.zeros((32768, 8), np.uint64)
    assert L.vkr_debug_wg_times(mode, buf.ctypes.data, buf.nbytes) == 0
    nt = min(nt, 32768)
    b = buf[:nt].astype(np.int64)
    t0 = b[:, 0].min()
    T = (b[:, :6] - t0) / 100.0   # microseconds
    dur = T[:, 5] - T[:, 0]
    n = b[:, 6] + b[:, 7]
    print("%s: %d workgroups, kernel span %.1f us; workgroup time median %.1f, p90 %.1f, p99 %.1f, max %.1f us; last start %.1f us" % (
        name, nt, T[:, 5].max(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), T[:, 0].max()))
    # how many workgroups are resident: started within the first 2 us, and at the quartiles of the span
    ev = np.sort(np.concatenate([np.stack([T[:, 0], np.ones(nt)], 1), np.stack([T[:, 5], -np.ones(nt)], 1)]), axis=0) if False else None
    starts, ends = np.sort(T[:, 0]), np.sort(T[:, 5])
    span = T[:, 5].max()
    res = [int(np.searchsorted(starts, q * span, side="right") - np.searchsorted(ends, q * span, side="right")) for q in (0.1, 0.25, 0.5, 0.75, 0.9)]
    print("   resident workgroups at 10/25/50/75/90 %% of the span: %s; started within 2 us: %d, within 5 us: %d; summed workgroup time %.0f us = %.0f resident on average"
          % (res, int((starts <= 2.0).sum()), int((starts <= 5.0).sum()), dur.sum(), dur.sum() / span))
    names = ("init+barrier", "header loads", "entries", "barrier", "store")
    for lo, hi in ((0, 32), (32, 128), (128, 512), (512, 2048), (2048, 10 ** 9)):
        m = (n >= lo) & (n < hi)
        if m.any():
            print("   bins of %5d..%-6s entries: %5d workgroups, mean %.1f us: " % (lo, hi if hi < 10 ** 9 else "", m.sum(), dur[m].mean()) +
                  "  ".join("%s %.1f" % (names[k], (T[m, k + 1] - T[m, k]).mean()) for k in range(5)))
