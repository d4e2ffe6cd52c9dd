#!/usr/bin/env python3
"""What the opaque camera pass of a 4K deferred frame walks, from a measurement build of the library
(make -C vulkanrenderer_amd/csrc EXTRA=-DVKR_RASTER_STATS): sub-triangles taken one per lane group and by whole
wavefronts, the pixels of their boxes, and the steps the wavefront path spends on them."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import fly_pose
from vulkanrenderer_amd import abi, renderer, scenes

packed = scenes.proxy_scene(tex_size=64)
ctx = renderer.Context(0); sc = renderer.Scene(ctx, packed)
L = renderer.lib()
if not hasattr(L, "vkr_debug_raster_stats"):
    sys.exit("library built without -DVKR_RASTER_STATS")
W, H = 3840, 2160
fo = abi.FrameOut()
for pi in (0, 128, 256, 400):
    L.vkr_debug_raster_stats(None, 1)
    sc.submit(abi.make_params(W, H, renderer.make_uniforms(W, H, fly_pose(pi), 1), deferred=True), fo); ctx.wait()
    st = np.zeros(8, np.uint64)
    L.vkr_debug_raster_stats(st.ctypes.data_as(C.c_void_p), 0)
    print("pose %3d: per lane group %d sub-triangles, %.1f box pixels each; per wavefront %d, %.0f box pixels each, %d block steps "
          "(row-major walk: %d)" % (pi, st[0], st[1] / max(1, st[0]), st[2], st[3] / max(1, st[2]), st[4], st[5]))
