#!/usr/bin/env python3
"""Writes tests/golden/micro_frames.npz: 64x36 frames of the micro scene in every render mode, two poses, as the CPU
oracle renders them (RGBA8 + the integer side buffers).  They pin the oracle against its own earlier output and give
the GPU parity tests a second, committed anchor."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import IDENTITY, pose_matrix  # noqa: E402

W, H, SHADOW = 64, 36, 128
MODES = {
    "deferred": dict(deferred=True), "forward": dict(debug=1), "mip": dict(debug=2), "depth": dict(debug=3),
    "deriv": dict(debug=4), "overdraw": dict(debug=5), "overshade": dict(debug=6), "albedo": dict(debug=7),
    "mosaic": dict(mosaic=True),
}
POSES = {"start": IDENTITY, "p1": pose_matrix(0.5, -0.3, -2.0, yaw=0.4, pitch=-0.15)}
BUFFERS = ("rgba8", "tri_id", "depth_bits", "mip_level", "stencil")


def render_all(scene, make_uniforms):
    from vulkanrenderer_amd import abi
    out = {}
    for mode, kw in MODES.items():
        for pose, m in POSES.items():
            p = abi.make_params(W, H, make_uniforms(W, H, m, kw.get("debug", 1)), shadow_size=SHADOW, **kw)
            r = scene.render(p, want=BUFFERS)
            for b in BUFFERS:
                out["%s/%s/%s" % (mode, pose, b)] = r[b]
    return out


def main():
    from oracle import oracle
    from vulkanrenderer_amd import scenes
    frames = render_all(oracle.OracleScene(scenes.micro_scene()), oracle.make_uniforms)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "micro_frames.npz"), **frames)
    print("wrote %d arrays" % len(frames))


if __name__ == "__main__":
    main()
