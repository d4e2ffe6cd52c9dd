"""GPU parity at the benchmark's own data: the whole SunTemple-proxy scene with the 1024^2 textures (11-level mip
chains) the timed frames sample, HIP path against the CPU oracle.  The other configuration tests use 64^2 textures,
where most pixels magnify; here nearly every pixel minifies, so the trilinear path, the level selection and the
integer mip-level buffer are exercised on the deep chains (debug.frag:19-47; levels = 32 - clz(w|h),
utils/vkimage.cpp:289-294).  Integer outputs bit for bit, colour within 1/255 and PSNR >= 50 dB.
"""
import numpy as np
import pytest

from conftest import IDENTITY
from test_configs_gpu import P1, P2, THREADS, check, make

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full1024():
    from oracle import oracle
    from vulkanrenderer_amd import renderer, scenes
    packed = scenes.proxy_scene(detail=1.0, tex_size=1024)
    ctx = renderer.Context(0)
    sc, osc = renderer.Scene(ctx, packed), oracle.OracleScene(packed)
    yield sc, osc, packed
    sc.close(); osc.close(); ctx.close()


@pytest.mark.parametrize("name,pose", [("I", IDENTITY), ("P1", P1), ("P2", P2)])
def test_mip_view_1080p_levels_bit_exact(full1024, name, pose):
    """View 2 at 1920x1080: the integer mip level of every pixel equals the oracle's, and the deep levels are in use."""
    sc, osc, _ = full1024
    want = ("rgba8", "tri_id", "depth_bits", "mip_level")
    p = make(1920, 1080, pose, debug=2)
    ref, got = osc.render(p, want=want, threads=THREADS), sc.render(p, want=want)
    hist = np.bincount(ref["mip_level"][ref["tri_id"] != 0xFFFFFFFF].ravel(), minlength=11)
    print("mip levels at pose %s: %s" % (name, hist.tolist()))
    assert hist.size == 11 and all(hist[l] > 1000 for l in (1, 2, 3, 4, 5, 6)) and hist[3:].sum() > hist.sum() // 2
    check(ref, got, "1024^2 textures, view 2, pose " + name)


@pytest.mark.parametrize("name,pose", [("I", IDENTITY), ("P1", P1)])
def test_deferred_1080p(full1024, name, pose):
    sc, osc, _ = full1024
    want = ("rgba8", "tri_id", "depth_bits", "gbuf_normal", "gbuf_albedo")
    p = make(1920, 1080, pose, deferred=True)
    ref, got = osc.render(p, want=want, threads=THREADS), sc.render(p, want=want)
    check(ref, got, "1024^2 textures, deferred, pose " + name)
    for k in ("gbuf_normal", "gbuf_albedo"):
        a, b = ref[k].view(np.float16).astype(np.float32), got[k].view(np.float16).astype(np.float32)
        assert np.all(np.abs(a - b) <= np.maximum(np.abs(a), 1.0) * 2.0 ** -9), k


def test_forward_720p_start_pose(full1024):
    """The reference's own CPU-runnable configuration (1280x720, forward + 2048^2 light pass, camera2world = I)."""
    sc, osc, _ = full1024
    want = ("rgba8", "tri_id", "depth_bits", "shadow_map")
    p = make(1280, 720, IDENTITY, shadow=2048)
    check(osc.render(p, want=want, threads=THREADS), sc.render(p, want=want), "1024^2 textures, forward 720p")


def test_headline_4k_deferred_fly_through_poses(full1024):
    """The timed workload itself: 3840x2160 deferred at poses of the benchmark's fly-through, two-kernel and fused."""
    from bench import fly_pose
    from vulkanrenderer_amd import abi
    sc, osc, _ = full1024
    want = ("rgba8", "tri_id", "depth_bits")
    for i in (40, 300):
        p = make(3840, 2160, fly_pose(i), deferred=True)
        ref, got = osc.render(p, want=want, threads=THREADS), sc.render(p, want=want)
        check(ref, got, "1024^2 textures, 4K deferred, fly-through pose %d" % i)
        fused = sc.render(make(3840, 2160, fly_pose(i), deferred=True, flags=abi.FRAME_FUSED_LIGHTING), want=want)
        check(ref, fused, "1024^2 textures, 4K fused lighting, fly-through pose %d" % i)


def test_mosaic_1080p_fly_through_pose(full1024):
    from bench import fly_pose
    sc, osc, _ = full1024
    p = make(1920, 1080, fly_pose(200), mosaic=True)
    check(osc.render(p, want=("rgba8",), threads=THREADS), sc.render(p, want=("rgba8",)), "1024^2 textures, mosaic")
