"""Sort-first partitions on one GPU: the rows each rank would render (bands, interleaved stripes) are rendered one
rank after the other and reassembled; the result must equal the frame rendered in one piece, byte for byte.  The RCCL
compositor (vkr_compositor_*) is exercised with a communicator of one rank: its rows travel through ncclSend / ncclRecv
to itself."""
import ctypes as C

import numpy as np
import pytest

from conftest import IDENTITY, pose_matrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    from vulkanrenderer_amd import renderer, scenes
    ctx = renderer.Context(0)
    sc = renderer.Scene(ctx, scenes.proxy_scene(detail=0.25, tex_size=64))
    yield ctx, sc
    sc.close(); ctx.close()


def make(W, H, pose, **kw):
    from vulkanrenderer_amd import abi, renderer
    return abi.make_params(W, H, renderer.make_uniforms(W, H, pose, kw.get("debug", 1)), shadow_size=512, **kw)


@pytest.mark.parametrize("mode,kw", [("deferred", dict(deferred=True)), ("fused", dict(deferred=True, flags=2)), ("forward", dict()),
                                     ("mip", dict(debug=2)), ("deriv", dict(debug=4)), ("overdraw", dict(debug=5)), ("overshade", dict(debug=6))])
def test_interleaved_stripes_reassemble_to_the_frame(scene, mode, kw):
    from vulkanrenderer_amd import abi
    from vulkanrenderer_amd.dist import stripe_rows
    ctx, sc = scene
    pose = pose_matrix(0.0, -1.0, -20.0, yaw=0.2)
    for (W, H, world, sh) in ((1920, 1080, 8, 64), (640, 360, 3, 32), (200, 120, 8, 32)):   # 1080 = 16 stripes of 64 + one of 56; ranks without rows
        want = ("rgba8", "tri_id", "depth_bits", "mip_level", "stencil")
        full = sc.render(make(W, H, pose, **kw), want=want)
        got = {k: np.zeros_like(v) for k, v in full.items()}
        for r in range(world):
            rows = stripe_rows(H, world, r, sh)
            part = sc.render(make(W, H, pose, stripes=abi.stripes(sh, world, r), **kw), want=want)
            for k in want:
                assert part[k].shape[0] == len(rows)
                got[k][rows] = part[k]
        for k in want:
            assert np.array_equal(full[k], got[k]), "%s %dx%d / %d ranks: %s differs at %d pixels" % (mode, W, H, world, k, int((full[k] != got[k]).sum()))


def test_stripes_are_validated(scene):
    from vulkanrenderer_amd import abi, renderer
    ctx, sc = scene
    for bad in (abi.stripes(48, 4, 0), abi.stripes(64, 4, 4), abi.stripes(0, 4, 1)):
        with pytest.raises(renderer.VkrError, match="stripes"):
            sc.render(make(320, 180, IDENTITY, deferred=True, stripes=bad))
    with pytest.raises(renderer.VkrError, match="stripes"):
        sc.render(make(320, 180, IDENTITY, mosaic=True, stripes=abi.stripes(64, 2, 0)))


def test_rccl_compositor_with_one_rank(scene):
    """A communicator of one rank: gather to root and the all-ranks variant move this rank's rows through RCCL (send
    to self) into their place in the frame, on the context's stream, behind the frame that produced them."""
    from conftest import DeviceBuffer
    from vulkanrenderer_amd import abi, renderer
    ctx, sc = scene
    W, H = 640, 360
    comp = renderer.Compositor(ctx, renderer.composite_unique_id(), 0, 1)
    local, frame = DeviceBuffer(W * H * 4), DeviceBuffer(W * H * 4)
    try:
        ref = sc.render(make(W, H, IDENTITY, deferred=True), want=("rgba8",))["rgba8"]
        fo = abi.FrameOut()
        fo.rgba8 = local.ptr
        for root, sh in ((0, 0), (-1, 0), (0, 64), (-1, 32)):
            frame.zero()
            sc.submit(make(W, H, IDENTITY, deferred=True), fo)                  # asynchronous
            comp.gather(W, H, local.ptr, frame.ptr, root=root, stripe_height=sh)  # queued behind it, no host wait
            ctx.wait()
            assert np.array_equal(frame.numpy((H, W, 4), np.uint8), ref), (root, sh)
        # rendered straight into place: nothing to move
        fo.rgba8 = frame.ptr
        frame.zero()
        sc.submit(make(W, H, IDENTITY, deferred=True), fo)
        comp.gather(W, H, frame.ptr, frame.ptr, root=0)
        ctx.wait()
        assert np.array_equal(frame.numpy((H, W, 4), np.uint8), ref)
        # the integer side buffers travel the same way: triangle ids (4 bytes per pixel) and the overdraw counter (1)
        side = sc.render(make(W, H, IDENTITY, debug=5), want=("stencil",))["stencil"]
        ids = sc.render(make(W, H, IDENTITY, deferred=True), want=("tri_id",))["tri_id"]
        lstc, fstc = DeviceBuffer(W * H), DeviceBuffer(W * H)
        try:
            fo = abi.FrameOut()
            fo.tri_id = local.ptr
            frame.zero()
            sc.submit(make(W, H, IDENTITY, deferred=True), fo)
            comp.gather_rows(W * 4, H, local.ptr, frame.ptr, root=0, stripe_height=32)
            ctx.wait()
            assert np.array_equal(frame.numpy((H, W), np.uint32), ids)
            fo = abi.FrameOut()
            fo.stencil = lstc.ptr
            fstc.zero()
            sc.submit(make(W, H, IDENTITY, debug=5), fo)
            comp.gather_rows(W, H, lstc.ptr, fstc.ptr, root=-1)
            ctx.wait()
            assert np.array_equal(fstc.numpy((H, W), np.uint8), side)
        finally:
            lstc.free(); fstc.free()
    finally:
        comp.close(); local.free(); frame.free()


def test_exchange_follows_the_frame_onto_its_lane(scene):
    """Two frames in flight: consecutive frames run on lanes (streams) of their own, each with its own rows and its own
    frame, and each exchange is queued on the stream of the frame it belongs to (vkr_context_frame_stream), so it
    neither waits for nor overtakes the other lane's frame.  Bands in place, bands through a separate buffer and
    stripes; the frames must be the ones rendered with one lane and a fence per frame."""
    from conftest import DeviceBuffer
    from vulkanrenderer_amd import abi, renderer
    ctx, sc = scene
    L = renderer.lib()
    W, H = 640, 360
    poses = [IDENTITY, pose_matrix(0.0, -1.0, -20.0, yaw=0.2), pose_matrix(0.0, -1.0, -40.0, yaw=-0.3), pose_matrix(1.0, -1.0, -60.0)]
    refs = [sc.render(make(W, H, p, deferred=True), want=("rgba8",))["rgba8"] for p in poses]
    comp = renderer.Compositor(ctx, renderer.composite_unique_id(), 0, 1)
    locals_, frames = [DeviceBuffer(W * H * 4) for _ in range(2)], [DeviceBuffer(W * H * 4) for _ in range(2)]
    try:
        ctx.set_frames_in_flight(2)
        lane0 = L.vkr_context_stream(ctx.h)
        for sh, in_place in ((0, False), (0, True), (64, False)):
            for f in frames:
                f.zero()
            got, streams = [], []
            for i, pose in enumerate(poses):
                k = i % 2
                if i >= 2:  # the lane's previous frame is read back before the lane is reused
                    assert L.vkr_wait_next_lane(ctx.h) == 0, L.vkr_last_error()
                    got.append(frames[k].numpy((H, W, 4), np.uint8).copy())
                fo = abi.FrameOut()
                fo.rgba8 = frames[k].ptr if in_place else locals_[k].ptr
                sc.submit(make(W, H, pose, deferred=True), fo)
                assert L.vkr_context_frame_lane(ctx.h) == k
                streams.append(L.vkr_context_frame_stream(ctx.h))
                comp.gather(W, H, fo.rgba8, frames[k].ptr, root=0, stripe_height=sh)
            ctx.wait()
            got += [frames[0].numpy((H, W, 4), np.uint8).copy(), frames[1].numpy((H, W, 4), np.uint8).copy()]
            assert streams[0] == lane0 and streams[1] != lane0 and streams[2] == streams[0] and streams[3] == streams[1]
            for i in range(4):
                assert np.array_equal(got[i], refs[i]), (sh, in_place, i)
    finally:
        ctx.wait()
        ctx.set_frames_in_flight(1)
        comp.close()
        for b in locals_ + frames:
            b.free()


def test_gather_refuses_rows_that_overlap_the_frame(scene):
    """`local` is either the rank's band in its place inside `frame` or a buffer apart from it."""
    from conftest import DeviceBuffer
    from vulkanrenderer_amd import renderer
    ctx, sc = scene
    W, H = 64, 64
    comp = renderer.Compositor(ctx, renderer.composite_unique_id(), 0, 1)
    frame = DeviceBuffer(W * H * 4 * 2)
    try:
        with pytest.raises(renderer.VkrError, match="overlaps"):
            comp.gather(W, H, frame.ptr + W * 4 * 8, frame.ptr, root=0)                      # a band that is not in its place
        with pytest.raises(renderer.VkrError, match="overlaps"):
            comp.gather(W, H, frame.ptr, frame.ptr, root=0, stripe_height=32)                # stripes are never in place
        with pytest.raises(renderer.VkrError, match="NULL"):
            comp.gather(W, H, None, frame.ptr, root=0)                                       # this rank owns rows
        comp.gather(W, H, frame.ptr + W * H * 4, frame.ptr, root=0)                          # apart: fine
        ctx.wait()
    finally:
        comp.close(); frame.free()
