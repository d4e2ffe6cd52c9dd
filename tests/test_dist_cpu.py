"""Multi-process (world_size 2, gloo, CPU) tests of the partitioning host logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vulkanrenderer_amd.dist import BandCompositor, band_rect, pose_indices


def test_band_rects_tile_the_frame():
    for (w, h) in ((3840, 2160), (1920, 1080), (64, 36), (17, 5)):
        for world in (1, 2, 3, 4, 8):
            rects = [band_rect(r, world, w, h) for r in range(world)]
            rows = np.zeros(h, np.int32)
            for (x0, y0, x1, y1) in rects:
                assert (x0, x1) == (0, w) and (y0 % 2 == 0 or y0 == y1) and 0 <= y0 <= y1 <= h
                rows[y0:y1] += 1
            assert np.all(rows == 1)


def test_pose_partition_covers_every_frame_once():
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in pose_indices(r, world, 512))
        assert got == list(range(512))
        assert max(len(pose_indices(r, world, 512)) for r in range(world)) == 512 // world


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, full):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W = full.shape[:2]
        comp = BandCompositor(W, H, "cpu")
        for frame in range(2):
            img = (full + frame) % 256          # stand-in for a renderer: each rank owns the rows of its rectangle
            x0, y0, x1, y1 = comp.rect
            comp.local.zero_()
            comp.local[: y1 - y0] = torch.from_numpy(img[y0:y1].astype(np.uint8))
            out = comp.gather()
            if rank == 0:
                assert out.shape == (H, W, 4)
                assert np.array_equal(out.numpy(), img.astype(np.uint8))
            else:
                assert out is None
        # interleaved stripes: the rows each rank owns come from the library's own partition function
        # (vkr_sortfirst_rows, what vkr_render's stripes and the RCCL compositor use), stored back to back
        from vulkanrenderer_amd import renderer
        from vulkanrenderer_amd.dist import HostCompositor, stripe_rows
        for sh in (32, 64):
            comp = HostCompositor(W, H, "cpu", stripe_height=sh)
            first, count = renderer.sortfirst_rows(H, world, rank, sh)
            rows = stripe_rows(H, world, rank, sh)
            assert count == len(rows) == comp.my_rows and (count == 0 or rows[0] == first)
            comp.local.zero_()
            if count:
                comp.local[:count] = torch.from_numpy(full[rows].astype(np.uint8))
            out = comp.gather()
            if rank == 0:
                assert np.array_equal(out.numpy(), full.astype(np.uint8))
        # every rank gets the frame
        comp = BandCompositor(W, H, "cpu", everywhere=True)
        x0, y0, x1, y1 = comp.rect
        comp.local.zero_()
        comp.local[: y1 - y0] = torch.from_numpy(full[y0:y1].astype(np.uint8))
        assert np.array_equal(comp.gather().numpy(), full.astype(np.uint8))
        # timing convention of bench.py: max over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [(36, 64), (37, 20), (150, 24)])
def test_sort_first_gather_world2(shape):
    H, W = shape
    rng = np.random.default_rng(2)
    full = rng.integers(0, 256, (H, W, 4)).astype(np.int64)
    mp.spawn(_worker, args=(2, _free_port(), full), nprocs=2, join=True)  # raises if a rank's assertions fail
