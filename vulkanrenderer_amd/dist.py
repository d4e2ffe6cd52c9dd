"""Partitioning of the frame workload over the GPUs of one node.

The reference renders on a single queue of a single device
(utils/vulkan_window.cpp:223-255); both partitionings below keep the scene
replicated per GPU and need no communication while rendering:

  poses       frame i of a camera fly-through goes to rank i % world
  sort-first  every rank renders one horizontal band of every frame
              (vkr_frame_params rect_*), and the RGBA8 bands are gathered on
              rank 0 - over RCCL/xGMI with the "nccl" backend, seven
              point-to-point transfers into the root, one per link.
"""
import torch
import torch.distributed as dist


def band_rect(rank, world, width, height):
    """Rectangle (x0, y0, x1, y1) of `rank`'s band: `world` horizontal bands of
    equal, even height (the last ones may be shorter or empty)."""
    band = (height + world - 1) // world
    band += band & 1
    y0, y1 = min(height, rank * band), min(height, (rank + 1) * band)
    return (0, y0, width, y1)


def pose_indices(rank, world, n_poses):
    """Frames of an n_poses fly-through rendered by `rank`."""
    return list(range(rank, n_poses, world))


class BandCompositor:
    """Gathers the per-rank RGBA8 bands of one frame into the full image on rank 0."""

    def __init__(self, width, height, device, rank=None, world=None, host_collective=False, everywhere=False):
        """host_collective: gather through host tensors (gloo rehearsals); the product path gathers device
        tensors over RCCL.  everywhere: every rank ends up with the full frame (one all-gather instead of the
        gather to rank 0), for hosts where each GPU drives a display of its own."""
        self.host = host_collective
        self.everywhere = everywhere
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.width, self.height = width, height
        self.rects = [band_rect(r, self.world, width, height) for r in range(self.world)]
        x0, y0, x1, y1 = self.rects[self.rank]
        # every rank contributes a buffer of the tallest band (gather needs equal sizes); short bands are padded
        self.band_h = max(r[3] - r[1] for r in self.rects)
        self.local = torch.zeros((self.band_h, width, 4), dtype=torch.uint8, device=device)
        gdev = "cpu" if host_collective else device
        self.gathered = ([torch.empty(self.local.shape, dtype=torch.uint8, device=gdev) for _ in range(self.world)]
                         if (self.rank == 0 or everywhere) else None)

    @property
    def rect(self):
        return self.rects[self.rank]

    def gather(self):
        """Returns the H x W x 4 frame on rank 0 (on every rank with `everywhere`), None elsewhere."""
        if self.world == 1:
            return self.local[: self.height]
        src = self.local.cpu() if self.host else self.local
        if self.everywhere:
            dist.all_gather(self.gathered, src)
        else:
            dist.gather(src, self.gathered, dst=0)
        if self.local.is_cuda and not self.host:
            # the collective is only enqueued (on torch's RCCL stream); `local` is also the renderer's output buffer,
            # written by the next frame on the renderer's own stream, so the send must have finished before this returns
            torch.cuda.current_stream().synchronize()
        if not self.everywhere and self.rank != 0:
            return None
        rows = [g[: r[3] - r[1]] for g, r in zip(self.gathered, self.rects) if r[3] > r[1]]
        return torch.cat(rows, dim=0)
