"""Partitioning of the frame workload over the GPUs of one node.

The reference renders on a single queue of a single device
(utils/vulkan_window.cpp:223-255); both partitionings below keep the scene
replicated per GPU and need no communication while rendering:

  poses       frame i of a camera fly-through goes to rank i % world
  sort-first  every rank renders its rows of every frame - one horizontal band
              (vkr_frame_params rect_*) or every world-th stripe of 32k rows
              (vkr_frame_params stripes) - and the RGBA8 rows meet on rank 0.

The product path for the exchange is the library's own compositor
(vkr_compositor_*: grouped RCCL send / receive on the renderer's stream, no
host wait, no per-frame allocation; `RcclCompositor` below is its thin caller).
`HostCompositor` does the same exchange through host tensors and any
torch.distributed backend; it exists for rehearsals of the multi-rank path on
boxes with fewer GPUs than ranks (gloo) and for the CPU tests.
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


def stripe_rows(height, world, rank, stripe_height):
    """Frame rows of `rank` under the interleaved-stripe partition (VKR_STRIPES), in the order they are stored."""
    rows = []
    for y in range(rank * stripe_height, height, world * stripe_height):
        rows.extend(range(y, min(height, y + stripe_height)))
    return rows


def rank_rows(height, world, rank, stripe_height=0):
    """Frame rows of `rank` in storage order, for either partition (stripe_height 0 = bands)."""
    if stripe_height:
        return stripe_rows(height, world, rank, stripe_height)
    _, y0, _, y1 = band_rect(rank, world, 1, height)
    return list(range(y0, y1))


def pose_indices(rank, world, n_poses):
    """Frames of an n_poses fly-through rendered by `rank`."""
    return list(range(rank, n_poses, world))


class HostCompositor:
    """Gathers the per-rank RGBA8 rows of one frame into the full image on rank 0 (or on every rank) through host
    tensors: the rehearsal / test stand-in for the library's RCCL compositor."""

    def __init__(self, width, height, device, rank=None, world=None, stripe_height=0, everywhere=False):
        self.everywhere = everywhere
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.width, self.height, self.stripe_height = width, height, stripe_height
        self.rows = [rank_rows(height, self.world, r, stripe_height) for r in range(self.world)]
        # every rank contributes a buffer of the largest share (gather needs equal sizes); short shares are padded
        self.max_rows = max(1, max(len(r) for r in self.rows))
        self.local = torch.zeros((self.max_rows, width, 4), dtype=torch.uint8, device=device)
        self._send = torch.zeros((self.max_rows, width, 4), dtype=torch.uint8)
        self.gathered = ([torch.empty(self._send.shape, dtype=torch.uint8) for _ in range(self.world)]
                         if (self.rank == 0 or everywhere) else None)
        self.frame = torch.zeros((height, width, 4), dtype=torch.uint8) if (self.rank == 0 or everywhere) else None
        self._index = [torch.tensor(r, dtype=torch.long) for r in self.rows]

    @property
    def rect(self):
        """This rank's band as a rectangle (bands only)."""
        assert not self.stripe_height
        return band_rect(self.rank, self.world, self.width, self.height)

    @property
    def my_rows(self):
        return len(self.rows[self.rank])

    def gather(self):
        """Returns the H x W x 4 frame (a host tensor) on rank 0 (on every rank with `everywhere`), None elsewhere.
        The caller has waited for the renderer; when this returns the rows have left `local`."""
        self._send.copy_(self.local)
        if self.world == 1:
            self.frame[self._index[0]] = self._send[: self.my_rows]
            return self.frame
        if self.everywhere:
            dist.all_gather(self.gathered, self._send)
        else:
            dist.gather(self._send, self.gathered, dst=0)
            if self.rank != 0:
                return None
        for g, idx in zip(self.gathered, self._index):
            if len(idx):
                self.frame[idx] = g[: len(idx)]
        return self.frame


# the name earlier callers used
BandCompositor = HostCompositor


class RcclCompositor:
    """Thin caller of the library's compositor (vkr_compositor_*): the rows go over RCCL / xGMI on the stream of the
    frame they belong to.  The unique id is created on rank 0 and broadcast over torch.distributed, which the host
    already has.  `lanes` = frames in flight on the context: every lane renders into rows of its own (`locals[k]`) and,
    on the receiving ranks, into a frame of its own (`frames[k]`), so the exchange of one frame runs under the
    rendering of the next."""

    def __init__(self, ctx, width, height, rank, world, device, stripe_height=0, everywhere=False, lanes=1):
        from . import renderer
        self.ctx, self.width, self.height, self.rank, self.world = ctx, width, height, rank, world
        self.stripe_height, self.everywhere = stripe_height, everywhere
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(renderer.composite_unique_id()), dtype=torch.uint8).clone()
        uid = uid.to(device)
        if world > 1:
            dist.broadcast(uid, src=0)
        self.comp = renderer.Compositor(ctx, bytes(uid.cpu().numpy().tobytes()), rank, world)
        first, count = renderer.sortfirst_rows(height, world, rank, stripe_height)
        self.first_row, self.my_rows = first, count
        receives = everywhere or rank == 0
        self.frames, self.locals = [], []
        for _ in range(max(1, lanes)):
            frame = torch.zeros((height, width, 4), dtype=torch.uint8, device=device) if receives else None
            if receives and not stripe_height:
                local = frame[first:first + count]  # a band is rendered straight into its place
            else:
                local = torch.zeros((max(count, 1), width, 4), dtype=torch.uint8, device=device)
            self.frames.append(frame)
            self.locals.append(local)
        self.frame, self.local = self.frames[0], self.locals[0]

    def gather(self, lane=0):
        """Queues the exchange behind the frame just submitted (which went to `lane`); returns at once.  `frames[lane]` is
        complete after ctx.wait()."""
        frame, local = self.frames[lane], self.locals[lane]
        self.comp.gather(self.width, self.height, local.data_ptr() if self.my_rows else None,
                         frame.data_ptr() if frame is not None else None, root=-1 if self.everywhere else 0, stripe_height=self.stripe_height)
        return frame

    def close(self):
        self.comp.close()
