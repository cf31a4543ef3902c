"""Sort-first multi-GPU rendering: one process per GPU, volume replicated, image rows split.

The reference drives a single GPU and queue (Source/VulkanApplication.cpp:316-321); this layer has
no counterpart there.  Pixels of the ray-march are independent (VolumeRendering.comp:159), so the
frame is cut into 16-row bands dealt round-robin to the ranks (interleaving balances the load when
skipping and early termination make some image regions cheaper than others).  Each rank renders
its bands with one launch into a dense buffer; the only exchange step is the gather of those
buffers to rank 0 — torch.distributed.gather, which is grouped send/recv on RCCL, i.e. one
point-to-point xGMI transfer of height*width*4/world bytes per peer — followed by one row
permutation on rank 0.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import band_row_indices, band_rows


def band_permutation(height, world, padded_rows):
    """perm[frame_row] = index of that row in the concatenation of the ranks' padded buffers."""
    perm = np.empty(height, np.int64)
    for r in range(world):
        rows = band_row_indices(height, r, world)
        perm[rows] = r * padded_rows + np.arange(rows.size)
    return perm


class SortFirstFrame:
    """Buffers and the exchange of one frame size.

    render_bands(part, parts, tensor) must fill `tensor` (uint8, [rows, width, 4], on `device`) with
    the part's bands; on a GPU it is VolumeRenderer.render_bands writing through the data pointer,
    in the CPU tests it is the checker.
    """

    def __init__(self, width, height, rank, world, device, group=None, host_staging=False):
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.device, self.group = torch.device(device), group
        # gloo cannot gather GPU tensors: a rehearsal of several ranks on one GPU stages through the host
        self.staging = host_staging
        self.rows = band_rows(height, rank, world)
        self.padded_rows = max(band_rows(height, r, world) for r in range(world))
        # equal-sized buffers on every rank: gather needs them.  Three of them, used in turn: a frame is
        # rendered while the one before it is still being finished on the renderer's second stream and the
        # one before that may still be in the gather.
        self.locals = [torch.zeros((self.padded_rows, width, 4), dtype=torch.uint8, device=self.device) for _ in range(3)]
        self.turn = 0
        self.recv = None
        self.frame = None
        if rank == 0:
            if world > 1:
                self.recv = torch.zeros((world, self.padded_rows, width, 4), dtype=torch.uint8, device=self.device)
                self.perm = torch.from_numpy(band_permutation(height, world, self.padded_rows)).to(self.device)
            self.frame = torch.zeros((height, width, 4), dtype=torch.uint8, device=self.device)
        # the fills above ran on torch's current stream, the renderer writes the buffers on its own
        # (non-blocking) stream: nothing else orders the two
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    @property
    def local(self):
        """The band buffer of the frame being rendered (the one the next exchange() sends)."""
        return self.locals[self.turn]

    def local_view(self):
        return self.local[: self.rows]

    def exchange(self):
        """Gathers every rank's bands on rank 0 and assembles the frame there (returns it on rank 0).
        Afterwards `local` is the next band buffer."""
        local = self.local
        self.turn = (self.turn + 1) % len(self.locals)
        if self.world == 1:
            # a single part's band buffer already is the frame, row for row
            self.frame.copy_(local[: self.height])
            return self.frame
        if self.staging:
            host = local.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, gather_list=parts, dst=0, group=self.group)
            if self.rank == 0:
                for r in range(self.world):
                    self.recv[r].copy_(parts[r])
        else:
            dist.gather(local, gather_list=list(self.recv.unbind(0)) if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank == 0:
            flat = self.recv.view(self.world * self.padded_rows, self.width, 4)
            torch.index_select(flat, 0, self.perm, out=self.frame)
            return self.frame
        return None
