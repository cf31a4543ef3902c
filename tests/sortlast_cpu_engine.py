"""The sort-last kernels restated on the CPU for the multi-process protocol tests.

SortLastFrame drives an "engine" through render_partial / composite_* / decide / chain_stage /
store_chain.  On a GPU the engine is the HIP library; here it is the checker (oracle/) plus a few
lines of numpy, so that the exchange schedule, the visibility order and the re-march protocol can
run over gloo on a machine without a GPU.  Test infrastructure only."""
import numpy as np
import torch

from oracle import vr_oracle as o


def to_unorm8(x):
    x = np.where(x > 0, x, 0.0)            # NaN and negatives to 0
    return np.rint(np.minimum(x, 1.0).astype(np.float32) * np.float32(255.0)).astype(np.uint8)


class OracleEngine:
    def __init__(self, volume, lut, location, to_world, width, height, samples, brick_lo, brick_hi, padded):
        self.vol, self.lut, self.loc, self.m = volume, lut, location, to_world
        self.width, self.height, self.samples = width, height, samples
        self.lo, self.hi = brick_lo, brick_hi
        nz, ny, nx = volume.shape
        d = o.march_direction(location, to_world, width, height, samples, (nx, ny, nz))
        self.dir = np.zeros((padded, 4), np.uint8)      # padding pixels: misses
        self.dir[: width * height] = d

    def render_partial(self, out):
        p = o.render_partial(self.vol, self.lut, self.loc, self.m, self.width, self.height, self.samples,
                             self.lo, self.hi, threads=1)
        out[: p.shape[0]].copy_(torch.from_numpy(p))

    @staticmethod
    def _under(front, back):
        res = np.empty_like(front)
        res[:, :3] = front[:, :3] + front[:, 3:4] * back[:, :3]
        res[:, 3] = front[:, 3] * back[:, 3]
        res[:, 4] = front[:, 4] + front[:, 3] * back[:, 4]
        return res

    def composite_pair(self, first, count, mine, theirs, axis, mine_is_low, out):
        d = self.dir[first: first + count]
        up, hit = d[:, axis].astype(bool), d[:, 3].astype(bool)
        a, b = mine.numpy().astype(np.float64), theirs.numpy().astype(np.float64)
        mine_front = (bool(mine_is_low) != up)[:, None]
        res = self._under(np.where(mine_front, a, b), np.where(mine_front, b, a))
        res[~hit] = 0.0
        out.copy_(torch.from_numpy(res.astype(np.float32)))

    def composite_ordered(self, first, count, parts, slot_of_box, grid, out):
        d = self.dir[first: first + count]
        hit = d[:, 3].astype(bool)
        p = parts.numpy().astype(np.float64)
        slots = slot_of_box.numpy()
        boxes = grid[0] * grid[1] * grid[2]
        res = np.zeros((count, 5))
        res[:, 3] = 1.0
        idx = np.arange(count)
        for rank in range(boxes - 1, -1, -1):            # front to back
            i, j, k = rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1])
            ci = np.where(d[:, 0] != 0, i, grid[0] - 1 - i)
            cj = np.where(d[:, 1] != 0, j, grid[1] - 1 - j)
            ck = np.where(d[:, 2] != 0, k, grid[2] - 1 - k)
            box = (ck * grid[1] + cj) * grid[0] + ci
            res = self._under(res, p[slots[box], idx])
        res[~hit] = 0.0
        out.copy_(torch.from_numpy(res.astype(np.float32)))

    def decide(self, first, count, partials, out_rgba8, undecided, undecided_count):
        p = partials.numpy().astype(np.float64)
        hit = self.dir[first: first + count, 3].astype(bool)
        x = np.clip(p[:, :3], 0.0, 1.0) * 255.0
        # generous: anything within 0.02 of a rounding boundary, or with light left over, is re-marched
        near = (np.abs(x - np.floor(x) - 0.5) < 0.02).any(axis=1) | (p[:, 3] * 255.0 > 1e-3)
        rgba = np.zeros((count, 4), np.uint8)
        rgba[:, :3] = to_unorm8(p[:, :3])
        rgba[:, 3] = 255
        todo = np.nonzero(near & hit)[0]
        out_rgba8.copy_(torch.from_numpy(rgba))
        undecided[: todo.size].copy_(torch.from_numpy((todo + first).astype(np.int32)))
        undecided_count[0] = int(todo.size)

    def chain_stage(self, pixels, colour_in, colour_out, stage, grid, box):
        c = o.chain_stage(self.vol, self.lut, self.loc, self.m, self.width, self.height, self.samples, self.lo,
                          self.hi, pixels.numpy().astype(np.uint32), colour_in.numpy(), stage, grid, box)
        colour_out.copy_(torch.from_numpy(c))

    def store_chain(self, pixels, colour, first, count, out_rgba8):
        px = pixels.numpy().astype(np.int64)
        sel = (px >= first) & (px < first + count)
        rgba = out_rgba8.numpy()
        rgba[px[sel] - first, :3] = to_unorm8(colour.numpy()[sel, :3])
