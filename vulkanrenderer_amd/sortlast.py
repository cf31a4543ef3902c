"""Sort-last multi-GPU rendering: one process per GPU, each holding one box of the volume.

The reference renders from a single 3-D image on a single GPU (Applications/VolumeRendering/Source/
main.cpp:244-258, Source/VulkanApplication.cpp:316-321); this layer has no counterpart there and
exists for volumes that should not be replicated.  The frame is still the one the shader defines
(Applications/VolumeRendering/Shader/VolumeRendering.comp:154-206).

A frame takes four steps (the kernels are behind the C ABI, see include/vr_abi.h):
  1. every rank marches the taps of its own box: a partial image of 5 floats per pixel;
  2. the partial images are exchanged and merged per pixel in visibility order.  Two schedules:
       "direct_send": the pixels are cut into `world` spans, one all_to_all hands every rank all
           boxes' partials of its span and one kernel merges them.  On MI355X every GPU has a
           direct xGMI link to each of its 7 peers, so all links carry traffic at once;
       "binary_swap": log2(world) rounds with partner rank ^ (1 << round), each round sending one
           half of the pixels still held and merging the half kept.  Same bytes in total, but one
           link busy per round; kept because it needs no all_to_all and for comparison.
     Either way a rank ends up with the merged partials of 1/world of the pixels;
  3. each rank turns its merged partials into RGBA8 and lists the pixels whose bytes they cannot
     decide (a rounding boundary within the error bound of the merged floats);
  4. the listed pixels (typically a fraction of a percent) are re-marched in the shader's own
     back-to-front float arithmetic, the colour travelling from box to box: `world` stages, each a
     kernel on every rank plus one all_reduce(SUM) of 16 bytes per listed pixel in which exactly
     one rank contributes a non-zero value.  Their bytes overwrite those of step 3.
Finally the spans are gathered on rank 0.  The result equals the single-GPU frame byte for byte.
"""
import numpy as np
import torch
import torch.distributed as dist

PARTIAL_FLOATS = 5


# ---- decomposition (pure host arithmetic) -----------------------------------------------------
def box_grid(world):
    """Boxes per axis (gx, gy, gz) with gx*gy*gz == world, as cubic as the factors allow."""
    grid = [1, 1, 1]
    n, p, factors = world, 2, []
    while n > 1:
        while n % p == 0:
            factors.append(p)
            n //= p
        p += 1
    for f in sorted(factors, reverse=True):
        k = grid.index(min(grid))
        grid[k] *= f
    return tuple(sorted(grid, reverse=True))


def box_of_rank(rank, grid):
    """Position (bx, by, bz) of a rank's box: x fastest."""
    return rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1])


def brick_box(bricks, grid, box):
    """Brick range (lo, hi) of a box when `bricks` (per axis) are split evenly over `grid`."""
    if any(b < g for b, g in zip(bricks, grid)):
        raise ValueError("fewer bricks than boxes along an axis: %r bricks, %r boxes" % (bricks, grid))
    lo = tuple((c * b) // g for c, b, g in zip(box, bricks, grid))
    hi = tuple(((c + 1) * b) // g for c, b, g in zip(box, bricks, grid))
    return lo, hi


class Cuts:
    """Where the volume's brick grid is cut into a grid of boxes: one set of planes across z, then for every
    z layer its own planes across y, then for every (z, y) cell its own planes across x (a k-d split).
    Boxes keep their grid coordinates, so the visibility order (lexicographic in mirrored (z, y, x)) and
    the binary-swap rounds (x pairs first, then y, then z: every merged pair is separated by one of
    these planes and its union is again a box) are those of an even grid."""

    def __init__(self, grid, z, y, x):
        self.grid = tuple(grid)
        self.z = [int(v) for v in z]                                   # gz + 1 brick indices
        self.y = [[int(v) for v in row] for row in y]                  # [gz][gy + 1]
        self.x = [[[int(v) for v in cell] for cell in row] for row in x]   # [gz][gy][gx + 1]

    def box(self, box):
        bx, by, bz = box
        return ((self.x[bz][by][bx], self.y[bz][by], self.z[bz]), (self.x[bz][by][bx + 1], self.y[bz][by + 1], self.z[bz + 1]))


def even_cuts(bricks, grid):
    """The even split brick_box() describes, as Cuts."""
    if any(b < g for b, g in zip(bricks, grid)):
        raise ValueError("fewer bricks than boxes along an axis: %r bricks, %r boxes" % (bricks, grid))
    edges = [[(c * b) // g for c in range(g + 1)] for b, g in zip(bricks, grid)]
    return Cuts(grid, edges[2], [edges[1]] * grid[2], [[edges[0]] * grid[1]] * grid[2])


def _balanced_edges(weights, parts, unit, total):
    """Cuts a 1-D array of weights (one per `unit` bricks, covering `total` bricks) into `parts` runs of about
    equal sum, every run at least one cell long; returns parts + 1 brick indices."""
    cells = len(weights)
    if cells < parts:
        raise ValueError("fewer count cells than boxes along an axis")
    cum = np.concatenate([[0.0], np.cumsum(np.asarray(weights, np.float64) + 1e-9)])
    edges = [0]
    for k in range(1, parts):
        target = cum[-1] * k / parts
        e = int(np.searchsorted(cum, target))
        if e > 0 and abs(cum[e - 1] - target) <= abs(cum[min(e, cells)] - target):
            e -= 1
        e = max(edges[-1] + 1, min(e, cells - (parts - k)))
        edges.append(e)
    edges.append(cells)
    return [min(e * unit, total) for e in edges[:-1]] + [total]


def balanced_cuts(counts, cell_bricks, bricks, grid):
    """Cuts with about the same number of occupied bricks in every box.  counts: [cz, cy, cx] occupied bricks
    per cube of cell_bricks^3 bricks (vr_occupied_brick_counts, summed over the contexts); bricks: the
    volume's brick grid (x, y, z); cut planes fall on cube boundaries."""
    counts = np.asarray(counts, np.float64)
    gx, gy, gz = grid
    z = _balanced_edges(counts.sum(axis=(1, 2)), gz, cell_bricks, bricks[2])
    ys, xs = [], []
    for k in range(gz):
        layer = counts[z[k] // cell_bricks: -(-z[k + 1] // cell_bricks)]
        y = _balanced_edges(layer.sum(axis=(0, 2)), gy, cell_bricks, bricks[1])
        ys.append(y)
        row = []
        for j in range(gy):
            cell = layer[:, y[j] // cell_bricks: -(-y[j + 1] // cell_bricks)]
            row.append(_balanced_edges(cell.sum(axis=(0, 1)), gx, cell_bricks, bricks[0]))
        xs.append(row)
    return Cuts(grid, z, ys, xs)


def swap_rounds(world, grid):
    """Binary swap: per round the axis whose boxes are merged.  Needs power-of-two grid sides."""
    if any(g & (g - 1) for g in grid):
        raise ValueError("binary swap needs a power-of-two number of boxes per axis")
    rounds = []
    for axis, g in enumerate(grid):
        rounds += [axis] * (g.bit_length() - 1)
    assert 1 << len(rounds) == world
    return rounds


def swap_span(rank, world, padded):
    """Pixel span (first, count) a rank holds after all binary-swap rounds."""
    first, count, r = 0, padded, 0
    while (1 << r) < world:
        count //= 2
        if (rank >> r) & 1:
            first += count
        r += 1
    return first, count


# ---- the per-frame driver -----------------------------------------------------------------------
class HipEngine:
    """The kernels of one context (VolumeRenderer) applied to torch tensors on its GPU."""

    def __init__(self, renderer, width, height):
        self.r, self.width, self.height = renderer, width, height
        self.stream = torch.cuda.ExternalStream(renderer.stream)

    def scope(self):
        # torch work and collectives are enqueued on the context's own stream: one ordered queue
        return torch.cuda.stream(self.stream)

    def render_partial(self, out):
        self.r.render_partial(self.width, self.height, out.data_ptr())

    def composite_pair(self, first, count, mine, theirs, axis, mine_is_low, out):
        self.r.composite_pair(self.width, self.height, first, count, mine.data_ptr(), theirs.data_ptr(), axis,
                              mine_is_low, out.data_ptr())

    def composite_ordered(self, first, count, parts, slot_of_box, grid, out):
        self.r.composite_ordered(self.width, self.height, first, count, parts.data_ptr(), slot_of_box.data_ptr(),
                                 grid, out.data_ptr())

    def decide(self, first, count, partials, out_rgba8, undecided, undecided_count):
        self.r.decide_partials(self.width, self.height, first, count, partials.data_ptr(), out_rgba8.data_ptr(),
                               undecided.data_ptr(), undecided_count.data_ptr())

    def chain_stage(self, pixels, colour_in, colour_out, stage, grid, box):
        self.r.chain_stage(self.width, self.height, pixels.data_ptr(), pixels.numel(), colour_in.data_ptr(),
                           colour_out.data_ptr(), stage, grid, box)

    def store_chain(self, pixels, colour, first, count, out_rgba8):
        self.r.store_chain(pixels.data_ptr(), colour.data_ptr(), pixels.numel(), first, count, out_rgba8.data_ptr())


class _NoScope:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class SortLastFrame:
    """Buffers and exchanges of one frame size for one rank.

    engine: the kernels (HipEngine on a GPU; the CPU tests plug in the checker).  Collectives run on
    `device` tensors unless host_staging is set (gloo with GPU tensors), in which case every exchanged
    buffer makes a round trip through host memory.
    """

    def __init__(self, engine, width, height, rank, world, device, grid=None, schedule="direct_send", group=None,
                 host_staging=False):
        self.e, self.width, self.height, self.rank, self.world = engine, width, height, rank, world
        self.device, self.group, self.schedule, self.staging = torch.device(device), group, schedule, host_staging
        self.grid = tuple(grid) if grid is not None else box_grid(world)
        assert self.grid[0] * self.grid[1] * self.grid[2] == world
        self.box = box_of_rank(rank, self.grid)
        # a single rank outside any process group needs no collectives; inside one it still issues them
        self.solo = world == 1 and not dist.is_initialized()
        self.pixels = width * height
        unit = world * 64
        self.padded = (self.pixels + unit - 1) // unit * unit
        self.span = self.padded // world
        if schedule == "binary_swap":
            self.rounds = swap_rounds(world, self.grid)
            self.first, _ = swap_span(rank, world, self.padded)
        elif schedule == "direct_send":
            self.first = rank * self.span
            # slot s of the all_to_all result is rank s's image; its box index is x-fastest == its rank
            self.slot_of_box = torch.arange(world, dtype=torch.int32, device=self.device)
        else:
            raise ValueError("schedule must be 'direct_send' or 'binary_swap'")
        f32 = dict(dtype=torch.float32, device=self.device)
        self.partial = torch.zeros((self.padded, PARTIAL_FLOATS), **f32)   # rows past `pixels` stay zero: misses
        self.work = [torch.zeros((self.padded // 2, PARTIAL_FLOATS), **f32) for _ in range(2)] \
            if schedule == "binary_swap" and world > 1 else None
        self.merged = torch.zeros((self.span, PARTIAL_FLOATS), **f32)
        self.bytes = torch.zeros((self.span, 4), dtype=torch.uint8, device=self.device)
        self.undecided = torch.zeros(self.span, dtype=torch.int32, device=self.device)
        self.undecided_count = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.frame = torch.zeros((self.padded, 4), dtype=torch.uint8, device=self.device) if rank == 0 else None
        self.remarched = 0      # pixels the last frame sent through step 4
        # the fills above ran on torch's current stream, the kernels that write these buffers run on the
        # context's own (non-blocking) stream: nothing else orders the two
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    # -- collectives, optionally staged through the host ------------------------------------------
    def _host(self, t):
        return t.cpu() if self.staging else t

    def _swap(self, send, partner):
        s = self._host(send.contiguous())
        r = torch.empty_like(s)
        ops = [dist.P2POp(dist.isend, s, partner, self.group), dist.P2POp(dist.irecv, r, partner, self.group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return r.to(self.device) if self.staging else r

    def _all_to_all(self, send):
        s = self._host(send)
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=self.group)
        r = r.view(self.world, self.span, PARTIAL_FLOATS)
        return r.to(self.device) if self.staging else r

    def _all_gather(self, t):
        s = self._host(t.contiguous().view(-1))
        out = torch.empty(self.world * s.numel(), dtype=s.dtype, device=s.device)
        dist.all_gather_into_tensor(out, s, group=self.group)
        out = out.view(self.world, -1)
        return out.to(self.device) if self.staging else out

    def _all_reduce_sum(self, t):
        if self.staging:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)

    # -- step 2 -------------------------------------------------------------------------------------
    def _merge_direct_send(self):
        if self.solo:
            self.e.composite_ordered(0, self.span, self.partial, self.slot_of_box, self.grid, self.merged)
            return
        parts = self._all_to_all(self.partial)          # [world, span, 5]: slot s = rank s's partials of my span
        self.e.composite_ordered(self.first, self.span, parts, self.slot_of_box, self.grid, self.merged)

    def _merge_binary_swap(self):
        held, first, count = self.partial, 0, self.padded
        for r, axis in enumerate(self.rounds):
            partner = self.rank ^ (1 << r)
            upper = (self.rank >> r) & 1
            count //= 2
            keep = count if upper else 0            # offsets inside `held`
            send = 0 if upper else count
            theirs = self._swap(held[send: send + count], partner)
            # a round reads `held` while it writes: results alternate between the two scratch buffers
            out = self.merged if count == self.span else self.work[r & 1][:count]
            self.e.composite_pair(first + keep, count, held[keep: keep + count], theirs, axis, not upper, out)
            held, first = out, first + keep
        if not self.rounds:
            self.merged.copy_(self.partial[: self.span])
        assert first == self.first and count == self.span

    # -- the frame ------------------------------------------------------------------------------------
    def render(self, mark=None):
        """Renders one frame with the engine's current camera; returns it (uint8 [height, width, 4],
        on `device`) on rank 0 and None elsewhere.  mark(name), if given, is called after each of the
        five stages has been enqueued (for timing with events on the engine's stream)."""
        mark = mark or (lambda name: None)
        scope = self.e.scope() if hasattr(self.e, "scope") else _NoScope()
        with scope:
            self.e.render_partial(self.partial)
            mark("partial_march")
            if self.schedule == "direct_send":
                self._merge_direct_send()
            else:
                self._merge_binary_swap()
            mark("exchange_merge")
            self.e.decide(self.first, self.span, self.merged, self.bytes, self.undecided, self.undecided_count)
            mark("decide")
            self._remarch()
            mark("remarch")
            frame = self._gather()
            mark("gather")
            return frame

    def _remarch(self):
        if self.solo:
            counts = self.undecided_count.view(1, 1)
            lists = self.undecided.view(1, -1)
        else:
            counts = self._all_gather(self.undecided_count)
        counts = [int(c) for c in counts.flatten().tolist()]
        self.remarched = sum(counts)
        if self.remarched == 0:
            return
        if not self.solo:
            longest = max(counts)
            lists = self._all_gather(self.undecided[:longest])
        pixels = torch.cat([lists[r, : counts[r]] for r in range(self.world)]).contiguous()
        colour = torch.zeros((pixels.numel(), 4), dtype=torch.float32, device=self.device)
        nxt = torch.empty_like(colour)
        for stage in range(self.world):
            self.e.chain_stage(pixels, colour, nxt, stage, self.grid, self.box)
            if not self.solo:
                self._all_reduce_sum(nxt)
            colour, nxt = nxt, colour
        self.e.store_chain(pixels, colour, self.first, self.span, self.bytes)

    def _gather(self):
        if self.solo:
            self.frame[: self.span].copy_(self.bytes)
        else:
            s = self._host(self.bytes)
            recv = None
            if self.rank == 0:
                recv = [torch.empty_like(s) for _ in range(self.world)]
            dist.gather(s, gather_list=recv, dst=0, group=self.group)
            if self.rank == 0:
                for r in range(self.world):
                    first = swap_span(r, self.world, self.padded)[0] if self.schedule == "binary_swap" else r * self.span
                    self.frame[first: first + self.span].copy_(recv[r])
        if self.rank == 0:
            return self.frame[: self.pixels].view(self.height, self.width, 4)
        return None


def upload_phantom_box(renderer, n, dtype_bytes, grid, box, device, seed=0x5EED1234, cuts=None):
    """Generates this rank's box of the n^3 synthetic phantom on its GPU and uploads it as the
    context's sub-volume (an even split of the brick grid unless `cuts` says otherwise).  The phantom's
    maximum is the dtype's (its brightest ellipsoid reaches it), so no reduction over the ranks is needed."""
    from . import brick_grid, subvolume_voxels
    lo, hi = cuts.box(box) if cuts is not None else brick_box(brick_grid(n, n, n), grid, box)
    v_lo, v_hi = subvolume_voxels(n, n, n, lo, hi)
    ext = [h - l for l, h in zip(v_lo, v_hi)]
    dtype = torch.uint8 if dtype_bytes == 1 else torch.int16      # raw bits; the kernels read them as unsigned
    raw = torch.empty((ext[2], ext[1], ext[0]), dtype=dtype, device=device)
    renderer.phantom_generate_box_device(raw.data_ptr(), dtype_bytes, n, v_lo, v_hi, seed)
    renderer.upload_subvolume_device(raw.data_ptr(), dtype_bytes, n, n, n, lo, hi, float(255 if dtype_bytes == 1 else 65535))
    renderer.synchronize()
    del raw
    return lo, hi


def upload_box_of(renderer, voxels, grid, box, device, max_value=None, cuts=None):
    """Cuts this rank's box out of a host array [nz, ny, nx] (uint8/uint16) and uploads it."""
    from . import brick_grid, subvolume_voxels
    voxels = np.asarray(voxels)
    nz, ny, nx = voxels.shape
    lo, hi = cuts.box(box) if cuts is not None else brick_box(brick_grid(nx, ny, nz), grid, box)
    v_lo, v_hi = subvolume_voxels(nx, ny, nz, lo, hi)
    sub = np.ascontiguousarray(voxels[v_lo[2]: v_hi[2], v_lo[1]: v_hi[1], v_lo[0]: v_hi[0]])
    if max_value is None:
        max_value = float(voxels.max())       # main.cpp:144: the maximum of the whole volume
    if voxels.dtype == np.uint16:
        t = torch.from_numpy(sub.view(np.int16)).to(device)
    else:
        t = torch.from_numpy(sub).to(device)
    torch.cuda.synchronize()      # the copy ran on torch's stream, the upload runs on the context's
    renderer.upload_subvolume_device(t.data_ptr(), voxels.dtype.itemsize, nx, ny, nz, lo, hi, max_value)
    renderer.synchronize()
    return lo, hi


def occupied_counts(renderer, n_bricks, cell_bricks, device, group=None, host_staging=False):
    """Occupied bricks per cube of cell_bricks^3 bricks, summed over the ranks' boxes: [cz, cy, cx] (numpy).
    The renderer must hold its (first, e.g. even) box and the transfer function the cut is for."""
    dims = [-(-b // cell_bricks) for b in n_bricks]
    counts = torch.zeros((dims[2], dims[1], dims[0]), dtype=torch.int32, device=device)
    torch.cuda.synchronize(device)
    renderer.occupied_brick_counts(cell_bricks, counts.data_ptr())
    renderer.synchronize()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if host_staging:
            h = counts.cpu()
            dist.all_reduce(h, group=group)
            return h.numpy()
        dist.all_reduce(counts, group=group)
    return counts.cpu().numpy()
