"""Sort-last multi-rank path on CPU over gloo: the exchange schedules (direct send, binary swap),
the per-pixel visibility order and the box-to-box re-march of undecided pixels, with the checker
standing in for the GPU kernels (the GPU tests pin those kernels to the checker).  The assembled
frame must equal the single-process frame byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import conftest as cf


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, width, height, samples, schedule, grid, camera, result_path, balanced=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vr_oracle as o
        from sortlast_cpu_engine import OracleEngine
        from vulkanrenderer_amd.sortlast import SortLastFrame, balanced_cuts, box_of_rank, brick_box

        vol, _ = o.normalize(o.phantom(n, np.uint8))
        lut = np.zeros((256, 4), np.float32)
        o.lut_update(lut, cf.TF_GRAPHS)
        vp, m = o.camera(*cf.CAMERAS[camera], width, height)
        bricks = ((n + 4 + 7) // 8,) * 3
        lo, hi = brick_box(bricks, grid, box_of_rank(rank, grid))
        if balanced:
            # uneven boxes: planes that balance the voxels above the table's threshold, per cube of 1^3 bricks
            vox = o.phantom(n, np.uint8)
            pad = np.zeros((bricks[2] * 8, bricks[1] * 8, bricks[0] * 8), np.uint8)
            pad[2:2 + n, 2:2 + n, 2:2 + n] = vox
            occ = (pad.reshape(bricks[2], 8, bricks[1], 8, bricks[0], 8) > 60).any(axis=(1, 3, 5)).astype(np.float64)
            occ[:, bricks[1] // 2:, : bricks[0] // 2] *= 5.0          # and lean to one side, so that the planes move
            cuts = balanced_cuts(occ, 1, bricks, grid)
            lo, hi = cuts.box(box_of_rank(rank, grid))
        sl = SortLastFrame(None, width, height, rank, world, "cpu", grid=grid, schedule=schedule)
        sl.e = OracleEngine(vol, lut, vp[4:8], m, width, height, samples, lo, hi, sl.padded)
        frame = sl.render()
        if rank == 0:
            full = o.render(vol, lut, vp[4:8], m, width, height, samples, threads=1)
            np.save(result_path, np.stack([frame.numpy(), full]))
            assert 0 < sl.remarched < width * height
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


CASES = [
    (2, "binary_swap", (2, 1, 1), "h30_v20"),
    (2, "direct_send", (1, 1, 2), "hm135_vm60"),
    (3, "direct_send", (1, 3, 1), "h30_v20"),
    (4, "binary_swap", (2, 2, 1), "hm135_vm60"),
    (8, "binary_swap", (2, 2, 2), "h30_v20"),
    (8, "direct_send", (2, 2, 2), "hm135_vm60"),
]


@pytest.mark.parametrize("world,schedule,grid,camera", CASES)
def test_sort_last_frame_equals_the_single_process_frame(tmp_path, world, schedule, grid, camera):
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), 32, 40, 36, 50, schedule, grid, camera, path), nprocs=world, join=True)
    got, want = np.load(path)
    assert want[..., :3].any()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("world,schedule,grid,camera", [(8, "binary_swap", (2, 2, 2), "hm135_vm60"), (8, "direct_send", (2, 2, 2), "h30_v20"),
                                                        (4, "binary_swap", (4, 1, 1), "h30_v20"), (6, "direct_send", (3, 2, 1), "top")])
def test_boxes_of_unequal_size(tmp_path, world, schedule, grid, camera):
    """Boxes cut by planes that balance the occupied bricks (a k-d split: z planes, per layer y planes, per
    cell x planes): both schedules and the re-march still give the single-process frame."""
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), 40, 40, 36, 50, schedule, grid, camera, path, True), nprocs=world, join=True)
    got, want = np.load(path)
    assert want[..., :3].any()
    assert np.array_equal(got, want)


def test_balanced_cuts_tile_the_grid():
    from vulkanrenderer_amd.sortlast import balanced_cuts, box_of_rank, brick_box, even_cuts
    rng = np.random.default_rng(3)
    bricks, cell = (65, 66, 70), 8
    counts = rng.integers(0, 50, size=(9, 9, 9)).astype(np.float64)
    counts[:, 5:, :] *= 6.0
    for grid in ((2, 2, 2), (4, 2, 1), (3, 1, 2), (1, 1, 1), (2, 1, 1)):
        cuts = balanced_cuts(counts, cell, bricks, grid)
        covered = np.zeros(bricks[::-1], np.int32)
        for r in range(grid[0] * grid[1] * grid[2]):
            lo, hi = cuts.box(box_of_rank(r, grid))
            assert all(h > l for l, h in zip(lo, hi))
            covered[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]] += 1
        assert (covered == 1).all(), grid
    # the heavy half gets the thinner boxes
    cuts = balanced_cuts(counts, cell, bricks, (1, 2, 1))
    (lo0, hi0), (lo1, hi1) = cuts.box((0, 0, 0)), cuts.box((0, 1, 0))
    assert hi0[1] - lo0[1] > hi1[1] - lo1[1]
    e = even_cuts((9, 17, 33), (2, 2, 2))
    for r in range(8):
        assert e.box(box_of_rank(r, (2, 2, 2))) == brick_box((9, 17, 33), (2, 2, 2), box_of_rank(r, (2, 2, 2)))
    with pytest.raises(ValueError):
        balanced_cuts(counts[:1], cell, (65, 66, 8), (1, 1, 2))


def test_decomposition_arithmetic():
    from vulkanrenderer_amd.sortlast import box_grid, box_of_rank, brick_box, swap_rounds, swap_span
    assert box_grid(8) == (2, 2, 2) and box_grid(4) == (2, 2, 1) and box_grid(2) == (2, 1, 1)
    assert box_grid(1) == (1, 1, 1) and box_grid(6) == (3, 2, 1) and box_grid(7) == (7, 1, 1)
    for world in (1, 2, 3, 4, 6, 8):
        grid = box_grid(world)
        bricks = (9, 17, 33)
        covered = np.zeros(bricks[::-1], np.int32)
        for r in range(world):
            lo, hi = brick_box(bricks, grid, box_of_rank(r, grid))
            covered[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]] += 1
        assert (covered == 1).all()              # the boxes tile the brick grid
    for world in (1, 2, 4, 8):
        padded = world * 64 * 5
        spans = sorted(swap_span(r, world, padded) for r in range(world))
        assert [s[0] for s in spans] == [k * padded // world for k in range(world)]
        assert len(swap_rounds(world, box_grid(world))) == world.bit_length() - 1
    with pytest.raises(ValueError):
        swap_rounds(6, (3, 2, 1))
    with pytest.raises(ValueError):
        brick_box((1, 9, 9), (2, 1, 1), (0, 0, 0))
