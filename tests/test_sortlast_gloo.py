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


def _worker(rank, world, port, n, width, height, samples, schedule, grid, camera, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vr_oracle as o
        from sortlast_cpu_engine import OracleEngine
        from vulkanrenderer_amd.sortlast import SortLastFrame, box_of_rank, brick_box

        vol, _ = o.normalize(o.phantom(n, np.uint8))
        lut = np.zeros((256, 4), np.float32)
        o.lut_update(lut, cf.TF_GRAPHS)
        vp, m = o.camera(*cf.CAMERAS[camera], width, height)
        bricks = ((n + 4 + 7) // 8,) * 3
        lo, hi = brick_box(bricks, grid, box_of_rank(rank, grid))
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
