"""Sort-first multi-rank path on CPU: world_size 2 and 3 over gloo.

Each rank renders its 16-row bands with the CPU checker (standing in for the GPU kernel, which
the GPU tests pin to the checker), the bands are gathered on rank 0 and assembled; the result must
be the single-process frame.  Exercises the band arithmetic exported by the C ABI (vr_band_rows),
its Python mirror, the padded gather and the row permutation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import conftest as cf


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, width, height, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vulkanrenderer_amd as vr
        from oracle import vr_oracle as o
        from vulkanrenderer_amd.sortfirst import SortFirstFrame

        vol, _ = o.normalize(o.phantom(24, np.uint8))
        lut = np.zeros((256, 4), np.float32)
        o.lut_update(lut, cf.TF_GRAPHS)
        vp, m = o.camera(*cf.CAMERAS["h30_v20"], width, height)
        full = o.render(vol, lut, vp[4:8], m, width, height, 40, threads=1)

        sf = SortFirstFrame(width, height, rank, world, "cpu")
        rows = vr.band_row_indices(height, rank, world)
        assert rows.size == sf.rows == vr.band_rows(height, rank, world)
        sf.local_view().copy_(torch.from_numpy(full[rows]))      # this rank's bands, densely packed
        frame = sf.exchange()
        if rank == 0:
            np.save(result_path, np.stack([frame.numpy(), full]))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height", [(2, 40, 50), (3, 33, 70), (2, 24, 16)])
def test_sort_first_gather_assembles_the_frame(tmp_path, world, width, height):
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, path), nprocs=world, join=True)
    got, want = np.load(path)
    assert np.array_equal(got, want)
    assert want[..., :3].any()


def test_band_partition_covers_every_row_once():
    import vulkanrenderer_amd as vr
    for height in (1, 15, 16, 17, 50, 2160):
        for parts in (1, 2, 3, 8):
            seen = np.concatenate([vr.band_row_indices(height, p, parts) for p in range(parts)])
            assert sorted(seen.tolist()) == list(range(height))
            for p in range(parts):
                assert vr.band_rows(height, p, parts) == vr.band_row_indices(height, p, parts).size
