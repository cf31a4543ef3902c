"""GPU tests of the sort-last path: every kernel is reached through the C ABI (libvr_hip.so).

One GPU stands in for several: each box of the split gets a context of its own on the same card.
The frame assembled from the boxes must equal, byte for byte, the frame a single context renders
from the whole volume, which the other GPU tests pin to the checker; the per-box partial results
are compared with the checker's directly."""
import math
import os
import socket

import numpy as np
import pytest

import conftest as cf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vr():
    import vulkanrenderer_amd as m
    return m


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available()
    return t


def make_boxes(vr, torch, vox, grid, graphs=cf.TF_GRAPHS, samples=60, border=(0, 0, 0, 0)):
    """One context per box of `grid`, each holding its part of the host volume `vox`."""
    from vulkanrenderer_amd import sortlast as sl
    ctxs = []
    for rank in range(grid[0] * grid[1] * grid[2]):
        c = vr.VolumeRenderer(0)
        sl.upload_box_of(c, vox, grid, sl.box_of_rank(rank, grid), "cuda")
        c.update_transfer_function(graphs, reset_state=True)
        c.set_sampling(samples, border)
        ctxs.append(c)
    return ctxs


def sort_last_frame(vr, torch, ctxs, grid, w, h, schedule="ordered"):
    """The four steps of a sort-last frame, the contexts taking turns on one GPU."""
    from vulkanrenderer_amd import sortlast as sl
    world = len(ctxs)
    n = w * h
    f32 = dict(dtype=torch.float32, device="cuda")
    parts = torch.zeros((world, n, 5), **f32)
    merged = torch.zeros((n, 5), **f32)
    rgba = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    todo = torch.zeros(n, dtype=torch.int32, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()        # torch's stream and the contexts' streams are ordered by hand here
    for r, c in enumerate(ctxs):
        c.render_partial(w, h, parts[r].data_ptr())
        c.synchronize()
    c0 = ctxs[0]
    if schedule == "ordered":
        slots = torch.arange(world, dtype=torch.int32, device="cuda")
        c0.composite_ordered(w, h, 0, n, parts.data_ptr(), slots.data_ptr(), grid, merged.data_ptr())
    else:
        # pairwise merges in binary-swap order, all pixels at every node
        held = [parts[r] for r in range(world)]
        for rnd, axis in enumerate(sl.swap_rounds(world, grid)):
            nxt = {}
            for r in range(world):
                if r & ((2 << rnd) - 1):       # only the lowest rank of each merged group carries on
                    continue
                out = torch.empty((n, 5), **f32)
                c0.composite_pair(w, h, 0, n, held[r].data_ptr(), held[r | (1 << rnd)].data_ptr(), axis, 1, out.data_ptr())
                nxt[r] = out
            held = nxt
        c0.synchronize()
        merged = held[0]
    c0.decide_partials(w, h, 0, n, merged.data_ptr(), rgba.data_ptr(), todo.data_ptr(), count.data_ptr())
    c0.synchronize()
    m = int(count.item())
    if m:
        pixels = todo[:m].contiguous()
        colour = torch.zeros((m, 4), **f32)
        for stage in range(world):
            total = torch.zeros((m, 4), **f32)
            torch.cuda.synchronize()
            for r, c in enumerate(ctxs):
                out = torch.empty((m, 4), **f32)
                c.chain_stage(w, h, pixels.data_ptr(), m, colour.data_ptr(), out.data_ptr(), stage, grid, sl.box_of_rank(r, grid))
                c.synchronize()
                total += out
            torch.cuda.synchronize()
            colour = total
        c0.store_chain(pixels.data_ptr(), colour.data_ptr(), m, 0, n, rgba.data_ptr())
        c0.synchronize()
    return rgba.cpu().numpy().reshape(h, w, 4), m, parts.cpu().numpy(), merged.cpu().numpy()


def whole_frame(vr, vox, cam, w, h, samples, graphs=cf.TF_GRAPHS, border=(0, 0, 0, 0), mode=1):
    with vr.VolumeRenderer(0) as c:
        c.upload_volume(vox)
        c.update_transfer_function(graphs, reset_state=True)
        c.set_sampling(samples, border)
        c.set_camera(*cam)
        c.set_mode(mode)
        return c.render(w, h)


@pytest.mark.parametrize("dtype,n,grid,schedule", [
    (np.uint16, 48, (2, 2, 2), "ordered"),
    (np.uint16, 48, (2, 2, 2), "pairs"),
    (np.uint8, 40, (3, 1, 2), "ordered"),
    (np.uint8, 40, (4, 1, 1), "pairs"),
    (np.uint16, 37, (1, 1, 1), "ordered"),
])
def test_boxes_assemble_the_single_context_frame(vr, torch, oracle, dtype, n, grid, schedule):
    vox = oracle.phantom(n, dtype)
    ctxs = make_boxes(vr, torch, vox, grid, samples=90)
    try:
        vol, _ = oracle.normalize(vox)
        lut = np.zeros((256, 4), np.float32)
        oracle.lut_update(lut, cf.TF_GRAPHS)
        for cname, cam in cf.CAMERAS.items():
            for c in ctxs:
                c.set_camera(*cam)
            w, h = 72, 56
            got, remarched, parts, merged = sort_last_frame(vr, torch, ctxs, grid, w, h, schedule)
            want = whole_frame(vr, vox, cam, w, h, 90)
            assert np.array_equal(got, want), (cname, cf.frame_report(got, want))
            vp, m = oracle.camera(cam[0], cam[1], w, h)
            assert np.array_equal(got, oracle.render(vol, lut, vp[4:8], m, w, h, 90)), cname
            assert remarched < 0.05 * w * h
            # the partial results themselves, box by box, against the checker's
            from vulkanrenderer_amd import sortlast as sl
            bricks = vr.brick_grid(n, n, n)
            for r in range(len(ctxs)):
                lo, hi = sl.brick_box(bricks, grid, sl.box_of_rank(r, grid))
                ref = oracle.render_partial(vol, lut, vp[4:8], m, w, h, 90, lo, hi)
                # a box drops a ray once its own taps made it darker than 2^-20; where exactly differs by a few taps
                assert np.allclose(parts[r], ref, rtol=2e-6, atol=3e-6), (cname, r, np.abs(parts[r] - ref).max())
            assert np.all(merged[:, 3] <= 2.0 ** -20)    # every hit ray reached its back tap's owner or went dark
    finally:
        for c in ctxs:
            c.close()


def test_tables_the_interval_argument_does_not_cover(vr, torch, oracle):
    # alpha above 1: partial sums mean nothing, every hit pixel must go through the box-to-box re-march
    graphs = [cf.TF_GRAPHS[0], cf.TF_GRAPHS[1], cf.TF_GRAPHS[2], [(0, 0.0), (120, 0.0), (300, 0.4), (483, 1.7)]]
    vox = oracle.phantom(32, np.uint8)
    grid = (2, 1, 2)
    ctxs = make_boxes(vr, torch, vox, grid, graphs=graphs, samples=50)
    try:
        cam = cf.CAMERAS["h30_v20"]
        for c in ctxs:
            c.set_camera(*cam)
        got, remarched, _, _ = sort_last_frame(vr, torch, ctxs, grid, 48, 40)
        vol, _ = oracle.normalize(vox)
        lut = np.zeros((256, 4), np.float32)
        oracle.lut_update(lut, graphs)
        vp, m = oracle.camera(cam[0], cam[1], 48, 40)
        assert remarched == oracle.count_hits(vp[4:8], m, 48, 40)
        assert np.array_equal(got, oracle.render(vol, lut, vp[4:8], m, 48, 40, 50))
    finally:
        for c in ctxs:
            c.close()


def test_opaque_border_and_noise_volume(vr, torch, oracle):
    rng = np.random.default_rng(5)
    vox = rng.integers(0, 3000, size=(19, 33, 26)).astype(np.uint16)
    grid = (2, 2, 1)
    for border in ((0, 0, 0, 1), (0.2, 0.4, 0.6, 0.3)):
        ctxs = make_boxes(vr, torch, vox, grid, samples=70, border=border)
        try:
            cam = cf.CAMERAS["hm135_vm60"]
            for c in ctxs:
                c.set_camera(*cam)
            got, _, _, _ = sort_last_frame(vr, torch, ctxs, grid, 64, 48)
            vol, _ = oracle.normalize(vox)
            lut = np.zeros((256, 4), np.float32)
            oracle.lut_update(lut, cf.TF_GRAPHS)
            vp, m = oracle.camera(cam[0], cam[1], 64, 48)
            assert np.array_equal(got, oracle.render(vol, lut, vp[4:8], m, 64, 48, 70, border=border)), border
        finally:
            for c in ctxs:
                c.close()


def test_phantom_box_equals_crop_of_the_phantom(vr, torch, oracle):
    n = 45
    full = oracle.phantom(n, np.uint16)
    with vr.VolumeRenderer(0) as c:
        for lo, hi in (((0, 0, 0), (n, n, n)), ((3, 10, 17), (40, 11, 45)), ((44, 44, 44), (45, 45, 45))):
            ext = [h - l for l, h in zip(lo, hi)]
            t = torch.zeros((ext[2], ext[1], ext[0]), dtype=torch.int16, device="cuda")
            c.phantom_generate_box_device(t.data_ptr(), 2, n, lo, hi)
            c.synchronize()
            got = t.cpu().numpy().view(np.uint16)
            assert np.array_equal(got, full[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]), (lo, hi)


def test_sub_volume_contexts_refuse_whole_frame_calls(vr, torch, oracle):
    vox = oracle.phantom(24, np.uint8)
    ctxs = make_boxes(vr, torch, vox, (2, 1, 1))
    try:
        with pytest.raises(vr.VrError):
            ctxs[0].render(32, 32)
        with pytest.raises(vr.VrError):
            ctxs[0].read_volume_normalized()
        t = torch.zeros(1024, dtype=torch.uint8, device="cuda")
        with pytest.raises(vr.VrError):      # brick box outside the grid
            ctxs[0].upload_subvolume_device(t.data_ptr(), 1, 24, 24, 24, (0, 0, 0), (9, 1, 1), 255.0)
        with pytest.raises(vr.VrError):      # the whole volume's maximum is required
            ctxs[0].upload_subvolume_device(t.data_ptr(), 1, 24, 24, 24, (0, 0, 0), (1, 1, 1), 0.0)
    finally:
        for c in ctxs:
            c.close()
    assert vr.brick_grid(24, 24, 24) == (4, 4, 4) and vr.brick_grid(2048, 2048, 2048) == (257, 257, 257)
    assert vr.subvolume_voxels(24, 24, 24, (0, 1, 3), (1, 3, 4)) == ((0, 6, 22), (7, 23, 24))


def test_tune_on_a_box_keeps_its_partial_image(vr, torch, oracle):
    """vr_tune on a context that holds a box picks the lane budget of the partial march by timing.  The
    partial image depends on it only in the last bits (which march sums a ray's taps, and the tap at which
    a dark ray is dropped); a split into bands is refused there."""
    vox = oracle.phantom(48, np.uint16)
    ctxs = make_boxes(vr, torch, vox, (2, 1, 1), samples=150)
    try:
        c = ctxs[1]
        c.set_camera(*cf.CAMERAS["h30_v20"])
        before = c.render_partial(160, 120)
        assert c.tune(160, 120) in (48, 64, 80, 96, 112, 128, 160)
        assert np.allclose(c.render_partial(160, 120), before, rtol=2e-6, atol=3e-6)
        with pytest.raises(vr.VrError):
            c.tune(160, 120, 0, 2)
    finally:
        for c in ctxs:
            c.close()


# ---- several processes on one GPU, exchanges over gloo through host memory ----------------------

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, width, height, samples, schedule, grid, cam, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vulkanrenderer_amd as vr
        from oracle import vr_oracle as o
        from vulkanrenderer_amd import sortlast as sl
        torch.cuda.set_device(0)
        with vr.VolumeRenderer(0) as c:
            sl.upload_phantom_box(c, n, 2, grid, sl.box_of_rank(rank, grid), "cuda")
            c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            c.set_sampling(samples)
            c.set_camera(*cam)
            frame = sl.SortLastFrame(sl.HipEngine(c, width, height), width, height, rank, world, "cuda", grid=grid,
                                     schedule=schedule, host_staging=True)
            out = frame.render()
            c.synchronize()
            if rank == 0:
                np.save(result_path, out.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,schedule,grid", [(2, "binary_swap", (1, 2, 1)), (4, "direct_send", (2, 1, 2)),
                                                 (8, "binary_swap", (2, 2, 2))])
def test_processes_sharing_one_gpu(vr, torch, oracle, tmp_path, world, schedule, grid):
    import torch.multiprocessing as mp
    n, w, h, samples = 64, 96, 80, 120
    cam = (math.radians(-70.0), math.radians(35.0))
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, w, h, samples, schedule, grid, cam, path), nprocs=world, join=True)
    got = np.load(path)
    want = whole_frame(vr, oracle.phantom(n, np.uint16), cam, w, h, samples)
    assert want[..., :3].any()
    assert np.array_equal(got, want), cf.frame_report(got, want)


@pytest.mark.parametrize("schedule", ["direct_send", "binary_swap"])
def test_single_rank_over_rccl(vr, torch, oracle, schedule):
    """The driver with device tensors in RCCL collectives (one rank: every collective is issued, on the
    context's stream, and degenerates to a copy)."""
    import torch.distributed as dist
    from vulkanrenderer_amd import sortlast as sl
    n, w, h, samples = 48, 80, 60, 100
    cam = cf.CAMERAS["h30_v20"]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        with vr.VolumeRenderer(0) as c:
            sl.upload_phantom_box(c, n, 2, (1, 1, 1), (0, 0, 0), "cuda")
            c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            c.set_sampling(samples)
            c.set_camera(*cam)
            frame = sl.SortLastFrame(sl.HipEngine(c, w, h), w, h, 0, 1, "cuda", schedule=schedule)
            assert not frame.solo
            got = frame.render()
            c.synchronize()
            got = got.cpu().numpy()
    finally:
        dist.destroy_process_group()
    want = whole_frame(vr, oracle.phantom(n, np.uint16), cam, w, h, samples)
    assert np.array_equal(got, want), cf.frame_report(got, want)


def test_multi_gpu_layer_behind_the_c_abi(vr, torch, oracle):
    """libvr_mg.so with one rank: every RCCL call of the sort-first and the sort-last frame is issued (sends
    to itself, all-gathers and all-reduces over one rank) and the frames equal the single-context frame."""
    from vulkanrenderer_amd import multigpu as mg
    from vulkanrenderer_amd import sortlast as sl
    n, w, h, samples = 64, 200, 150, 160
    cam = cf.CAMERAS["hm135_vm60"]
    vox = oracle.phantom(n, np.uint16)
    want = whole_frame(vr, vox, cam, w, h, samples)
    frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with vr.VolumeRenderer(0) as c:
        c.upload_volume(vox)
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        c.set_camera(*cam)
        c.set_mode(vr.MODE_FAST)
        with mg.MultiGpu(c, mg.unique_id(), 0, 1) as m:
            for pipeline in (1, 2):
                c.set_pipeline(pipeline)
                frame.zero_()
                torch.cuda.synchronize()
                m.render_sort_first(w, h, frame.data_ptr())
                m.synchronize()
                assert np.array_equal(frame.cpu().numpy(), want), pipeline
                # five frames of an orbit back to back into three buffers (the gathers run beside the next marches,
                # band buffers in turn); the host waits once, torch's stream through vrmg_wait_frame
                cams = [(cam[0] + 0.2 * k, cam[1] + 0.05 * k) for k in range(5)]
                singles = []
                for k in (2, 3, 4):
                    c.set_camera(*cams[k])
                    singles.append(c.render(w, h))
                bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(3)]
                torch.cuda.synchronize()
                for k in range(5):
                    c.set_camera(*cams[k])
                    m.render_sort_first(w, h, bufs[k % 3].data_ptr())
                m.wait_frame(torch.cuda.current_stream().cuda_stream)
                last = bufs[4 % 3].clone()          # on torch's stream, which now follows the last gather
                m.synchronize()
                torch.cuda.synchronize()
                assert np.array_equal(last.cpu().numpy(), singles[2]), pipeline
                assert np.array_equal(bufs[2 % 3].cpu().numpy(), singles[0]) and np.array_equal(bufs[3 % 3].cpu().numpy(), singles[1]), pipeline
                c.set_camera(*cam)
            with pytest.raises(vr.VrError):
                m.render_sort_last(w, h, mg.DIRECT_SEND, frame.data_ptr())       # no box grid yet
    with vr.VolumeRenderer(0) as c:
        sl.upload_box_of(c, vox, (1, 1, 1), (0, 0, 0), "cuda")
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        c.set_camera(*cam)
        with mg.MultiGpu(c, mg.unique_id(), 0, 1) as m:
            with pytest.raises(vr.VrError):
                m.set_box_grid((2, 1, 1))
            m.set_box_grid((1, 1, 1))
            for schedule in (mg.DIRECT_SEND, mg.BINARY_SWAP):
                frame.zero_()
                torch.cuda.synchronize()
                remarched = m.render_sort_last(w, h, schedule, frame.data_ptr())
                m.synchronize()
                assert np.array_equal(frame.cpu().numpy(), want), schedule
                assert 0 < remarched < 0.05 * w * h
            # a larger frame after a smaller one: the buffers grow (and the ranks agree on it) in the middle of a run
            w2, h2 = 333, 251
            want2 = whole_frame(vr, vox, cam, w2, h2, samples)
            big = torch.zeros((h2, w2, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for schedule in (mg.BINARY_SWAP, mg.DIRECT_SEND):
                m.render_sort_last(w2, h2, schedule, big.data_ptr())
                m.synchronize()
                assert np.array_equal(big.cpu().numpy(), want2), schedule


def test_multi_gpu_layer_reports_failures(vr, torch, oracle, monkeypatch):
    """An allocation that fails while a frame's buffers grow (sort-first bands, sort-last frame buffers, the
    re-march's lists): the call reports it after the ranks' agreement, nothing is left half sized, and the
    same frame goes through once memory is there.  One rank: the agreement is an all-reduce over itself."""
    from vulkanrenderer_amd import multigpu as mg
    from vulkanrenderer_amd import sortlast as sl
    n, w, h, samples = 64, 200, 150, 160
    cam = cf.CAMERAS["hm135_vm60"]
    vox = oracle.phantom(n, np.uint16)
    want = whole_frame(vr, vox, cam, w, h, samples)
    frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def refused(nth, call):
        monkeypatch.setenv("VRMG_TEST_FAIL_ALLOC", str(nth))
        try:
            with pytest.raises(vr.VrError) as info:
                call()
        finally:
            monkeypatch.delenv("VRMG_TEST_FAIL_ALLOC")
        return str(info.value)

    with vr.VolumeRenderer(0) as c:
        c.upload_volume(vox)
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        c.set_camera(*cam)
        c.set_mode(vr.MODE_FAST)
        with mg.MultiGpu(c, mg.unique_id(), 0, 1) as m:
            for nth in (2, 1):                                   # the second of three band buffers, twice (the others have grown)
                assert "memory" in refused(nth, lambda: m.render_sort_first(w, h, frame.data_ptr()))
                m.synchronize()
            m.render_sort_first(w, h, frame.data_ptr())
            m.synchronize()
            assert np.array_equal(frame.cpu().numpy(), want)
    with vr.VolumeRenderer(0) as c:
        sl.upload_box_of(c, vox, (1, 1, 1), (0, 0, 0), "cuda")
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        c.set_camera(*cam)
        with mg.MultiGpu(c, mg.unique_id(), 0, 1) as m:
            m.set_box_grid((1, 1, 1))
            # what has grown stays: direct send first fails on the partial image, then (with it in place) on the first
            # buffer of the re-march; binary swap after it only needs its two work buffers
            for schedule, nths in ((mg.DIRECT_SEND, (1, 2)), (mg.BINARY_SWAP, (2, 1))):
                for nth in nths:
                    msg = refused(nth, lambda: m.render_sort_last(w, h, schedule, frame.data_ptr()))
                    print("schedule %d, allocation %d refused: %s" % (schedule, nth, msg))
                    assert "memory" in msg, msg
                    m.synchronize()
                frame.zero_()
                torch.cuda.synchronize()
                assert m.render_sort_last(w, h, schedule, frame.data_ptr()) > 0
                m.synchronize()
                assert np.array_equal(frame.cpu().numpy(), want), schedule


def test_multi_gpu_example_program(tmp_path):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "multi_gpu")
    if not os.path.exists(exe):
        pytest.fail("examples/multi_gpu is not built: make -C vulkanrenderer_amd/csrc")
    res = subprocess.run([exe, "--rank", "0", "--world", "1", "--id-file", str(tmp_path / "id"), "--phantom", "64", "--size", "160x120",
                          "--out", str(tmp_path / "frame.ppm")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("identical to the single-GPU frame") == 3, res.stdout
    assert os.path.getsize(str(tmp_path / "frame.ppm")) > 160 * 120 * 3


def test_occupied_brick_counts(vr, torch, oracle):
    """vr_occupied_brick_counts: the boxes' counts add up to the whole volume's, which equal numpy's count of the
    bricks whose value range reaches the table's first non-transparent entry."""
    from vulkanrenderer_amd import sortlast as sl
    n, cell = 72, 2
    vox = oracle.phantom(n, np.uint16)
    nb = vr.brick_grid(n, n, n)
    with vr.VolumeRenderer(0) as c:
        c.upload_volume(vox)
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        whole = sl.occupied_counts(c, nb, cell, "cuda")
    total = np.zeros_like(whole)
    grid = (2, 2, 1)
    for r in range(4):
        with vr.VolumeRenderer(0) as c:
            sl.upload_box_of(c, vox, grid, sl.box_of_rank(r, grid), "cuda")
            c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            total += sl.occupied_counts(c, nb, cell, "cuda")
    assert np.array_equal(total, whole) and whole.sum() > 0
    assert whole.shape == tuple(-(-b // cell) for b in nb[::-1])
    # the exterior of the phantom is exactly zero and transparent: whole corners of the grid are empty
    assert whole[0, 0, 0] == 0 and whole[whole.shape[0] // 2, whole.shape[1] // 2, whole.shape[2] // 2] > 0
    cuts = sl.balanced_cuts(whole, cell, nb, (2, 2, 2))
    ctxs = []
    try:
        for r in range(8):
            c = vr.VolumeRenderer(0)
            sl.upload_box_of(c, vox, (2, 2, 2), sl.box_of_rank(r, (2, 2, 2)), "cuda", cuts=cuts)
            c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            c.set_sampling(150)
            c.set_camera(*cf.CAMERAS["h30_v20"])
            ctxs.append(c)
        got, _, _, _ = sort_last_frame(vr, torch, ctxs, (2, 2, 2), 96, 72)
    finally:
        for c in ctxs:
            c.close()
    assert np.array_equal(got, whole_frame(vr, vox, cf.CAMERAS["h30_v20"], 96, 72, 150))


def test_full_size_split_equals_whole(vr, torch):
    """1024^3 u16 at 1920x1080, eight boxes: the frame equals the single-context one."""
    from vulkanrenderer_amd import sortlast as sl
    n, w, h = 1024, 1920, 1080
    samples = int(math.ceil(2 * math.sqrt(3) * n)) - 1
    cam = (math.radians(40.0), math.radians(20.0))
    grid = (2, 2, 2)
    ctxs = []
    try:
        for r in range(8):
            c = vr.VolumeRenderer(0)
            sl.upload_phantom_box(c, n, 2, grid, sl.box_of_rank(r, grid), "cuda")
            c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            c.set_sampling(samples)
            c.set_camera(*cam)
            ctxs.append(c)
        got, remarched, _, _ = sort_last_frame(vr, torch, ctxs, grid, w, h)
    finally:
        for c in ctxs:
            c.close()
    with vr.VolumeRenderer(0) as c:
        raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
        c.phantom_generate_device(raw.data_ptr(), 2, n)
        c.upload_volume_device(raw.data_ptr(), 2, n, n, n)
        del raw
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        c.set_camera(*cam)
        c.set_mode(vr.MODE_FAST)
        want = c.render(w, h)
    print("re-marched pixels: %d of %d" % (remarched, w * h))
    assert want[..., :3].any()
    assert np.array_equal(got, want), cf.frame_report(got, want)
    assert remarched < 0.02 * w * h


# ---- the sort-last configuration at its own size: every box of the split resident on one GPU ------------

class BoxesOnOneGpu:
    """The boxes of an n^3 phantom as separate contexts on one GPU, and the steps of a sort-last frame over
    them with the buffers kept on the device (4K frames: 1.3 GB of partial images)."""

    def __init__(self, vr, torch, n, grid, samples, balance="even"):
        from vulkanrenderer_amd import sortlast as sl
        self.vr, self.torch, self.sl, self.n, self.grid = vr, torch, sl, n, tuple(grid)
        self.world = grid[0] * grid[1] * grid[2]
        self.ctxs = []
        self.cuts = None
        for r in range(self.world):
            c = vr.VolumeRenderer(0)
            self.ctxs.append(c)
            sl.upload_phantom_box(c, n, 2, self.grid, sl.box_of_rank(r, self.grid), "cuda")
            torch.cuda.empty_cache()
            c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            c.set_sampling(samples)
        if balance == "occupied":
            # what bench.py --balance occupied does over the ranks, here over the contexts: count the occupied
            # bricks of the even boxes, cut where they balance, load the boxes of that split instead
            nb = vr.brick_grid(n, n, n)
            cell = max(1, min(8, min(b // (2 * g) for b, g in zip(nb, self.grid))))
            counts = sum(sl.occupied_counts(c, nb, cell, "cuda") for c in self.ctxs)
            self.cuts = sl.balanced_cuts(counts, cell, nb, self.grid)
            for r, c in enumerate(self.ctxs):
                sl.upload_phantom_box(c, n, 2, self.grid, sl.box_of_rank(r, self.grid), "cuda", cuts=self.cuts)
                torch.cuda.empty_cache()

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []
        self.torch.cuda.empty_cache()

    def set_camera(self, cam):
        for c in self.ctxs:
            c.set_camera(*cam)

    def partials(self, w, h):
        torch = self.torch
        parts = torch.zeros((self.world, w * h, 5), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for r, c in enumerate(self.ctxs):
            c.render_partial(w, h, parts[r].data_ptr())
            c.synchronize()
        return parts

    def merge(self, parts, w, h, schedule):
        torch, c0, n = self.torch, self.ctxs[0], w * h
        if schedule == "direct_send":
            merged = torch.zeros((n, 5), dtype=torch.float32, device="cuda")
            slots = torch.arange(self.world, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            c0.composite_ordered(w, h, 0, n, parts.data_ptr(), slots.data_ptr(), self.grid, merged.data_ptr())
            c0.synchronize()
            return merged
        held = {r: parts[r] for r in range(self.world)}
        for rnd, axis in enumerate(self.sl.swap_rounds(self.world, self.grid)):
            nxt = {}
            for r in sorted(held):
                if r & (1 << rnd):
                    continue
                out = torch.empty((n, 5), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                c0.composite_pair(w, h, 0, n, held[r].data_ptr(), held[r | (1 << rnd)].data_ptr(), axis, 1, out.data_ptr())
                c0.synchronize()
                nxt[r] = out
            held = nxt
        return held[0]

    def chain(self, pixels, w, h):
        """The listed pixels in the shader's own order, the float colour handed from box to box back to front."""
        torch, m = self.torch, pixels.numel()
        colour = torch.zeros((m, 4), dtype=torch.float32, device="cuda")
        for stage in range(self.world):
            total = torch.zeros((m, 4), dtype=torch.float32, device="cuda")
            out = torch.empty((m, 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            for r, c in enumerate(self.ctxs):
                c.chain_stage(w, h, pixels.data_ptr(), m, colour.data_ptr(), out.data_ptr(), stage, self.grid,
                              self.sl.box_of_rank(r, self.grid))
                c.synchronize()
                total += out
                torch.cuda.synchronize()
            colour = total
        return colour

    def frame(self, parts, w, h, schedule):
        """Steps 2 to 4 on the partial images: the frame (device, uint8 [h, w, 4]) and how many pixels were re-marched."""
        torch, c0, n = self.torch, self.ctxs[0], w * h
        merged = self.merge(parts, w, h, schedule)
        rgba = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        todo = torch.zeros(n, dtype=torch.int32, device="cuda")
        count = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        c0.decide_partials(w, h, 0, n, merged.data_ptr(), rgba.data_ptr(), todo.data_ptr(), count.data_ptr())
        c0.synchronize()
        m = int(count.item())
        if m:
            pixels = todo[:m].contiguous()
            colour = self.chain(pixels, w, h)
            c0.store_chain(pixels.data_ptr(), colour.data_ptr(), m, 0, n, rgba.data_ptr())
            c0.synchronize()
        return rgba.view(h, w, 4), m

    def literal_rect(self, w, h, rect):
        """Every pixel of the rectangle through the box-to-box chain: the shader's loop order, no early decision."""
        torch = self.torch
        x0, y0, x1, y1 = rect
        ys = torch.arange(y0, y1, dtype=torch.int64, device="cuda").view(-1, 1)
        xs = torch.arange(x0, x1, dtype=torch.int64, device="cuda").view(1, -1)
        pixels = (ys * w + xs).reshape(-1).to(torch.int32).contiguous()
        colour = self.chain(pixels, w, h)
        out = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        self.ctxs[0].store_chain(pixels.data_ptr(), colour.data_ptr(), pixels.numel(), 0, w * h, out.data_ptr())
        self.ctxs[0].synchronize()
        return out.view(h, w, 4)[y0:y1, x0:x1].cpu().numpy()


def _parity_rects(frame, w, h):
    """Four rectangles on the object's outline and two inside it (the image centre, where the cut planes of a
    2x2x2 split cross for the default camera, and one off centre)."""
    outline = cf.silhouette_rects(frame, size=96, count=4)[:4]
    inside = [(w // 2 - 48, h // 2 - 48, w // 2 + 48, h // 2 + 48), (w // 2 + 200, h // 2 - 160, w // 2 + 296, h // 2 - 64)]
    return outline, inside


@pytest.mark.parametrize("balance", ["even", "occupied"])
def test_config5_at_its_own_size(vr, torch, balance):
    """BASELINE configs[4] as the 8-GPU run holds it: the 4096^3 uint16 phantom in 2x2x2 boxes of bricks (207 GB,
    all eight resident on this one GPU), 3840x2160, 14188 samples per ray.  The whole volume does not fit beside
    its boxes, so the frame is checked against the shader's own loop order where that can be had: every pixel of
    rectangles on the outline and inside the object goes through the box-to-box chain (vr_chain_stage, no early
    decision, the float colour handed on in the shader's blend order, VolumeRendering.comp:183-202) and must give
    the bytes of the fast frame; both exchange schedules must give the same frame.  One camera looks along
    two of the cut planes (rays of the centre row and column run inside them)."""
    n, w, h = 4096, 3840, 2160
    samples = int(math.ceil(2 * math.sqrt(3) * n)) - 1
    assert samples == 14188
    free, total = torch.cuda.mem_get_info()
    if total < 250e9:
        pytest.skip("needs the 288 GB of an MI355X: %.0f GB here" % (total / 1e9))
    boxes = BoxesOnOneGpu(vr, torch, n, (2, 2, 2), samples, balance=balance)
    try:
        if balance == "occupied":
            lo, hi = boxes.cuts.box((0, 0, 0))
            print("4096^3 balanced cuts: box (0,0,0) = bricks %s..%s" % (lo, hi))
        for cam in ((0.0, 0.0), (math.radians(40.0), math.radians(20.0)), (math.radians(-110.0), math.radians(-35.0))):
            boxes.set_camera(cam)
            parts = boxes.partials(w, h)
            direct, remarched = boxes.frame(parts, w, h, "direct_send")
            swap, remarched_swap = boxes.frame(parts, w, h, "binary_swap")
            del parts
            assert torch.equal(direct, swap), cam
            fast = direct.cpu().numpy()
            del direct, swap
            assert fast[..., :3].any() and np.all(fast[..., 3] == 255)
            assert 0 < remarched < 0.02 * w * h and 0 < remarched_swap < 0.02 * w * h
            outline, inside = _parity_rects(fast, w, h)
            assert len(outline) >= 4
            for kind, rects in (("outline", outline), ("inside", inside)):
                for rect in rects:
                    x0, y0, x1, y1 = rect
                    crop = fast[y0:y1, x0:x1]
                    lit = crop[..., :3].any(axis=-1)
                    assert lit.any(), (cam, rect)
                    if kind == "outline":
                        assert not lit.all(), (cam, rect)          # object and background
                    literal = boxes.literal_rect(w, h, rect)
                    assert np.array_equal(literal, crop), (balance, cam, kind, rect, cf.frame_report(literal, crop))
            print("4096^3 %s, camera (%.0f, %.0f): %d + %d rectangles equal the chained loop, %d pixels re-marched"
                  % (balance, math.degrees(cam[0]), math.degrees(cam[1]), len(outline), len(inside), remarched))
    finally:
        boxes.close()


def test_config5_split_equals_whole_at_3200(vr, torch):
    """The largest cube whose whole volume and whose eight boxes both fit one GPU (one after the other): 3200^3
    uint16 at 3840x2160, 11085 samples per ray.  The frame assembled from the boxes (balanced cuts, both exchange
    schedules; even cuts for one camera) equals the whole-volume vr_render frame byte for byte, all 8.3 M pixels."""
    n, w, h = 3200, 3840, 2160
    samples = int(math.ceil(2 * math.sqrt(3) * n)) - 1
    free, total = torch.cuda.mem_get_info()
    if total < 250e9:
        pytest.skip("needs the 288 GB of an MI355X: %.0f GB here" % (total / 1e9))
    cams = ((math.radians(40.0), math.radians(20.0)), (0.0, 0.0))
    want = []
    with vr.VolumeRenderer(0) as c:
        raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        c.phantom_generate_device(raw.data_ptr(), 2, n)
        c.upload_volume_device(raw.data_ptr(), 2, n, n, n)
        c.synchronize()
        del raw
        torch.cuda.empty_cache()
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        for cam in cams:
            c.set_camera(*cam)
            c.set_mode(vr.MODE_FAST)
            fast = c.render(w, h)
            c.set_mode(vr.MODE_REFERENCE_SKIP)
            assert np.array_equal(c.render(w, h), fast), cam
            want.append(fast)
    torch.cuda.empty_cache()
    for balance, use in (("occupied", cams), ("even", cams[:1])):
        boxes = BoxesOnOneGpu(vr, torch, n, (2, 2, 2), samples, balance=balance)
        try:
            for cam in use:
                boxes.set_camera(cam)
                parts = boxes.partials(w, h)
                for schedule in ("direct_send", "binary_swap"):
                    got, remarched = boxes.frame(parts, w, h, schedule)
                    got = got.cpu().numpy()
                    ref = want[cams.index(cam)]
                    assert ref[..., :3].any()
                    assert np.array_equal(got, ref), (balance, cam, schedule, cf.frame_report(got, ref))
                    assert 0 < remarched < 0.02 * w * h
                del parts
        finally:
            boxes.close()
