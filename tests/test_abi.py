"""The C-ABI library loads and exports every function include/vr_abi.h declares; host-only entry
points (no GPU needed) agree with the checker."""
import ctypes
import os
import re

import numpy as np
import pytest

import conftest as cf
import vulkanrenderer_amd as vr


def declared_functions():
    text = open(os.path.join(cf.ROOT, "include", "vr_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 20
    L = ctypes.CDLL(vr.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    assert sorted(vr.ABI_SYMBOLS) == names


def test_camera_uniforms_bitwise_equal_checker(oracle):
    for h, v in cf.CAMERAS.values():
        for (w, hgt) in ((512, 512), (1280, 720), (3840, 2160), (33, 17)):
            vp, m = vr.camera_uniforms(h, v, w, hgt)
            ovp, om = oracle.camera(h, v, w, hgt)
            assert cf.same_bits(vp, ovp) and cf.same_bits(m, om)


def test_camera_uniforms_rejects_degenerate_extent():
    vp = np.zeros(20, np.float32)
    m = np.zeros(16, np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    assert vr.lib().vr_camera_uniforms(0.0, 0.0, 1, 720, vp.ctypes.data_as(fp), m.ctypes.data_as(fp)) == -1


def test_control_point_intensity(oracle):
    for y, hgt in ((0.0, 300.0), (299.0, 300.0), (17.0, 123.0)):
        assert vr.control_point_intensity(y, hgt) == oracle.control_point_intensity(y, hgt)


def test_binding_argument_counts_match_the_header():
    """Every prototype of the header has as many parameters as the ctypes binding declares for it."""
    text = open(os.path.join(cf.ROOT, "include", "vr_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    L = vr.lib()
    checked = 0
    for name, params in re.findall(r"\b(vr_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = params.strip()
        count = 0 if params in ("", "void") else len(params.split(","))
        argtypes = getattr(L, name).argtypes
        assert argtypes is not None, "no argtypes for %s" % name
        assert len(argtypes) == count, (name, len(argtypes), count)
        checked += 1
    assert checked == len(vr.ABI_SYMBOLS)


def test_multi_gpu_library_exports_every_declared_symbol():
    """libvr_mg.so (include/vr_mg.h): loads, exports every declared function, binding arities match."""
    from vulkanrenderer_amd import multigpu as mg
    text = open(os.path.join(cf.ROOT, "include", "vr_mg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = re.findall(r"\b(vrmg_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    assert sorted(n for n, _ in protos) == sorted(mg.MG_SYMBOLS)
    L = mg.lib()
    for name, params in protos:
        params = params.strip()
        count = 0 if params in ("", "void") else len(params.split(","))
        assert len(getattr(L, name).argtypes) == count, name
    assert mg.UNIQUE_ID_BYTES == 128


# ---- the sort-last schedule of libvr_mg.so as pure host arithmetic ------------------------------------------

@pytest.mark.parametrize("world,grid", [(1, (1, 1, 1)), (2, (2, 1, 1)), (2, (1, 1, 2)), (4, (2, 2, 1)), (4, (4, 1, 1)), (4, (1, 2, 2)),
                                        (8, (2, 2, 2)), (8, (4, 2, 1)), (8, (8, 1, 1)), (8, (1, 2, 4)), (16, (4, 2, 2))])
def test_binary_swap_plan_of_every_rank(world, grid):
    """vrmg_plan_sort_last for all ranks of a world at once: the rounds pair up (my partner's partner is me, what I
    send is what my partner merges with what it keeps, on the same pixels), every merged pair is split by the
    round's axis, the buffers a round reads and writes differ, after the last round the ranks' spans tile the
    padded frame, and everything equals the Python driver's arithmetic (sortlast.swap_rounds / swap_span)."""
    from vulkanrenderer_amd import multigpu as mg
    from vulkanrenderer_amd import sortlast as sl
    width, height = 333, 211           # not a multiple of 64 * world: padding
    plans = [mg.plan_sort_last(r, world, grid, mg.BINARY_SWAP, width, height) for r in range(world)]
    rounds = sl.swap_rounds(world, grid)
    unit = world * 64
    padded = (width * height + unit - 1) // unit * unit
    held = [(0, padded)] * world       # frame pixels [first, first + count) a rank holds, starting with everything
    where = [-1] * world               # which buffer holds them: -1 the partial image, 0 / 1 scratch, 2 the final span
    for p in plans:
        assert (p.pixels, p.padded, p.span, p.rounds) == (width * height, padded, padded // world, len(rounds))
    for r, axis in enumerate(rounds):
        nxt, nwhere = list(held), list(where)
        for rank, p in enumerate(plans):
            rd = p.round[r]
            partner = plans[rd.partner].round[r]
            assert rd.partner == rank ^ (1 << r) and partner.partner == rank
            assert rd.axis == axis == partner.axis and rd.count == partner.count == held[rank][1] // 2
            assert held[rank] == held[rd.partner]                          # both hold the same pixels before the round
            # the half I send is the half my partner keeps, and the other way round
            assert rd.send_offset == partner.keep_offset and rd.keep_offset == partner.send_offset
            assert {rd.keep_offset, rd.send_offset} == {0, rd.count}
            assert rd.first_pixel == held[rank][0] + rd.keep_offset
            # the two ranks differ in exactly the coordinate of the round's axis, and the low one says so
            mine, theirs = sl.box_of_rank(rank, grid), sl.box_of_rank(rd.partner, grid)
            assert [a != b for a, b in zip(mine, theirs)] == [k == axis for k in range(3)]
            assert rd.mine_is_low == int(mine[axis] < theirs[axis]) and partner.mine_is_low == 1 - rd.mine_is_low
            # a round never writes the buffer it reads
            assert rd.out_buffer != where[rank] and rd.out_buffer == (2 if rd.count == p.span else r & 1)
            nxt[rank] = (rd.first_pixel, rd.count)
            nwhere[rank] = rd.out_buffer
        held, where = nxt, nwhere
    covered = sorted(held)
    assert [c for _, c in covered] == [padded // world] * world
    assert [f for f, _ in covered] == [k * (padded // world) for k in range(world)]       # the spans tile the frame
    for rank, p in enumerate(plans):
        assert (p.first, p.span) == held[rank] == sl.swap_span(rank, world, padded)
        assert tuple(p.box) == sl.box_of_rank(rank, grid)
        assert world == 1 or where[rank] == 2
    # direct send: consecutive spans, no rounds
    for rank in range(world):
        p = mg.plan_sort_last(rank, world, grid, mg.DIRECT_SEND, width, height)
        assert (p.rounds, p.first, p.span, p.padded) == (0, rank * (padded // world), padded // world, padded)


def test_sort_last_plan_rejects_what_the_frame_rejects():
    from vulkanrenderer_amd import multigpu as mg
    import vulkanrenderer_amd as vr
    for args in ((0, 6, (3, 2, 1), mg.BINARY_SWAP, 64, 64),      # not a power of two per axis
                 (0, 4, (2, 2, 2), mg.DIRECT_SEND, 64, 64),      # grid does not match the world
                 (4, 4, (2, 2, 1), mg.DIRECT_SEND, 64, 64),      # rank outside the world
                 (0, 2, (2, 1, 1), 7, 64, 64),                   # unknown schedule
                 (0, 2, (2, 1, 1), mg.DIRECT_SEND, 1, 64)):      # image too small
        with pytest.raises(vr.VrError):
            mg.plan_sort_last(*args)
    assert mg.plan_sort_last(5, 6, (3, 2, 1), mg.DIRECT_SEND, 64, 64).box[:] == [2, 1, 0]
