#!/usr/bin/env python3
"""Makes the real-data fixture of the GPU tests: a block of the reference's CThead dataset and the frames
the CPU checker renders from it with the reference's own defaults.

Input (read only where the reference checkout exists): Applications/VolumeRendering/Dependencies/VolumeData,
113 headerless slices of 256x256 big-endian uint16.  Kept: the central 88x88 pixels of every slice, still
big-endian, exactly as they sit in the files (1.75 MB raw) -- the slice loader needs square slices, so the
block is cut in x and y only.  The test writes them back out as slice files CThead.1 ... CThead.113 and
loads those through vr_load_volume_slices.

Expected frames: the reference's window (1280x720, main.cpp:30), its shader constant samples = 200
(VolumeRendering.comp:173) and its default camera (both sliders at 0, main.cpp:69,77), with the measurement
transfer function, for both readings of the sampler's border colour (main.cpp:380), and with the
editor's default graphs (two points at intensity 0, main.cpp:974-992: an all-zero table, a black frame).
Run from the repository root: python tests/golden/make_cthead_crop.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import vr_oracle as o  # noqa: E402
import conftest as cf  # noqa: E402

DATA = "/root/reference/Applications/VolumeRendering/Dependencies/VolumeData"
SIDE, CROP, SLICES = 256, 88, 113
SIZE = (1280, 720)
SAMPLES = 200
CAMERA = (0.0, 0.0)
BORDERS = {"transparent": (0.0, 0.0, 0.0, 0.0), "opaque": (0.0, 0.0, 0.0, 1.0)}


def write_slices(block, directory):
    """block: [slices, side, side] big-endian uint16 -> files CThead.<k>, k from 1."""
    for k in range(block.shape[0]):
        block[k].astype(">u2").tofile(os.path.join(directory, "CThead.%d" % (k + 1)))


def frames_of(directory):
    vol, mx = o.load_slices(directory)
    vp, m = o.camera(CAMERA[0], CAMERA[1], *SIZE)
    out = {}
    lut = np.zeros((256, 4), np.float32)
    o.lut_update(lut, cf.TF_GRAPHS)
    for name, border in BORDERS.items():
        out["frame_" + name] = o.render(vol, lut, vp[4:8], m, SIZE[0], SIZE[1], SAMPLES, border=border)
    default = np.zeros((256, 4), np.float32)
    o.lut_update(default, cf.lut_cases()["default_two_points"][0])
    out["frame_default_graphs"] = o.render(vol, default, vp[4:8], m, SIZE[0], SIZE[1], SAMPLES)
    return out, vol.shape, mx


def main():
    o.build()
    lo = (SIDE - CROP) // 2
    block = np.empty((SLICES, CROP, CROP), ">u2")
    for k in range(SLICES):
        s = np.fromfile(os.path.join(DATA, "CThead.%d" % (k + 1)), ">u2").reshape(SIDE, SIDE)
        block[k] = s[lo:lo + CROP, lo:lo + CROP]
    with tempfile.TemporaryDirectory() as d:
        write_slices(block, d)
        frames, shape, mx = frames_of(d)
    np.savez_compressed(os.path.join(HERE, "cthead_crop.npz"), slices=block, max_value=np.float32(mx), **frames)
    print("wrote cthead_crop.npz: block", shape, "max", mx, {k: int(v[..., :3].any(axis=-1).sum()) for k, v in frames.items()})


if __name__ == "__main__":
    main()
