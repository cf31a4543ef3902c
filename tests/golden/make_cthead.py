#!/usr/bin/env python3
"""Renders the golden frame of the reference's CThead dataset with the CPU checker.

The dataset (Applications/VolumeRendering/Dependencies/VolumeData, 113 headerless slices of 256x256
big-endian uint16) exists only where the reference checkout does and is not copied: this script
reads it there through the checker's slice loader and commits only the 128x72 RGBA8 frame rendered
from it.  Run from the repository root: python tests/golden/make_cthead.py
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import vr_oracle as o  # noqa: E402
import conftest as cf  # noqa: E402

DATA = "/root/reference/Applications/VolumeRendering/Dependencies/VolumeData"
CAMERA = (math.radians(30.0), math.radians(20.0))
SIZE = (128, 72)
SAMPLES = 200


def render(vol):
    lut = np.zeros((256, 4), np.float32)
    o.lut_update(lut, cf.TF_GRAPHS)
    vp, m = o.camera(CAMERA[0], CAMERA[1], *SIZE)
    return o.render(vol, lut, vp[4:8], m, SIZE[0], SIZE[1], SAMPLES)


def main():
    o.build()
    vol, max_value = o.load_slices(DATA)
    np.savez_compressed(os.path.join(HERE, "cthead_frame.npz"), frame=render(vol))
    print("wrote cthead_frame.npz from", vol.shape, "max", max_value)


if __name__ == "__main__":
    main()
