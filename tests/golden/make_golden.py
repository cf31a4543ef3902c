#!/usr/bin/env python3
"""Regenerates the committed golden vectors from the CPU checker (oracle/).

The reference ships no tests or expected outputs and cannot be built in this image, so these
vectors come from the restatement itself: they pin it against accidental change and give the
GPU tests fixed expected data.  Run from the repository root: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import vr_oracle as o  # noqa: E402
import conftest as cf  # noqa: E402


def main():
    o.build()
    out = {}
    for name, updates in cf.lut_cases().items():
        lut = np.zeros((256, 4), np.float32)
        for graphs in updates:
            o.lut_update(lut, graphs)
        out["lut_" + name] = lut
    for cname, (h, v) in cf.CAMERAS.items():
        for (w, hgt) in ((512, 512), (1280, 720)):
            vp, m = o.camera(h, v, w, hgt)
            out["cam_%s_%dx%d_vp" % (cname, w, hgt)] = vp
            out["cam_%s_%dx%d_m" % (cname, w, hgt)] = m
    lut = out["lut_four_points"]
    for dtype, n, tag in ((np.uint8, 32, "u8_32"), (np.uint16, 64, "u16_64")):
        vol, _ = o.normalize(o.phantom(n, dtype))
        for cname, (h, v) in cf.CAMERAS.items():
            vp, m = o.camera(h, v, 64, 64)
            out["frame_%s_%s" % (tag, cname)] = o.render(vol, lut, vp[4:8], m, 64, 64, 200)
    # the shader's commented-out lighting switched on, the reference's constants and a second set
    vol, _ = o.normalize(o.phantom(32, np.uint8))
    for cname, (h, v) in cf.CAMERAS.items():
        vp, m = o.camera(h, v, 64, 64)
        for sname, scene in cf.SCENES.items():
            out["shaded_u8_32_%s_%s" % (cname, sname)] = o.render(vol, lut, vp[4:8], m, 64, 64, 200, scene=scene)
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
