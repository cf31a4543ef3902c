import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The measurement transfer function: four control points per channel, x_last = 483.
TF_X = (0.0, 120.0, 300.0, 483.0)
TF_GRAPHS = [
    list(zip(TF_X, (0.0, 0.9, 1.0, 1.0))),
    list(zip(TF_X, (0.0, 0.3, 0.8, 1.0))),
    list(zip(TF_X, (0.0, 0.1, 0.5, 1.0))),
    list(zip(TF_X, (0.0, 0.0, 0.05, 0.6))),
]

# Transfer-function cases: name -> list of successive updates (each four graphs), applied to a
# zeroed table in order without resetting in between.
def lut_cases():
    h = 300.0  # histogram child window height used for the default end points
    from oracle import vr_oracle as o
    low = o.control_point_intensity(h - 1, h)  # default end points sit on the floor: intensity 0
    default = [[(0.0, low), (482.0, low)]] * 4
    frac = [[(0.0, 0.1), (37.0, 0.8), (111.0, 0.2), (333.0, 0.95)]] * 3 + [[(0.0, 0.0), (50.0, 0.0), (333.0, 1.0)]]
    # a duplicated x in the middle is overwritten by the next segment; at the end it stays (0/0)
    dup = [[(0.0, 0.2), (100.0, 0.4), (100.0, 0.9), (400.0, 0.7), (400.0, 1.0)]] * 4
    many = [[(float(x), ((x * 37) % 101) / 100.0) for x in range(0, 481, 16)]] * 4
    return {
        "default_two_points": [default],
        "four_points": [TF_GRAPHS],
        "fractional_remap": [frac],
        "duplicate_x_nan": [dup],
        "stale_state": [TF_GRAPHS, frac, default],
        "single_channel_then_rest": [[TF_GRAPHS[0], [], [], []], [[], TF_GRAPHS[1], TF_GRAPHS[2], TF_GRAPHS[3]]],
        "many_points": [many],
    }


CAMERAS = {
    "front": (0.0, 0.0),
    "h30_v20": (math.radians(30.0), math.radians(20.0)),
    "hm135_vm60": (math.radians(-135.0), math.radians(-60.0)),
    "top": (math.radians(90.0), math.radians(90.0)),
}


# push constants `Scene` (ambient, diffuse, specular, shininess): the reference's values (main.cpp:90-98)
# and a second set inside the sliders' ranges (main.cpp:1212-1215)
SCENES = {
    "reference": (1.0, 1.0, 1.0, 100.0),
    "soft": (0.4, 0.7, 0.25, 12.0),
}


def same_bits(a, b):
    """Bitwise equality of float32 arrays, any NaN equal to any NaN."""
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    if a.shape != b.shape:
        return False
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | both_nan))


def frame_report(got, want):
    """max |delta| over channels and fraction of pixels that are identical."""
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    exact = float(np.mean(np.all(d == 0, axis=-1)))
    return int(d.max()), exact


@pytest.fixture(scope="session")
def oracle():
    from oracle import vr_oracle as o
    o.build()
    return o


def silhouette_rects(frame, size=128, count=4):
    """Rectangles of size x size pixels centred on the object's outline in a rendered frame: where a row or
    a column of the frame goes from background to object."""
    h, w = frame.shape[:2]
    lit = frame[..., :3].any(axis=-1)
    rects = []

    def add(cx, cy):
        x0 = int(min(max(cx - size // 2, 0), w - size)) & ~7
        y0 = int(min(max(cy - size // 2, 0), h - size)) & ~7
        rects.append((x0, y0, x0 + size, y0 + size))
    for frac in np.linspace(0.3, 0.7, count // 2):
        row = int(h * frac)
        cols = np.flatnonzero(lit[row])
        if cols.size:
            add(cols[0], row)
            add(cols[-1], row)
    col = w // 2
    rows = np.flatnonzero(lit[:, col])
    if rows.size:
        add(col, rows[0])
        add(col, rows[-1])
    return rects
