"""Times the work one GPU of a sort-last split does per frame: the partial march of its box, the
merge of `boxes` partial images of its pixel span, the decision and (with --chain) its share of the
re-march.  The exchanges themselves need the other GPUs and are not part of this measurement.

  python tools/sortlast_rank_time.py --volume 4096 --grid 2 2 2 --box 0 0 0 --width 3840 --height 2160
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkanrenderer_amd as vr                      # noqa: E402
from vulkanrenderer_amd import sortlast as sl        # noqa: E402

TF = [[(0, 0.0), (120, 0.9), (300, 1.0), (483, 1.0)], [(0, 0.0), (120, 0.3), (300, 0.8), (483, 1.0)],
      [(0, 0.0), (120, 0.1), (300, 0.5), (483, 1.0)], [(0, 0.0), (120, 0.0), (300, 0.05), (483, 0.6)]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=2048)
    ap.add_argument("--grid", type=int, nargs=3, default=[2, 2, 2])
    ap.add_argument("--box", type=int, nargs=3, default=None, help="one box; default: every box in turn")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--step-deg", type=float, default=15.0)
    ap.add_argument("--balance", choices=("even", "occupied"), default="occupied",
                    help="even: equal brick counts per box; occupied: k-d cuts with equal numbers of occupied bricks")
    ap.add_argument("--cell", type=int, default=8, help="occupied bricks are counted per cube of this many bricks")
    a = ap.parse_args()
    grid, n, w, h = tuple(a.grid), a.volume, a.width, a.height
    world = grid[0] * grid[1] * grid[2]
    samples = int(math.ceil(2 * math.sqrt(3) * n)) - 1
    boxes = [tuple(a.box)] if a.box else [sl.box_of_rank(r, grid) for r in range(world)]
    pixels = w * h
    out = {"volume": n, "grid": grid, "frame": [w, h], "samples_per_ray": samples, "balance": a.balance, "boxes": []}
    cuts = None
    if a.balance == "occupied":
        # one box after the other (one GPU stands in for all): occupied bricks of the even boxes, then the cuts
        nb = vr.brick_grid(n, n, n)
        total = None
        for r in range(world):
            with vr.VolumeRenderer(0) as c:
                sl.upload_phantom_box(c, n, 2, grid, sl.box_of_rank(r, grid), "cuda")
                c.update_transfer_function(TF, reset_state=True)
                part = sl.occupied_counts(c, nb, a.cell, "cuda")
                total = part if total is None else total + part
        cuts = sl.balanced_cuts(total, a.cell, nb, grid)
        out["occupied_bricks"] = int(total.sum())
    for box in boxes:
        with vr.VolumeRenderer(0) as c:
            lo, hi = sl.upload_phantom_box(c, n, 2, grid, box, "cuda", cuts=cuts)
            c.update_transfer_function(TF, reset_state=True)
            c.set_sampling(samples)
            stream = torch.cuda.ExternalStream(c.stream)
            partial = torch.zeros((pixels, 5), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            ms = []
            with torch.cuda.stream(stream):
                for k in range(a.frames + 2):
                    c.set_camera(math.radians(-180.0 + k * a.step_deg), math.radians(20.0))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    c.render_partial(w, h, partial.data_ptr())
                    e1.record()
                    e1.synchronize()
                    if k >= 2:
                        ms.append(e0.elapsed_time(e1))
            ms = np.asarray(ms)
            # the re-march of undecided pixels: one call per stage, here for 0.15 % of the pixels picked at random
            rng = np.random.default_rng(3)
            todo = torch.from_numpy(rng.choice(pixels, size=max(1, int(pixels * 0.0015)), replace=False).astype(np.int32)).to("cuda")
            cin = torch.zeros((todo.numel(), 4), dtype=torch.float32, device="cuda")
            cout = torch.zeros_like(cin)
            torch.cuda.synchronize()
            chain = []
            with torch.cuda.stream(stream):
                for stage in range(world):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    c.chain_stage(w, h, todo.data_ptr(), todo.numel(), cin.data_ptr(), cout.data_ptr(), stage, grid, box)
                    e1.record()
                    e1.synchronize()
                    chain.append(e0.elapsed_time(e1))
            out["boxes"].append({"box": box, "bricks": [lo, hi], "partial_ms_mean": float(ms.mean()),
                                 "partial_ms_max": float(ms.max()), "partial_ms_min": float(ms.min()),
                                 "chain_stage_ms": [round(x, 3) for x in chain], "chain_pixels": int(todo.numel())})
            print(json.dumps(out["boxes"][-1]), flush=True)
    # merge + decide of one rank's span (world images of pixels/world partials), synthetic partials
    with vr.VolumeRenderer(0) as c:
        sl.upload_phantom_box(c, 64, 2, (1, 1, 1), (0, 0, 0), "cuda")     # any volume: only its size enters the merge
        c.update_transfer_function(TF, reset_state=True)
        c.set_sampling(samples)
        c.set_camera(0.5, 0.3)
        span = (pixels + world - 1) // world
        parts = torch.rand((world, span, 5), dtype=torch.float32, device="cuda")
        slots = torch.arange(world, dtype=torch.int32, device="cuda")
        merged = torch.zeros((span, 5), dtype=torch.float32, device="cuda")
        rgba = torch.zeros((span, 4), dtype=torch.uint8, device="cuda")
        todo = torch.zeros(span, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        stream = torch.cuda.ExternalStream(c.stream)
        with torch.cuda.stream(stream):
            for k in range(3):
                e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e0.record()
                c.composite_ordered(w, h, 0, span, parts.data_ptr(), slots.data_ptr(), grid, merged.data_ptr())
                e1.record()
                c.decide_partials(w, h, 0, span, merged.data_ptr(), rgba.data_ptr(), todo.data_ptr(), cnt.data_ptr())
                e2.record()
                e2.synchronize()
        out["merge_ms"] = e0.elapsed_time(e1)
        out["decide_ms"] = e1.elapsed_time(e2)
        out["exchange_bytes_per_rank"] = (world - 1) * span * 20
    means = [b["partial_ms_mean"] for b in out["boxes"]]
    out["partial_ms_max_over_mean"] = max(means) / (sum(means) / len(means))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
